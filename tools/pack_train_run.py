#!/usr/bin/env python3
"""Condense a tools/run_walk.sh output directory (gpurun_out/train_a1_<tag>/) into one JSON under profiles/.
Usage: tools/pack_train_run.py gpurun_out/train_a1_r06_mfma_tgs profiles/r06_train_a1_3000_mfma_tgs.json "tools/run_walk.sh 3000 mfma tgs"
"""
import json
import os
import sys


def last_json(path):
    for ln in reversed(open(path).read().strip().splitlines()):
        if ln.startswith("{"):
            return json.loads(ln)
    raise SystemExit("no JSON line in " + path)


def main(src, dst, command):
    out = {"command": command, "train": last_json(os.path.join(src, "train_summary.json")), "play": {}}
    for t in ("heightfield", "flat", "trimesh"):
        p = os.path.join(src, f"play_{t}.json")
        if os.path.exists(p):
            out["play"][t] = last_json(p)
    json.dump(out, open(dst, "w"), indent=1)
    tr = out["train"]
    print(dst, "%.1f s" % tr["seconds"], {t: round(v["speed_along_cmd_over_cmd"], 3) for t, v in out["play"].items()},
          {t: v["falls_per_env_per_1000_steps"] for t, v in out["play"].items()})


if __name__ == "__main__":
    main(*sys.argv[1:4])
