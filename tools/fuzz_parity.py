#!/usr/bin/env python3
"""Long differential run of the fused steps against the CPU oracle (test infrastructure: uses oracle/ like tests/ do).

The `-m gpu` parity tests compare a few dozen to 150 vec-steps; this runs the same comparison for thousands of steps and
more envs -- every tensor, bit for bit, checked every `--every` steps -- on every kernel form of both tasks.

    python tools/fuzz_parity.py [--steps 2000] [--envs 384]        # on the MI355X box
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch


def run(steps=2000, envs=384, every=100, link_envs=128):
    """Every kernel form of both tasks against the oracle for `steps` vec-steps; returns one record per form (raises on the
    first tensor that differs).  tests/test_gpu_parity.py runs a trimmed version under `pytest -m gpu`."""
    from types import SimpleNamespace
    args = SimpleNamespace(steps=steps, envs=envs, every=every)
    from oracle import pyoracle as oracle
    import test_gpu_parity as T
    from shifu_amd import _abi
    out = []
    for group in (32, "chain32", "chain16", "pgs", "tgs"):       # "pgs" / "tgs": k_a1_chain_pgs / _tgs, the velocity-level solve (chain mapping, 32 lanes)
        t0 = time.time()
        cm, sp, tp, terr, hs, bufs, sim, task, rng = T._a1_setup(args.envs, True, seed=123, group="chain32" if group in ("pgs", "tgs") else group, env_off=777,
                                                                 **({"solver": group} if group in ("pgs", "tgs") else {}))
        resets = 0
        for it in range(args.steps):
            raw = (2 * rng.random((args.envs, cm.blob.nd)) - 1).astype(np.float32) * (1.5 if it % 400 < 200 else 0.3)
            task.step(torch.from_numpy(raw).cuda())
            oracle.a1_step(cm.blob, sp, tp, args.envs, 777, bufs, raw, terrain=terr, heights=hs)
            resets += int(bufs["reset"].sum())
            if it % args.every == args.every - 1:
                T._compare(sim, task, bufs, f"a1 {group} step {it}")
        out.append({"task": "a1", "kernel": str(group), "envs": args.envs, "steps": args.steps, "resets": resets, "equal": True,
                    "seconds": round(time.time() - t0, 1)})
        print(json.dumps(out[-1]), flush=True)
    from shifu_amd.abb_task import box_desc
    from shifu_amd.gym.abb_fused import FusedAbbEnv
    extra = [box_desc([0.05, 0.05, 0.02], 0.0, 0.5, True, [0.25, 0.25, 0.11])]
    for name, kw in (("split", dict(group=16, link_contacts=False)), ("chain16", dict(group=16, mapping="chain")),
                     ("chain32", dict(group=32, link_contacts=False)),
                     ("levels16", dict(group=16, mapping="body", link_contacts=False)),
                     ("generic32", dict(group=32, extra_boxes=extra, link_contacts=False)),
                     ("link16", dict(group=16, link_contacts=True)), ("link16-body", dict(group=16, link_contacts=True, mapping="body")),
                     ("link32", dict(group=32, link_contacts=True)),
                     ("link32-generic", dict(group=32, link_contacts=True, extra_boxes=extra)),
                     ("pgs", dict(link_contacts=False, solver="pgs", mapping="body")), ("pgs-link", dict(link_contacts=True, solver="pgs", mapping="body")),
                     # k_abb_step_ws_hard (round 6): arm wave + box wave, the solve regrouped at 32 lanes per env
                     ("pgs-split", dict(link_contacts=False, solver="pgs")), ("pgs-link-split", dict(link_contacts=True, solver="pgs")),
                     ("tgs-link-split", dict(link_contacts=True, solver="tgs"))):
        kw.setdefault("solver", "compliant")
        t0 = time.time()
        n = args.envs if "link" not in name else min(args.envs, link_envs)
        env = FusedAbbEnv(num_envs=n, seed=31, **kw)       # "pgs*": the velocity-level solve; the others name kernel forms of the compliant law
        bufs = {k: env.sim.tensors[t].cpu().numpy().copy() for k, t in T._ABB_SIM_T.items()}
        bufs.update({k: env.task.tensors[t].cpu().numpy().copy() for k, t in T._ABB_T.items()})
        rng = np.random.default_rng(5)
        resets = 0
        for it in range(args.steps):
            raw = (2 * rng.random((n, 3)) - 1).astype(np.float32) * 1.3
            raw[: n // 2, 0] = np.abs(raw[: n // 2, 0])               # half the arms keep pushing +x (rod against the cube)
            env.task.step(torch.from_numpy(raw).cuda())
            oracle.abb_step(env.cm.blob, env.sim_params, env.boxes, env.task_params, n, 0, bufs, raw)
            resets += int(bufs["reset"].sum())
            if it % args.every == args.every - 1:
                torch.cuda.synchronize()
                for k, t in list(T._ABB_SIM_T.items()) + list(T._ABB_T.items()):
                    got = (env.sim.tensors if k in T._ABB_SIM_T else env.task.tensors)[t].cpu().numpy().reshape(bufs[k].shape)
                    np.testing.assert_array_equal(got, bufs[k], err_msg=f"abb {name}: {k} step {it}")
        out.append({"task": "abb", "kernel": name, "mapping": env.mapping, "envs": n, "steps": args.steps, "resets": resets, "equal": True,
                    "seconds": round(time.time() - t0, 1)})
        print(json.dumps(out[-1]), flush=True)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--envs", type=int, default=384)
    ap.add_argument("--every", type=int, default=100)
    args = ap.parse_args()
    run(args.steps, args.envs, args.every)


if __name__ == "__main__":
    main()
