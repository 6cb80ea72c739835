#!/usr/bin/env python3
"""Condense a tools/profile.sh output directory into a small markdown + csv under profiles/.
Usage: tools/summarize_prof.py gpurun_out/prof_<tag> profiles/<name> [traffic-key] [kernel-substring]

`traffic-key` (default: basename of <name>) is the entry written to profiles/traffic.json, which bench.py replays --
labelled with its source -- for the counter-derived fields it cannot measure in-run: `<kernel>_<workload>_g<lanes>`,
e.g. k_a1_step_terrain_g32."""
import collections
import csv
import json
import os
import shutil
import sys


def agg(path):
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    if not os.path.exists(path):
        return d
    for r in csv.DictReader(open(path)):
        d[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return d


def main(src, dst, key=None, kern="k_a1_step"):
    key = key or os.path.basename(dst)
    os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
    shutil.copy(os.path.join(src, "stats", "a1_kernel_stats.csv"), dst + "_kernel_stats.csv")
    rows = list(csv.DictReader(open(os.path.join(src, "stats", "a1_kernel_stats.csv"))))
    out = ["# rocprofv3 summary: " + os.path.basename(src), "",
           "Command: `tools/profile.sh` (rocprofv3 --kernel-trace --stats, then separate --pmc passes) around",
           "`python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline ...`; 4096 envs, 1x MI355X.", ""]
    bj = os.path.join(src, "bench_stats.json")
    kernel_sha = kernel_symbol = None
    if os.path.exists(bj):
        try:
            b = json.loads(open(bj).read().strip().splitlines()[-1])
            kernel_sha, kernel_symbol = b["roofline"].get("kernel_code_sha"), b["roofline"].get("kernel_symbol")
            out += ["bench.py line under the profiler: value = %.3e env-steps/s, kernel_ms (HIP events) = %.4f" %
                    (b["value"], b["roofline"]["kernel_ms"]), ""]
        except Exception:
            pass
    out += ["## Kernel time (--kernel-trace --stats)", "", "| kernel | calls | avg ns | % |", "|---|---|---|---|"]
    for r in rows[:6]:
        out.append("| `%s` | %s | %.0f | %s |" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]), r["Percentage"]))
    out += ["", "## PMC counters, mean per launch (separate passes)", ""]
    for f in ("pmc_sq", "pmc_lds", "pmc_fetch", "pmc_write"):
        d = agg(os.path.join(src, f, "a1_counter_collection.csv"))
        for k, v in d.items():
            if kern in k or "k_sim_step" in k:
                out.append("**%s** `%s`" % (f, k[:60]))
                out.append("")
                for c, x in sorted(v.items()):
                    out.append("- %s = %.4g (n=%d)" % (c, sum(x) / len(x), len(x)))
                out.append("")
    # HBM traffic per launch of the fused kernel, corrected as MI355X_MICROARCH.md (HBM section) prescribes:
    # FETCH_SIZE under-reports by 2x on gfx950 (wide reads), WRITE_SIZE as is; both in KiB.
    fetch = write = None
    for f, ckey in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
        for k, v in agg(os.path.join(src, f, "a1_counter_collection.csv")).items():
            if kern in k and ckey in v:
                val = sum(v[ckey]) / len(v[ckey])
                if ckey == "FETCH_SIZE":
                    fetch = val
                else:
                    write = val
    if fetch is not None and write is not None:
        traffic = (2.0 * fetch + write) * 1024.0
        out += ["## HBM traffic per launch", "",
                "2 x FETCH_SIZE + WRITE_SIZE = %.2f MB (FETCH %.0f KiB, WRITE %.0f KiB)" % (traffic / 1e6, fetch, write), ""]
        tj = os.path.join(os.path.dirname(dst) or ".", "traffic.json")
        db = json.load(open(tj)) if os.path.exists(tj) else {}
        ent = {"fetch_kib": fetch, "write_kib": write, "traffic_bytes": traffic,
               # machine-code hash of the kernel these counters were collected from (the bench line under the profiler carries it):
               # bench.py prints counters_stale when the library it runs holds another build of the kernel
               "kernel_code_sha": kernel_sha, "kernel_symbol": kernel_symbol,
               "source": "profiles/%s_summary.md: rocprofv3 --pmc passes of this bench command (2 x FETCH_SIZE + WRITE_SIZE, "
                         "gfx950 correction), replayed by bench.py -- not measured in the run that prints it" % os.path.basename(dst)}
        sq = {}
        for k, v in agg(os.path.join(src, "pmc_sq", "a1_counter_collection.csv")).items():
            if kern in k:
                sq.update({c: sum(x) / len(x) for c, x in v.items()})
        for k, v in agg(os.path.join(src, "pmc_lds", "a1_counter_collection.csv")).items():
            if kern in k:
                sq.update({c: sum(x) / len(x) for c, x in v.items()})
        if sq.get("SQ_WAVE_CYCLES"):
            ent["valu_issue_frac"] = sq.get("SQ_ACTIVE_INST_VALU", 0.0) / sq["SQ_WAVE_CYCLES"]
            if "SQ_WAIT_ANY" in sq:
                ent["wait_frac"] = sq["SQ_WAIT_ANY"] / sq["SQ_WAVE_CYCLES"]
            ent["valu_insts_per_launch"] = sq.get("SQ_INSTS_VALU")
            ent["lds_insts_per_launch"] = sq.get("SQ_INSTS_LDS")
        db[key] = ent
        json.dump(db, open(tj, "w"), indent=1, sort_keys=True)
    open(dst + "_summary.md", "w").write("\n".join(out) + "\n")
    print("\n".join(out))


if __name__ == "__main__":
    main(*sys.argv[1:5])
