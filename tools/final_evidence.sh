#!/bin/bash
# On the GPU box (gpurun): everything the round's numbers are quoted from, in one pass -> gpurun_out/final/.
# Afterwards (here): copy bench_*.json to profiles/r06_bench_*.json, run tools/summarize_prof.py on the prof_* dirs.
# Usage: tools/final_evidence.sh [part ...]   parts: tests bench sweep hooks phase prof prof_a1 prof_abb soak fuzz  (default: all but prof_a1 / prof_abb / fuzz)
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/final
mkdir -p "$OUT"
cd "$REPO"
PARTS=${*:-tests bench sweep hooks phase prof soak}
has() { [[ " $PARTS " == *" $1 "* ]]; }
B="--steps 500 --warmup 50"
if has tests; then
  python -m pytest tests -m gpu -q 2>&1 | tail -3 > "$OUT/pytest_gpu.txt"
  python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.txt" 2>&1
  cat "$OUT/pytest_gpu.txt"; tail -2 "$OUT/smoke.txt"
fi
if has bench; then
  # the default solver (TGS, the reference's PhysX settings: solver_type = 1) on every workload, with the CPU baseline ...
  for W in terrain flat trimesh abb; do
    python bench.py --workload $W $B > "$OUT/bench_$W.json" 2> "$OUT/bench_$W.err"
  done
  # ... PGS (solver_type = 0) on the same ...
  for W in terrain flat trimesh abb; do
    python bench.py --workload $W --solver pgs $B --no-cpu-baseline > "$OUT/bench_${W}_pgs.json" 2>/dev/null
  done
  # ... the sixteen-constraint cap, config 5 on the run-time-shaped kernel and with the links as convex hulls
  python bench.py --max-contacts 16 $B --no-cpu-baseline --no-other-solver > "$OUT/bench_terrain_k16.json" 2>/dev/null
  python bench.py --workload abb --solver tgs --mapping body $B --no-cpu-baseline --no-other-solver > "$OUT/bench_abb_body.json" 2>/dev/null
  python bench.py --workload abb --link-shapes hull $B --no-cpu-baseline --no-other-solver > "$OUT/bench_abb_hull.json" 2>/dev/null
  python bench.py --steps 20 --warmup 5 > "$OUT/bench_terrain_driver_shape.json" 2>/dev/null
  python bench.py --workload trimesh --self-collision $B --no-cpu-baseline --no-other-solver > "$OUT/bench_trimesh_selfcollision.json" 2>/dev/null
  python bench.py --workload abb --no-link-contacts $B --no-cpu-baseline > "$OUT/bench_abb_rod_only.json" 2>/dev/null
  # ... and the compliant law of rounds 1-4 (opt-in) on the same
  for W in terrain flat trimesh abb; do
    python bench.py --workload $W --solver compliant $B --no-cpu-baseline > "$OUT/bench_${W}_compliant.json" 2>/dev/null
  done
  python bench.py --workload trimesh --self-collision --solver compliant $B --no-cpu-baseline > "$OUT/bench_trimesh_selfcollision_compliant.json" 2>/dev/null
  python bench.py --workload abb --no-link-contacts --solver compliant $B --no-cpu-baseline > "$OUT/bench_abb_rod_only_compliant.json" 2>/dev/null
  for f in "$OUT"/bench_*.json; do echo -n "$(basename $f .json) "; tail -1 "$f" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], '%.4g' % d['value'], d['roofline'].get('kernel'), d['roofline'].get('kernel_ms'))"; done
fi
if has sweep; then
  # env-count sweep: where the throughput of one GPU saturates, both solvers
  for S in tgs pgs compliant; do for N in 1024 2048 4096 8192 16384 32768; do
    python bench.py --envs $N --solver $S --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 |
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$S', $N, d['ms_per_step'], '%.4g' % d['value'])"
  done; done > "$OUT/env_count_sweep.txt"
  cat "$OUT/env_count_sweep.txt"
fi
if has hooks; then
  for k in 1 2 3 4 5; do python tools/bench_hook_envs.py --steps 300 --graph-hooks 2>/dev/null | grep "^{"; done > "$OUT/hook_envs.txt"
  for k in 1 2 3; do SHIFU_AMD_SOLVER=compliant python tools/bench_hook_envs.py --steps 300 --graph-hooks 2>/dev/null | grep "^{"; done > "$OUT/hook_envs_compliant.txt"
  tail -2 "$OUT/hook_envs.txt" | cut -c1-400
fi
if has phase; then
  python tools/phase_clock.py 32 200 --chain --pgs > "$OUT/phase_a1_chain_pgs.txt" 2>&1
  python tools/phase_clock.py 32 200 --chain --pgs --self --trimesh > "$OUT/phase_a1_chain_pgs_tw_self.txt" 2>&1
  python tools/phase_clock.py 32 200 --chain > "$OUT/phase_a1_chain_g32.txt" 2>&1
  python tools/phase_clock.py 16 100 --abb --split --link --pgs > "$OUT/phase_abb_ws_hard.txt" 2>&1
  python tools/phase_clock.py 32 100 --abb --link --pgs > "$OUT/phase_abb_pgs_link.txt" 2>&1
fi
if has prof_a1; then
  bash tools/profile.sh r06_a1_tgs > /dev/null 2>&1
  bash tools/profile.sh r06_a1_pgs --solver pgs > /dev/null 2>&1
  find "$REPO/gpurun_out" -name "*kernel_trace.csv" -size +4M -delete
fi
if has fuzz; then
  python tools/fuzz_parity.py --steps 2000 --envs 384 2>&1 | grep "^{" > "$OUT/fuzz.txt"
  wc -l "$OUT/fuzz.txt"
fi
if has prof_abb; then
  bash tools/profile.sh r06_abb_tgs --workload abb > /dev/null 2>&1
  find "$REPO/gpurun_out" -name "*kernel_trace.csv" -size +4M -delete
fi
if has prof; then
  bash tools/profile.sh r06_a1_tgs > /dev/null 2>&1                                                    # the default: k_a1_chain_tgs<false,false>
  bash tools/profile.sh r06_a1_pgs --solver pgs > /dev/null 2>&1                                       # k_a1_chain_pgs<false,false>
  bash tools/profile.sh r06_a1_tgs_tw_self --workload trimesh --self-collision > /dev/null 2>&1         # the reference's effective scene: <true,true>
  bash tools/profile.sh r06_a1_compliant --solver compliant > /dev/null 2>&1
  bash tools/profile.sh r06_abb_tgs --workload abb > /dev/null 2>&1                                     # config 5 under TGS: k_abb_step_ws_hard<true>
  bash tools/profile.sh r06_abb_pgs --workload abb --solver pgs > /dev/null 2>&1
  bash tools/profile.sh r06_abb_compliant --workload abb --solver compliant > /dev/null 2>&1            # k_abb_step_ws<512,true>
  find "$REPO/gpurun_out" -name "*kernel_trace.csv" -size +4M -delete     # keep the merge under gpurun's 64 MiB
fi
if has soak; then
  python tools/soak.py 5000 > "$OUT/soak.txt" 2>&1
  grep done "$OUT/soak.txt"
fi
du -sh "$REPO/gpurun_out"
