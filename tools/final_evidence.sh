#!/bin/bash
# On the GPU box (gpurun): everything the round's numbers are quoted from, in one pass -> gpurun_out/final/.
# Afterwards (here): copy bench_*.json to profiles/r03_bench_*.json, run tools/summarize_prof.py on the prof_* dirs.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/final
mkdir -p "$OUT"
cd "$REPO"
python -m pytest tests -m gpu -q 2>&1 | tail -3 > "$OUT/pytest_gpu.txt"
python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.txt" 2>&1
for W in terrain flat trimesh abb; do
  python bench.py --workload $W --steps 500 --warmup 50 > "$OUT/bench_$W.json" 2> "$OUT/bench_$W.err"
done
python bench.py --self-collision --steps 500 --warmup 50 --no-cpu-baseline > "$OUT/bench_terrain_selfcollision.json" 2>/dev/null
python bench.py --workload trimesh --self-collision --steps 500 --warmup 50 --no-cpu-baseline > "$OUT/bench_trimesh_selfcollision.json" 2>/dev/null
python bench.py --steps 20 --warmup 5 > "$OUT/bench_terrain_driver_shape.json" 2>/dev/null
python bench.py --actions torch --steps 500 --warmup 50 --no-cpu-baseline > "$OUT/bench_terrain_torch_actions.json" 2>/dev/null
python bench.py --mapping body --group 32 --steps 500 --warmup 50 --no-cpu-baseline > "$OUT/bench_terrain_body_g32.json" 2>/dev/null
python bench.py --mapping chain --group 16 --steps 500 --warmup 50 --no-cpu-baseline > "$OUT/bench_terrain_chain_g16.json" 2>/dev/null
python bench.py --workload abb --link-contacts --steps 300 --warmup 30 --no-cpu-baseline > "$OUT/bench_abb_link_contacts.json" 2>/dev/null
python bench.py --workload abb --mapping chain --steps 500 --warmup 50 --no-cpu-baseline > "$OUT/bench_abb_chain.json" 2>/dev/null
python bench.py --workload abb --mapping body --steps 500 --warmup 50 --no-cpu-baseline > "$OUT/bench_abb_levels.json" 2>/dev/null
for k in 1 2 3; do python tools/bench_hook_envs.py --steps 300 2>&1 | grep "^{" >> "$OUT/hook_envs.txt"; done
python tools/phase_clock.py 32 200 > "$OUT/phase_a1_body_g32.txt" 2>&1
python tools/phase_clock.py 32 200 --chain > "$OUT/phase_a1_chain_g32.txt" 2>&1
python tools/phase_clock.py 16 200 --abb > "$OUT/phase_abb.txt" 2>&1
python tools/phase_clock.py 16 200 --abb --split > "$OUT/phase_abb_split.txt" 2>&1
bash tools/profile.sh r03_a1 > /dev/null 2>&1                                       # the default: chain per lane, 32 lanes
bash tools/profile.sh r03_a1_body --mapping body --group 32 > /dev/null 2>&1
bash tools/profile.sh r03_abb --workload abb > /dev/null 2>&1
find "$REPO/gpurun_out" -name "*kernel_trace.csv" -size +4M -delete     # keep the merge under gpurun's 64 MiB
cat "$OUT/pytest_gpu.txt"; tail -2 "$OUT/smoke.txt"
for W in terrain flat trimesh abb terrain_selfcollision trimesh_selfcollision terrain_driver_shape terrain_torch_actions terrain_body_g32 terrain_chain_g16 abb_link_contacts abb_chain abb_levels; do tail -1 "$OUT/bench_$W.json" | cut -c1-200; done
du -sh "$REPO/gpurun_out"
