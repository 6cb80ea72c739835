#!/bin/bash
# On the GPU box (gpurun): everything the round's numbers are quoted from, in one pass -> gpurun_out/final/.
# Afterwards (here): copy bench_*.json to profiles/r04_bench_*.json, run tools/summarize_prof.py on the prof_* dirs.
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
python bench.py --workload abb --no-link-contacts --steps 500 --warmup 50 --no-cpu-baseline > "$OUT/bench_abb_rod_only.json" 2>/dev/null
python bench.py --self-collision --steps 500 --warmup 50 --no-cpu-baseline > "$OUT/bench_terrain_selfcollision.json" 2>/dev/null
python bench.py --workload trimesh --self-collision --steps 500 --warmup 50 --no-cpu-baseline > "$OUT/bench_trimesh_selfcollision.json" 2>/dev/null
python bench.py --workload trimesh --self-collision --mapping body --steps 500 --warmup 50 --no-cpu-baseline > "$OUT/bench_trimesh_selfcollision_body.json" 2>/dev/null
python bench.py --steps 20 --warmup 5 > "$OUT/bench_terrain_driver_shape.json" 2>/dev/null
python bench.py --mapping body --group 32 --steps 500 --warmup 50 --no-cpu-baseline > "$OUT/bench_terrain_body_g32.json" 2>/dev/null
python bench.py --mapping chain --group 16 --steps 500 --warmup 50 --no-cpu-baseline > "$OUT/bench_terrain_chain_g16.json" 2>/dev/null
python bench.py --workload abb --no-link-contacts --mapping chain --steps 500 --warmup 50 --no-cpu-baseline > "$OUT/bench_abb_rod_only_chain.json" 2>/dev/null
python bench.py --workload abb --group 32 --steps 300 --warmup 30 --no-cpu-baseline > "$OUT/bench_abb_g32.json" 2>/dev/null
python tools/mlp_probe.py --check > "$OUT/mlp_probe.json" 2> "$OUT/mlp_probe_check.txt"
python tools/mlp_probe.py --bf16 > "$OUT/mlp_probe_bf16.json" 2>/dev/null
python tools/mlp_probe.py --tiled > "$OUT/mlp_probe_tiled.json" 2>/dev/null
python tools/mlp_probe.py 4096 > "$OUT/mlp_probe_4096.json" 2>/dev/null
for k in 1 2 3; do python tools/bench_hook_envs.py --steps 300 --graph-hooks 2>/dev/null | grep "^{"; done > "$OUT/hook_envs.txt"
python tools/phase_clock.py 32 200 --chain > "$OUT/phase_a1_chain_g32.txt" 2>&1
python tools/phase_clock.py 16 200 --abb --link > "$OUT/phase_abb_link_g16.txt" 2>&1
python tools/phase_clock.py 16 200 --abb --split > "$OUT/phase_abb_split.txt" 2>&1
bash tools/profile.sh r04_a1 > /dev/null 2>&1                                       # the default: chain per lane, 32 lanes
bash tools/profile.sh r04_abb --workload abb > /dev/null 2>&1                        # config 5 as the reference runs it (link contacts)
bash tools/profile.sh r04_abb_rod_only --workload abb --no-link-contacts > /dev/null 2>&1
find "$REPO/gpurun_out" -name "*kernel_trace.csv" -size +4M -delete     # keep the merge under gpurun's 64 MiB
cat "$OUT/pytest_gpu.txt"; tail -2 "$OUT/smoke.txt"
for W in terrain flat trimesh abb abb_rod_only terrain_selfcollision trimesh_selfcollision trimesh_selfcollision_body terrain_driver_shape terrain_body_g32 terrain_chain_g16 abb_rod_only_chain abb_g32; do echo -n "$W "; tail -1 "$OUT/bench_$W.json" | cut -c80-200; done
du -sh "$REPO/gpurun_out"
