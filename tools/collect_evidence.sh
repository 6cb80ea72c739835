#!/bin/bash
# Here (no GPU), after a tools/final_evidence.sh pass came back in gpurun_out/: copy what is quoted into profiles/<round>_*
# and condense the rocprofv3 directories (tools/summarize_prof.py also stamps profiles/traffic.json with the kernels' code hashes).
# Usage: tools/collect_evidence.sh [round tag, default r06]
set -u
R=${1:-r06}
cd "$(dirname "$0")/.."
F=gpurun_out/final
for f in $F/bench_*.json; do
  [ "$f" -nt tools/final_evidence.sh ] || continue          # (older passes leave files in the same directory)
  n=$(basename "$f" .json); tail -1 "$f" > profiles/${R}_$n.json
done
{ cat $F/pytest_gpu.txt; tail -2 $F/smoke.txt; } > profiles/${R}_pytest_gpu.txt
cp $F/soak.txt profiles/${R}_soak.txt
for p in a1_chain_pgs a1_chain_pgs_tw_self a1_chain_g32 abb_pgs_link abb_ws_hard; do
  [ $F/phase_$p.txt -nt tools/final_evidence.sh ] && grep -v "amdgpu.ids" $F/phase_$p.txt > profiles/${R}_phase_clock_$p.txt
done
{ echo "# Env-count sweep, one MI355X (tools/final_evidence.sh sweep: bench.py --envs N --solver S --steps 200 --warmup 20)"; echo
  echo "| solver | envs | ms per vec-step | env-steps/s |"; echo "|---|---|---|---|"
  awk '{printf "| %s | %s | %.4f | %s |\n", $1, $2, $3, $4}' $F/env_count_sweep.txt; } > profiles/${R}_env_count_sweep.md
{ echo "# Hook path, 4096 envs, tools/bench_hook_envs.py --steps 300 --graph-hooks (five runs under the default solver, three under the compliant law)"; echo
  echo '```'; cat $F/hook_envs.txt; echo '```'; echo; echo "compliant:"; echo; echo '```'; cat $F/hook_envs_compliant.txt; echo '```'; } > profiles/${R}_hook_envs.md
S=tools/summarize_prof.py
python $S gpurun_out/prof_${R}_a1_tgs profiles/${R}_a1_chain_tgs k_a1_step_terrain_g32_chain_tgs k_a1_chain_tgs > /dev/null
python $S gpurun_out/prof_${R}_a1_pgs profiles/${R}_a1_chain_pgs k_a1_step_terrain_g32_chain_pgs k_a1_chain_pgs > /dev/null
python $S gpurun_out/prof_${R}_a1_tgs_tw_self profiles/${R}_a1_chain_tgs_tw_self k_a1_step_trimesh_g32_chain_tgs_self k_a1_chain_tgs > /dev/null
python $S gpurun_out/prof_${R}_a1_compliant profiles/${R}_a1_chain k_a1_step_terrain_g32_chain k_a1_chain > /dev/null
python $S gpurun_out/prof_${R}_abb_tgs profiles/${R}_abb_tgs k_abb_step_abb_g16_split_tgs_link k_abb_step_ws_hard > /dev/null
python $S gpurun_out/prof_${R}_abb_pgs profiles/${R}_abb_pgs k_abb_step_abb_g16_split_pgs_link k_abb_step_ws_hard > /dev/null
python $S gpurun_out/prof_${R}_abb_compliant profiles/${R}_abb_step_ws_link k_abb_step_abb_g16_split_link k_abb_step_ws > /dev/null
ls profiles | grep "^${R}_" | wc -l
