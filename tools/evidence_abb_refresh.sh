set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/final; mkdir -p $OUT
B="--steps 500 --warmup 50"
bash tools/profile.sh r06_abb_tgs --workload abb > /dev/null 2>&1
bash tools/profile.sh r06_abb_pgs --workload abb --solver pgs > /dev/null 2>&1
find gpurun_out -name "*kernel_trace.csv" -size +4M -delete
python tools/phase_clock.py 16 100 --abb --split --link --pgs > $OUT/phase_abb_ws_hard.txt 2>&1
python bench.py --workload abb --no-link-contacts $B --no-cpu-baseline > $OUT/bench_abb_rod_only.json 2>/dev/null
python bench.py --workload abb --solver tgs --mapping body $B --no-cpu-baseline --no-other-solver > $OUT/bench_abb_body.json 2>/dev/null
bash tools/run_walk.sh 3000 mfma tgs > $OUT/run_walk_tgs.txt 2>&1
python tools/fuzz_parity.py --steps 2000 --envs 384 2>&1 | grep "^{" > $OUT/fuzz.txt
wc -l $OUT/fuzz.txt; tail -3 $OUT/run_walk_tgs.txt | cut -c1-300
