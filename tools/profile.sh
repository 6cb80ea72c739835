#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + PMC passes of bench.py.
# Usage: tools/profile.sh <tag> [bench args...]    -> gpurun_out/prof_<tag>/
# then (here): tools/summarize_prof.py gpurun_out/prof_<tag> profiles/<name> <traffic-key> <kernel-substring>
set -u
TAG=${1:-r01}; shift || true
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 200 --warmup 20 --no-cpu-baseline $*"
rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o a1 --output-format csv -- python3 $REPO/bench.py $ARGS > "$OUT/bench_stats.json" 2> "$OUT/stats.err"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d "$OUT/pmc_sq" -o a1 --output-format csv -- python3 $REPO/bench.py --steps 40 --warmup 10 --no-cpu-baseline $* > /dev/null 2> "$OUT/pmc_sq.err"
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE -d "$OUT/pmc_lds" -o a1 --output-format csv -- python3 $REPO/bench.py --steps 40 --warmup 10 --no-cpu-baseline $* > /dev/null 2> "$OUT/pmc_lds.err"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o a1 --output-format csv -- python3 $REPO/bench.py --steps 40 --warmup 10 --no-cpu-baseline $* > /dev/null 2> "$OUT/pmc_fetch.err"
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/pmc_write" -o a1 --output-format csv -- python3 $REPO/bench.py --steps 40 --warmup 10 --no-cpu-baseline $* > /dev/null 2> "$OUT/pmc_write.err"
find "$OUT" -name "*.csv" | head -30
ls -la "$OUT"/*/ 2>/dev/null | head -40
