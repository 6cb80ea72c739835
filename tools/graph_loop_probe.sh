#!/bin/bash
# Real-loop probe of the captured PPO update (tools/train_a1.py, 22 iterations, seed 1): parameter checksum per variant.
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/graph_loop2.txt
run() { name=$1; shift
  out=$(env "$@" python tools/train_a1.py --iters 22 --seed 1 --quiet $EXTRA 2>/dev/null | grep "^{" | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['param_checksum'][:2])")
  echo "$name $EXTRA: $out" >> $OUT
}
EXTRA="" run eager X=1
for k in 1 2 3; do
EXTRA="--graph-update" run "none_1stream_memcpy#$k" SHIFU_AMD_REPLAY_MODE=none SHIFU_AMD_TWO_STREAM_UPDATE=0
EXTRA="--graph-update" run "none_1stream_kernelcopy#$k" SHIFU_AMD_REPLAY_MODE=none SHIFU_AMD_TWO_STREAM_UPDATE=0 SHIFU_AMD_IDX_COPY=kernel
EXTRA="--graph-update" run "none_2stream_memcpy#$k" SHIFU_AMD_REPLAY_MODE=none
done
