#!/bin/bash
# On the GPU box: the A1 schedule with the reference's collision filter 0 (self-collision ON, units.py:68) for ITERS
# iterations, then the deterministic policy played with self-collision on.  Output in gpurun_out/train_a1_selfcol/.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
ITERS=${1:-1000}
OUT=$REPO/gpurun_out/train_a1_selfcol
mkdir -p "$OUT" /tmp/train_a1_sc
cd "$REPO"
python tools/train_a1.py --iters "$ITERS" --graph --quiet --self-collision --log /tmp/train_a1_sc > "$OUT/train_summary.json" 2> "$OUT/train.err"
cp /tmp/train_a1_sc/progress.jsonl "$OUT/progress.jsonl"
for T in heightfield flat; do
  python tools/play_a1.py /tmp/train_a1_sc/model_"$ITERS".pt --envs 1024 --steps 500 --terrain $T --self-collision > "$OUT/play_$T.json" 2>> "$OUT/train.err"
done
cut -c1-600 "$OUT/train_summary.json"
cat "$OUT"/play_*.json
