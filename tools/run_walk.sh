#!/bin/bash
# On the GPU box (gpurun): the reference's A1 training schedule (A1PPOConfig.runner.max_iterations = 3000, 24 steps x 4096
# envs per iteration; usage: run_walk.sh [iterations] [torch|mfma] [pgs|compliant] ['extra trainer flags']; reference README.md:49 "A1 conditional walking ... 47.97 minutes") on the fused env, then the
# deterministic policy rolled for 500 steps (run_mode='play').  Checkpoints stay in /tmp (they exceed what gpurun copies
# back); the curve, the final model and the play reports land in gpurun_out/train_a1_r04/.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
ITERS=${1:-3000}
MLP=${2:-torch}        # torch: stock fp32 GEMMs; mfma: csrc/shf_mlp.hip layers + captured PPO update
SOLVER=${3:-tgs}       # contact solver: tgs (the reference's PhysX settings: solver_type = 1) | pgs | compliant (rounds 1-4)
EXTRA=${4:-}            # further tools/train_a1.py flags (e.g. --torch-loss)
TAG=$(echo "$EXTRA" | tr -d ' -')
OUT=$REPO/gpurun_out/train_a1_r06_${MLP}_$SOLVER${TAG:+_$TAG}
mkdir -p "$OUT" /tmp/train_a1
cd "$REPO"
python tools/train_a1.py --iters "$ITERS" --graph --quiet --mlp "$MLP" --solver "$SOLVER" $EXTRA --log /tmp/train_a1 > "$OUT/train_summary.json" 2> "$OUT/train.err"
cp /tmp/train_a1/progress.jsonl "$OUT/progress.jsonl"
cp /tmp/train_a1/model_"$ITERS".pt "$OUT/model_$ITERS.pt"
for T in heightfield flat trimesh; do
  python tools/play_a1.py "$OUT/model_$ITERS.pt" --envs 1024 --steps 500 --terrain $T --solver "$SOLVER" > "$OUT/play_$T.json" 2>> "$OUT/train.err"
done
cat "$OUT/train_summary.json" | cut -c1-600
cat "$OUT"/play_*.json
