#!/bin/bash
# On the GPU box (gpurun): the 3000-iteration A1 schedule for several seeds with one layer backend, final return / noise
# std / wall time per seed, and the deterministic policy played on the height field.
#   usage: run_seeds.sh mfma|torch seed [seed ...]        -> gpurun_out/seeds_<backend>/seed_<S>.{json,play.json}
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
MLP=$1; shift
OUT=$REPO/gpurun_out/seeds_$MLP${OUT_TAG:-}
mkdir -p "$OUT"
cd "$REPO"
for S in "$@"; do
  rm -rf /tmp/seed_run; mkdir -p /tmp/seed_run
  python tools/train_a1.py --iters 3000 --graph --quiet --mlp "$MLP" --seed "$S" --log /tmp/seed_run 2> "$OUT/seed_$S.err" | grep '^{' > "$OUT/seed_$S.json"
  python tools/play_a1.py /tmp/seed_run/model_3000.pt --envs 1024 --steps 500 --terrain heightfield 2>> "$OUT/seed_$S.err" | grep '^{' > "$OUT/seed_$S.play.json"
  python - "$OUT/seed_$S.json" "$OUT/seed_$S.play.json" "$S" <<'PY'
import json, sys
t = json.load(open(sys.argv[1])); p = json.load(open(sys.argv[2])); c = t["curve"][-1]
print(f"seed {sys.argv[3]}: return {c['mean_reward']:.0f} (std {c['std']:.2f}), {t['seconds']:.1f} s, learn {1e3 * t['mean_learn_s']:.1f} ms; "
      f"play {p['speed_along_cmd_over_cmd']:.2f} / {p['mean_lin_vel_error_m_s']:.2f}")
PY
done
