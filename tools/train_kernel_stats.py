#!/usr/bin/env python3
"""GPU time per PPO iteration by kernel family, from `rocprofv3 --kernel-trace --stats` of a short training run.

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_train -o t --output-format csv -- \
        python3 $REPO/tools/train_a1.py --iters 30 --graph --mlp mfma --seed 1 --quiet --log /tmp/pt
    python tools/train_kernel_stats.py gpurun_out/prof_train/t_kernel_stats.csv 30 > profiles/rNN_train_kernel_stats.md
"""
import csv
import sys

FAMILIES = [("k_a1_chain", "env step"), ("k_mlp_gemm", "k_mlp_gemm (weight gradient)"), ("k_mlp_panel", "k_mlp_panel (forward / input gradient)"),
            ("k_mlp_chain", "k_mlp_chain (rollout inference)"), ("k_mlp_reduce", "k_mlp_reduce_slices"), ("k_mlp_pack", "k_mlp_pack"),
            ("k_ppo_loss", "k_ppo_loss* (one-pass loss + gradient)"), ("multi_tensor", "Adam / grad norm (multi-tensor)"),
            ("FusedAdam", "Adam / grad norm (multi-tensor)"), ("k_", "other repo kernels (gather, gae, copies, bookkeeping, lr)"),
            ("copyBuffer", "copies / fills"), ("fillBuffer", "copies / fills"), ("FillFunctor", "copies / fills")]


def main(path, iters):
    fam = {}
    for r in csv.DictReader(open(path)):
        name, calls, total = r["Name"], int(r["Calls"]), float(r["TotalDurationNs"])
        label = next((lab for key, lab in FAMILIES if key in name), "small torch kernels")
        c, t = fam.get(label, (0, 0.0))
        fam[label] = (c + calls, t + total)
    tot = sum(t for _, t in fam.values())
    print("| kernels | launches / iteration | ms / iteration | % |\n|---|---:|---:|---:|")
    for lab, (c, t) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        print(f"| {lab} | {c / iters:.0f} | {t / iters / 1e6:.2f} | {100 * t / tot:.1f} |")
    print(f"| **sum** | {sum(c for c, _ in fam.values()) / iters:.0f} | {tot / iters / 1e6:.2f} | |")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]))
