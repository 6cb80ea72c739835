#!/usr/bin/env python3
"""Per-layer timing of the trainer's MFMA kernels against the stock fp32 library GEMM path (torch -> hipBLASLt), on the
A1 ActorCritic's shapes at the PPO mini-batch size (24 576 rows).  Prints one JSON object: per op microseconds,
TFLOP/s, and the fraction of the dense bf16 MFMA peak (2.5 PFLOP/s, MI355X_MICROARCH.md)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

PEAK_BF16 = 2500.0


def timed(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3   # us


def main():
    from shifu_amd.rl.mfma_linear import MfmaLinear
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 24576
    out = {"rows": M, "layers": []}
    tot = {"mfma": 0.0, "torch": 0.0}
    for K, N, elu in ((259, 512, True), (512, 256, True), (256, 128, True), (128, 12, False)):
        x = torch.randn(M, K, device="cuda", requires_grad=True)
        g = torch.randn(M, N, device="cuda")
        row = {"K": K, "N": N}
        for name, lin in (("mfma", MfmaLinear(K, N, elu=elu).cuda()), ("torch", torch.nn.Linear(K, N).cuda())):
            act = (lambda t: t) if (name == "mfma" or not elu) else torch.nn.functional.elu
            fwd = timed(lambda: act(lin(x)))
            y = act(lin(x))
            def bwd():
                x.grad = None
                lin.zero_grad(set_to_none=True)
                y.backward(g, retain_graph=True)
            b = timed(bwd)
            flops_f, flops_b = 2.0 * M * K * N, 4.0 * M * K * N
            row[name] = {"fwd_us": fwd, "bwd_us": b, "fwd_tflops": flops_f / fwd / 1e6, "bwd_tflops": flops_b / b / 1e6}
            tot[name] += fwd + b
        row["mfma"]["fwd_frac_of_bf16_peak"] = row["mfma"]["fwd_tflops"] / PEAK_BF16
        out["layers"].append(row)
    out["fwd_plus_bwd_us_all_layers"] = tot
    print(json.dumps(out))


if __name__ == "__main__":
    main()
