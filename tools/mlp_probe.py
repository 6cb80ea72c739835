#!/usr/bin/env python3
"""Kernel-only timing of csrc/shf_mlp.hip through the C ABI (no autograd around it): forward, input gradient and weight
gradient of each A1 ActorCritic layer at the PPO mini-batch size, HIP-event time per call.

    python tools/mlp_probe.py [rows] [--bf16]        # product library (default operand precision bf16x3)
    SHIFU_AMD_LIB=... python tools/mlp_probe.py      # an experiment build (tools/mlp_probe.py build <name> <-Dflags...>)
"""
import ctypes as C
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build(name, flags):
    from shifu_amd import build as b
    out = os.path.join(ROOT, "shifu_amd", f"libshifu_amd_exp_{name}.so")
    b.compile_all(b.FLAGS, out, extra=list(flags))
    print(out)


def main():
    import torch
    from shifu_amd._lib import lib
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    M = int(args[0]) if args else 24576
    L = lib()
    precision = "bf16" if "--bf16" in sys.argv else "bf16x3"
    L.shf_mlp_set_precision(0 if precision == "bf16" else 1)
    p = lambda t: C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def timed(fn, iters=40):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / iters * 1e3

    out = {"rows": M, "precision": precision, "layers": []}
    for K, N, act in ((259, 512, 1), (512, 256, 1), (256, 128, 1), (128, 12, 0)):
        x, w, b = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda") * 0.05, torch.randn(N, device="cuda")
        y, g = torch.empty(M, N, device="cuda"), torch.randn(M, N, device="cuda")
        gx, gw, gb = torch.empty_like(x), torch.empty_like(w), torch.empty_like(b)
        n = C.c_int64()
        L.shf_mlp_backward_weight_workspace(M, K, N, C.byref(n))
        ws = torch.empty(n.value, device="cuda")
        yp = p(y) if act else None
        f = timed(lambda: L.shf_mlp_linear_forward(p(x), p(w), p(b), p(y), M, K, N, act, st))
        bi = timed(lambda: L.shf_mlp_linear_backward_input(p(g), yp, p(w), p(gx), M, K, N, st))
        bw = timed(lambda: L.shf_mlp_linear_backward_weight(p(g), yp, p(x), p(gw), p(gb), p(ws), M, K, N, st))
        byt_f = 4.0 * (M * K + N * K + M * N)
        out["layers"].append({"K": K, "N": N, "fwd_us": round(f, 1), "bwd_input_us": round(bi, 1), "bwd_weight_us": round(bw, 1),
                              "fwd_tflops": round(2.0 * M * K * N / f / 1e6, 1), "fwd_alg_GBps": round(byt_f / f / 1e3, 0)})
    out["total_us"] = round(sum(l["fwd_us"] + l["bwd_input_us"] + l["bwd_weight_us"] for l in out["layers"]), 1)
    print(json.dumps(out))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "build":
        build(sys.argv[2], sys.argv[3:])
    else:
        main()
