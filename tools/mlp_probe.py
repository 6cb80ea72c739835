#!/usr/bin/env python3
"""Kernel-only timing of csrc/shf_mlp.hip through the C ABI (no autograd around it): forward, input gradient and weight
gradient of each A1 ActorCritic layer at the PPO mini-batch size, HIP-event time per call.

    python tools/mlp_probe.py [rows] [--bf16] [--tiled] [--check]   # product library (default operand precision bf16x3)
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

    tiled = "--tiled" in sys.argv          # the round-2 tiled GEMM for forward / input gradient instead of the row-panel kernels
    out = {"rows": M, "precision": precision, "kernels": "tiled" if tiled else "panel", "layers": []}
    for K, N, act in ((259, 512, 1), (512, 256, 1), (256, 128, 1), (128, 12, 0)):
        x, w, b = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda") * 0.05, torch.randn(N, device="cuda")
        y, g = torch.empty(M, N, device="cuda"), torch.randn(M, N, device="cuda")
        gx, gw, gb = torch.empty_like(x), torch.empty_like(w), torch.empty_like(b)
        n = C.c_int64()
        L.shf_mlp_backward_weight_workspace(M, K, N, C.byref(n))
        ws = torch.empty(n.value, device="cuda")
        yp = p(y) if act else None
        if tiled:
            f = timed(lambda: L.shf_mlp_linear_forward(p(x), p(w), p(b), p(y), M, K, N, act, st))
            bi = timed(lambda: L.shf_mlp_linear_backward_input(p(g), yp, p(w), p(gx), M, K, N, st))
            pk = 0.0
        else:
            nb = C.c_int64()
            L.shf_mlp_pack_bytes(K, N, C.byref(nb))
            pack = torch.empty(nb.value, device="cuda", dtype=torch.uint8)
            pk = timed(lambda: L.shf_mlp_pack_weights(p(w), p(pack), K, N, st))
            f = timed(lambda: L.shf_mlp_panel_forward(p(x), p(pack), p(b), p(y), M, K, N, act, st))
            bi = timed(lambda: L.shf_mlp_panel_backward_input(p(g), yp, p(pack), p(gx), M, K, N, st))
            if "--check" in sys.argv:       # the panel kernels against the tiled ones, bit for bit
                y2, gx2 = torch.empty_like(y), torch.empty_like(gx)
                L.shf_mlp_linear_forward(p(x), p(w), p(b), p(y2), M, K, N, act, st)
                L.shf_mlp_panel_forward(p(x), p(pack), p(b), p(y), M, K, N, act, st)
                yq = p(y2) if act else None
                L.shf_mlp_linear_backward_input(p(g), yq, p(w), p(gx2), M, K, N, st)
                L.shf_mlp_panel_backward_input(p(g), yq, p(pack), p(gx), M, K, N, st)
                torch.cuda.synchronize()
                print("check", K, N, "forward equal:", bool(torch.equal(y, y2)), "input gradient equal:", bool(torch.equal(gx, gx2)),
                      float((y - y2).abs().max()), float((gx - gx2).abs().max()), file=sys.stderr)
        bw = timed(lambda: L.shf_mlp_linear_backward_weight(p(g), yp, p(x), p(gw), p(gb), p(ws), M, K, N, st))
        byt_f = 4.0 * (M * K + N * K + M * N)
        out["layers"].append({"K": K, "N": N, "pack_us": round(pk, 1), "fwd_us": round(f, 1), "bwd_input_us": round(bi, 1), "bwd_weight_us": round(bw, 1),
                              "fwd_tflops": round(2.0 * M * K * N / f / 1e6, 1), "fwd_alg_GBps": round(byt_f / f / 1e3, 0)})
    if not tiled:
        # the four layers' forward as one chained launch (k_mlp_chain): every activation kept (a training pass) / only the output
        from shifu_amd.rl import mfma_linear as ML
        dims = (259, 512, 256, 128, 12)
        net = ML.MfmaMLP(*[m for i in range(4) for m in (ML.MfmaLinear(dims[i], dims[i + 1], elu=i < 3), torch.nn.Identity())][:-1]).cuda()
        ls = [m for m in net if isinstance(m, ML.MfmaLinear)]
        x = torch.randn(M, 259, device="cuda")
        ML.refresh_packs(net)
        ys = [torch.empty(M, d, device="cuda") for d in dims[1:]]
        out["chain_fwd_all_kept_us"] = round(timed(lambda: ML._chain_call(x, ls, [m._pack for m in ls], ys)), 1)
        out["chain_fwd_output_only_us"] = round(timed(lambda: ML._chain_call(x, ls, [m._pack for m in ls], [None, None, None, ys[-1]])), 1)
        out["layerwise_fwd_us"] = round(sum(l["fwd_us"] for l in out["layers"]), 1)
    out["total_us"] = round(sum(l["pack_us"] + l["fwd_us"] + l["bwd_input_us"] + l["bwd_weight_us"] for l in out["layers"]), 1)
    print(json.dumps(out))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "build":
        build(sys.argv[2], sys.argv[3:])
    else:
        main()
