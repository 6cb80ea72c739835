#!/usr/bin/env python3
"""Small-graph probes for the back-to-back replay fault of the captured PPO update.

tools/hipgraph_order_probe.py showed that kernel-only graphs (linear chain, fork/join) are ordered correctly against
work queued behind the launch.  The captured update also contains what torch turns into MEMCPY nodes: `clone()` /
`contiguous()` / `copy_` between contiguous same-dtype device tensors go through hipMemcpyAsync, which capture records as
a D2D memcpy node, not a kernel.  This probe builds <= 6-node graphs around such nodes and replays them back to back with
an eager writer of the graph's input in between -- any replay whose result does not match its own input is a violation.

    python tools/hipgraph_memcpy_probe.py [reps=400] [n=1<<22]
"""
import json
import sys

import torch


def probe(kind, reps, n, wait):
    dev = "cuda:0"
    src = torch.zeros(n, device=dev)
    a = torch.zeros(n, device=dev)
    b = torch.zeros(n, device=dev)
    c = torch.zeros(n, device=dev)
    out = torch.zeros(reps, 64, device=dev)
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        if kind == "kernel-memcpy-kernel":          # k: a = src + 1 ; m: b <- a ; k: c = b * 2
            torch.add(src, 1.0, out=a)
            b.copy_(a)                               # contiguous same dtype: hipMemcpyAsync -> memcpy node
            torch.mul(b, 2.0, out=c)
        elif kind == "kernel-memcpy(leaf)":          # k: a = src + 1 ; m: c <- a  (the memcpy is the graph's last node)
            torch.add(src, 1.0, out=a)
            c.copy_(a)
        elif kind == "memcpy(root)-kernel":          # m: a <- src ; k: c = a + 1   (the memcpy is the graph's first node)
            a.copy_(src)
            torch.add(a, 1.0, out=c)
        elif kind == "memset-kernel":                # memset node, then kernels
            a.zero_()
            a.add_(src)
            torch.add(a, 1.0, out=c)
        elif kind == "kernels-only":
            torch.add(src, 1.0, out=a)
            torch.mul(a, 1.0, out=b)
            torch.add(b, 0.0, out=c)
        else:
            raise ValueError(kind)
    want = {"kernel-memcpy-kernel": lambda r: 2.0 * (r + 1.0), "kernel-memcpy(leaf)": lambda r: r + 1.0,
            "memcpy(root)-kernel": lambda r: r + 1.0, "memset-kernel": lambda r: r + 1.0, "kernels-only": lambda r: r + 1.0}[kind]
    for r in range(reps):
        src.fill_(float(r))                          # eager write of the graph's input, right behind the previous replay
        g.replay()
        if wait:
            torch.cuda.synchronize()
        out[r].copy_(c[:: n // 64][:64])             # eager read of the graph's output, right behind this replay
    torch.cuda.synchronize()
    exp = torch.tensor([want(float(r)) for r in range(reps)], device=dev).unsqueeze(1)
    return int((out != exp).any(dim=1).sum())


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 22
    for kind in ("kernels-only", "kernel-memcpy-kernel", "kernel-memcpy(leaf)", "memcpy(root)-kernel", "memset-kernel"):
        rec = {"graph": kind, "replays": reps, "floats": n,
               "wrong_back_to_back": probe(kind, reps, n, False), "wrong_with_host_wait": probe(kind, reps, n, True)}
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
