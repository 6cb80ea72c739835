#!/usr/bin/env python3
"""Is work queued behind a hipGraph launch ordered after the graph's last node?  (rl/ppo.py waits on the host after
every replay of the captured update because, without the wait, training drifted from the eager result.)  Result on
ROCm 7.2 / MI355X: yes -- linear chains and fork-join graphs replayed back to back show no violation, so whatever breaks
the back-to-back replay of the captured PPO update is not this simple.

A graph of `depth` dependent in-place additions on a buffer is replayed, and an eager copy of the buffer is queued on the
same stream right behind it -- with an eager write to the graph's *input* queued right behind that (what the trainer
does: the next mini-batch's indices).  Every copy must hold the full sum; violations are counted."""
import sys
import torch


def main():
    depth = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    n = 1 << 20
    dev = "cuda:0"
    src = torch.zeros(n, device=dev)
    acc = torch.zeros(n, device=dev)
    out = torch.zeros(reps, n // 4096, device=dev)
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    fork = len(sys.argv) > 3 and sys.argv[3] == "fork"
    side = torch.cuda.Stream()
    acc2 = torch.zeros(n, device=dev)
    with torch.cuda.graph(g):
        acc.copy_(src)
        if fork:
            # a fork / join inside the graph, as the autograd engine makes when a backward runs on its own stream
            cur = torch.cuda.current_stream()
            acc2.copy_(src)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                for _ in range(depth):
                    acc2.add_(0.5)
            for _ in range(depth):
                acc.add_(0.5)
            cur.wait_stream(side)
            acc.add_(acc2).sub_(src)
        else:
            for _ in range(depth):
                acc.add_(1.0)
    bad = {}
    for mode in ("back-to-back", "host wait after replay"):
        for r in range(reps):
            src.fill_(float(r))                      # eager write of the graph's input
            g.replay()
            if mode != "back-to-back":
                torch.cuda.synchronize()
            out[r].copy_(acc[::4096])                # eager read of the graph's output, queued right behind the launch
        torch.cuda.synchronize()
        want = torch.arange(reps, device=dev, dtype=torch.float32).unsqueeze(1) + depth
        bad[mode] = int((out != want).any(dim=1).sum())
    print({"fork_join": fork, "depth": depth, "replays": reps, "replays_with_a_wrong_copy": bad})


if __name__ == "__main__":
    main()
