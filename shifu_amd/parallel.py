"""Multi-GPU: envs shard trivially, one process per GPU; the only exchange is the
episode-statistics all-gather.

The reference is single-process (device 'cuda:0' hard-coded, env_config.py:15); its
logging computes mean(sum[env_ids]) / T per reward term (shifu/gym/env.py:149-158).
Across shards the *sums and counts* are gathered -- a mean of per-rank means would be
wrong for unequal reset counts -- and the global mean is formed from them.  The payload
is 8 floats per rank: latency-bound on xGMI, no bucketing or overlap to speak of.
Backend "nccl" is RCCL on ROCm; the same code runs on gloo for the CPU tests.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.distributed as dist


def shard_range(total_envs: int, rank: int, world_size: int):
    """Contiguous global env-id slice owned by `rank` (equal shards)."""
    if total_envs % world_size:
        raise ValueError("total_envs must be divisible by world_size")
    n = total_envs // world_size
    return rank * n, (rank + 1) * n


def gather_episode_stats(local: torch.Tensor, group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """all_gather of a small fixed-size stats vector -> (world, len) tensor.
    Layout of `local` for the A1 task: [6 reward-term sums, level sum, finished count]."""
    if not (dist.is_available() and dist.is_initialized()):
        return local.unsqueeze(0)
    world = dist.get_world_size(group)
    parts = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(parts, local.contiguous(), group=group)
    return torch.stack(parts)


def global_episode_means(gathered: torch.Tensor, names: List[str], max_episode_length_s: float,
                         envs_per_rank: int) -> Dict[str, torch.Tensor]:
    """extras["episode"] for the whole job from per-rank (sums..., level_sum, count)."""
    tot = gathered.sum(0)
    cnt = tot[len(names) + 1].clamp(min=1.0)
    ep = {n: tot[k] / cnt / max_episode_length_s for k, n in enumerate(names)}
    ep["terrain_levels"] = tot[len(names)] / float(envs_per_rank * gathered.shape[0])
    return ep


# ---- trainer side: data-parallel PPO over env shards ----
def world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def collectives_on() -> bool:
    """True when the data-parallel collectives run: more than one rank -- or one rank with SHIFU_AMD_FORCE_DIST=1
    (testing only: walks the RCCL code path on a single-GPU box)."""
    import os
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get("SHIFU_AMD_FORCE_DIST", "0") == "1"


def average_(t: torch.Tensor) -> torch.Tensor:
    """In-place mean over ranks (identity in a single process)."""
    if collectives_on():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        t /= world_size()
    return t


def average_gradients(params, bucket_bytes: int = 64 << 20) -> None:
    """Mean of .grad over ranks with as few collectives as the bucket size allows.  The A1 actor-critic is
    0.6 M parameters (2.5 MB): one all-reduce per mini-batch -- on xGMI's per-link-bound ring a single large
    message beats one per layer."""
    grads = [p.grad for p in params if p.grad is not None]
    if not grads or not collectives_on():
        return
    bucket, size = [], 0
    def flush():
        if not bucket:
            return
        flat = torch.cat([g.reshape(-1) for g in bucket])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat /= world_size()
        off = 0
        for g in bucket:
            g.copy_(flat[off:off + g.numel()].view_as(g))
            off += g.numel()
    for g in grads:
        if size + g.numel() * g.element_size() > bucket_bytes and bucket:
            flush()
            bucket, size = [], 0
        bucket.append(g)
        size += g.numel() * g.element_size()
    flush()


def broadcast_parameters(module: torch.nn.Module, src: int = 0) -> None:
    """Same initial weights on every rank."""
    if not collectives_on():
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src)


# ---- launching one process per GPU (bench.py --gpus N, tools/train_a1.py --gpus N) ----
def launch_ranks(script: str, argv: List[str], nproc: int) -> int:
    """Start `script` once per GPU through torch.distributed.run (--standalone: rendezvous on 127.0.0.1, a port the launcher binds itself) as CHILD processes and
    return the launcher's exit code: torchrun tears the other ranks down when one fails and exits non-zero, so a rank that
    dies takes the job's exit status with it instead of leaving the others waiting in a collective.  Call this before
    anything in the calling process has touched the GPU (the children initialise it themselves; a process that has
    initialised the GPU must never exec another program on this pool)."""
    import subprocess
    import sys
    # --standalone: the launcher binds its own rendezvous port (no pick-a-free-port-then-hand-it-over race between two launches
    # on one node); --local-addr: the container's hostname may not resolve
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           f"--nproc-per-node={nproc}", script] + list(argv)
    return subprocess.call(cmd)


def init_ranks(device: torch.device, backend: str = "nccl", timeout_s: float = 300.0) -> None:
    """init_process_group with a finite timeout: a rank that never arrives makes the others fail (non-zero exit through the
    launcher) after `timeout_s` instead of hanging the node.  SHIFU_AMD_DIST_TIMEOUT_S overrides it."""
    import datetime
    import os
    t = datetime.timedelta(seconds=float(os.environ.get("SHIFU_AMD_DIST_TIMEOUT_S", timeout_s)))
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=device, timeout=t)
    else:
        dist.init_process_group(backend, timeout=t)


def device_identity(device: torch.device) -> str:
    """What tells one GPU of a node from another in a log line: the PCI bus id where torch exposes it, else the UUID, else
    the ordinal."""
    p = torch.cuda.get_device_properties(device)
    for attr in ("pci_bus_id", "uuid"):
        v = getattr(p, attr, None)
        if v is not None:
            if attr == "pci_bus_id":
                return "pci %04x:%02x:%02x.0" % (int(getattr(p, "pci_domain_id", 0)), int(v), int(getattr(p, "pci_device_id", 0)))
            return f"uuid {v}"
    return f"cuda:{device.index}"


def gather_rank_reports(report: dict) -> List[dict]:
    """Every rank's small report dict on every rank, rank order (all_gather_object; a single process: [report])."""
    if not (dist.is_available() and dist.is_initialized()):
        return [report]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, report)
    return out
