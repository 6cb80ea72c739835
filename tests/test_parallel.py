"""N>1 path on CPU: two gloo ranks shard the env range and all-gather (sum, count) episode
statistics exactly like bench.py / FusedA1Env do over RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from shifu_amd.parallel import gather_episode_stats, global_episode_means, shard_range

NAMES = ["tracking_lin_vel", "tracking_ang_vel", "stabilizing_base", "smoothing_action", "leg_collision",
         "torques_penalize"]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total_envs, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = shard_range(total_envs, rank, world)
        rng = np.random.default_rng(0)                       # every rank draws the GLOBAL arrays ...
        sums = rng.uniform(-50, 50, (6, total_envs)).astype(np.float32)
        done = rng.random(total_envs) < (0.1 if rank == 0 else 0.6)   # ... but finishes unevenly
        done_all = np.concatenate([(np.random.default_rng(0).random(total_envs) < 0.1)[:total_envs // 2],
                                   (np.random.default_rng(0).random(total_envs) < 0.6)[total_envs // 2:]])
        levels = rng.integers(0, 10, total_envs).astype(np.float32)
        local = np.zeros(8, np.float32)
        local[:6] = (sums[:, lo:hi] * done_all[lo:hi]).sum(1)
        local[6] = levels[lo:hi].sum()
        local[7] = done_all[lo:hi].sum()
        gathered = gather_episode_stats(torch.from_numpy(local))
        ep = global_episode_means(gathered, NAMES, 10.0, hi - lo)
        if rank == 0:
            ref = {n: float((sums[k] * done_all).sum() / max(done_all.sum(), 1) / 10.0) for k, n in enumerate(NAMES)}
            ref["terrain_levels"] = float(levels.mean())
            q.put((gathered.shape, {k: float(v) for k, v in ep.items()}, ref, float(local[7])))
    finally:
        dist.destroy_process_group()


def test_two_rank_gather_of_episode_stats():
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 64, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    shape, ep, ref, cnt0 = q.get()
    assert tuple(shape) == (2, 8)
    for k in ref:   # mean over ALL finished episodes, not a mean of per-rank means
        assert ep[k] == pytest.approx(ref[k], rel=1e-5, abs=1e-6), k


def test_shard_range_and_single_process_gather():
    assert shard_range(32768, 3, 8) == (12288, 16384)
    with pytest.raises(ValueError):
        shard_range(10, 0, 3)
    x = torch.arange(8, dtype=torch.float32)
    assert gather_episode_stats(x).shape == (1, 8)          # no process group: identity


def test_global_types_follow_global_env_ids():
    """terrain_types = floor(i / (N_total / num_cols)) must be computed on global ids so a shard
    sees the columns the unsharded run would give it (isaac_gym.py:342-344)."""
    total, cols = 32768, 20
    full = torch.div(torch.arange(total), total / cols, rounding_mode='floor').long()
    lo, hi = shard_range(total, 5, 8)
    shard = torch.div(torch.arange(lo, hi), total / cols, rounding_mode='floor').long()
    assert torch.equal(full[lo:hi], shard)


def test_launch_ranks_owns_its_rendezvous_port(tmp_path):
    """shifu_amd.parallel.launch_ranks (what `bench.py --gpus N` and tools/train_a1.py become when started bare): two gloo ranks
    through torch.distributed.run --standalone -- the launcher binds the rendezvous port itself, so two launches at once cannot
    be handed the same "free" port -- and a rank that exits non-zero fails the job."""
    import subprocess
    import sys
    script = tmp_path / "ranks.py"
    script.write_text(
        "import os, sys\n"
        "import torch.distributed as dist\n"
        "if len(sys.argv) > 1 and os.environ['RANK'] == sys.argv[1]:\n"
        "    raise SystemExit(3)\n"
        "dist.init_process_group('gloo')\n"
        "assert dist.get_world_size() == 2 and os.environ['MASTER_ADDR'] == '127.0.0.1'\n"
        "dist.barrier()\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = "import sys; from shifu_amd.parallel import launch_ranks; raise SystemExit(launch_ranks(sys.argv[1], sys.argv[2:], 2))"
    two = [subprocess.Popen([sys.executable, "-c", code, str(script)], cwd=root, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
           for _ in range(2)]                                     # two launches side by side
    assert [p.wait(timeout=240) for p in two] == [0, 0]
    bad = subprocess.run([sys.executable, "-c", code, str(script), "1"], cwd=root, capture_output=True, timeout=240)
    assert bad.returncode != 0
