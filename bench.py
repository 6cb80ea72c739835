#!/usr/bin/env python3
"""Headline benchmark: env-steps/s of the vectorised A1 env step (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W [--workload terrain|flat] [--envs 4096]

One "step" = one ShifuVecEnv.step over all envs of this rank with random actions
`2*rand-1` (the reference's run_mode='random' driver, shifu/runner/policy_runner.py:33-41):
5 physics sub-steps (Q1), get_heights, termination, 6 reward terms, on-device resets,
259-dim observation -- all inside shf_a1_step -- plus the episode-stat reduction and,
for N>1, an RCCL all-gather of the (sum,count) episode statistics every 24 steps
(the only cross-rank traffic: envs shard with no data-path collective, "weak" scaling).

Rank 0 prints ONE JSON line.  `roofline.achieved` = B_alg x envs / mean duration of the
fused kernel, measured with HIP events on the launch stream inside the timed region;
`cpu_baseline` times the CPU oracle (a port, not the reference: Isaac Gym is not
installable) on a bounded sample of the same workload on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Algorithmic HBM bytes per env-step (restated in DESIGN.md section 5)
B_ALG = {"terrain": 5539, "flat": 5019, "trimesh": 5539}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s achievable)


def cpu_baseline(workload: str, seconds_budget: float = 15.0):
    """The oracle's fused A1 step on host cores, OpenMP over envs.  Bounded sample."""
    from oracle import pyoracle
    from shifu_amd import _abi
    from shifu_amd.a1_task import a1_task_params, height_points
    from shifu_amd.backend import default_sim_params
    from shifu_amd.gym.a1_fused import default_terrain_cfg
    from shifu_amd.model import asset_path, compile_urdf
    from shifu_amd.utils.terrain import Terrain
    pyoracle.build()
    n = 1024
    cm = compile_urdf(asset_path("a1.urdf"), default_dof_drive_mode=_abi.DOF_MODE_EFFORT)
    m = cm.blob
    sp = default_sim_params()
    ct = default_terrain_cfg()
    np.random.seed(42)
    if workload in ("terrain", "trimesh"):
        ter = Terrain(ct, n)
        hs, origins = np.ascontiguousarray(ter.heightsamples), ter.env_origins.astype(np.float32)
    else:
        hs = np.zeros((1300, 2100), np.int16)
        origins = np.zeros((ct.num_rows, ct.num_cols, 3), np.float32)
        for i in range(ct.num_rows):
            for j in range(ct.num_cols):
                origins[i, j] = [(i + 0.5) * 8, (j + 0.5) * 8, 0]
    terr = _abi.ShfTerrain()
    terr.rows, terr.cols, terr.hscale, terr.vscale, terr.border, terr.friction = hs.shape[0], hs.shape[1], 0.1, 0.005, 25.0, 1.0
    if workload == "trimesh":
        from shifu_amd.isaacgym.terrain_utils import pack_trimesh_samples, trimesh_warp_map
        terr.warped = 1
        hs = pack_trimesh_samples(hs, trimesh_warp_map(hs, 0.1, 0.005, ct.slope_treshold))
    tp = a1_task_params(cm)
    nb, nd, P = m.nb, m.nd, tp.num_height_points
    rng = np.random.default_rng(0)
    types = (np.arange(n) * ct.num_cols // n).astype(np.int64)
    b = dict(dof_state=np.zeros((n * nd, 2), np.float32), root_state=np.zeros((n, 13), np.float32),
             body_state=np.zeros((n * nb, 13), np.float32), contact=np.zeros((n * nb, 3), np.float32),
             friction=rng.uniform(0.5, 1.25, n).astype(np.float32), actions=np.zeros((n, nd), np.float32),
             obs=np.zeros((n, 12 + 5 * nd + P), np.float32), rew=np.zeros(n, np.float32), reset=np.zeros(n, np.uint8),
             timeout=np.zeros(n, np.uint8), ep_len=np.zeros(n, np.int64), command=np.zeros((n, 3), np.float32),
             history=np.zeros((n, nd, 3), np.float32), rew_sums=np.zeros((6, n), np.float32),
             torques=np.zeros((n, nd), np.float32), base_vel=np.zeros((n, 9), np.float32),
             heights=np.zeros((n, P), np.float32), hpoints=height_points(), push=np.zeros((n, nb, 3), np.float32),
             origins=np.ascontiguousarray(origins[0, types]), levels=np.zeros(n, np.int64), types=types,
             torigins=origins, reset_count=np.zeros(n, np.int32), done_sums=np.zeros((8, n), np.float32))
    b["dof_state"][:, 0] = np.tile(np.array([tp.default_dof_pos[d] for d in range(nd)], np.float32), n)
    b["root_state"][:, :3] = b["origins"] + np.array([0, 0, 0.42], np.float32)
    b["root_state"][:, 6] = 1.0
    def run(nt, k):
        t = time.perf_counter()
        for _ in range(k):
            raw = (2 * rng.random((n, nd)) - 1).astype(np.float32)
            pyoracle.a1_step(m, sp, tp, n, 0, b, raw, terrain=terr, heights=hs, nthreads=nt)
        return time.perf_counter() - t

    # os.cpu_count() can exceed what the container may use: pick the thread count that is fastest
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cands = sorted({c for c in (1, 4, 8, 16, 32, 64, 128, avail) if 1 <= c <= avail})
    best, best_t = 1, None
    for c in cands:
        run(c, 4)  # the first calls at a new team size pay thread start-up
        t = run(c, 3)
        if best_t is None or t < best_t:
            best, best_t = c, t
    cores = best
    steps, t0 = 0, time.perf_counter()
    while True:
        run(cores, 1)
        steps += 1
        el = time.perf_counter() - t0
        if el > seconds_budget or steps >= 2000:
            break
    return {"value": n * steps / el, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"{n} envs x {steps} vec-steps of the same A1 {workload} workload, oracle/shf_oracle.c (f32, "
                      f"OpenMP over envs); Isaac Gym CPU PhysX pipeline unavailable (not installable offline)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--envs", type=int, default=4096, help="envs per GPU")
    ap.add_argument("--workload", choices=["terrain", "flat", "trimesh"], default="terrain",
                    help="terrain = config 3 (height field); trimesh = the same samples as the mesh with vertical risers")
    ap.add_argument("--group", type=int, default=32, help="lanes per env: 64 = one wavefront per env, 32 = two envs per wavefront (fastest measured, DESIGN.md 6)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--decimation", type=int, default=4, help="(experiments only) control.decimation")
    ap.add_argument("--no-extra-substep", action="store_true", help="(experiments only) drop the Q1 sub-step")
    ap.add_argument("--log-interval", type=int, default=24, help="all-gather period (num_steps_per_env)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started bare: become the launcher (a child process per rank; nothing here has touched the GPU yet)
        import socket
        import subprocess
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the backend has no CPU fallback")
    # one rank per GPU; SHIFU_AMD_DIST_BACKEND=gloo (testing only) lets several ranks share a GPU to exercise the
    # N>1 control flow on a single-GPU box -- RCCL itself refuses two ranks on one device
    backend = os.environ.get("SHIFU_AMD_DIST_BACKEND", "nccl")
    local_rank = local_rank % torch.cuda.device_count() if backend != "nccl" else local_rank
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from shifu_amd import _abi
    from shifu_amd.gym.a1_fused import FusedA1Env
    from shifu_amd.parallel import gather_episode_stats

    env = FusedA1Env(num_envs=args.envs, device=dev, terrain={"terrain": "heightfield", "flat": "flat", "trimesh": "trimesh"}[args.workload],
                     seed=42, rank=rank, world_size=world, group=args.group, decimation=args.decimation,
                     extra_substep=not args.no_extra_substep)
    gen = torch.Generator(device=dev)
    gen.manual_seed(42 + rank)
    N, A = env.num_envs, env.num_actions
    env.reset()

    actions = torch.empty(N, A, device=dev)

    def vec_step(i, ev=None):
        actions.uniform_(-1.0, 1.0, generator=gen)   # = 2*rand-1 of policy_runner.py:40, one kernel, no allocation
        if ev is not None:
            ev[0].record()
        env.task.launch_step(actions)
        if ev is not None:
            ev[1].record()
        slot = env.task.launch_stats()
        if world > 1 and (i + 1) % args.log_interval == 0:
            gather_episode_stats(env.task.tensors[_abi.A1_STATS][slot][:8])
        return slot

    for i in range(args.warmup):
        vec_step(i)
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        vec_step(i, events[i])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kern_ms = sum(a.elapsed_time(b) for a, b in events) / args.steps

    t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    finite = bool(torch.isfinite(env.obs_buf).all().item())
    resets = int(env.task.tensors[_abi.A1_RESET_COUNT].sum().item())

    if rank == 0:
        total_envs = N * world
        value = total_envs * args.steps / elapsed
        b_alg = B_ALG[args.workload]
        wl_name = {"terrain": "procedural heightfield 1300x2100 (config 3)", "flat": "all-zero heightfield (config 2)",
                   "trimesh": "procedural terrain 1300x2100 as trimesh with vertical risers (the reference's effective A1 terrain)"}[args.workload]
        achieved = b_alg * N / (kern_ms * 1e-3) / 1e9
        traffic = None   # HBM bytes per launch from the committed rocprofv3 PMC passes (tools/profile.sh)
        try:
            tdb = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            key = "r01_a1_step_g%d" % args.group
            if args.workload == "terrain" and N == 4096 and key in tdb:
                traffic = tdb[key]["traffic_bytes"]
        except Exception:
            pass
        out = {
            "metric": "env-steps/sec (whole node), A1 12-dof 4096 envs/GPU", "value": value, "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"a1_conditional {wl_name}, "
                                   f"{N} envs/GPU, random actions, 5 substeps/env-step (dt 5 ms), resets on",
                       "envs_per_gpu": N, "total_envs": total_envs, "substeps_per_env_step": 5,
                       "lanes_per_env": args.group, "parallelism": f"env-sharded x{world}, all-gather of episode stats every {args.log_interval} steps",
                       "substeps_per_s": value * 5, "obs_finite": finite, "episodes_reset_rank0": resets},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "k_a1_step", "kernel_ms": kern_ms, "alg_bytes_per_env_step": b_alg,
                         "note": "latency/ALU-bound by design: ~5.5 KB compulsory traffic per env-step (DESIGN.md 5)"},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.workload)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
