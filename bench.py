#!/usr/bin/env python3
"""Headline benchmark: env-steps/s of the vectorised env step (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W [--workload terrain|flat|trimesh|abb] [--envs 4096]

One "step" = one ShifuVecEnv.step over all envs of this rank with random actions
`2*rand-1` (the reference's run_mode='random' driver, shifu/runner/policy_runner.py:33-41).
A1 workloads (configs 2-4): 5 physics sub-steps (Q1), get_heights, termination, 6 reward terms,
on-device resets, 259-dim observation and the episode-statistics reduction -- all inside
shf_a1_step.  `abb` (config 5, examples/abb_pushbox_vision/a_prior_stage.py:67-135): in-kernel
IK, 6 sub-steps of 20 ms with box contacts, refresh, termination, rewards, re-spawn, observation
inside shf_abb_step.  For N>1 an RCCL all-gather of the (sum,count) episode statistics every 24
steps is the only cross-rank traffic: envs shard with no data-path collective ("weak" scaling).

Rank 0 prints ONE JSON line.  `roofline.achieved` = B_alg x envs / mean duration of the fused
kernel, measured with HIP events on the launch stream inside the timed region; `roofline.secondary`
prices the same launch against the fp32 vector peak with the algorithm's counted flops and states
the occupancy; `cpu_baseline` times the CPU oracle (a port, not the reference: Isaac Gym is not
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
B_ALG = {"terrain": 5539, "flat": 5019, "trimesh": 5539, "abb": 2140}
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_ACHIEVABLE_GBS = 6290.0   # MI355X_MICROARCH.md: 6.29 TB/s measured (float4 copy) -- the figure BASELINE.md section 2 quotes fractions of
VALU_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32 vector (non-matrix) peak
NUM_CUS, SIMDS_PER_CU = 256, 4
MAX_CLOCK_HZ = 2.4e9       # MI355X_MICROARCH.md: max clock
# tools/valu_microbench (profiles/r03_valu_microbench.md), measured on the MI355X: a SIMD issues one wave64 fp32 VALU
# instruction per ~2 clocks when two or more waves feed it (2.30 at two waves, 1.93 at four; the guide's "2 cyc (SIMD-32)"),
# whatever the number of active lanes; ONE wave alone gets one issue slot per 4.57 clocks, dependent or not.
VALU_ISSUE_CLOCKS_SIMD = 2.0
VALU_ISSUE_CLOCKS_ONE_WAVE = 4.57
WL_NAME = {"terrain": "a1_conditional procedural heightfield 1300x2100 (config 3)",
           "flat": "a1_conditional all-zero heightfield (config 2)",
           "trimesh": "a1_conditional procedural terrain 1300x2100 as trimesh with vertical risers (the reference's effective A1 terrain)",
           "abb": "abb_pushbox prior stage: ABB arm + table + free cube + goal pad (config 5)"}


# ----------------------------------------------------------------------------- CPU oracle workloads --
def oracle_workload(workload: str, n: int, seed: int = 0, solver_kw=None, link_contacts: bool = True, link_shapes: str = "box"):
    """The same workload on NumPy buffers for the CPU oracle: returns step(nthreads, count_flops=False) -> None."""
    from oracle import pyoracle
    from shifu_amd import _abi
    from shifu_amd.backend import default_sim_params
    pyoracle.build()
    rng = np.random.default_rng(seed)
    if workload == "abb":
        from shifu_amd.abb_task import ABB_BASE_POS, abb_boxes, abb_model, abb_task_params
        cm = abb_model(link_contacts=link_contacts, link_shapes=link_shapes)      # the scene the GPU leg runs
        m, boxes = cm.blob, abb_boxes()
        flags = _abi.SCENE_FACE_MANIFOLD if (link_contacts and link_shapes == "hull") else 0
        sp = default_sim_params(dt=0.02, **(solver_kw or {}))
        tp = abb_task_params(cm)
        nb, nd, A = m.nb, m.nd, 4
        B = nb + 3
        root = np.zeros((n * A, 13), np.float32)
        root[:, 6] = 1.0
        root[0::A, :3] = ABB_BASE_POS
        for k, b in enumerate(boxes):
            root[1 + k::A, :3] = list(b.pos)
        bufs = dict(dof_state=np.zeros((n * nd, 2), np.float32), root_state=root,
                    body_state=np.zeros((n * B, 13), np.float32), contact=np.zeros((n * B, 3), np.float32),
                    jacobian=np.zeros((n, nb - 1, 6, nd), np.float32), friction=np.ones(n, np.float32),
                    actions=np.zeros((n, 3), np.float32), obs=np.zeros((n, 6), np.float32), rew=np.zeros(n, np.float32),
                    reset=np.zeros(n, np.uint8), timeout=np.zeros(n, np.uint8), success=np.zeros(n, np.uint8),
                    ep_len=np.zeros(n, np.int64), rew_sums=np.zeros((2, n), np.float32),
                    dof_targets=np.zeros((n, nd), np.float32), reset_count=np.zeros(n, np.int32),
                    done_sums=np.zeros((4, n), np.float32))
        bufs["dof_state"][:, 0] = np.tile(np.array([tp.default_dof_pos[d] for d in range(nd)], np.float32), n)
        bufs["body_state"][:, 6] = 1.0

        def step(nthreads, count_flops=False):
            raw = (2 * rng.random((n, 3)) - 1).astype(np.float32)
            with pyoracle.scene_extras(hulls=cm.hulls, flags=flags):
                pyoracle.abb_step(m, sp, boxes, tp, n, 0, bufs, raw, nthreads=nthreads, count_flops=count_flops)
        return step
    from shifu_amd.a1_task import a1_task_params, height_points
    from shifu_amd.gym.a1_fused import default_terrain_cfg
    from shifu_amd.model import asset_path, compile_urdf
    from shifu_amd.utils.terrain import Terrain
    cm = compile_urdf(asset_path("a1.urdf"), default_dof_drive_mode=_abi.DOF_MODE_EFFORT)
    m = cm.blob
    for d in range(m.nd):
        m.damping[d] = 0.5                       # dof_props['damping'] as FusedA1Env sets it
    sp = default_sim_params(**(solver_kw or {}))
    ct = default_terrain_cfg()
    if workload in ("terrain", "trimesh"):
        state = np.random.get_state()
        np.random.seed(42)
        ter = Terrain(ct, n)
        np.random.set_state(state)
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

    def step(nthreads, count_flops=False):
        raw = (2 * rng.random((n, nd)) - 1).astype(np.float32)
        pyoracle.a1_step(m, sp, tp, n, 0, b, raw, terrain=terr, heights=hs, nthreads=nthreads, count_flops=count_flops)
    return step


def count_flops(workload: str, n: int = 64, warm: int = 40, steps: int = 20, solver_kw=None, **scene) -> float:
    """Floating-point operations per env-step of the algorithm (add/sub/mul/div/sqrt = 1, fma = 2), counted by running
    the oracle compiled with a counting real type (oracle/flopcount.cpp) on `n` envs in the workload's steady state
    (robots standing / stumbling on the terrain with random actions, resets included)."""
    from oracle import pyoracle
    if scene.get("link_shapes") == "hull":
        raise RuntimeError("the counting build has no hull entry points")
    step = oracle_workload(workload, n, seed=1, solver_kw=solver_kw, **scene)
    for _ in range(warm):
        step(1)
    L = pyoracle.flop_lib()
    L.shf_flopcount_read(1)
    for _ in range(steps):
        step(1, count_flops=True)
    return L.shf_flopcount_read(1) / float(n * steps)


def usable_cores() -> int:
    """Cores this process can actually run on: the affinity mask, capped by the cgroup CPU quota (a container that sees
    256 logical CPUs but is throttled to a few would otherwise be timed with 256 threads fighting over them)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]          # cgroup v2
        if q != "max":
            n = min(n, max(1, int(float(q) / float(per) + 0.5)))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())       # cgroup v1
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, int(q / per + 0.5)))
        except Exception:
            pass
    return n


def cpu_baseline(workload: str, seconds_budget: float = 16.0, solver_kw=None, **scene):
    """The oracle's fused env step on host cores.  Deterministic thread counts: one thread, and every core this process
    may use (OpenMP over envs, 4096 envs so that each thread has tens of envs per step).  Bounded sample."""
    avail = usable_cores()

    def timed(n, threads, budget):
        step = oracle_workload(workload, n, solver_kw=solver_kw, **scene)
        for _ in range(3):
            step(threads)          # thread start-up, page faults
        k, t0 = 0, time.perf_counter()
        while True:
            step(threads)
            k += 1
            el = time.perf_counter() - t0
            if el > budget or k >= 2000:
                return n * k / el, k
    v1, k1 = timed(256, 1, seconds_budget * 0.25)
    # every usable core, and half / a quarter of them (SMT siblings and memory-bound phases can make fewer threads
    # faster): the same three counts on every box, the best one reported, all three listed
    tried = {}
    for c in sorted({avail, max(1, avail // 2), max(1, avail // 4)}, reverse=True):
        tried[c] = timed(4096, c, seconds_budget * 0.25)
    cores = max(tried, key=lambda c: tried[c][0])
    vall, kall = tried[cores]
    what = "ABB push-box" if workload == "abb" else f"A1 {workload}"
    try:                          # the same oracle, built with a counting real type: flops per env-step of the algorithm
        flops, flops_src = count_flops(workload, solver_kw=solver_kw, **scene), "counted in the cpu_baseline leg: oracle/flopcount.cpp (add/sub/mul/div/sqrt = 1, fma = 2)"
    except Exception as e:        # the counting build needs g++ on the box
        flops, flops_src = None, f"unavailable: {e}"
    return {"value": vall, "flops_per_env_step": flops, "flops_source": flops_src, "unit": "env-steps/s", "cores": cores, "kind": "port", "value_1thread": v1,
            "threads_tried": {str(c): v[0] for c, v in tried.items()},
            "sample": f"oracle/shf_oracle.c (f32) on the same {what} workload: 4096 envs x {kall} vec-steps on {cores} threads "
                      f"(OpenMP over envs; best of {sorted(tried)} threads, {avail} usable cores) = `value`; 256 envs x {k1} "
                      f"vec-steps on 1 thread = `value_1thread`.  kind=port because the reference's CPU pipeline (Isaac Gym "
                      f"sim_device=cpu, use_gpu_pipeline=False) is a closed binary that cannot be installed offline"}


def committed_profile(kernel_key: str):
    """Counter-derived facts that bench.py cannot measure in-run (rocprofv3 --pmc needs its own passes): replayed from
    the committed summaries under profiles/ and labelled as such."""
    try:
        db = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        return db.get(kernel_key)
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--test-fail-rank", type=int, default=None, help=argparse.SUPPRESS)     # tests only: this rank exits with status 3
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--envs", type=int, default=4096, help="envs per GPU")
    ap.add_argument("--workload", choices=["terrain", "flat", "trimesh", "abb"], default="terrain",
                    help="terrain = config 3 (height field); trimesh = the same samples as the mesh with vertical risers; "
                         "abb = config 5")
    ap.add_argument("--group", type=int, default=None, help="lanes per env: 64 = one wavefront per env; default 32 for the A1 workloads, 16 for abb: the fastest measured (DESIGN.md 6)")
    ap.add_argument("--actions", choices=["kernel", "torch"], default="kernel",
                    help="random actions drawn inside the fused launch (default) or by a torch uniform_ launch before it")
    ap.add_argument("--mapping", choices=["chain", "body", "split"], default=None,
                    help="A1 workloads: lane = kinematic chain (default, with --self-collision too at 32 lanes per env; csrc/shf_chain.h) "
                         "or lane = rigid body (the general kernels).  Kernel selection only: results are bit-identical")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--solver", choices=["pgs", "tgs", "compliant"], default=None,
                    help="contact solver (ShfSimParams.solver): pgs = the velocity-level projected Gauss-Seidel solve with the reference's PhysX "
                         "settings (env_config.py:50-58: 8 + 1 iterations), the default of the A1 workloads (without --self-collision, chain "
                         "mapping at 32 lanes); compliant = rounds 1-4's spring-damper law (config 5 / abb: the only one built)")
    ap.add_argument("--no-other-solver", action="store_true",
                    help="skip the `other_solver` leg (N = 1, --solver not given: the same workload under the solver that is NOT the "
                         "default, timed the same way after the main measurement and reported beside it)")
    ap.add_argument("--pos-iters", type=int, default=8, help="physx.num_position_iterations (pgs)")
    ap.add_argument("--vel-iters", type=int, default=1, help="physx.num_velocity_iterations (pgs)")
    ap.add_argument("--max-contacts", type=int, default=None,
                    help="ShfSimParams.max_contacts (pgs): constraints the solve holds per env and sub-step, the deepest candidates -- 8 (default) "
                         "on k_a1_chain_pgs, up to 16 on k_a1_chain_pgs16 (A1 workloads)")
    ap.add_argument("--self-collision", action="store_true",
                    help="A1 workloads: collide the robot's own links (capsule pairs; the reference's collision filter 0, "
                         "units.py:68) -- off in the headline configuration, whose BASELINE entry names height-field contact")
    ap.add_argument("--no-link-contacts", action="store_true",
                    help="abb workload: the rod against the cube only (the scene benchmarked in rounds 1-3) instead of the "
                         "reference's scene with every link colliding; implied by --mapping chain / split, which are compiled for it")
    ap.add_argument("--link-contacts", action="store_true",
                    help="abb workload: the arm's links (box stand-ins for their mesh colliders) and the rod also collide with the "
                         "table, the cube and the goal pad (SURVEY 8f f3, ShfModel.link_collide) -- the default for this workload since round 4 (the flag is kept for old command lines)")
    ap.add_argument("--link-shapes", choices=["box", "hull"], default="box",
                    help="abb workload: the arm's links as the bounding boxes of their mesh colliders (default: the kernels compiled for this scene) or "
                         "as the convex hulls of the reference's collision meshes, reduced to <= 32 vertices (abb_rod_isaac.urdf:38-113), through the "
                         "convex narrow phase with the clipped face manifold on (csrc/shf_hull.h; run-time-shaped kernels)")
    ap.add_argument("--graph", action="store_true", help="(experiments: slower, and back-to-back graph replays are not trustworthy on this stack, profiles/r02_mlp_probe.md) replay the vec-step from a captured hipGraph instead of launching it "
                    "eagerly (measured slower on ROCm 7.2: 84.7 vs 73.9 us per vec-step, profiles/r02_bench_*.json)")
    ap.add_argument("--decimation", type=int, default=4, help="(experiments only) control.decimation")
    ap.add_argument("--no-extra-substep", action="store_true", help="(experiments only) drop the Q1 sub-step")
    ap.add_argument("--log-interval", type=int, default=24, help="all-gather period (num_steps_per_env)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started bare: become the launcher (a child process per rank; nothing here has touched the GPU yet).  A rank that
        # exits non-zero fails the whole job with it (shifu_amd/parallel.py: launch_ranks)
        from shifu_amd.parallel import launch_ranks
        raise SystemExit(launch_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus))

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.test_fail_rank is not None and args.test_fail_rank == rank and world > 1:
        raise SystemExit(3)      # tests/test_gpu_bench.py (hidden flag): a rank that dies must fail the whole job
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the backend has no CPU fallback")
    # one rank per GPU; SHIFU_AMD_DIST_BACKEND=gloo (testing only) lets several ranks share a GPU to exercise the
    # N>1 control flow on a single-GPU box -- RCCL itself refuses two ranks on one device
    backend = os.environ.get("SHIFU_AMD_DIST_BACKEND", "nccl")
    local_rank = local_rank % torch.cuda.device_count() if backend != "nccl" else local_rank
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # SHIFU_AMD_FORCE_DIST=1 (testing only): take the N>1 code path -- process group, barriers, the all-gather of episode
    # statistics, the MAX all-reduce of the elapsed time -- with a single rank, e.g. to exercise RCCL on a 1-GPU box
    use_dist = world > 1 or os.environ.get("SHIFU_AMD_FORCE_DIST", "0") == "1"
    from shifu_amd import _abi
    from shifu_amd.parallel import device_identity, gather_episode_stats, gather_rank_reports, init_ranks
    if use_dist:
        init_ranks(dev, backend)       # finite timeout: a rank that never arrives fails the job instead of hanging it

    solver_given = args.solver is not None or args.mapping is not None or args.group is not None or args.link_shapes != "box" or args.max_contacts is not None
    abb = args.workload == "abb"
    if abb:
        if args.link_contacts and (args.no_link_contacts or args.mapping == "chain"):
            raise SystemExit("bench.py: --link-contacts contradicts --no-link-contacts / --mapping chain (compiled for the rod-only scene)")
        args.link_contacts = not (args.no_link_contacts or args.mapping == "chain")
        if args.solver is None and args.mapping is None and args.group is None:
            args.solver = "tgs"      # FusedAbbEnv's own default: the reference's PhysX settings, solver_type = 1
        if args.solver in ("pgs", "tgs"):
            # the shipped scene: arm wave + box wave, the solve regrouped at 32 lanes per env (k_abb_step_ws_hard); --mapping body
            # (and hulls): the run-time-shaped kernel at 32 lanes per env
            mapping = "body" if (args.link_shapes == "hull" or args.mapping == "body" or args.group == 32) else "split"
            group = 16 if mapping == "split" else 32
        else:
            mapping = args.mapping or ("split" if (args.group or 16) == 16 else ("chain" if (not args.link_contacts and (args.group or 16) == 32) else "body"))
            group = args.group or 16
            if args.link_shapes == "hull":
                mapping = "body"          # the convex narrow phase lives in the run-time-shaped kernels
    else:    # the fused A1 env's own default: the chain-per-lane kernel at 32 lanes when there is no self-collision
        mapping = args.mapping or ("chain" if ((args.group or 32) == 32 or ((args.group or 32) == 16 and not args.self_collision)) else "body")
        group = args.group or 32
    if abb:
        from shifu_amd.gym.abb_fused import FusedAbbEnv
        env = FusedAbbEnv(num_envs=args.envs, device=dev, seed=42, rank=rank, world_size=world, group=group,
                          link_contacts=args.link_contacts, mapping=mapping, link_shapes=args.link_shapes,
                          **({} if args.solver is None else {"solver": args.solver}))
        args.solver, group, mapping = env.solver, env.sim.group, env.mapping
        stats_t, count_t, kernel = _abi.ABB_STATS, _abi.ABB_RESET_COUNT, "k_abb_step"
        substeps = 6
    else:
        from shifu_amd.gym.a1_fused import FusedA1Env
        env = FusedA1Env(num_envs=args.envs, device=dev, terrain={"terrain": "heightfield", "flat": "flat", "trimesh": "trimesh"}[args.workload],
                         seed=42, rank=rank, world_size=world, group=group, mapping=mapping, decimation=args.decimation,
                         extra_substep=not args.no_extra_substep, self_collision=args.self_collision, solver=args.solver,
                         solver_kw={"pos_iters": args.pos_iters, "vel_iters": args.vel_iters, **({} if args.max_contacts is None else {"max_contacts": args.max_contacts})})
        stats_t, count_t, kernel = _abi.A1_STATS, _abi.A1_RESET_COUNT, "k_a1_step"
        substeps = args.decimation + (0 if args.no_extra_substep else 1)
        args.solver = env.solver
    gen = torch.Generator(device=dev)
    gen.manual_seed(42 + rank)
    N, A = env.num_envs, env.num_actions
    env.reset()
    actions = torch.empty(N, A, device=dev)
    nslots = env.task.tensors[stats_t].shape[0]

    def eager_step(ev=None):
        if args.actions == "kernel":
            # run_policy('random'): U(-1, 1) drawn inside the fused launch (counter-based, keyed by global env id and
            # vec-step: BASELINE.md section 3) -- one launch per vec-step
            if ev is not None:
                ev[0].record()
            slot = env.task.step_random()
        else:
            actions.uniform_(-1.0, 1.0, generator=gen)   # = 2*rand-1 of policy_runner.py:40, one kernel, no allocation
            if ev is not None:
                ev[0].record()
            slot = env.task.step(actions)                # the fused kernel (episode statistics included)
        if ev is not None:
            ev[1].record()
        return slot

    # The vec-step is two launches (action draw + fused step, which now carries the episode statistics).  --graph
    # replays them as one captured hipGraph (possible because no launch argument depends on the step index).  HIP
    # events cannot be recorded inside a replay, so the kernel's own duration is always measured on a dedicated eager
    # pass of the same K steps after the timed region.
    use_graph = args.graph
    gathers = {"warmup": 0, "timed": 0}

    gather_events = []          # HIP events around every timed all-gather: its latency on this rank's stream

    def maybe_gather(global_step, slot, phase):
        # extras["episode"] logging cadence (policy_config.py:36: num_steps_per_env = 24), counted from the first step of
        # the run so that the driver's shape (--warmup 5 --steps 20) times exactly one all-gather
        if use_dist and (global_step + 1) % args.log_interval == 0:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) if phase == "timed" else None
            if ev:
                ev[0].record()
            gather_episode_stats(env.task.tensors[stats_t][slot][:env.task.num_sums])
            if ev:
                ev[1].record()
                gather_events.append(ev)
            gathers[phase] += 1

    for i in range(args.warmup):
        maybe_gather(i, eager_step(), "warmup")
    graph = None
    if use_graph:
        assert args.actions == "torch", "--graph captures [uniform_, fused step]: use --actions torch"
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        graph.register_generator_state(gen)
        with torch.cuda.graph(graph):
            actions.uniform_(-1.0, 1.0, generator=gen)
            env.task.launch_step(actions)
        env.task.graph_slot_stride = 1

    def vec_step(i):
        if graph is not None:
            graph.replay()
            slot = env.task.advance_slot()
        else:
            slot = eager_step()
        maybe_gather(args.warmup + i, slot, "timed")
        return slot

    torch.cuda.synchronize()
    # constraints / contacts beyond the per-env limits (SHF_T_DROPPED: counted by the kernels, never cleared here): the count
    # before the timed region, read again after it -- one D2H on each side, outside the clock
    dropped_t = env.sim.tensors.get(_abi.T_DROPPED)
    dropped0 = int(dropped_t.sum().item()) if dropped_t is not None else None
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        vec_step(i)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    dropped1 = int(dropped_t.sum().item()) if dropped_t is not None else None
    envs_dropping = int((dropped_t > 0).sum().item()) if dropped_t is not None else None

    # duration of the fused kernel alone: HIP events on the launch stream around each of K eager launches
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    if args.actions == "kernel":
        # the vec-step IS the kernel launch: one pair of events around K back-to-back launches (an event pair around every
        # launch adds ~2 us of event processing to each and reads higher than rocprofv3's per-kernel duration)
        events[0][0].record()
        for i in range(args.steps):
            eager_step()
        events[0][1].record()
        torch.cuda.synchronize()
        kern_ms = events[0][0].elapsed_time(events[0][1]) / args.steps
    else:
        for i in range(args.steps):
            eager_step(events[i])
        torch.cuda.synchronize()
        kern_ms = sum(a.elapsed_time(b) for a, b in events) / args.steps

    # candidate constraints per env and sub-step BEFORE the max_contacts cap (SHF_T_CONTACT_HIST), from a third, untimed pass of
    # up to 100 steps with the histogram bound (binding it costs the kernel one read-modify-write per env and sub-step: kept out of
    # the timed region and of the kernel_ms pass)
    cand_hist = None
    hard = args.solver in ("pgs", "tgs")
    if hard and graph is None:
        ht = env.sim.bind_contact_hist(True)
        for i in range(min(args.steps, 100)):
            eager_step()
        torch.cuda.synchronize()
        cand_hist = ht[:, :-1].sum(0).cpu().numpy().astype(np.float64)
        env.sim.bind_contact_hist(False)

    # every rank's own elapsed time: the MAX is the job's time (the contract), MIN / MAX together show the spread
    rank_elapsed = elapsed
    t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    tmin = t.clone()
    dist_world = 1
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
        dist_world = dist.get_world_size()
    elapsed, elapsed_min = float(t.item()), float(tmin.item())
    # every rank's own report, on rank 0's line: which GPU it ran on, its own clock, what its all-gathers cost
    gather_ms = [a.elapsed_time(b) for a, b in gather_events]
    reports = gather_rank_reports({"rank": rank, "local_rank": local_rank, "device": device_identity(dev), "pid": os.getpid(),
                                   "ms_per_step": rank_elapsed / args.steps * 1e3,
                                   "all_gather_ms": gather_ms, "all_gather_ms_mean": (sum(gather_ms) / len(gather_ms) if gather_ms else None)})
    finite = bool(torch.isfinite(env.obs_buf).all().item())
    resets = int(env.task.tensors[count_t].sum().item())

    if rank == 0:
        total_envs = N * world
        value = total_envs * args.steps / elapsed
        b_alg = B_ALG[args.workload]
        achieved = b_alg * N / (kern_ms * 1e-3) / 1e9
        prof = committed_profile(f"{kernel}_{args.workload}_g{group}" + ("_" + mapping if mapping != "body" else "") + ("_" + args.solver if args.solver in ("pgs", "tgs") else "") + ("_link" if (abb and args.link_contacts) else "")) if (N == 4096 and not args.self_collision) else None
        res = {}
        try:
            res = json.load(open(os.path.join(ROOT, "shifu_amd", "libshifu_amd.resources.json")))
        except Exception:
            pass
        entry = env.task.kernel_symbol() if hasattr(env.task, "kernel_symbol") else None
        vg = next((v for k, v in res.items() if entry and k.startswith(entry)), None)
        waves = (N + (64 // group) - 1) // (64 // group) if group < 64 else N
        if mapping == "split":
            waves *= 2          # every env group has an arm wave and a box wave (k_abb_step_ws)
        secondary = {"bound": "valu_fp32", "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "waves_per_simd": waves / float(NUM_CUS * SIMDS_PER_CU),
                     "lanes_per_env": group, "vgprs": None if vg is None else vg.get("vgprs"),
                     "scratch_bytes_per_lane": None if vg is None else vg.get("scratch")}
        if prof:
            secondary.update({"valu_issue_frac": prof.get("valu_issue_frac"), "wait_frac": prof.get("wait_frac"),
                              "counter_source": prof.get("source")})
            if prof.get("valu_insts_per_launch"):
                # VALU issue slots, priced with the measured issue rates (tools/valu_microbench): (i) the chip-wide peak,
                # one wave-instruction per SIMD per 2 clocks; (ii) the floor a single wave's own issue rate puts under the
                # launch: its VALU instructions x 4.57 clocks (a wave cannot issue faster whatever else the SIMD does).
                # Instruction count: SQ_INSTS_VALU of the committed PMC pass (replayed); duration: measured in this run.
                issue_peak = NUM_CUS * SIMDS_PER_CU * MAX_CLOCK_HZ / VALU_ISSUE_CLOCKS_SIMD
                issued = prof["valu_insts_per_launch"] / (kern_ms * 1e-3)
                per_wave = prof["valu_insts_per_launch"] / float(waves)
                one_wave_floor_ms = per_wave * VALU_ISSUE_CLOCKS_ONE_WAVE / MAX_CLOCK_HZ * 1e3
                secondary["valu_issue"] = {"wave_instructions_per_launch": prof["valu_insts_per_launch"],
                                           "achieved": issued, "peak": issue_peak, "unit": "wave-instructions/s",
                                           "frac": issued / issue_peak,
                                           "valu_instructions_per_wave": per_wave,
                                           "single_wave_issue_floor_ms": one_wave_floor_ms,
                                           "single_wave_issue_frac": one_wave_floor_ms / kern_ms,
                                           "peak_source": "profiles/r03_valu_microbench.md: v_fma_f32 wave64 = 4.57 clk per wave alone, "
                                                          "2.30 / 1.93 clk per SIMD at 2 / 4 waves; peak priced at 2 clk (MI355X_MICROARCH.md) and 2.4 GHz",
                                           "note": "the launch is bound by the dependent chain of each wave (LDS hand-offs, "
                                                   "IEEE div/sqrt sequences), not by chip-wide issue slots: see single_wave_issue_frac"}
        out = {
            "metric": ("env-steps/sec (whole node), ABB push-box 6-dof arm + free cube, 4096 envs/GPU" if abb else
                       "env-steps/sec (whole node), A1 12-dof 4096 envs/GPU"), "value": value, "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            # what the process group itself reports (proves that the collective backend saw N ranks) and the spread of the
            # per-rank clocks around the same K steps; `ms_per_step` is the slowest rank's
            "rccl_world_size": (dist_world if (use_dist and backend == "nccl") else None),
            "dist": {"backend": (backend if use_dist else None), "world_size": dist_world,
                     "ms_per_step_rank_min": elapsed_min / args.steps * 1e3, "ms_per_step_rank_max": elapsed / args.steps * 1e3,
                     "ms_per_step_rank0": rank_elapsed / args.steps * 1e3,
                     # one entry per rank (all_gather_object): distinct devices prove that N GPUs ran; all_gather_ms = HIP
                     # events around each timed all-gather of the episode statistics on that rank's stream
                     "ranks": reports,
                     "distinct_devices": len({r["device"] for r in reports}),
                     "all_gather_ms_max": max([x for r in reports for x in r["all_gather_ms"]], default=None)},
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{WL_NAME[args.workload]}, {N} envs/GPU, random actions, {substeps} substeps/env-step "
                                   f"(dt {'20' if abb else '5'} ms), resets on, contact solver: "
                                   + (f"velocity-level {'TGS (physx.solver_type = 1: sub-stepped sweeps)' if args.solver == 'tgs' else 'PGS (physx.solver_type = 0)'} {args.pos_iters} + {args.vel_iters} iterations (the reference's physx settings, env_config.py:50-58), at most {int(env.sim_params.max_contacts) or 8} constraints per env" if args.solver in ("pgs", "tgs")
                                      else "compliant spring-damper law (rounds 1-4)")
                                   + (((", link contacts ON (arm links as the reduced convex hulls of their collision meshes + rod vs table / cube / goal pad; clipped face manifolds on)" if args.link_shapes == "hull" else ", link contacts ON (arm links + rod vs table / cube / goal pad); the arm's own links do not collide with each other (the reference's collision filter 0 lets them: the gym facade's hook path has those 46 capsule pairs on)") if args.link_contacts else
                                       ", arm collider: the rod against the cube (link contacts OFF, see --link-contacts)") if abb else (", self-collision ON (capsule pairs, the reference's collision filter 0)" if args.self_collision
                                                      else ", self-collision OFF (the reference has it on: units.py:68; see --self-collision)")),
                       "envs_per_gpu": N, "total_envs": total_envs, "substeps_per_env_step": substeps,
                       "lanes_per_env": group, "lane_mapping": mapping, "self_collision": bool(args.self_collision) and not abb, "vec_step": "hipGraph replay of [uniform_, fused step]" if graph is not None else ("one launch: fused step with the U(-1,1) actions of run_policy('random') drawn in-kernel (counter-based, keyed by global env id and vec-step)" if args.actions == "kernel" else "two eager launches: torch uniform_ + fused step"),
                       "parallelism": f"env-sharded x{world}, all-gather of episode stats every {args.log_interval} steps",
                       "substeps_per_s": value * substeps, "obs_finite": finite, "episodes_reset_rank0": resets,
                       # candidates beyond ShfSimParams.max_contacts (PGS) / the self- and link-contact slot limits, per env and vec-step
                       # over the timed region on rank 0 (SHF_T_DROPPED deltas); envs_ever_dropping: envs whose counter is non-zero
                       "max_contacts": (int(env.sim_params.max_contacts) or 8) if args.solver in ("pgs", "tgs") else None,
                       "dropped_constraints_per_env_step": None if dropped0 is None else (dropped1 - dropped0) / float(N * args.steps),
                       "envs_ever_dropping_frac": None if envs_dropping is None else envs_dropping / float(N),
                       # share of (env, sub-step) pairs that offered k candidate constraints to the solve, k = 0 .. 24, last bin: more
                       # (rank 0, an untimed pass with SHF_T_CONTACT_HIST bound); mean; share above the cap
                       "candidates_per_substep_hist": None if cand_hist is None else [round(float(x), 5) for x in cand_hist / max(cand_hist.sum(), 1.0)],
                       "candidates_per_substep_mean": None if cand_hist is None else float((cand_hist * np.arange(len(cand_hist))).sum() / max(cand_hist.sum(), 1.0)),
                       "substeps_above_cap_frac": None if cand_hist is None else float(cand_hist[((int(env.sim_params.max_contacts) or 8) + 1):].sum() / max(cand_hist.sum(), 1.0)),
                       "gathers_in_timed_region": gathers["timed"], "gathers_in_warmup": gathers["warmup"]},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "peak_achievable": HBM_ACHIEVABLE_GBS, "frac_of_achievable": achieved / HBM_ACHIEVABLE_GBS,
                         "traffic": None if not prof else prof.get("traffic_bytes"),
                         "traffic_source": None if not prof else prof.get("source"),
                         # the loaded kernel's machine-code hash (shifu_amd/build.py: kernel_code_hashes) against the one the replayed
                         # counters were collected from: true = the counters describe another build of this kernel (re-profile)
                         "kernel_symbol": entry, "kernel_code_sha": None if vg is None else vg.get("code_sha"),
                         "counters_kernel_sha": None if not prof else prof.get("kernel_code_sha"),
                         "counters_stale": None if not prof else (prof.get("kernel_code_sha") is None or vg is None or prof.get("kernel_code_sha") != vg.get("code_sha")),
                         "kernel": (("k_a1_chain_" + args.solver + ("16" if int(env.sim_params.max_contacts) > 8 else "")) if (kernel == "k_a1_step" and args.solver in ("pgs", "tgs")) else "k_a1_chain" if (kernel == "k_a1_step" and mapping == "chain") else "k_abb_step_ws" if (kernel == "k_abb_step" and mapping == "split") else "k_abb_step_pgs_wide" if (abb and "pgs_wide" in env.task.kernel_symbol()) else kernel), "kernel_ms": kern_ms, "alg_bytes_per_env_step": b_alg,
                         "note": "latency/ALU-bound by design: a few KB of compulsory traffic per env-step (DESIGN.md 5)" +
                                 ("; alg_bytes counts the env's own tensors (state in, state / body_state / contact / Jacobian / obs out): link contacts "
                                  "add work on the LDS-resident model and scene, not tensors, so B_alg is the rod-only scene's" if (abb and args.link_contacts) else ""),
                         "secondary": secondary},
        }
        if kern_ms > out["ms_per_step"]:
            # a kernel cannot take longer than the step that contains it: the two are different passes of K launches (the
            # timed region by the host clock, then a second pass bracketed by HIP events) and differ by run-to-run noise
            out["roofline"]["kernel_ms_note"] = ("kernel_ms (HIP events, second pass of K launches) reads above ms_per_step (host clock, "
                                                 "timed pass): pass-to-pass noise, not a longer kernel; roofline.achieved uses kernel_ms, "
                                                 "the more conservative of the two")
        if world == 1 and not solver_given and not args.no_other_solver and args.actions == "kernel" and graph is None:
            # the driver runs the default command only: the opt-in compliant law of rounds 1-4 on the same workload, same K and
            # W, same clock, so that one line carries both (a fresh env; the main measurement above is already taken)
            other = "compliant" if args.solver in ("pgs", "tgs") else "pgs"
            if abb:
                env2 = FusedAbbEnv(num_envs=args.envs, device=dev, seed=42, link_contacts=args.link_contacts, solver=other, link_shapes=args.link_shapes)
            else:
                env2 = FusedA1Env(num_envs=args.envs, device=dev, terrain={"terrain": "heightfield", "flat": "flat", "trimesh": "trimesh"}[args.workload],
                                  seed=42, decimation=args.decimation, extra_substep=not args.no_extra_substep,
                                  self_collision=args.self_collision, solver=other)
            env2.reset()
            for _ in range(args.warmup):
                env2.task.step_random()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                env2.task.step_random()
            torch.cuda.synchronize()
            e2 = time.perf_counter() - t1
            out["other_solver"] = {"solver": env2.solver, "ms_per_step": e2 / args.steps * 1e3, "value": N * args.steps / e2, "unit": "env-steps/s",
                                   "kernel_symbol": env2.task.kernel_symbol(), "lanes_per_env": env2.sim.group, "lane_mapping": env2.mapping,
                                   "obs_finite": bool(torch.isfinite(env2.obs_buf).all().item()),
                                   "note": "the same workload under the solver that is not the default, timed the same way (host clock, K steps after W "
                                           "warm-up steps); compliant = the spring-damper law of rounds 1-4 (opt-in: --solver compliant)"}
        if not args.no_cpu_baseline and world == 1:
            scene_kw = {"link_contacts": bool(args.link_contacts), "link_shapes": args.link_shapes} if abb else {}
            cb = out["cpu_baseline"] = cpu_baseline(args.workload, solver_kw={"solver": args.solver, "pos_iters": args.pos_iters, "vel_iters": args.vel_iters, **({} if args.max_contacts is None else {"max_contacts": args.max_contacts})}, **scene_kw)     # the only leg that touches oracle/
            f_alg = cb["flops_per_env_step"]
            secondary.update({"flops_alg_per_env_step": f_alg, "flops_source": cb["flops_source"]})
            if f_alg is not None:
                secondary["achieved"] = f_alg * N / (kern_ms * 1e-3) / 1e12
                secondary["frac"] = secondary["achieved"] / VALU_PEAK_TFLOPS
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
