"""HIP kernels vs the CPU oracle on identical inputs, through the C ABI.

The kernels are specified to reproduce the oracle's float arithmetic operation
for operation (shifu_amd/csrc/shf_device.h), so the comparisons below are
*bit-exact* for every float tensor, not just within the north-star's 1e-4
relative tolerance over 1000 steps -- which follows a fortiori and is also
asserted explicitly in test_a1_1000_steps_within_north_star_tolerance.
"""
import os
import sys

import numpy as np
import pytest

from shifu_amd import _abi
from tests import helpers as H

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU is visible (these tests never fall back to CPU)")


def _terrain(rng, rows=60, cols=70, rough=True):
    t = _abi.ShfTerrain()
    t.rows, t.cols, t.hscale, t.vscale, t.border, t.friction = rows, cols, 0.1, 0.005, 2.0, 1.0
    h = np.zeros((rows, cols), np.int16)
    if rough:
        h[:] = rng.integers(-12, 12, size=(rows, cols))
        h[20:30, 20:40] += 40   # a plateau with vertical faces
        h[35:50, 10:30] = (np.arange(15)[:, None] * 6).astype(np.int16)  # a ramp
    return t, h


def _random_states(m, n, rng, z_lo=0.15, z_hi=0.45, xy_hi=3.0):
    dof = np.zeros((n * m.nd, 2), np.float32)
    root = np.zeros((n, 13), np.float32)
    q0 = np.array(H_DEFAULT_Q, np.float32)
    for e in range(n):
        dof[e * m.nd:(e + 1) * m.nd, 0] = q0 + rng.uniform(-0.4, 0.4, m.nd)
        dof[e * m.nd:(e + 1) * m.nd, 1] = rng.uniform(-3, 3, m.nd)
        quat = np.array([0, 0, 0, 1.0]) + rng.normal(0, 0.25, 4)
        quat /= np.linalg.norm(quat)
        root[e] = np.concatenate([[rng.uniform(0.5, xy_hi), rng.uniform(0.5, xy_hi), rng.uniform(z_lo, z_hi)], quat,
                                  rng.uniform(-1, 1, 3), rng.uniform(-2, 2, 3)])
    return dof, root


H_DEFAULT_Q = [0.1, 0.8, -1.5, 0.1, 0.8, -1.5, -0.1, 0.8, -1.5, -0.1, 0.8, -1.5]


def _make_sim(cm, sp, n, terrain=None, heights=None, group=64, env_off=0, warp=None):
    """group: lanes per env of the body-per-lane kernels, or 'chain16' / 'chain32' for the chain-per-lane fused A1 step."""
    from shifu_amd.backend import Sim
    mapping = "body"
    if isinstance(group, str):
        mapping, group = "chain", int(group[len("chain"):])
    sim = Sim(sp, "cuda:0")
    if terrain is None:
        sim.set_plane(1.0)
    else:
        sim.set_heightfield(heights, terrain.hscale, terrain.vscale, terrain.border, terrain.friction, warp=warp)
    sim.set_articulation(cm.blob)
    sim.finalize(n, env_off, group=group, mapping=mapping)
    return sim


@pytest.mark.parametrize("group", [64, 32])
@pytest.mark.parametrize("rough", [False, True])
def test_simulate_matches_oracle_bitwise(oracle, rough, group):
    """gym.simulate parity: 64 envs in and out of contact, explicit efforts, external pushes, 30 sub-steps."""
    _need_gpu()
    rng = np.random.default_rng(11)
    cm = H.a1_model()
    m = cm.blob
    sp = H.sim_params(angular_damping=0.5)
    n = 64
    terr, hs = _terrain(rng, rough=rough)
    dof, root = _random_states(m, n, rng)
    fr = rng.uniform(0.5, 1.25, n).astype(np.float32)
    sim = _make_sim(cm, sp, n, terr if rough else None, hs if rough else None, group=group)
    T = sim.tensors
    T[_abi.T_SIM_DOF].copy_(torch.from_numpy(dof))
    T[_abi.T_SIM_ROOT].copy_(torch.from_numpy(root))
    T[_abi.T_FRICTION].copy_(torch.from_numpy(fr))
    for it in range(30):
        eff = rng.uniform(-25, 25, n * m.nd).astype(np.float32)
        force = np.zeros((n * m.nb, 3), np.float32)
        push = it % 3 == 0
        if push:
            force[::m.nb] = rng.uniform(-5, 5, (n, 3))
            force[3::m.nb] = rng.uniform(-2, 2, (n, 3))   # also a leg body
            sim.apply_body_force(torch.from_numpy(force).cuda())
        sim.set_dof_command(_abi.T_EFFORT, torch.from_numpy(eff).cuda())
        sim.step()
        sim.refresh(_abi.REFRESH_ALL)
        contact, bstate = oracle.step(m, sp, n, dof, root, terrain=terr if rough else None, heights=hs if rough else None,
                                      effort=eff, body_force=force if push else None, friction=fr, want_contact=True,
                                      want_body_state=True)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(T[_abi.T_DOF_STATE].cpu().numpy(), dof, err_msg=f"dof_state step {it}")
        np.testing.assert_array_equal(T[_abi.T_ROOT_STATE].cpu().numpy(), root, err_msg=f"root_state step {it}")
        np.testing.assert_array_equal(T[_abi.T_CONTACT].cpu().numpy(), contact, err_msg=f"contact step {it}")
        np.testing.assert_array_equal(T[_abi.T_BODY_STATE].cpu().numpy(), bstate, err_msg=f"body_state step {it}")
    assert np.isfinite(root).all()
    assert (np.abs(contact).sum(1) > 0).any(), "test must exercise contacts"


@pytest.mark.parametrize("solver", ["compliant", "pgs"])
def test_force_at_position_matches_oracle_bitwise(oracle, solver):
    """gym.apply_rigid_body_force_at_pos_tensors(force, pos) (robot.py:231-236, apply_force_on_base(force, pos)): the force
    acts at the given world point -- on the base off its centre, and on a leg body -- for the sub-step that consumes it."""
    _need_gpu()
    rng = np.random.default_rng(12)
    cm = H.a1_model()
    m = cm.blob
    # ("pgs": the A1 with a wrench at a point takes the run-time-shaped kernel with the generic solve, k_sim_step<32,..,HARD> --
    # the chain kernel is compiled for centre-of-mass forces; robots near the ground so that constraints are in play too)
    sp = H.sim_params(angular_damping=0.5, solver=solver)
    n = 48
    dof, root = _random_states(m, n, rng)
    root[:, 2] += 1.0 if solver == "compliant" else 0.0  # in flight: what moves the bodies is the applied wrench
    sim = _make_sim(cm, sp, n, None, None, group=32)
    T = sim.tensors
    T[_abi.T_SIM_DOF].copy_(torch.from_numpy(dof))
    T[_abi.T_SIM_ROOT].copy_(torch.from_numpy(root))
    for it in range(12):
        force = np.zeros((n * m.nb, 3), np.float32)
        pos = np.zeros((n * m.nb, 3), np.float32)
        force[::m.nb] = rng.uniform(-20, 20, (n, 3))
        pos[::m.nb] = root[:, :3] + rng.uniform(-0.3, 0.3, (n, 3)).astype(np.float32)
        force[5::m.nb] = rng.uniform(-5, 5, (n, 3))
        pos[5::m.nb] = root[:, :3] + rng.uniform(-0.4, 0.4, (n, 3)).astype(np.float32)
        at_pos = it % 2 == 0                              # alternate with the centre-of-mass form: the switch must not stick
        sim.apply_body_force(torch.from_numpy(force).cuda(), torch.from_numpy(pos).cuda() if at_pos else None)
        sim.step()
        sim.refresh(_abi.REFRESH_ALL)
        oracle.step(m, sp, n, dof, root, body_force=force, body_force_pos=pos if at_pos else None)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(T[_abi.T_DOF_STATE].cpu().numpy(), dof, err_msg=f"dof_state step {it}")
        np.testing.assert_array_equal(T[_abi.T_ROOT_STATE].cpu().numpy(), root, err_msg=f"root_state step {it}")
    assert np.isfinite(root).all() and np.abs(root[:, 10:13]).max() > 0.5, "off-centre pushes must spin the trunk"


def _a1_buffers(cm, tp, n, rng, terr_rows, terr_cols):
    m = cm.blob
    nb, nd = m.nb, m.nd
    from shifu_amd.a1_task import height_points
    P = tp.num_height_points
    b = {}
    dof = np.zeros((n * nd, 2), np.float32)
    dof[:, 0] = np.tile(np.array(H_DEFAULT_Q, np.float32), n)
    torigins = np.zeros((tp.max_terrain_level, tp.num_terrain_cols, 3), np.float32)
    for i in range(tp.max_terrain_level):
        for j in range(tp.num_terrain_cols):
            torigins[i, j] = [1.5 + 0.3 * i, 1.0 + 0.15 * j, 0.02 * ((i + j) % 3)]
    levels = rng.integers(0, tp.max_terrain_level, n).astype(np.int64)
    types = (np.arange(n) * tp.num_terrain_cols // n).astype(np.int64)
    origins = torigins[levels, types].copy()
    root = np.zeros((n, 13), np.float32)
    root[:, :3] = origins + np.array([0, 0, 0.42], np.float32)
    root[:, 6] = 1.0
    b["dof_state"], b["root_state"] = dof, root
    b["body_state"] = np.zeros((n * nb, 13), np.float32)
    b["contact"] = np.zeros((n * nb, 3), np.float32)
    b["friction"] = rng.uniform(0.5, 1.25, n).astype(np.float32)
    b["actions"] = np.zeros((n, nd), np.float32)
    b["obs"] = np.zeros((n, 12 + 2 * nd + 3 * nd + P), np.float32)
    b["rew"] = np.zeros(n, np.float32)
    b["reset"] = np.zeros(n, np.uint8)
    b["timeout"] = np.zeros(n, np.uint8)
    b["ep_len"] = rng.integers(0, 500, n).astype(np.int64)   # staggered so time-outs happen early
    b["command"] = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
    b["history"] = np.zeros((n, nd, 3), np.float32)
    b["rew_sums"] = np.zeros((6, n), np.float32)
    b["torques"] = np.zeros((n, nd), np.float32)
    b["base_vel"] = np.zeros((n, 9), np.float32)
    b["heights"] = np.zeros((n, P), np.float32)
    b["hpoints"] = height_points()
    push = np.zeros((n, nb, 3), np.float32)
    push[:, tp.base_body] = rng.uniform(-5, 5, (n, 3))
    b["push"] = push
    b["origins"] = origins
    b["levels"], b["types"], b["torigins"] = levels, types, torigins
    b["reset_count"] = np.zeros(n, np.int32)
    b["done_sums"] = np.zeros((8, n), np.float32)
    return b


_SIM_T = {"dof_state": _abi.T_DOF_STATE, "root_state": _abi.T_ROOT_STATE, "body_state": _abi.T_BODY_STATE,
          "contact": _abi.T_CONTACT, "friction": _abi.T_FRICTION}
_A1_T = {"actions": _abi.A1_ACTIONS, "obs": _abi.A1_OBS, "rew": _abi.A1_REW, "reset": _abi.A1_RESET,
         "timeout": _abi.A1_TIMEOUT, "ep_len": _abi.A1_EP_LEN, "command": _abi.A1_COMMAND,
         "history": _abi.A1_HISTORY, "rew_sums": _abi.A1_REW_SUMS, "torques": _abi.A1_TORQUES,
         "base_vel": _abi.A1_BASE_VEL, "heights": _abi.A1_HEIGHTS, "hpoints": _abi.A1_HPOINTS, "push": _abi.A1_PUSH,
         "origins": _abi.A1_ORIGINS, "levels": _abi.A1_LEVELS, "types": _abi.A1_TYPES, "torigins": _abi.A1_TORIGINS,
         "reset_count": _abi.A1_RESET_COUNT, "done_sums": _abi.A1_DONE_SUMS}


def _upload(sim, task, bufs):
    for k, tid in _SIM_T.items():
        sim.tensors[tid].copy_(torch.from_numpy(bufs[k]).reshape(sim.tensors[tid].shape))
    for k, tid in _A1_T.items():
        task.tensors[tid].copy_(torch.from_numpy(bufs[k]).reshape(task.tensors[tid].shape))


def _compare(sim, task, bufs, tag, exact=True, tol=0.0):
    torch.cuda.synchronize()
    worst = 0.0
    for k, tid in list(_SIM_T.items()) + list(_A1_T.items()):
        t = (sim.tensors if k in _SIM_T else task.tensors)[tid].cpu().numpy().reshape(bufs[k].shape)
        if exact:
            np.testing.assert_array_equal(t, bufs[k], err_msg=f"{k} {tag}")
        elif t.dtype.kind == "f":
            d = np.abs(t - bufs[k]) / np.maximum(1.0, np.abs(bufs[k]))
            worst = max(worst, float(d.max()))
            assert d.max() <= tol, f"{k} {tag}: rel err {d.max()}"
        else:
            np.testing.assert_array_equal(t, bufs[k], err_msg=f"{k} {tag}")
    return worst


def _pgs_group(group):
    """'pgs' in a group parametrisation: the chain kernel at 32 lanes under SHF_SOLVER_PGS (k_a1_chain_pgs), the default of the
    fused A1 env since round 5 -> (group, sim_params keywords)."""
    return ("chain32", {"solver": "pgs"}) if group == "pgs" else (group, {})


def _a1_setup(n, rough, seed=5, group=64, env_off=0, cm=None, **spkw):
    from shifu_amd.a1_task import a1_task_params
    from shifu_amd.backend import A1Task
    rng = np.random.default_rng(seed)
    cm = cm or H.a1_model()
    sp = H.sim_params(angular_damping=0.5, **spkw)
    tp = a1_task_params(cm, num_rows=4, num_cols=5, env_length=0.8)
    terr, hs = _terrain(rng, rows=80, cols=60, rough=rough)
    bufs = _a1_buffers(cm, tp, n, rng, terr.rows, terr.cols)
    sim = _make_sim(cm, sp, n, terr, hs, group=group, env_off=env_off)
    task = A1Task(sim, tp)
    _upload(sim, task, bufs)
    return cm, sp, tp, terr, hs, bufs, sim, task, rng


@pytest.mark.parametrize("group", [64, 32, "chain16", "chain32"])
@pytest.mark.parametrize("rough", [False, True])
def test_fused_a1_step_matches_oracle_bitwise(oracle, rough, group):
    """ShifuVecEnv.step for A1Conditional: physics x5, heights, termination, six reward
    terms, on-device reset with curriculum, observations, history -- 120 vec-steps."""
    _need_gpu()
    n = 96
    cm, sp, tp, terr, hs, bufs, sim, task, rng = _a1_setup(n, rough, group=group, env_off=1000)
    resets = 0
    for it in range(120):
        raw = (2 * rng.random((n, cm.blob.nd)) - 1).astype(np.float32) * 1.5
        slot = task.step(torch.from_numpy(raw).cuda())
        oracle.a1_step(cm.blob, sp, tp, n, 1000, bufs, raw, terrain=terr, heights=hs)
        _compare(sim, task, bufs, f"step {it}")
        stats = task.tensors[_abi.A1_STATS][slot].cpu().numpy()
        np.testing.assert_array_equal(stats, oracle.a1_stats(tp, n, bufs["done_sums"]), err_msg=f"stats step {it}")
        resets += int(bufs["reset"].sum())
    assert resets > n // 4, "the run must exercise resets (time-outs and base contacts)"
    assert np.isfinite(bufs["obs"]).all()


@pytest.mark.parametrize("kmax", [8, 3, 16, 12])
@pytest.mark.parametrize("rough", [False, True])
def test_fused_a1_step_with_the_velocity_level_solve_matches_oracle_bitwise(oracle, rough, kmax):
    """The same step under ShfSimParams.solver = SHF_SOLVER_PGS (the reference's PhysX settings, env_config.py:50-58: 8 + 1
    sweeps): candidate selection, response matrix by impulse propagation, projected Gauss-Seidel, the two impulse passes --
    k_a1_chain_pgs against the oracle's hard_solve, every tensor, 120 vec-steps with falls and resets.  kmax = 3: more
    candidates than the solve holds on most steps (the deepest are kept, the others counted).  kmax = 16 / 12 (round 6): up to
    sixteen constraints per env on k_a1_chain_pgs16 -- the response matrix's upper triangle, packed over the contact slots and the
    pose / rate / exchange slots."""
    _need_gpu()
    n = 96
    cm, sp, tp, terr, hs, bufs, sim, task, rng = _a1_setup(n, rough, group="chain32", env_off=1000, solver="pgs", max_contacts=kmax)
    assert task.kernel_symbol().startswith("_Z16k_a1_chain_pgs16" if kmax > 8 else "_Z14k_a1_chain_pgsI")
    oracle.dropped(reset=True)
    resets = 0
    for it in range(120):
        raw = (2 * rng.random((n, cm.blob.nd)) - 1).astype(np.float32) * 1.5
        slot = task.step(torch.from_numpy(raw).cuda())
        oracle.a1_step(cm.blob, sp, tp, n, 1000, bufs, raw, terrain=terr, heights=hs)
        _compare(sim, task, bufs, f"step {it}")
        stats = task.tensors[_abi.A1_STATS][slot].cpu().numpy()
        np.testing.assert_array_equal(stats, oracle.a1_stats(tp, n, bufs["done_sums"]), err_msg=f"stats step {it}")
        resets += int(bufs["reset"].sum())
    assert resets > n // 4 and np.isfinite(bufs["obs"]).all()
    assert np.abs(bufs["contact"]).max() > 10.0, "the feet carry the robots"
    torch.cuda.synchronize()
    d = oracle.dropped()
    assert int(sim.tensors[_abi.T_DROPPED].sum()) == d
    assert d > (20000 if kmax == 3 else 0) or kmax > 8, d
    if kmax > 8:      # the cap at 8 would have dropped far more on the same run (the stumbling robots offer 9 .. 16 candidates)
        assert d < 10000, d


@pytest.mark.parametrize("kmax", [8, 16])
def test_fused_a1_step_with_the_velocity_level_solve_at_full_size(oracle, kmax):
    """... and at BASELINE's env count: 4096 envs x 30 vec-steps, every tensor bit for bit (8 and 16 constraints per env)."""
    _need_gpu()
    n = 4096
    cm, sp, tp, terr, hs, bufs, sim, task, rng = _a1_setup(n, True, seed=91, group="chain32", env_off=8192, solver="pgs", max_contacts=kmax)
    bufs["ep_len"][:] = rng.integers(900, 1001, n)
    _upload(sim, task, bufs)
    resets = 0
    for it in range(30):
        raw = (2 * rng.random((n, cm.blob.nd)) - 1).astype(np.float32)
        slot = task.step(torch.from_numpy(raw).cuda())
        oracle.a1_step(cm.blob, sp, tp, n, 8192, bufs, raw, terrain=terr, heights=hs)
        if it % 10 == 9:
            _compare(sim, task, bufs, f"step {it}")
            np.testing.assert_array_equal(task.tensors[_abi.A1_STATS][slot].cpu().numpy(), oracle.a1_stats(tp, n, bufs["done_sums"]))
        resets += int(bufs["reset"].sum())
    assert resets > 100


@pytest.mark.parametrize("group", [32, "chain32"])
def test_fused_a1_step_matches_oracle_bitwise_at_full_size(oracle, group):
    """BASELINE's env count (4096 per GPU): the fused A1 step against the oracle (OpenMP over envs), every tensor bit for
    bit, 30 vec-steps with resets -- not only through size-independent properties."""
    _need_gpu()
    n = 4096
    cm, sp, tp, terr, hs, bufs, sim, task, rng = _a1_setup(n, True, seed=91, group=group, env_off=8192)
    bufs["ep_len"][:] = rng.integers(900, 1001, n)       # time-outs inside the window for a good share of the envs
    _upload(sim, task, bufs)
    resets = 0
    for it in range(30):
        raw = (2 * rng.random((n, cm.blob.nd)) - 1).astype(np.float32)
        slot = task.step(torch.from_numpy(raw).cuda())
        oracle.a1_step(cm.blob, sp, tp, n, 8192, bufs, raw, terrain=terr, heights=hs)
        if it % 10 == 9:
            _compare(sim, task, bufs, f"step {it}")
            np.testing.assert_array_equal(task.tensors[_abi.A1_STATS][slot].cpu().numpy(), oracle.a1_stats(tp, n, bufs["done_sums"]))
        resets += int(bufs["reset"].sum())
    assert resets > 100


@pytest.mark.parametrize("terrain", ["heightfield", "flat"])
def test_fused_a1_env_on_the_benchmark_scene_matches_oracle_bitwise_at_full_size(oracle, terrain):
    """BASELINE configs 3 and 2 exactly as bench.py builds them -- FusedA1Env(4096) on the procedural 1300 x 2100 height
    field (NumPy seed 42, curriculum layout) / on the all-zero map, default kernel (chain per lane, 32 lanes) -- against the
    oracle for 30 vec-steps with time-outs and falls: every tensor, bit for bit."""
    _need_gpu()
    from shifu_amd.gym.a1_fused import FusedA1Env
    n = 4096
    env = FusedA1Env(num_envs=n, terrain=terrain, seed=42)
    assert env.mapping == "chain" and env.group == 32
    S, T = env.sim.tensors, env.task.tensors
    assert tuple(S[_abi.T_HEIGHTS].shape) == (1300, 2100)
    g = torch.Generator().manual_seed(3)
    # spread the envs over the curriculum's rows so that every sub-terrain kind is stood on, and bring time-outs into the window
    T[_abi.A1_LEVELS].copy_(torch.randint(0, 10, (n,), generator=g))
    lv, ty = T[_abi.A1_LEVELS].cpu(), T[_abi.A1_TYPES].cpu()
    T[_abi.A1_ORIGINS].copy_(T[_abi.A1_TORIGINS].cpu()[lv, ty])
    env.task.reset_all()
    T[_abi.A1_EP_LEN].copy_(torch.randint(470, 501, (n,), generator=g))
    torch.cuda.synchronize()
    bufs = {k: S[t].cpu().numpy().copy() for k, t in _SIM_T.items()}
    bufs.update({k: T[t].cpu().numpy().copy() for k, t in _A1_T.items()})
    hs = S[_abi.T_HEIGHTS].cpu().numpy().copy()
    assert (terrain == "flat") == (not hs.any())
    rng = np.random.default_rng(17)
    resets, timeouts = 0, 0
    for it in range(30):
        raw = (2 * rng.random((n, 12)) - 1).astype(np.float32)
        slot = env.task.step(torch.from_numpy(raw).cuda())
        oracle.a1_step(env.cm.blob, env.sim_params, env.task_params, n, 0, bufs, raw, terrain=env.sim.terrain, heights=hs)
        if it % 10 == 9:
            _compare(env.sim, env.task, bufs, f"{terrain} step {it}")
            np.testing.assert_array_equal(T[_abi.A1_STATS][slot].cpu().numpy(), oracle.a1_stats(env.task_params, n, bufs["done_sums"]))
        resets += int(bufs["reset"].sum())
        timeouts += int(bufs["timeout"].sum())
    assert resets > 500 and timeouts > 200
    if terrain == "heightfield":
        assert np.unique(bufs["heights"]).size > 100, "the robots stand on real relief"


def test_long_differential_run_of_every_kernel_form(oracle):
    """tools/fuzz_parity.py trimmed to fit the suite: 320 vec-steps x 192 envs (64 with link contacts) of all nineteen kernel
    forms of both tasks -- twelve of the compliant law, seven of the velocity-level solve (A1 chain PGS / TGS, config 5 on the run-time-shaped kernel
    and on k_abb_step_ws_hard) -- against the oracle, every tensor compared every 80 steps, through hundreds of resets
    (the 2000-step run is profiles/r03_fuzz_parity.txt)."""
    _need_gpu()
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_parity
    out = fuzz_parity.run(steps=320, envs=192, every=80, link_envs=64)
    assert len(out) == 19 and all(r["equal"] for r in out)
    assert sum(r["resets"] for r in out if r["task"] == "a1") > 300 and sum(r["resets"] for r in out if r["task"] == "abb") > 300


@pytest.mark.parametrize("solver", ["tgs", "pgs", "compliant", "pgs-body"])
@pytest.mark.parametrize("link", [True, False])
def test_fused_abb_step_matches_oracle_bitwise_at_full_size(oracle, link, solver):
    """The same for config 5 -- the reference's scene, every link colliding (the default of FusedAbbEnv and of
    bench.py --workload abb), and the rod-only scene -- at 4096 envs, 30 vec-steps: under the default solver (PGS: the
    run-time-shaped body-per-lane kernel with the generic solve) and under the compliant law (the arm wave + box wave kernel)."""
    _need_gpu()
    from shifu_amd.gym.abb_fused import FusedAbbEnv
    n = 4096
    body = solver.endswith("-body")       # the run-time-shaped kernel (what any other arm / scene runs on) instead of the default
    solver = solver.split("-")[0]
    env = FusedAbbEnv(num_envs=n, seed=23, link_contacts=link, solver=solver, **({"mapping": "body"} if body else {}))
    assert env.solver == solver and env.link_contacts == link
    if solver in ("pgs", "tgs") and not body:
        # the default since round 6: arm wave + box wave, the solve regrouped at 32 lanes per env
        assert env.mapping == "split" and env.sim.group == 16 and env.sim_params.solver == (_abi.SOLVER_TGS if solver == "tgs" else _abi.SOLVER_PGS)
        assert env.task.kernel_symbol() == f"_Z18k_abb_step_ws_hardILb{int(link)}EE"
    elif solver in ("pgs", "tgs"):
        assert env.mapping == "body" and env.sim.group == 32 and env.sim_params.solver == (_abi.SOLVER_TGS if solver == "tgs" else _abi.SOLVER_PGS)
        # link contacts: sixteen envs per workgroup of 512 threads (8.7 KB of LDS per env: 4096 envs resident at once); the rod-only
        # scene: eight per workgroup of 256, two workgroups per CU
        assert env.task.kernel_symbol() == ("_Z19k_abb_step_pgs_wideILb1EE" if link else "_Z10k_abb_stepILi32E7DynDims8DynSceneLb0ELi0ELb1ELb0EE")
    else:
        assert env.mapping == "split" and env.sim.group == 16
        assert env.task.kernel_symbol() == ("_Z13k_abb_step_wsILi512ELb1EE" if link else "_Z13k_abb_step_wsILi256ELb0EE")
    env.task.tensors[_abi.ABB_EP_LEN].copy_(torch.randint(150, 201, (n,)))
    torch.cuda.synchronize()
    bufs = {k: env.sim.tensors[t].cpu().numpy().copy() for k, t in _ABB_SIM_T.items()}
    bufs.update({k: env.task.tensors[t].cpu().numpy().copy() for k, t in _ABB_T.items()})
    rng = np.random.default_rng(4)
    resets = 0
    for it in range(30):
        raw = (2 * rng.random((n, 3)) - 1).astype(np.float32)
        env.task.step(torch.from_numpy(raw).cuda())
        oracle.abb_step(env.cm.blob, env.sim_params, env.boxes, env.task_params, n, 0, bufs, raw)
        resets += int(bufs["reset"].sum())
    torch.cuda.synchronize()
    for k, t in list(_ABB_SIM_T.items()) + list(_ABB_T.items()):
        got = (env.sim.tensors if k in _ABB_SIM_T else env.task.tensors)[t].cpu().numpy().reshape(bufs[k].shape)
        np.testing.assert_array_equal(got, bufs[k], err_msg=k)
    assert resets > 100


@pytest.mark.parametrize("group", [32, "chain16", "chain32", "pgs"])
def test_fused_a1_step_push_on_every_body(oracle, group):
    """rand_force_buf is (N, bodies, 3) (a1_conditional.py:82-87): a user may push any body, welded feet included --
    forces on every reported body, folded into its moving body in body order, on both lane mappings."""
    _need_gpu()
    n = 40
    group, kw = _pgs_group(group)
    cm, sp, tp, terr, hs, bufs, sim, task, rng = _a1_setup(n, True, seed=77, group=group, **kw)
    bufs["push"][:] = rng.uniform(-4, 4, bufs["push"].shape).astype(np.float32)
    bufs["push"][::3, 5] = 0.0          # some bodies of some envs unpushed (skipped, not added as zeros)
    _upload(sim, task, bufs)
    for it in range(40):
        raw = (2 * rng.random((n, cm.blob.nd)) - 1).astype(np.float32)
        task.step(torch.from_numpy(raw).cuda())
        oracle.a1_step(cm.blob, sp, tp, n, 0, bufs, raw, terrain=terr, heights=hs)
        _compare(sim, task, bufs, f"step {it}")
    assert np.isfinite(bufs["obs"]).all()


@pytest.mark.parametrize("group", [64, "chain16", "pgs"])
def test_a1_1000_steps_vs_oracle_at_north_star_tolerance(oracle, group):
    """Per-step dof_pos / dof_vel / root_state within the north-star's 1e-4 relative over 1000 steps -- of the ORACLE
    (Isaac Gym itself is a closed binary that is absent).  Checked every 50 steps."""
    _need_gpu()
    n = 32
    group, kw = _pgs_group(group)
    cm, sp, tp, terr, hs, bufs, sim, task, rng = _a1_setup(n, True, seed=9, group=group, **kw)
    worst = 0.0
    for it in range(1000):
        raw = (2 * rng.random((n, cm.blob.nd)) - 1).astype(np.float32)
        task.step(torch.from_numpy(raw).cuda())
        oracle.a1_step(cm.blob, sp, tp, n, 0, bufs, raw, terrain=terr, heights=hs)
        if it % 50 == 49:
            worst = max(worst, _compare(sim, task, bufs, f"step {it}", exact=False, tol=1e-4))
    assert worst <= 1e-4


def test_shard_invariance_of_fused_step():
    """env e of a shard starting at global id g must evolve exactly like
    global env g+e of an unsharded run (RNG keyed by global id)."""
    _need_gpu()
    n = 64
    cm, sp, tp, terr, hs, bufs, sim, task, rng = _a1_setup(n, True, seed=21, env_off=0)
    half = {}
    for k, v in bufs.items():
        if k in ("hpoints", "torigins"):
            half[k] = v.copy()
        elif k in ("rew_sums", "done_sums"):
            half[k] = np.ascontiguousarray(v[:, n // 2:])
        else:
            rows = v.shape[0] // n
            half[k] = np.ascontiguousarray(v[n // 2 * rows:])
    from shifu_amd.backend import A1Task
    sim2 = _make_sim(cm, sp, n // 2, terr, hs, env_off=n // 2)
    task2 = A1Task(sim2, tp)
    _upload(sim2, task2, half)
    for it in range(150):
        raw = (2 * rng.random((n, cm.blob.nd)) - 1).astype(np.float32) * 1.5
        task.step(torch.from_numpy(raw).cuda())
        task2.step(torch.from_numpy(raw[n // 2:]).cuda())
    torch.cuda.synchronize()
    for tid in (_abi.A1_OBS, _abi.A1_REW, _abi.A1_COMMAND, _abi.A1_EP_LEN, _abi.A1_RESET_COUNT):
        a = task.tensors[tid][n // 2:].cpu().numpy()
        b = task2.tensors[tid].cpu().numpy()
        np.testing.assert_array_equal(a, b)
    assert int(task.tensors[_abi.A1_RESET_COUNT].sum()) > 0


def test_backend_fails_loudly_without_library(monkeypatch):
    """No silent CPU fallback: a missing .so must raise."""
    import shifu_amd._lib as L
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "_PATH", "/nonexistent/libshifu_amd.so")
    with pytest.raises(L.BackendError):
        L.lib()


@pytest.mark.parametrize("group", [64, 32])
def test_abb_scene_matches_oracle_bitwise(oracle, group):
    """Config 5 scene (reference examples/abb_pushbox_vision/task_config.py:13-84): fixed-base 6-dof arm
    under implicit POS drives, fixed table, free cube, fixed goal pad; cube-on-table and rod-vs-cube
    contacts; rigid_body_state incl. the box rows and the fixed-base Jacobian tensor."""
    _need_gpu()
    from shifu_amd.abb_task import ABB_BASE_POS, ABB_DEFAULT_DOF_POS, abb_boxes, abb_model
    from shifu_amd.backend import Sim
    rng = np.random.default_rng(17)
    cm = abb_model()
    m = cm.blob
    sp = H.sim_params(dt=0.02, angular_damping=0.5)
    boxes = abb_boxes()
    n, A, B = 32, 4, m.nb + 3
    dof = np.zeros((n * m.nd, 2), np.float32)
    dof[:, 0] = np.tile(np.array(ABB_DEFAULT_DOF_POS, np.float32), n) + rng.uniform(-0.05, 0.05, n * m.nd)
    root = np.zeros((n * A, 13), np.float32)
    root[:, 6] = 1.0
    root[0::A, :3] = ABB_BASE_POS
    root[1::A, :3] = [0, 0, 0.05]
    root[2::A, :3] = np.stack([rng.uniform(0.03, 0.08, n), rng.uniform(-0.03, 0.03, n), rng.uniform(0.125, 0.14, n)], 1)
    yaw = rng.uniform(-np.pi, np.pi, n)
    root[2::A, 5], root[2::A, 6] = np.sin(yaw / 2), np.cos(yaw / 2)
    root[3::A, :3] = np.stack([rng.uniform(-0.1, 0.1, n), rng.uniform(-0.1, 0.1, n), np.full(n, 0.1)], 1)
    fr = np.ones(n, np.float32)
    sim = Sim(sp, "cuda:0")
    sim.set_plane(1.0)
    sim.set_articulation(m)
    for b in boxes:
        sim.add_box(b)
    sim.finalize(n, 0, group=group)
    T = sim.tensors
    T[_abi.T_SIM_DOF].copy_(torch.from_numpy(dof))
    T[_abi.T_SIM_ROOT].copy_(torch.from_numpy(root))
    tgt = dof[:, 0].copy()
    touched = False
    for it in range(60):
        if it % 6 == 0:   # drive the rod towards +x: joint_2/joint_3 forward a little each env step
            tgt = dof[:, 0].copy()
            tgt[1::m.nd] += 0.02
            tgt[2::m.nd] -= 0.01
            tgt += rng.uniform(-0.005, 0.005, tgt.shape).astype(np.float32)
        sim.set_dof_command(_abi.T_POS_TARGET, torch.from_numpy(tgt).cuda())
        sim.step()
        sim.refresh(_abi.REFRESH_ALL)
        contact, bstate, jac = oracle.scene_step(m, sp, boxes, n, dof, root, pos_target=tgt, friction=fr,
                                                 want_jacobian=True)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(T[_abi.T_DOF_STATE].cpu().numpy(), dof, err_msg=f"dof step {it}")
        np.testing.assert_array_equal(T[_abi.T_ROOT_STATE].cpu().numpy(), root, err_msg=f"root step {it}")
        np.testing.assert_array_equal(T[_abi.T_CONTACT].cpu().numpy(), contact, err_msg=f"contact step {it}")
        np.testing.assert_array_equal(T[_abi.T_BODY_STATE].cpu().numpy(), bstate, err_msg=f"body step {it}")
        np.testing.assert_array_equal(T[_abi.T_JACOBIAN].cpu().numpy(), jac, err_msg=f"jacobian step {it}")
        touched |= bool(np.abs(contact.reshape(n, B, 3)[:, :m.nb]).sum() > 0)
    assert np.isfinite(root).all() and np.isfinite(dof).all()
    cube = root[2::A]
    assert (np.abs(contact.reshape(n, B, 3)[:, m.nb + 1, 2] - 0.981) < 0.2).mean() > 0.5   # cubes rest on the table
    assert touched, "the rod must have touched a cube"
    # This blind joint ramp drives every rod tip through the table top (the arm's spheres collide with free boxes only),
    # and a 0.1 kg cube pinned between rod and table is felt by the arm at only m / (m + dt beta) of its reaction (the
    # staggered coupling of shf_boxes.h): a cube the rod comes down on is squeezed into the table.  Cubes pushed from
    # the side -- what the task does, its rod tip stays in z in [0.11, 0.14] (task_config.py:63-64) -- stay on it.
    assert (cube[:, 2] > 0.12).mean() >= 0.9 and (cube[:, 2] > 0.10).all(), cube[:, 2]


def _scene_on_gpu(cm, sp, boxes, roots, n, group):
    from shifu_amd.backend import Sim
    m = cm.blob
    A = 1 + len(boxes)
    dof = np.zeros((n * m.nd, 2), np.float32)
    root = np.zeros((n * A, 13), np.float32)
    root[:, 6] = 1.0
    for k, p in enumerate(roots):
        root[1 + k::A, :3] = p
    sim = Sim(sp, "cuda:0")
    sim.set_plane(1.0)
    sim.set_articulation(m)
    for b in boxes:
        sim.add_box(b)
    sim.finalize(n, 0, group=group)
    sim.tensors[_abi.T_SIM_DOF].copy_(torch.from_numpy(dof))
    sim.tensors[_abi.T_SIM_ROOT].copy_(torch.from_numpy(root))
    return sim, dof, root


@pytest.mark.parametrize("group", [16, 64])
def test_link_contacts_match_oracle_bitwise(oracle, group):
    """SURVEY 8f f3 (ShfModel.link_collide): link box corners against a free cube (pair law, several slots on one box) and
    against a fixed table, cube corners inside a link's box volume, a capsule against the fixed table, more contacts than
    slots (dropped and counted per env) -- the scenes of tests/test_link_contacts.py through shf_sim_step, every tensor
    against the oracle bit for bit, per-env perturbed so that the envs of a wavefront differ."""
    _need_gpu()
    from shifu_amd.abb_task import box_desc
    from tests import kat_models as K
    sp = H.sim_params(angular_damping=0.5)
    more = "".join('<collision><origin xyz="%g 0 0.5"/><geometry><box size="0.04 0.04 0.04"/></geometry></collision>' % x for x in (0.1, -0.1))
    scenes = [
        ("ram pushes cube", K.box_pusher_model(), [box_desc((0.1, 0.1, 0.1), 0.5, 0.6, False, (0.2, 0.0, 0.05))], [(0.2, 0.0, 0.0499)], 0.4, 160),
        ("ram on table", K.box_pusher_model(centre=(0.0, 0.0, 0.5), axis="0 0 -1"), [box_desc((0.6, 0.6, 0.1), 0.0, 0.5, True, (0.0, 0.0, 0.3))], [(0.0, 0.0, 0.3)], 0.5, 120),
        ("cube on anvil", K.box_pusher_model(size=(0.3, 0.3, 0.1), centre=(0.0, 0.0, 0.05)), [box_desc((0.05, 0.05, 0.05), 0.2, 0.6, False, (0.0, 0.0, 0.125))], [(0.02, 0.01, 0.1249)], 0.05, 120),
        ("capsule on table", K.box_pusher_model(size=None, axis="0 0 -1", capsule=((0.0, 0.0, 0.5), (0.0, 0.0, 0.4), 0.02)), [box_desc((0.6, 0.6, 0.1), 0.0, 0.5, True, (0.0, 0.0, 0.2))], [(0.0, 0.0, 0.2)], 0.5, 120),
        ("24 corners in a box", K.box_pusher_model(size=(0.04, 0.04, 0.04), centre=(0.0, 0.0, 0.5), extra_shapes=more), [box_desc((1.0, 1.0, 0.2), 0.0, 0.5, True, (0.0, 0.0, 0.43))], [(0.0, 0.0, 0.43)], 0.0, 6),
    ]
    rng = np.random.default_rng(3)
    for name, cm, boxes, roots, v, steps in scenes:
        n = 9
        m = cm.blob
        assert m.link_collide == 1
        sim, dof, root = _scene_on_gpu(cm, sp, boxes, roots, n, group)
        A = 1 + len(boxes)
        # the envs of a wavefront must differ: shift the free boxes a little, vary the drive speed
        root[1::A, 0] += rng.uniform(-0.004, 0.004, n).astype(np.float32)
        vt = (v * rng.uniform(0.6, 1.0, n * m.nd)).astype(np.float32)
        sim.tensors[_abi.T_SIM_ROOT].copy_(torch.from_numpy(root))
        oracle.dropped(reset=True)
        sim.tensors[_abi.T_DROPPED].zero_()
        seen = False
        for it in range(steps):
            sim.set_dof_command(_abi.T_VEL_TARGET, torch.from_numpy(vt).cuda())
            sim.step()
            sim.refresh(_abi.REFRESH_ALL)
            contact, bstate, _ = oracle.scene_step(m, sp, boxes, n, dof, root, vel_target=vt, friction=np.ones(n, np.float32))
            torch.cuda.synchronize()
            np.testing.assert_array_equal(sim.tensors[_abi.T_DOF_STATE].cpu().numpy(), dof, err_msg=f"{name}: dof step {it}")
            np.testing.assert_array_equal(sim.tensors[_abi.T_ROOT_STATE].cpu().numpy(), root, err_msg=f"{name}: root step {it}")
            np.testing.assert_array_equal(sim.tensors[_abi.T_CONTACT].cpu().numpy(), contact, err_msg=f"{name}: contact step {it}")
            seen |= bool(np.abs(contact.reshape(n, -1, 3)[:, m.nb - 1]).sum() > 0)
        assert seen, f"{name}: the link never touched"
        assert int(sim.tensors[_abi.T_DROPPED].sum()) == oracle.dropped(reset=True), name
        if name.startswith("24"):
            assert (sim.tensors[_abi.T_DROPPED].cpu().numpy() == steps * 8).all()
        sim.destroy() if hasattr(sim, "destroy") else None


@pytest.mark.parametrize("group", [16, 32, 64])
def test_edge_edge_contacts_match_oracle_bitwise(oracle, group):
    """SURVEY 8f f3, edge-edge box contact (box_box_edge: the separating-axis edge case), the GPU twin of
    tests/test_link_contacts.py::test_a_box_rests_crosswise_on_another_boxs_edge: (1) a free bar set down crosswise, ridge to
    ridge, on a fixed box turned 45 degrees -- no vertex of either inside the other, only the crossing edges carry it; (2) the
    ram's box, turned 45 degrees about its travel axis, driven down onto the ridge of a fixed box turned 45 degrees the other
    way: a link volume's edge across a box actor's edge (family E of the link contacts).  shf_sim_step against the oracle,
    every tensor bit for bit, the envs of a wavefront perturbed; both scenes must have carried load through the edge slot."""
    _need_gpu()
    from shifu_amd.abb_task import box_desc
    from tests import kat_models as K
    from tests.test_link_contacts import _quat
    sp = H.sim_params(angular_damping=0.5)
    s2 = np.sqrt(2.0)
    rng = np.random.default_rng(8)
    # (1) bar on ridge
    ridge = box_desc((0.4, 0.1, 0.1), 0.0, 0.8, True, (0.0, 0.0, 0.3), _quat([1, 0, 0], np.pi / 4))
    bar = box_desc((0.1, 0.4, 0.1), 0.5, 0.8, False, (0.0, 0.0, 0.0), _quat([0, 1, 0], np.pi / 4))
    cm = K.box_pusher_model(centre=(2.0, 0.0, 0.5))
    m = cm.blob
    n = 9
    z0 = 0.3 + 0.1 * s2 + 0.002
    sim, dof, root = _scene_on_gpu(cm, sp, [ridge, bar], [(0.0, 0.0, 0.3), (0.01, 0.0, z0)], n, group)
    root[1::3, 3:7] = ridge.quat[:]
    root[2::3, 3:7] = bar.quat[:]
    root[2::3, 0] += rng.uniform(-0.05, 0.05, n).astype(np.float32)          # anywhere along the ridge
    root[2::3, 1] += rng.uniform(-0.002, 0.002, n).astype(np.float32)        # slightly off balance: the bars tip over at different times
    sim.tensors[_abi.T_SIM_ROOT].copy_(torch.from_numpy(root))
    vt = np.zeros(n * m.nd, np.float32)
    carried = 0
    for it in range(150):
        sim.set_dof_command(_abi.T_VEL_TARGET, torch.from_numpy(vt).cuda())
        sim.step()
        sim.refresh(_abi.REFRESH_ALL)
        contact, bstate, _ = oracle.scene_step(m, sp, [ridge, bar], n, dof, root, vel_target=vt, friction=np.ones(n, np.float32))
        torch.cuda.synchronize()
        np.testing.assert_array_equal(sim.tensors[_abi.T_ROOT_STATE].cpu().numpy(), root, err_msg=f"bar on ridge: root step {it}")
        np.testing.assert_array_equal(sim.tensors[_abi.T_CONTACT].cpu().numpy(), contact, err_msg=f"bar on ridge: contact step {it}")
        carried += int((contact.reshape(n, -1, 3)[:, m.nb + 1, 2] > 2.0).sum())
    assert carried > 40 * n, carried                          # (m g = 4.9 N through the one edge-edge slot, for most of the run)
    assert (root[2::3, 2] > 0.3).all()                        # nobody fell through the ridge
    sim.destroy() if hasattr(sim, "destroy") else None
    # (2) a link volume's edge across a fixed box's edge
    cm = K.box_pusher_model(size=(0.3, 0.06, 0.06), centre=(0.0, 0.0, 0.5), axis="0 0 -1", shape_rpy=(np.pi / 4, 0.0, 0.0))
    m = cm.blob
    assert m.link_collide == 1 and m.nabox == 1
    anvil = box_desc((0.06, 0.4, 0.06), 0.0, 0.5, True, (0.0, 0.0, 0.3), _quat([0, 1, 0], np.pi / 4))
    sim, dof, root = _scene_on_gpu(cm, sp, [anvil], [(0.0, 0.0, 0.3)], n, group)
    root[1::2, 3:7] = anvil.quat[:]
    root[1::2, 0] += rng.uniform(-0.02, 0.02, n).astype(np.float32)
    sim.tensors[_abi.T_SIM_ROOT].copy_(torch.from_numpy(root))
    vt = (0.5 * rng.uniform(0.6, 1.0, n * m.nd)).astype(np.float32)
    pushed = 0
    for it in range(140):
        sim.set_dof_command(_abi.T_VEL_TARGET, torch.from_numpy(vt).cuda())
        sim.step()
        sim.refresh(_abi.REFRESH_ALL)
        contact, bstate, _ = oracle.scene_step(m, sp, [anvil], n, dof, root, vel_target=vt, friction=np.ones(n, np.float32))
        torch.cuda.synchronize()
        np.testing.assert_array_equal(sim.tensors[_abi.T_DOF_STATE].cpu().numpy(), dof, err_msg=f"ram edge on ridge: dof step {it}")
        np.testing.assert_array_equal(sim.tensors[_abi.T_CONTACT].cpu().numpy(), contact, err_msg=f"ram edge on ridge: contact step {it}")
        pushed += int((contact.reshape(n, -1, 3)[:, m.nb - 1, 2] > 10.0).sum())
    assert pushed > 20 * n, pushed
    qd = dof.reshape(n, m.nd, 2)[:, 0, 1]
    assert (np.abs(qd) < 0.05).all(), qd                      # stalled on the ridge (without the edge pass it sails through)


@pytest.mark.parametrize("group", [16, 64])
def test_capsule_line_contact_joint_law_matches_oracle_bitwise(oracle, group):
    """The capsule pusher of tests/test_contact_kats.py lying along the cube's face -- a line contact: two contact points
    (ShfModel.sph_part) eliminated against the cube together (pair_law_joint) -- centred and 8 cm off the centre line, and
    turned by 25 degrees (a single point, the independent law): shf_sim_step against the oracle, every tensor bit for bit."""
    _need_gpu()
    from shifu_amd.abb_task import box_desc
    from tests import kat_models as K
    sp = H.sim_params()
    rng = np.random.default_rng(11)
    for name, yaw, y0, steps in (("parallel", 0.0, 0.0, 420), ("parallel, off centre", 0.0, 0.08, 420), ("turned", np.deg2rad(25.0), 0.0, 300)):
        cm = K.pusher_model(yaw=yaw)
        m = cm.blob
        assert m.nsph == 2 and [m.sph_part[0], m.sph_part[1]] == [0, 1]
        n = 9
        boxes = [box_desc((0.1, 0.1, 0.1), 0.5, 0.6, False, (0.12, y0, 0.05))]
        sim, dof, root = _scene_on_gpu(cm, sp, boxes, [(0.12, y0, 0.05 - 0.5 * K.G / (4 * K.K_N))], n, group)
        root[1::2, 1] += rng.uniform(-0.003, 0.003, n).astype(np.float32)       # the envs of a wavefront differ
        vt = (0.05 * rng.uniform(0.7, 1.0, n * m.nd)).astype(np.float32)
        sim.tensors[_abi.T_SIM_ROOT].copy_(torch.from_numpy(root))
        both = 0
        for it in range(steps):
            sim.set_dof_command(_abi.T_VEL_TARGET, torch.from_numpy(vt).cuda())
            sim.step()
            sim.refresh(_abi.REFRESH_ALL)
            contact, bstate, _ = oracle.scene_step(m, sp, boxes, n, dof, root, vel_target=vt, friction=np.ones(n, np.float32))
            torch.cuda.synchronize()
            np.testing.assert_array_equal(sim.tensors[_abi.T_DOF_STATE].cpu().numpy(), dof, err_msg=f"{name}: dof step {it}")
            np.testing.assert_array_equal(sim.tensors[_abi.T_ROOT_STATE].cpu().numpy(), root, err_msg=f"{name}: root step {it}")
            np.testing.assert_array_equal(sim.tensors[_abi.T_CONTACT].cpu().numpy(), contact, err_msg=f"{name}: contact step {it}")
            both += int((np.abs(contact.reshape(n, -1, 3)[:, m.nb - 1]).sum(1) > 0).sum())
        assert both > n * 50, f"{name}: the capsule never pushed"
        cube = root.reshape(n, 2, 13)[:, 1]
        if yaw == 0.0:     # (float32, perturbed envs: the float64 known answer is tests/test_contact_kats.py's)
            assert (np.abs(cube[:, 12]) < 2e-2).all(), f"{name}: a line contact does not turn the cube: {cube[:, 12]}"
        sim.destroy() if hasattr(sim, "destroy") else None


def test_abb_scene_with_link_contacts_matches_oracle_bitwise(oracle):
    """The config-5 scene with the arm's links (box stand-ins for their mesh colliders, shifu_amd/assets/
    abb_link_boxes.json) and the rod colliding with table, cube and goal pad: the blind joint ramp that drove every rod
    tip through the table top in the plain scene is now stopped by it, bit for bit as the oracle says."""
    _need_gpu()
    from shifu_amd.abb_task import ABB_BASE_POS, ABB_DEFAULT_DOF_POS, abb_boxes, abb_model
    rng = np.random.default_rng(17)
    cm = abb_model(link_contacts=True)
    m = cm.blob
    assert m.link_collide == 1 and m.nabox == 7 and m.np == 59
    sp = H.sim_params(dt=0.02, angular_damping=0.5)
    boxes = abb_boxes()
    n, A, B = 24, 4, m.nb + 3
    sim, dof, root = _scene_on_gpu(cm, sp, boxes, [b.pos for b in boxes], n, 32)
    dof[:, 0] = np.tile(np.array(ABB_DEFAULT_DOF_POS, np.float32), n) + rng.uniform(-0.05, 0.05, n * m.nd)
    root[0::A, :3] = ABB_BASE_POS
    root[2::A, :3] = np.stack([rng.uniform(0.03, 0.08, n), rng.uniform(-0.03, 0.03, n), rng.uniform(0.125, 0.14, n)], 1)
    sim.tensors[_abi.T_SIM_DOF].copy_(torch.from_numpy(dof))
    sim.tensors[_abi.T_SIM_ROOT].copy_(torch.from_numpy(root))
    fr = np.ones(n, np.float32)
    tgt = dof[:, 0].copy()
    table_touch = False
    for it in range(90):
        if it % 6 == 0:
            tgt = dof[:, 0].copy()
            tgt[1::m.nd] += 0.02
            tgt[2::m.nd] -= 0.01
        sim.set_dof_command(_abi.T_POS_TARGET, torch.from_numpy(tgt).cuda())
        sim.step()
        sim.refresh(_abi.REFRESH_ALL)
        contact, bstate, jac = oracle.scene_step(m, sp, boxes, n, dof, root, pos_target=tgt, friction=fr, want_jacobian=True)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(sim.tensors[_abi.T_DOF_STATE].cpu().numpy(), dof, err_msg=f"dof step {it}")
        np.testing.assert_array_equal(sim.tensors[_abi.T_ROOT_STATE].cpu().numpy(), root, err_msg=f"root step {it}")
        np.testing.assert_array_equal(sim.tensors[_abi.T_CONTACT].cpu().numpy(), contact, err_msg=f"contact step {it}")
        np.testing.assert_array_equal(sim.tensors[_abi.T_BODY_STATE].cpu().numpy(), bstate, err_msg=f"body step {it}")
        table_touch |= bool((contact.reshape(n, B, 3)[:, :m.nb, 2] > 1.0).any())
    assert table_touch, "the ramp must have pressed the arm onto the table"
    tip = bstate.reshape(n, B, 13)[:, m.nb - 1, 2]
    assert (tip > 0.1 - 0.02).all(), f"rod tips stay above the table top (z = 0.1): {tip.min()}"
    assert np.isfinite(root).all() and np.isfinite(dof).all()


@pytest.mark.parametrize("step_kernel", ["body", "split", "split-tgs"])
@pytest.mark.parametrize("link", [False, True])
def test_abb_scene_under_the_velocity_level_solve_matches_oracle_bitwise(oracle, link, step_kernel):
    """The config-5 scene under ShfSimParams.solver = SHF_SOLVER_PGS through gym.simulate: the body-per-lane sub-step with the
    generic solve (csrc/shf_hard.h, k_sim_step<32, BOX, SELF, LINK, HARD>) -- the free cube as a solver body of its own (its
    corners against the table by signed distance, the rod's capsule ends against the cube with impulses on both sides), with
    link contacts the arm's volumes and points against table / cube / pad -- every tensor, bit for bit, 90 sub-steps of the
    blind joint ramp that presses rod and links onto cube and table."""
    _need_gpu()
    from shifu_amd.abb_task import ABB_BASE_POS, ABB_DEFAULT_DOF_POS, abb_boxes, abb_model
    rng = np.random.default_rng(17)
    cm = abb_model(link_contacts=link)
    m = cm.blob
    sp = H.sim_params(dt=0.02, angular_damping=0.5, solver="tgs" if step_kernel.endswith("tgs") else "pgs")
    boxes = abb_boxes()
    n, A, B = 24, 4, m.nb + 3
    sim, dof, root = _scene_on_gpu(cm, sp, boxes, [b.pos for b in boxes], n, 32)
    # "split": gym.simulate on the kernel compiled for this arm + scene (k_sim_step_ws_hard: arm wave + box wave, the solve
    # regrouped -- what the gym facade selects after prepare_sim); "body": the run-time-shaped kernel any scene runs on
    from shifu_amd._lib import lib
    assert lib().shf_sim_step_split_supported(sim._h) == 1
    assert sim.use_split_step() if step_kernel.startswith("split") else sim.mapping == "body"
    dof[:, 0] = np.tile(np.array(ABB_DEFAULT_DOF_POS, np.float32), n) + rng.uniform(-0.05, 0.05, n * m.nd)
    root[0::A, :3] = ABB_BASE_POS
    root[2::A, :3] = np.stack([rng.uniform(0.03, 0.08, n), rng.uniform(-0.03, 0.03, n), rng.uniform(0.125, 0.14, n)], 1)
    yaw = rng.uniform(-np.pi, np.pi, n)
    root[2::A, 5], root[2::A, 6] = np.sin(yaw / 2), np.cos(yaw / 2)
    sim.tensors[_abi.T_SIM_DOF].copy_(torch.from_numpy(dof))
    sim.tensors[_abi.T_SIM_ROOT].copy_(torch.from_numpy(root))
    fr = np.ones(n, np.float32)
    tgt = dof[:, 0].copy()
    touched = cube_held = False
    oracle.dropped(reset=True)
    for it in range(90):
        if it % 6 == 0:
            tgt = dof[:, 0].copy()
            tgt[1::m.nd] += 0.02
            tgt[2::m.nd] -= 0.01
        sim.set_dof_command(_abi.T_POS_TARGET, torch.from_numpy(tgt).cuda())
        sim.step()
        sim.refresh(_abi.REFRESH_ALL)
        contact, bstate, jac = oracle.scene_step(m, sp, boxes, n, dof, root, pos_target=tgt, friction=fr, want_jacobian=True)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(sim.tensors[_abi.T_DOF_STATE].cpu().numpy(), dof, err_msg=f"dof step {it}")
        np.testing.assert_array_equal(sim.tensors[_abi.T_ROOT_STATE].cpu().numpy(), root, err_msg=f"root step {it}")
        np.testing.assert_array_equal(sim.tensors[_abi.T_CONTACT].cpu().numpy(), contact, err_msg=f"contact step {it}")
        np.testing.assert_array_equal(sim.tensors[_abi.T_BODY_STATE].cpu().numpy(), bstate, err_msg=f"body step {it}")
        c = contact.reshape(n, B, 3)
        assert np.isfinite(c).all() and np.isfinite(dof).all(), f"step {it}"      # (assert_array_equal counts nan == nan)
        touched |= bool(np.abs(c[:, :m.nb]).sum() > 0)
        cube_held |= bool((np.abs(c[:, m.nb + 1, 2] - 0.981) < 0.05).mean() > 0.5)
    assert touched, "the arm met a cube / the table"
    assert cube_held, "the table carries the cubes' weight"
    assert np.isfinite(root).all() and np.isfinite(dof).all()
    torch.cuda.synchronize()
    assert int(sim.tensors[_abi.T_DROPPED].sum()) == oracle.dropped()


def test_generic_articulation_under_the_velocity_level_solve_matches_oracle_bitwise(oracle):
    """An articulation that is NOT the A1's shape -- the 13-body variant (feet merged into the shanks), self-collision on,
    thrown onto rough terrain -- through gym.simulate under SHF_SOLVER_PGS: the generic solve without box actors (floating
    root: its LDL^T factors answer every column), capsule-pair constraints between two bodies of the tree."""
    _need_gpu()
    from shifu_amd.model import asset_path, compile_urdf
    cm = compile_urdf(asset_path("a1.urdf"), honour_dont_collapse=False, default_dof_drive_mode=_abi.DOF_MODE_EFFORT, self_collision=True)
    m = cm.blob
    assert m.nb == 13
    for d in range(m.nd):
        m.damping[d] = 0.5
    rng = np.random.default_rng(34)
    sp = H.sim_params(angular_damping=0.5, solver="pgs")
    n = 40
    terr, hs = _terrain(rng, rough=True)
    dof, root = _random_states(m, n, rng, z_lo=0.1, z_hi=0.5)
    lo, up = np.array(m.lower[:m.nd]), np.array(m.upper[:m.nd])
    dof[:, 0] = rng.uniform(lo, up, (n, m.nd)).astype(np.float32).reshape(-1)
    dof[:, 1] = rng.uniform(-6, 6, n * m.nd)
    fr = rng.uniform(0.5, 1.25, n).astype(np.float32)
    sim = _make_sim(cm, sp, n, terr, hs, group=32)
    T = sim.tensors
    T[_abi.T_SIM_DOF].copy_(torch.from_numpy(dof))
    T[_abi.T_SIM_ROOT].copy_(torch.from_numpy(root))
    T[_abi.T_FRICTION].copy_(torch.from_numpy(fr))
    hits = 0
    for it in range(60):
        eff = rng.uniform(-25, 25, n * m.nd).astype(np.float32)
        sim.set_dof_command(_abi.T_EFFORT, torch.from_numpy(eff).cuda())
        sim.step()
        sim.refresh(_abi.REFRESH_ALL)
        contact, bstate = oracle.step(m, sp, n, dof, root, effort=eff, friction=fr, terrain=terr, heights=hs, want_contact=True,
                                      want_body_state=True)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(T[_abi.T_DOF_STATE].cpu().numpy(), dof, err_msg=f"dof step {it}")
        np.testing.assert_array_equal(T[_abi.T_ROOT_STATE].cpu().numpy(), root, err_msg=f"root step {it}")
        np.testing.assert_array_equal(T[_abi.T_CONTACT].cpu().numpy(), contact, err_msg=f"contact step {it}")
        hits += int((np.abs(contact).sum(1) > 0).sum())
    assert hits > 500 and np.isfinite(root).all()
    sim.destroy()


@pytest.mark.parametrize("mapping", ["split", "body"])
@pytest.mark.parametrize("link", [False, True])
def test_fused_abb_step_under_the_velocity_level_solve_matches_oracle_bitwise(oracle, link, mapping):
    """ShifuVecEnv.step for AbbPushBox with FusedAbbEnv(solver='pgs') -- k_abb_step<32, DynDims, DynScene, LINK, 0, HARD> --
    against the oracle: 120 vec-steps with re-spawns, every tensor."""
    _need_gpu()
    from shifu_amd.gym.abb_fused import FusedAbbEnv
    n = 48
    env = FusedAbbEnv(num_envs=n, seed=11, link_contacts=link, solver="pgs", **({"mapping": "body"} if mapping == "body" else {}))
    assert env.sim_params.solver == _abi.SOLVER_PGS and env.mapping == mapping
    assert ("Lb1ELb0EE" in env.task.kernel_symbol() or "pgs_wide" in env.task.kernel_symbol()) if mapping == "body" else ("ws_hard" in env.task.kernel_symbol())
    bufs = {k: env.sim.tensors[t].cpu().numpy().copy() for k, t in _ABB_SIM_T.items()}
    bufs.update({k: env.task.tensors[t].cpu().numpy().copy() for k, t in _ABB_T.items()})
    rng = np.random.default_rng(2)
    resets = 0
    for it in range(120):
        raw = (2 * rng.random((n, 3)) - 1).astype(np.float32) * 1.3
        env.task.step(torch.from_numpy(raw).cuda())
        oracle.abb_step(env.cm.blob, env.sim_params, env.boxes, env.task_params, n, 0, bufs, raw)
        if it % 10 == 9 or it < 3:
            torch.cuda.synchronize()
            for k, t in list(_ABB_SIM_T.items()) + list(_ABB_T.items()):
                got = (env.sim.tensors if k in _ABB_SIM_T else env.task.tensors)[t].cpu().numpy().reshape(bufs[k].shape)
                np.testing.assert_array_equal(got, bufs[k], err_msg=f"{k} step {it}")
        resets += int(bufs["reset"].sum())
    assert np.isfinite(bufs["obs"]).all()


_ABB_SIM_T = {"dof_state": _abi.T_DOF_STATE, "root_state": _abi.T_ROOT_STATE, "body_state": _abi.T_BODY_STATE,
              "contact": _abi.T_CONTACT, "jacobian": _abi.T_JACOBIAN, "friction": _abi.T_FRICTION}
_ABB_T = {"actions": _abi.ABB_ACTIONS, "obs": _abi.ABB_OBS, "rew": _abi.ABB_REW, "reset": _abi.ABB_RESET,
          "timeout": _abi.ABB_TIMEOUT, "success": _abi.ABB_SUCCESS, "ep_len": _abi.ABB_EP_LEN,
          "rew_sums": _abi.ABB_REW_SUMS, "dof_targets": _abi.ABB_DOF_TARGETS, "reset_count": _abi.ABB_RESET_COUNT,
          "done_sums": _abi.ABB_DONE_SUMS}


@pytest.mark.parametrize("group,generic", [(64, False), (32, False), (16, False), (16, "chain"), (32, "levels"), (16, "levels"), (32, True),
                                           (64, True), (16, "link"), (16, "link-split"), (32, "link"), (64, "link"), (32, "link-generic")])
def test_fused_abb_step_matches_oracle_bitwise(oracle, group, generic):
    """ShifuVecEnv.step for AbbPushBox (config 5): in-kernel damped-least-squares IK on the Jacobian
    tensor, 6 sub-steps with implicit POS drives and box contacts, refresh, termination, rewards,
    Philox re-spawn of cube / goal, observation -- 150 vec-steps, every tensor compared exactly."""
    _need_gpu()
    from shifu_amd.gym.abb_fused import FusedAbbEnv
    n = 48
    # generic: a fourth (fixed, out of reach) box makes the scene differ from the compile-time ABB scene, so the step
    # runs on k_abb_step<G, DynDims, DynScene> (LDS flags) instead of <G, AbbDims, AbbScene> (ballots): same results
    from shifu_amd.abb_task import box_desc
    # "link": the arm's links and rod also collide with table / cube / goal pad (ShfModel.link_collide): the run-time-shaped
    # kernel with the link-contact pass
    # ("link": the shipped arm and scene, compile-time shaped -- the default of FusedAbbEnv since round 4; "link-generic": a
    # fourth box puts the same on the run-time-shaped instantiation)
    link_generic = generic == "link-generic"
    link_split = generic == "link-split"          # arm wave + box wave, the link passes on the box wave (k_abb_step_ws<512, true>)
    link = generic in ("link", "link-split") or link_generic
    # "levels": the compile-time arm and scene on the level-by-level sub-step (mapping 'body'); False at 16 / 32 lanes takes
    # the default, the arm's recursions on one lane (csrc/shf_arm.h)
    levels = generic == "levels"
    # False at 16 lanes takes the default there: arm and boxes of an env on different waves of the workgroup
    # (k_abb_step_ws, mapping 'split'); "chain": the one-wave form of the same at 16 lanes
    chain = generic == "chain"
    split = generic is False and group == 16
    generic = generic is True or link_generic
    extra = [box_desc([0.05, 0.05, 0.02], 0.0, 0.5, True, [0.25, 0.25, 0.11])] if generic else []
    env = FusedAbbEnv(num_envs=n, seed=11, group=group, extra_boxes=extra, link_contacts=link,
                      mapping="body" if (levels or (link and not link_split)) else ("chain" if chain else ("split" if link_split else None)),
                      solver="compliant")       # (the kernel forms of the compliant law; the velocity-level solve has its own tests)
    assert env.mapping == ("split" if (split or link_split) else "chain" if (not generic and not link and not levels and group < 64) else "body")
    assert ("FixedDims" in env.task.kernel_symbol() or split or link_split) != bool(generic)
    assert ("Lb1ELi0ELb0ELb0EE" in env.task.kernel_symbol()) == (link and not generic and not link_split)
    assert env.task.kernel_symbol().endswith("Li6ELb0ELb0EE") == (env.mapping == "chain")
    env.task.tensors[_abi.ABB_EP_LEN].copy_(torch.randint(0, 200, (n,)))   # staggered time-outs
    torch.cuda.synchronize()
    bufs = {k: env.sim.tensors[t].cpu().numpy().copy() for k, t in _ABB_SIM_T.items()}
    bufs.update({k: env.task.tensors[t].cpu().numpy().copy() for k, t in _ABB_T.items()})
    rng = np.random.default_rng(2)
    resets = successes = 0
    for it in range(150):
        raw = (2 * rng.random((n, 3)) - 1).astype(np.float32) * 1.3
        raw[: n // 2, 0] = np.abs(raw[: n // 2, 0])          # half the arms keep pushing +x
        slot = env.task.step(torch.from_numpy(raw).cuda())
        oracle.abb_step(env.cm.blob, env.sim_params, env.boxes, env.task_params, n, 0, bufs, raw)
        torch.cuda.synchronize()
        for k, t in list(_ABB_SIM_T.items()) + list(_ABB_T.items()):
            got = (env.sim.tensors if k in _ABB_SIM_T else env.task.tensors)[t].cpu().numpy().reshape(bufs[k].shape)
            np.testing.assert_array_equal(got, bufs[k], err_msg=f"{k} step {it}")
        np.testing.assert_array_equal(env.task.tensors[_abi.ABB_STATS][slot].cpu().numpy(),
                                      oracle.abb_stats(env.task_params, n, bufs["done_sums"]))
        resets += int(bufs["reset"].sum()); successes += int(bufs["success"].sum())
    assert resets > 10, "the run must exercise resets"
    assert np.isfinite(bufs["obs"]).all() and np.isfinite(bufs["root_state"]).all()
    cube_moved = np.abs(bufs["root_state"].reshape(n, -1, 13)[:, 2, 7:10]).sum() > 0 or resets > 0
    assert cube_moved


@pytest.mark.parametrize("group", [32, "chain16", "pgs"])
@pytest.mark.parametrize("n", [1, 5, 37])
def test_ragged_env_counts(oracle, n, group):
    """Env counts that do not fill a block (4, 8 or 16 envs per 256-thread block; an odd count leaves half a wavefront of the
    two-envs-per-wave kernels empty): no out-of-range lanes."""
    _need_gpu()
    group, kw = _pgs_group(group)
    cm, sp, tp, terr, hs, bufs, sim, task, rng = _a1_setup(n, True, seed=30 + n, group=group, **kw)
    for it in range(8):
        raw = (2 * rng.random((n, cm.blob.nd)) - 1).astype(np.float32)
        task.step(torch.from_numpy(raw).cuda())
        oracle.a1_step(cm.blob, sp, tp, n, 0, bufs, raw, terrain=terr, heights=hs)
    _compare(sim, task, bufs, f"n={n}")


@pytest.mark.parametrize("kind", ["a1", "a1-chain", "a1-pgs", "abb-split", "abb-levels"])
def test_random_action_step_matches_oracle_bitwise(oracle, kind):
    """shf_a1_step_random / shf_abb_step_random -- run_policy('random') with the U(-1, 1) actions drawn inside the launch
    (Philox4x32-10, counter = global env id, vec-step index, dof) -- against the oracle fed with the actions the oracle's
    own restatement of that generator gives (oracle.random_actions): every tensor bit for bit, global ids offset as on
    rank 1 of a sharded run."""
    _need_gpu()
    n, off = 21, 4096
    if kind.startswith("a1"):
        from shifu_amd.gym.a1_fused import FusedA1Env
        solver = "pgs" if kind.endswith("pgs") else "compliant"
        env = FusedA1Env(num_envs=n, rank=1, world_size=2, seed=9, group=32, mapping="body" if kind == "a1" else "chain", solver=solver)
        off = env.env_id_offset
        env.reset()
    else:
        from shifu_amd.gym.abb_fused import FusedAbbEnv
        env = FusedAbbEnv(num_envs=n, rank=1, world_size=2, seed=9, group=16, link_contacts=False, mapping="split" if kind.endswith("split") else "body")
        off = env.env_id_offset
    # the generator itself: two runs of the oracle's restatement agree with each other and differ between steps / envs
    a0 = oracle.random_actions(9, n, off, 0, env.num_actions)
    a1 = oracle.random_actions(9, n, off, 1, env.num_actions)
    assert a0.shape == (n, env.num_actions) and (np.abs(a0) <= 1).all() and not np.array_equal(a0, a1)
    assert len(np.unique(a0)) > 0.9 * a0.size
    if kind.startswith("abb"):
        bufs = {k: env.sim.tensors[t].cpu().numpy().copy() for k, t in _ABB_SIM_T.items()}
        bufs.update({k: env.task.tensors[t].cpu().numpy().copy() for k, t in _ABB_T.items()})
        for it in range(20):
            step = env.task.step_index
            env.task.step_random()
            raw = oracle.random_actions(env.task_params.seed, n, off, step, 3)
            oracle.abb_step(env.cm.blob, env.sim_params, env.boxes, env.task_params, n, off, bufs, raw)
        torch.cuda.synchronize()
        for k, t in list(_ABB_SIM_T.items()) + list(_ABB_T.items()):
            got = (env.sim.tensors if k in _ABB_SIM_T else env.task.tensors)[t].cpu().numpy().reshape(bufs[k].shape)
            np.testing.assert_array_equal(got, bufs[k], err_msg=f"{kind}: {k}")
        return
    # A1: two envs built alike -- one stepped with in-kernel actions, the other fed the oracle's actions through the
    # tensor entry point (itself held to the oracle by the tests above): identical tensors
    ref = FusedA1Env(num_envs=n, rank=1, world_size=2, seed=9, group=32, mapping="chain" if solver == "pgs" else "body", solver=solver)
    ref.reset()
    for it in range(25):
        step = env.task.step_index
        assert step == ref.task.step_index
        env.step_random()
        raw = oracle.random_actions(env.task_params.seed, n, off, step, 12)
        ref.step(torch.from_numpy(raw).cuda())
    torch.cuda.synchronize()
    for tid in range(_abi.A1_COUNT):
        if tid in (_abi.A1_PARAMS,):
            continue
        a, b = env.task.tensors[tid], ref.task.tensors[tid]
        assert torch.equal(a, b), f"{kind}: task tensor {tid}"
    for tid in (_abi.T_DOF_STATE, _abi.T_ROOT_STATE, _abi.T_BODY_STATE, _abi.T_CONTACT):
        assert torch.equal(env.sim.tensors[tid], ref.sim.tensors[tid]), f"{kind}: sim tensor {tid}"
    assert int(ref.task.tensors[_abi.A1_RESET_COUNT].sum()) >= 0


@pytest.mark.parametrize("mapping,group", [("split", 16), ("chain", 16), ("chain", 32), ("pgs-link", 32), ("pgs", 32), ("pgs-link-split", 16), ("pgs-split", 16)])
@pytest.mark.parametrize("n", [1, 5, 13, 37])
def test_ragged_env_counts_abb(oracle, n, mapping, group):
    """The fused ABB step with env counts that leave lanes -- and, in the two-wave kernel, whole waves -- without an env
    (8 envs per workgroup there: the waves of dead envs still have to meet every workgroup barrier).  "pgs-link": the
    velocity-level solve at sixteen envs per 512-thread workgroup (k_abb_step_pgs_wide); "pgs": rod-only, eight per workgroup."""
    _need_gpu()
    from shifu_amd.gym.abb_fused import FusedAbbEnv
    if mapping.endswith("-split"):        # k_abb_step_ws_hard: sixteen envs per workgroup, regrouped for the solve
        env = FusedAbbEnv(num_envs=n, seed=17 + n, solver="pgs", link_contacts="link" in mapping)
        assert "ws_hard" in env.task.kernel_symbol()
        mapping = "split"
    elif mapping.startswith("pgs"):
        env = FusedAbbEnv(num_envs=n, seed=17 + n, solver="pgs", link_contacts=mapping == "pgs-link", mapping="body")
        assert ("pgs_wide" in env.task.kernel_symbol()) == (mapping == "pgs-link")
        mapping = "body"
    else:
        env = FusedAbbEnv(num_envs=n, seed=17 + n, group=group, mapping=mapping)
    assert env.mapping == mapping
    bufs = {k: env.sim.tensors[t].cpu().numpy().copy() for k, t in _ABB_SIM_T.items()}
    bufs.update({k: env.task.tensors[t].cpu().numpy().copy() for k, t in _ABB_T.items()})
    rng = np.random.default_rng(n)
    for it in range(12):
        raw = (2 * rng.random((n, 3)) - 1).astype(np.float32)
        env.task.step(torch.from_numpy(raw).cuda())
        oracle.abb_step(env.cm.blob, env.sim_params, env.boxes, env.task_params, n, 0, bufs, raw)
    torch.cuda.synchronize()
    for k, t in list(_ABB_SIM_T.items()) + list(_ABB_T.items()):
        got = (env.sim.tensors if k in _ABB_SIM_T else env.task.tensors)[t].cpu().numpy().reshape(bufs[k].shape)
        np.testing.assert_array_equal(got, bufs[k], err_msg=f"{k} n={n}")


def test_split_mapping_is_refused_for_other_shapes():
    """SHF_MAP_CHAIN_SPLIT is compiled for the ABB arm in its scene: the A1 is refused when the mapping is set, the ABB with a
    fourth box at its first step -- loudly, not by falling back."""
    _need_gpu()
    from shifu_amd._lib import BackendError
    from shifu_amd.abb_task import box_desc
    from shifu_amd.gym.a1_fused import FusedA1Env
    from shifu_amd.gym.abb_fused import FusedAbbEnv
    with pytest.raises(BackendError, match="split chain mapping"):
        FusedA1Env(num_envs=8, group=16, mapping="split")
    extra = [box_desc([0.05, 0.05, 0.02], 0.0, 0.5, True, [0.25, 0.25, 0.11])]
    env = FusedAbbEnv(num_envs=8, group=16, mapping="split", link_contacts=False, extra_boxes=extra)
    with pytest.raises(BackendError, match="chain mapping needs"):
        env.task.step(torch.zeros(8, 3, device="cuda"))
    env = FusedAbbEnv(num_envs=8, group=16, mapping="split", link_contacts=True, extra_boxes=extra)     # (k_abb_step_ws<512, true>)
    with pytest.raises(BackendError, match="split mapping with link contacts needs"):
        env.task.step(torch.zeros(8, 3, device="cuda"))


@pytest.mark.parametrize("group", [32, "chain", "pgs"])
def test_full_size_determinism_and_shard_invariance(group):
    """BASELINE size (4096 envs, procedural 1300x2100 height map): two runs are bit-identical, and
    two 2048-env shards with global-id offsets reproduce the 4096-env run env for env.  32: the body-per-lane kernel;
    "chain": the chain-per-lane kernel at 16 lanes per env."""
    _need_gpu()
    from shifu_amd.gym.a1_fused import FusedA1Env
    N, steps = 4096, 40
    g = torch.Generator(device="cuda:0")

    def run(num, rank, world, acts=None):
        env = (FusedA1Env(num_envs=num, rank=rank, world_size=world, seed=42, mapping="chain", group=16) if group == "chain" else
               FusedA1Env(num_envs=num, rank=rank, world_size=world, seed=42) if group == "pgs" else      # the env's defaults: k_a1_chain_pgs
               FusedA1Env(num_envs=num, rank=rank, world_size=world, seed=42, group=group, mapping="body"))
        assert env.mapping == ("body" if group == 32 else "chain") and (env.sim_params.solver == _abi.SOLVER_TGS) == (group == "pgs")      # ("pgs": the env's default solve -- TGS since round 6)
        env.reset()
        out = []
        for k in range(steps):
            a = acts[k][rank * num:(rank + 1) * num].contiguous()
            obs, _, rew, done, _ = env.step(a)
            out.append((obs.clone(), rew.clone(), done.clone()))
        res = (torch.stack([o for o, _, _ in out]), torch.stack([r for _, r, _ in out]),
               torch.stack([d for _, _, d in out]), env.task.tensors[_abi.A1_RESET_COUNT].clone())
        env.destroy()
        return res

    g.manual_seed(1)
    acts = [2 * torch.rand(N, 12, device="cuda:0", generator=g) - 1 for _ in range(steps)]
    full1 = run(N, 0, 1, acts)
    full2 = run(N, 0, 1, acts)
    for a, b in zip(full1, full2):
        assert torch.equal(a, b), "run-to-run determinism"
    halves = [run(N // 2, r, 2, acts) for r in range(2)]
    for k in range(4):
        cat = torch.cat([halves[0][k], halves[1][k]], dim=-1 if k in (1, 2) else (1 if k == 0 else 0))
        assert torch.equal(cat, full1[k]), f"shard invariance, output {k}"
    assert torch.isfinite(full1[0]).all()
    assert int(full1[3].sum()) > N, "resets must have happened"


@pytest.mark.parametrize("group", [16, 32])
def test_fused_step_generic_dimension_path(oracle, group):
    """A robot that is not the compiled-in A1 layout (here: the A1 with its feet collapsed into the shanks, 13 bodies)
    takes the run-time-dimension instantiation of the fused kernel -- level-loop kinematics, rolled contact loops,
    flag-driven folds -- and, with 13 bodies, also fits four envs per wavefront."""
    _need_gpu()
    from shifu_amd.model import asset_path, compile_urdf
    cm = compile_urdf(asset_path("a1.urdf"), default_dof_drive_mode=_abi.DOF_MODE_EFFORT, honour_dont_collapse=False)
    assert cm.blob.nb == 13
    n = 40
    cm, sp, tp, terr, hs, bufs, sim, task, rng = _a1_setup(n, True, seed=21, group=group, cm=cm)
    for it in range(60):
        raw = (2 * rng.random((n, cm.blob.nd)) - 1).astype(np.float32) * 1.5
        task.step(torch.from_numpy(raw).cuda())
        oracle.a1_step(cm.blob, sp, tp, n, 0, bufs, raw, terrain=terr, heights=hs)
        _compare(sim, task, bufs, f"step {it}")
    assert np.isfinite(bufs["obs"]).all()


@pytest.mark.parametrize("group", [32, 16, "chain16", "chain32", "chain32-self", "pgs", "pgs-self"])
def test_trimesh_terrain_matches_oracle_bitwise(oracle, group):
    """the trimesh form of the terrain (vertical risers at steep steps; ShfTerrain.warped) -- simulate
    and the fused step against the oracle, on a terrain whose plateau and noise shift many vertices."""
    _need_gpu()
    from shifu_amd.a1_task import a1_task_params
    from shifu_amd.backend import A1Task
    from shifu_amd.isaacgym.terrain_utils import pack_trimesh_samples, trimesh_warp_map
    from shifu_amd.model import asset_path, compile_urdf
    rng = np.random.default_rng(31)
    # "chain32-self": the reference's effective A1 scene -- trimesh terrain (task_config.py:53) with every link colliding
    # (units.py:68) -- on the chain-per-lane kernel
    pgs = group in ("pgs", "pgs-self")          # ... and under the velocity-level solve: k_a1_chain_pgs<true, SELF>
    selfc = group in ("chain32-self", "pgs-self")
    if selfc or pgs:
        group = "chain32"
    if selfc:
        cm = compile_urdf(asset_path("a1.urdf"), default_dof_drive_mode=_abi.DOF_MODE_EFFORT, self_collision=True)
        for d in range(cm.blob.nd):
            cm.blob.damping[d] = 0.5
    elif group in (32, "chain16", "chain32"):
        cm = H.a1_model()
    else:
        cm = compile_urdf(asset_path("a1.urdf"), default_dof_drive_mode=_abi.DOF_MODE_EFFORT, honour_dont_collapse=False)
    sp = H.sim_params(angular_damping=0.5, solver="pgs" if pgs else "compliant")
    tp = a1_task_params(cm, num_rows=4, num_cols=5, env_length=0.8)
    terr, hs = _terrain(rng, rows=80, cols=60, rough=True)
    warp = trimesh_warp_map(hs, terr.hscale, terr.vscale, 0.75)
    assert ((warp & 15) != 5).mean() > 0.02, "the test terrain must move vertices"
    terr.warped = 1
    packed = pack_trimesh_samples(hs, warp)
    n = 48
    bufs = _a1_buffers(cm, tp, n, rng, terr.rows, terr.cols)
    sim = _make_sim(cm, sp, n, terr, hs, group=group, warp=warp)
    assert sim.terrain.warped == 1 and sim.tensors[_abi.T_HEIGHTS].numel() == packed.size
    task = A1Task(sim, tp)
    if selfc or pgs:
        assert task.kernel_symbol() == ("_Z14k_a1_chain_pgsILb1ELb%dEE" % int(selfc) if pgs else "_Z10k_a1_chainILi32ELb1ELb1EE")
    _upload(sim, task, bufs)
    for it in range(60):
        raw = (2 * rng.random((n, cm.blob.nd)) - 1).astype(np.float32) * (2.0 if selfc else 1.5)
        task.step(torch.from_numpy(raw).cuda())
        oracle.a1_step(cm.blob, sp, tp, n, 0, bufs, raw, terrain=terr, heights=packed)
        _compare(sim, task, bufs, f"step {it}")
    assert np.isfinite(bufs["obs"]).all()
    # and it is a different surface from the height field: drop the same robots once on each
    terr0, _ = _terrain(np.random.default_rng(31), rows=80, cols=60, rough=True)
    dw, rw = _random_states(cm.blob, n, np.random.default_rng(5), z_lo=0.05, z_hi=0.25, xy_hi=5.0)
    dh, rh = dw.copy(), rw.copy()
    oracle.step(cm.blob, sp, n, dw, rw, nsteps=3, terrain=terr, heights=packed)
    oracle.step(cm.blob, sp, n, dh, rh, nsteps=3, terrain=terr0, heights=hs)
    assert not np.array_equal(rw, rh)


# ---------------------------------------------------------------- contact KATs on the GPU --
def _kat_run(oracle, cm, sp, dof0, root0, steps, mu_shape=1.0, effort=None, group=16):
    """The same single-env scenario on the HIP kernel (through the C ABI) and on the float oracle: asserts bit-equal
    state every step and returns the trajectory."""
    m = cm.blob
    dof, root = dof0.astype(np.float32).copy(), root0.astype(np.float32).copy()
    fr = np.full(1, mu_shape, np.float32)
    sim = _make_sim(cm, sp, 1, group=group)
    T = sim.tensors
    T[_abi.T_SIM_DOF].copy_(torch.from_numpy(dof))
    T[_abi.T_SIM_ROOT].copy_(torch.from_numpy(root))
    T[_abi.T_FRICTION].copy_(torch.from_numpy(fr))
    if effort is not None:
        sim.set_dof_command(_abi.T_EFFORT, torch.from_numpy(effort).cuda())
    traj_r, traj_q = [], []
    for k in range(steps):
        sim.step()
        oracle.step(m, sp, 1, dof, root, effort=effort, friction=fr)
        if k % 10 == 9 or k == steps - 1:
            sim.refresh(_abi.REFRESH_DOF | _abi.REFRESH_ROOT)
            torch.cuda.synchronize()
            np.testing.assert_array_equal(T[_abi.T_ROOT_STATE].cpu().numpy(), root, err_msg=f"root step {k}")
            if m.nd:
                np.testing.assert_array_equal(T[_abi.T_DOF_STATE].cpu().numpy(), dof, err_msg=f"dof step {k}")
        traj_r.append(root[0].copy()); traj_q.append(dof[:, 0].copy() if m.nd else None)
    sim.destroy()
    return np.array(traj_r), traj_q


def test_contact_kats_stick_slip_impact_on_the_gpu(oracle):
    """tests/test_contact_kats.py on the HIP path: the kernel equals the float oracle bit for bit on every scenario,
    and the analytic answers are asserted on what the GPU produced: creep v_eps tan(theta)/mu below the friction
    angle, Coulomb acceleration above it, mu g deceleration and stopping distance, inelastic sphere impact with
    m g / k rest penetration."""
    _need_gpu()
    from tests import kat_models as K
    blk = K.block_model()
    nodof = np.zeros((0, 2), np.float32)

    def block(theta, mu_shape, steps, lin=(0, 0, 0)):
        sp = H.sim_params(gravity=(K.G * np.sin(theta), 0.0, -K.G * np.cos(theta)))
        pen = 2.0 * K.G * np.cos(theta) / (4 * K.K_N)
        return _kat_run(oracle, blk, sp, nodof, K.root_row((0, 0, 0.05 - pen), lin=lin), steps, mu_shape)[0]
    tr = block(np.arctan(0.3), 0.2, 600)
    assert abs(tr[-1, 7] - K.V_EPS * 0.3 / 0.6) < 0.02 * K.V_EPS and np.abs(tr[-1, 10:13]).max() < 1e-4
    th = np.arctan(1.0)
    tr = block(th, 0.2, 200)
    a = K.G * (np.sin(th) - 0.6 * np.cos(th))
    assert abs((tr[-1, 7] - tr[99, 7]) / 0.5 - a) < 0.012 * a
    tr = block(0.0, 0.6, 200, lin=(1.0, 0, 0))
    d = 1.0 / (2 * 0.8 * K.G)
    assert d < tr[-1, 0] < 1.08 * d and abs(tr[-1, 7]) < 1e-4
    tr = _kat_run(oracle, K.ball_model(), H.sim_params(), nodof, K.root_row((0, 0, 0.55)), 400)[0]
    z, vz = tr[:, 2] - 0.05, tr[:, 9]
    hit = int(np.argmax(z < 0))
    assert vz[hit:].max() < 0.05 * 3.13 and z[hit:].max() < 0.0 and z.min() > -0.003
    assert abs(-z[-1] - K.G / K.K_N) < 0.05 * K.G / K.K_N


def test_contact_kats_joint_limit_on_the_gpu(oracle):
    _need_gpu()
    from tests import kat_models as K
    cm = K.limit_model()
    tr, q = _kat_run(oracle, cm, H.sim_params(), np.zeros((1, 2), np.float32), K.root_row((0, 0, 1.0)), 600,
                     effort=np.full(1, 30.0, np.float32))
    assert abs(q[-1][0] - (0.5 + 30.0 / 2000.0)) < 2e-4 and max(x[0] for x in q) < 0.58


# -------------------------------------------------------------------- self-collision --
@pytest.mark.parametrize("group,dyn", [(32, False), ("chain32", False), (32, True), (16, True)])
def test_self_collision_matches_oracle_bitwise(oracle, group, dyn):
    """capsule-pair self-collision (collision filter 0, reference units.py:68) on the HIP path equals the
    oracle bit for bit -- gym.simulate from states with the legs folded through each other, then the fused A1 step with
    large actions; A1 instantiation and the run-time-dimension one (a 13-body variant for 16 lanes)."""
    _need_gpu()
    from shifu_amd.model import asset_path, compile_urdf
    kw = dict(default_dof_drive_mode=_abi.DOF_MODE_EFFORT, self_collision=True)
    if dyn:
        cm = compile_urdf(asset_path("a1.urdf"), honour_dont_collapse=False, **kw)      # feet merged into the shanks: 13 bodies,
        assert cm.blob.nb == 13                                                         # not the compile-time A1 shape
    else:
        cm = compile_urdf(asset_path("a1.urdf"), **kw)
    m = cm.blob
    assert m.self_collide == 1 and m.npair > 40
    for d in range(m.nd):
        m.damping[d] = 0.5
    rng = np.random.default_rng(21)
    sp = H.sim_params(angular_damping=0.5)
    n = 48
    terr, hs = _terrain(rng, rough=False)
    dof, root = _random_states(m, n, rng, z_lo=0.5, z_hi=0.9)
    lo, up = np.array(m.lower[:m.nd]), np.array(m.upper[:m.nd])
    q = rng.uniform(lo, up, (n, m.nd)).astype(np.float32)          # anywhere inside the joint limits: legs cross
    dof[:, 0] = q.reshape(-1)
    dof[:, 1] = rng.uniform(-6, 6, n * m.nd)
    fr = rng.uniform(0.5, 1.25, n).astype(np.float32)
    sim = _make_sim(cm, sp, n, None, None, group=group)
    T = sim.tensors
    T[_abi.T_SIM_DOF].copy_(torch.from_numpy(dof))
    T[_abi.T_SIM_ROOT].copy_(torch.from_numpy(root))
    T[_abi.T_FRICTION].copy_(torch.from_numpy(fr))
    selfhits = 0
    for it in range(40):
        eff = rng.uniform(-25, 25, n * m.nd).astype(np.float32)
        sim.set_dof_command(_abi.T_EFFORT, torch.from_numpy(eff).cuda())
        sim.step()
        sim.refresh(_abi.REFRESH_ALL)
        contact, bstate = oracle.step(m, sp, n, dof, root, effort=eff, friction=fr, want_contact=True, want_body_state=True)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(T[_abi.T_DOF_STATE].cpu().numpy(), dof, err_msg=f"dof step {it}")
        np.testing.assert_array_equal(T[_abi.T_ROOT_STATE].cpu().numpy(), root, err_msg=f"root step {it}")
        np.testing.assert_array_equal(T[_abi.T_CONTACT].cpu().numpy(), contact, err_msg=f"contact step {it}")
        selfhits += int((np.abs(contact).sum(1) > 0).sum())       # the bodies are in the air: every force is a self-contact
    assert selfhits > 200, selfhits
    sim.destroy()
    if group == 16:
        return
    # the fused step with self-collision on
    cm2, sp2, tp, terr, hs, bufs, sim, task, rng = _a1_setup(64, True, group=group, cm=cm)
    hits = 0
    for it in range(60):
        raw = (2 * rng.random((64, m.nd)) - 1).astype(np.float32) * 2.0
        task.step(torch.from_numpy(raw).cuda())
        oracle.a1_step(m, sp2, tp, 64, 0, bufs, raw, terrain=terr, heights=hs)
        _compare(sim, task, bufs, f"fused self-collision step {it}")
    if group == "chain32":     # the chain-per-lane kernel with the self-collision pass (round 4): k_a1_chain<32, TW, true>
        assert task.kernel_symbol() == "_Z10k_a1_chainILi32ELb0ELb1EE"
    else:
        assert "self" in task.kernel_symbol() and ("DynDims" in task.kernel_symbol()) == dyn


@pytest.mark.parametrize("kmax", [8, 16])
@pytest.mark.parametrize("selfc", [False, True])
def test_velocity_level_solve_simulate_and_self_collision_match_oracle_bitwise(oracle, selfc, kmax):
    """ShfSimParams.solver = SHF_SOLVER_PGS through gym.simulate (k_sim_step_chain_pgs: the hook path's sub-step) from thrown,
    folded states with random efforts and pushes -- ground contacts and, with self-collision, capsule-pair constraints between
    two bodies of the tree (impulses on both sides of the response matrix) -- then the fused step with self-collision on the
    trimesh-free rough terrain: every tensor, bit for bit."""
    _need_gpu()
    from shifu_amd.model import asset_path, compile_urdf
    cm = compile_urdf(asset_path("a1.urdf"), default_dof_drive_mode=_abi.DOF_MODE_EFFORT, self_collision=selfc)
    m = cm.blob
    for d in range(m.nd):
        m.damping[d] = 0.5
    rng = np.random.default_rng(33)
    sp = H.sim_params(angular_damping=0.5, solver="pgs", max_contacts=kmax)
    n = 48
    terr, hs = _terrain(rng, rough=True)
    dof, root = _random_states(m, n, rng, z_lo=0.1, z_hi=0.5)
    lo, up = np.array(m.lower[:m.nd]), np.array(m.upper[:m.nd])
    dof[:, 0] = rng.uniform(lo, up, (n, m.nd)).astype(np.float32).reshape(-1)
    dof[:, 1] = rng.uniform(-6, 6, n * m.nd)
    fr = rng.uniform(0.5, 1.25, n).astype(np.float32)
    sim = _make_sim(cm, sp, n, terr, hs, group=32)
    T = sim.tensors
    T[_abi.T_SIM_DOF].copy_(torch.from_numpy(dof))
    T[_abi.T_SIM_ROOT].copy_(torch.from_numpy(root))
    T[_abi.T_FRICTION].copy_(torch.from_numpy(fr))
    hits = 0
    for it in range(60):
        eff = rng.uniform(-25, 25, n * m.nd).astype(np.float32)
        push = np.zeros((n * m.nb, 3), np.float32)
        push[::m.nb] = rng.uniform(-20, 20, (n, 3))
        sim.set_dof_command(_abi.T_EFFORT, torch.from_numpy(eff).cuda())
        sim.apply_body_force(torch.from_numpy(push).cuda())
        sim.step()
        sim.refresh(_abi.REFRESH_ALL)
        contact, bstate = oracle.step(m, sp, n, dof, root, effort=eff, friction=fr, body_force=push, terrain=terr, heights=hs,
                                      want_contact=True, want_body_state=True)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(T[_abi.T_DOF_STATE].cpu().numpy(), dof, err_msg=f"dof step {it}")
        np.testing.assert_array_equal(T[_abi.T_ROOT_STATE].cpu().numpy(), root, err_msg=f"root step {it}")
        np.testing.assert_array_equal(T[_abi.T_CONTACT].cpu().numpy(), contact, err_msg=f"contact step {it}")
        hits += int((np.abs(contact).sum(1) > 0).sum())
    assert hits > 500 and np.isfinite(root).all(), hits
    sim.destroy()
    if not selfc:
        return
    cm2, sp2, tp, terr, hs, bufs, sim, task, rng = _a1_setup(64, True, group="chain32", cm=cm, solver="pgs", max_contacts=kmax)
    for it in range(60):
        raw = (2 * rng.random((64, m.nd)) - 1).astype(np.float32) * 2.0
        task.step(torch.from_numpy(raw).cuda())
        oracle.a1_step(m, sp2, tp, 64, 0, bufs, raw, terrain=terr, heights=hs)
        _compare(sim, task, bufs, f"fused self-collision step {it}")
    assert task.kernel_symbol().startswith("_Z16k_a1_chain_pgs16ILb0ELb1E" if kmax > 8 else "_Z14k_a1_chain_pgsILb0ELb1E")


@pytest.mark.parametrize("solver", ["compliant", "pgs"])
def test_velocity_drive_matches_oracle_bitwise(oracle, solver):
    """Robot._internal_motor_step's DOF_MODE_VEL branch (reference shifu/units/robot.py:55-64; no shipped config uses it):
    an implicit velocity drive (kd (v* - qd), effort-limited) on the ABB arm, targets through
    gym.set_dof_velocity_target_tensor, HIP against the oracle; the joints reach the commanded speeds."""
    _need_gpu()
    from shifu_amd.abb_task import ABB_DEFAULT_DOF_POS, abb_model
    rng = np.random.default_rng(4)
    cm = abb_model(kp=0.0, kd=60.0)
    m = cm.blob
    for d in range(m.nd):
        m.drive_mode[d] = _abi.DOF_MODE_VEL
    sp = H.sim_params(dt=0.02, solver=solver)        # ("pgs": a fixed-base arm on the run-time-shaped kernel with the generic solve)
    n = 24
    dof = np.zeros((n * m.nd, 2), np.float32)
    dof[:, 0] = np.tile(np.array(ABB_DEFAULT_DOF_POS, np.float32), n)
    root = np.zeros((n, 13), np.float32); root[:, 6] = 1.0
    vt = rng.uniform(-0.4, 0.4, n * m.nd).astype(np.float32)
    vt[5::m.nd] = 0.0                                     # the dummy tip joint has effort 0: it cannot be driven
    sim = _make_sim(cm, sp, n, group=32)
    T = sim.tensors
    T[_abi.T_SIM_DOF].copy_(torch.from_numpy(dof))
    T[_abi.T_SIM_ROOT].copy_(torch.from_numpy(root))
    sim.set_dof_command(_abi.T_VEL_TARGET, torch.from_numpy(vt).cuda())
    for it in range(25):
        sim.step()
        sim.refresh(_abi.REFRESH_DOF)
        oracle.step(m, sp, n, dof, root, vel_target=vt)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(T[_abi.T_DOF_STATE].cpu().numpy(), dof, err_msg=f"dof step {it}")
    qd = dof[:, 1].reshape(n, m.nd)
    err = np.abs(qd[:, :5] - vt.reshape(n, m.nd)[:, :5])              # kd 60 against the arm's coupling torques: a few cm/s of droop
    assert err.max() < 0.1 and err.mean() < 0.03, (err.max(), err.mean())


def test_config5_full_size_is_deterministic_and_shard_invariant():
    """BASELINE config 5 at its full size (4096 envs), where the oracle would take minutes: size-independent properties --
    two runs give bit-identical tensors, and two 2048-env shards (ranks 0 / 1 of 2) reproduce the unsharded run."""
    _need_gpu()
    from shifu_amd.gym.abb_fused import FusedAbbEnv
    N, K = 4096, 40
    g = torch.Generator(device="cuda:0").manual_seed(5)
    acts = [2 * torch.rand(N, 3, device="cuda:0", generator=g) - 1 for _ in range(K)]

    def run(num, rank, world, sl):
        env = FusedAbbEnv(num_envs=num, seed=3, rank=rank, world_size=world, episode_length_s=2.0)
        for a in acts:
            env.step(a[sl].contiguous())
        torch.cuda.synchronize()
        out = {k: v.clone() for k, v in env.task.tensors.items() if k not in (_abi.ABB_PARAMS, _abi.ABB_STATS, _abi.ABB_STATS_ACC, _abi.ABB_COUNT)}
        out.update({100 + k: env.sim.tensors[k].clone() for k in (_abi.T_DOF_STATE, _abi.T_ROOT_STATE, _abi.T_BODY_STATE, _abi.T_CONTACT)})
        resets = int(env.task.tensors[_abi.ABB_RESET_COUNT].sum())
        env.destroy()
        return out, resets
    a, ra = run(N, 0, 1, slice(0, N))
    b, _ = run(N, 0, 1, slice(0, N))
    assert ra > N and all(torch.equal(a[k], b[k]) for k in a)            # run-to-run identical, resets exercised
    for r in (0, 1):
        s, _ = run(N // 2, r, 2, slice(r * N // 2, (r + 1) * N // 2))
        for k, v in s.items():
            full = a[k]
            if full.shape[0] == v.shape[0] * 2:
                assert torch.equal(v, full[r * v.shape[0]:(r + 1) * v.shape[0]]), f"rank {r} tensor {k}"
            elif full.dim() == 2 and full.shape[1] == v.shape[1] * 2:        # (rows, N) layouts: rew_sums, done_sums
                assert torch.equal(v, full[:, r * v.shape[1]:(r + 1) * v.shape[1]]), f"rank {r} tensor {k}"
            else:
                raise AssertionError(f"unexpected layout for tensor {k}: {tuple(full.shape)} vs {tuple(v.shape)}")


def test_mesh_collider_model_steps_bit_exact_on_the_gpu(oracle, tmp_path):
    """SURVEY 8f f3, mesh shapes: a link whose collision geometry is an STL (convex hull -> eight sample points, hull-derived
    inertia; shifu_amd/model.py) tumbles onto the ground on the HIP path exactly as in the oracle, and comes to rest."""
    _need_gpu()
    from shifu_amd.model import compile_urdf
    from tests.test_model import MESH_URDF, _write_box_stl
    _write_box_stl(tmp_path / "box.stl", (0.3, 0.2, 0.1))
    (tmp_path / "m.urdf").write_text(MESH_URDF.format(inertial="", geom='<mesh filename="box.stl"/>'))
    cm = compile_urdf(str(tmp_path / "m.urdf"))
    assert cm.blob.np == 8
    from tests import kat_models as K
    q = np.array([0.3, 0.2, 0.1, 0.92]); q /= np.linalg.norm(q)
    tr, _ = _kat_run(oracle, cm, H.sim_params(), np.zeros((0, 2), np.float32), K.root_row((0, 0, 0.4), quat=q, ang=(1.0, -2.0, 0.5)), 500)
    assert np.abs(tr[-1, 7:13]).max() < 5e-3 and 0.04 < tr[-1, 2] < 0.2          # at rest on one of its faces


def test_many_contact_points_generic_path(oracle, tmp_path):
    """SURVEY 8f f3 'arbitrary URDFs': a ten-link floating chain with two boxes per link -- 160 contact sample points, what
    anymal.urdf (143) or a cabinet of mesh hulls (164) need -- on the run-time-dimensioned kernels, bit-exact with the oracle
    while it drops onto rough ground and folds up."""
    _need_gpu()
    from shifu_amd.model import compile_urdf
    links = ['<robot name="snake">']
    for k in range(10):
        links.append(f'<link name="l{k}"><inertial><mass value="0.8"/><inertia ixx="0.004" ixy="0" ixz="0" iyy="0.004" iyz="0" izz="0.004"/></inertial>'
                     f'<collision><origin xyz="0.05 0 0"/><geometry><box size="0.08 0.05 0.04"/></geometry></collision>'
                     f'<collision><origin xyz="0.14 0 0"/><geometry><box size="0.08 0.04 0.05"/></geometry></collision></link>')
        if k:
            ax = "0 1 0" if k % 2 else "0 0 1"
            links.append(f'<joint name="j{k}" type="revolute"><parent link="l{k - 1}"/><child link="l{k}"/><origin xyz="0.2 0 0"/>'
                         f'<axis xyz="{ax}"/><limit effort="20" lower="-1.2" upper="1.2" velocity="30"/></joint>')
    links.append("</robot>")
    (tmp_path / "snake.urdf").write_text("\n".join(links))
    cm = compile_urdf(str(tmp_path / "snake.urdf"), default_dof_drive_mode=_abi.DOF_MODE_EFFORT)
    m = cm.blob
    assert m.np == 160 and m.nb == 10 and m.nd == 9
    for d in range(m.nd):
        m.damping[d] = 0.2
    rng = np.random.default_rng(3)
    sp = H.sim_params(angular_damping=0.5)
    n = 24
    terr, hs = _terrain(rng, rough=True)
    dof = np.zeros((n * m.nd, 2), np.float32)
    dof[:, 0] = rng.uniform(-0.6, 0.6, n * m.nd); dof[:, 1] = rng.uniform(-2, 2, n * m.nd)
    root = np.zeros((n, 13), np.float32)
    root[:, 0:2] = rng.uniform(0.5, 3.0, (n, 2)); root[:, 2] = rng.uniform(0.3, 0.6, n)
    q = rng.normal(size=(n, 4)); root[:, 3:7] = q / np.linalg.norm(q, axis=1, keepdims=True)
    root[:, 7:13] = rng.uniform(-1, 1, (n, 6))
    fr = rng.uniform(0.5, 1.25, n).astype(np.float32)
    sim = _make_sim(cm, sp, n, terr, hs, group=32)
    T = sim.tensors
    T[_abi.T_SIM_DOF].copy_(torch.from_numpy(dof))
    T[_abi.T_SIM_ROOT].copy_(torch.from_numpy(root))
    T[_abi.T_FRICTION].copy_(torch.from_numpy(fr))
    touched = 0
    for it in range(60):
        eff = rng.uniform(-3, 3, n * m.nd).astype(np.float32)
        sim.set_dof_command(_abi.T_EFFORT, torch.from_numpy(eff).cuda())
        sim.step()
        sim.refresh(_abi.REFRESH_ALL)
        contact, _ = oracle.step(m, sp, n, dof, root, effort=eff, friction=fr, terrain=terr, heights=hs, want_contact=True,
                                 want_body_state=True)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(T[_abi.T_DOF_STATE].cpu().numpy(), dof, err_msg=f"dof step {it}")
        np.testing.assert_array_equal(T[_abi.T_ROOT_STATE].cpu().numpy(), root, err_msg=f"root step {it}")
        np.testing.assert_array_equal(T[_abi.T_CONTACT].cpu().numpy(), contact, err_msg=f"contact step {it}")
        touched += int((np.abs(contact).sum(1) > 0).sum())
    assert touched > 500 and np.isfinite(root).all()
    sim.destroy()


@pytest.mark.parametrize("form", ["chain32-pgs", "chain32-pgs-self", "chain32", "chain16", 32, 64, "32-self"])
def test_per_env_link_masses_fused_a1_step_matches_oracle_bitwise(oracle, form):
    """SHF_T_BODY_MASS_SCALE (gym.set_actor_rigid_body_properties(..., recomputeInertia=True), shifu/units/units.py:104-110): each
    env its own factor per body on mass and inertia tensor, read in the inertia phase of every kernel form of the fused A1 step
    -- against the oracle given the same rows, every tensor, through falls and resets."""
    _need_gpu()
    n = 64
    selfc = isinstance(form, str) and form.endswith("-self")
    base = form[:-5] if selfc else form
    pgs = isinstance(base, str) and base.endswith("-pgs")
    group = base[:-4] if pgs else base
    group = int(group) if isinstance(group, str) and group.isdigit() else group
    cm = H.a1_model(self_collision=True) if selfc else None
    cm, sp, tp, terr, hs, bufs, sim, task, rng = _a1_setup(n, True, group=group, env_off=300, cm=cm, solver="pgs" if pgs else "compliant")
    scale = rng.uniform(0.6, 1.6, (n, cm.blob.nb)).astype(np.float32)
    scale[:4] = 1.0
    bound = sim.set_body_mass_scale(scale)
    assert bound.shape == (n, cm.blob.nb)
    resets = 0
    with oracle.body_mass_scale(scale):
        for it in range(80):
            raw = (2 * rng.random((n, cm.blob.nd)) - 1).astype(np.float32) * 1.5
            task.step(torch.from_numpy(raw).cuda())
            oracle.a1_step(cm.blob, sp, tp, n, 300, bufs, raw, terrain=terr, heights=hs)
            if it % 8 == 7 or it < 2:
                _compare(sim, task, bufs, f"{form} step {it}")
            resets += int(bufs["reset"].sum())
    assert resets > 4 and np.isfinite(bufs["obs"]).all()
    # the factors matter: the same run without them ends elsewhere
    sim.set_body_mass_scale(None)
    task.step(torch.from_numpy(raw).cuda())
    oracle.a1_step(cm.blob, sp, tp, n, 300, bufs, raw, terrain=terr, heights=hs)
    _compare(sim, task, bufs, f"{form} unbound")

@pytest.mark.parametrize("group,solver", [(64, "compliant"), (32, "compliant"), (32, "pgs")])
def test_per_env_link_masses_simulate_matches_oracle_bitwise(oracle, group, solver):
    """... and under gym.simulate (k_sim_step, the body-per-lane sub-step; solver 'pgs' on an A1 takes k_sim_step_chain_pgs)."""
    _need_gpu()
    rng = np.random.default_rng(12)
    cm = H.a1_model()
    m = cm.blob
    sp = H.sim_params(angular_damping=0.5, solver=solver)
    n = 64
    terr, hs = _terrain(rng, rough=True)
    dof, root = _random_states(m, n, rng)
    sim = _make_sim(cm, sp, n, terr, hs, group=group)
    scale = rng.uniform(0.5, 2.0, (n, m.nb)).astype(np.float32)
    sim.set_body_mass_scale(torch.from_numpy(scale))
    T = sim.tensors
    T[_abi.T_SIM_DOF].copy_(torch.from_numpy(dof))
    T[_abi.T_SIM_ROOT].copy_(torch.from_numpy(root))
    with oracle.body_mass_scale(scale):
        for it in range(40):
            eff = rng.uniform(-25, 25, n * m.nd).astype(np.float32)
            sim.set_dof_command(_abi.T_EFFORT, torch.from_numpy(eff).cuda())
            sim.step()
            sim.refresh(_abi.REFRESH_ALL)
            contact, bstate = oracle.step(m, sp, n, dof, root, terrain=terr, heights=hs, effort=eff, want_contact=True, want_body_state=True)
            torch.cuda.synchronize()
            np.testing.assert_array_equal(T[_abi.T_DOF_STATE].cpu().numpy(), dof, err_msg=f"dof_state step {it}")
            np.testing.assert_array_equal(T[_abi.T_ROOT_STATE].cpu().numpy(), root, err_msg=f"root_state step {it}")
            np.testing.assert_array_equal(T[_abi.T_CONTACT].cpu().numpy(), contact, err_msg=f"contact step {it}")
            np.testing.assert_array_equal(T[_abi.T_BODY_STATE].cpu().numpy(), bstate, err_msg=f"body_state step {it}")
    assert np.isfinite(root).all() and (np.abs(contact).sum(1) > 0).any()

@pytest.mark.parametrize("kw", [dict(solver="pgs", link_contacts=True), dict(solver="pgs", link_contacts=False),
                                dict(solver="compliant", group=16, link_contacts=True), dict(solver="compliant", group=16, link_contacts=False),
                                dict(solver="compliant", group=16, mapping="chain"), dict(solver="compliant", group=32, mapping="body", link_contacts=True),
                                dict(solver="compliant", group=16, mapping="body", link_contacts=False)],
                         ids=["pgs-link", "pgs", "split-link", "split", "chain16", "body32-link", "levels16"])
def test_per_env_link_masses_fused_abb_step_matches_oracle_bitwise(oracle, kw):
    """... and on the arm of the push-box task: every kernel form of the fused ABB step (two waves per env, arm recursions on
    one lane, body per lane; both solvers) with a factor per env and link."""
    _need_gpu()
    from shifu_amd.gym.abb_fused import FusedAbbEnv
    n = 40
    env = FusedAbbEnv(num_envs=n, seed=11, **kw)
    rng = np.random.default_rng(4)
    scale = rng.uniform(0.5, 2.0, (n, env.cm.blob.nb)).astype(np.float32)
    env.sim.set_body_mass_scale(scale)
    bufs = {k: env.sim.tensors[t].cpu().numpy().copy() for k, t in _ABB_SIM_T.items()}
    bufs.update({k: env.task.tensors[t].cpu().numpy().copy() for k, t in _ABB_T.items()})
    with oracle.body_mass_scale(scale):
        for it in range(60):
            raw = (2 * rng.random((n, 3)) - 1).astype(np.float32) * 1.3
            env.task.step(torch.from_numpy(raw).cuda())
            oracle.abb_step(env.cm.blob, env.sim_params, env.boxes, env.task_params, n, 0, bufs, raw)
            if it % 10 == 9 or it < 3:
                torch.cuda.synchronize()
                for k, t in list(_ABB_SIM_T.items()) + list(_ABB_T.items()):
                    got = (env.sim.tensors if k in _ABB_SIM_T else env.task.tensors)[t].cpu().numpy().reshape(bufs[k].shape)
                    np.testing.assert_array_equal(got, bufs[k], err_msg=f"{k} step {it}")
    assert np.isfinite(bufs["obs"]).all()


def test_velocity_level_solve_keeps_every_net_contact_force_in_the_friction_cone_at_full_size():
    """A size-independent property of the solve at BASELINE's size (4096 envs, flat ground, random actions, falls and resets):
    every constraint's impulse leaves the sweeps inside its Coulomb cone (the last visit projects it), the cone is convex and
    on flat ground all of a body's contacts share the normal, so the NET contact force on every body pushes up and
    |F_xy| <= mu F_z -- with mu = (shape friction of the env + terrain friction) / 2.  Under the compliant law this does not
    hold (its friction is a viscous law capped per point), which the last lines check as a control."""
    _need_gpu()
    from shifu_amd.gym.a1_fused import FusedA1Env
    n = 4096
    worst = {}
    for solver in ("pgs", "compliant"):
        env = FusedA1Env(num_envs=n, terrain="flat", seed=7, solver=solver)
        assert env.solver == solver
        env.reset()
        mu = 0.5 * (env.sim.tensors[_abi.T_FRICTION].view(n, 1) + float(env.sim.terrain.friction if env.sim.terrain is not None else 1.0))
        g = torch.Generator(device="cuda:0").manual_seed(3)
        loaded, w = 0, 0.0
        for it in range(60):
            env.step(2 * torch.rand(n, 12, device="cuda:0", generator=g) - 1)
            F = env.contact_state.view(n, -1, 3)
            fz, fxy = F[:, :, 2], F[:, :, :2].norm(dim=-1)
            assert torch.isfinite(F).all()
            if solver == "pgs":
                assert float(fz.min()) >= -1e-4, (it, float(fz.min()))
            excess = (fxy - mu * fz.clamp_min(0.0)) / (1.0 + fz.abs())
            w = max(w, float(excess.max()))
            loaded += int((fz > 1.0).sum())
        worst[solver] = w
        assert loaded > 10 * n, "the run must put weight on the ground"
        env.destroy()
    assert worst["pgs"] <= 1e-4, worst
    assert worst["compliant"] > 1e-3, worst       # the control: the property is the solver's, not the scene's


@pytest.mark.parametrize("pos_iters,vel_iters,kmax", [(4, 0, 8), (1, 2, 5), (8, 1, 1), (2, 0, 3), (12, 3, 8)])
def test_velocity_level_solve_other_iteration_counts_match_oracle_bitwise(oracle, pos_iters, vel_iters, kmax):
    """physx.num_position_iterations / num_velocity_iterations / the constraint limit away from the reference's 8 + 1 / 8: no
    velocity iterations at all (one impulse set through the tree, poses and velocities from it), several, a single
    constraint per env -- the fused A1 step (k_a1_chain_pgs) and gym.simulate on a generic articulation (k_sim_step<32,..,HARD>
    through a 13-body A1) against the oracle, every tensor."""
    _need_gpu()
    n = 64
    kw = dict(solver="pgs", pos_iters=pos_iters, vel_iters=vel_iters, max_contacts=kmax)
    cm, sp, tp, terr, hs, bufs, sim, task, rng = _a1_setup(n, True, group="chain32", env_off=50, **kw)
    assert sp.pos_iters == pos_iters and sp.vel_iters == vel_iters and sp.max_contacts == kmax
    for it in range(60):
        raw = (2 * rng.random((n, cm.blob.nd)) - 1).astype(np.float32) * 1.5
        task.step(torch.from_numpy(raw).cuda())
        oracle.a1_step(cm.blob, sp, tp, n, 50, bufs, raw, terrain=terr, heights=hs)
        if it % 6 == 5 or it < 2:
            _compare(sim, task, bufs, f"a1 step {it}")
    assert np.isfinite(bufs["obs"]).all() and np.abs(bufs["contact"]).max() > 5.0
    # the run-time-shaped kernel: the A1 with its feet merged into the shanks (13 bodies: not the chain kernel's shape)
    from shifu_amd.model import asset_path, compile_urdf
    cg = compile_urdf(asset_path("a1.urdf"), default_dof_drive_mode=_abi.DOF_MODE_EFFORT, honour_dont_collapse=False)
    m = cg.blob
    assert m.nb == 13
    spg = H.sim_params(angular_damping=0.5, **kw)
    rng = np.random.default_rng(8)
    terr, hs = _terrain(rng, rough=True)
    dof, root = _random_states(m, n, rng)
    simg = _make_sim(cg, spg, n, terr, hs, group=32)
    T = simg.tensors
    T[_abi.T_SIM_DOF].copy_(torch.from_numpy(dof))
    T[_abi.T_SIM_ROOT].copy_(torch.from_numpy(root))
    for it in range(40):
        eff = rng.uniform(-20, 20, n * m.nd).astype(np.float32)
        simg.set_dof_command(_abi.T_EFFORT, torch.from_numpy(eff).cuda())
        simg.step()
        simg.refresh(_abi.REFRESH_ALL)
        contact, bstate = oracle.step(m, spg, n, dof, root, terrain=terr, heights=hs, effort=eff, want_contact=True, want_body_state=True)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(T[_abi.T_DOF_STATE].cpu().numpy(), dof, err_msg=f"generic dof step {it}")
        np.testing.assert_array_equal(T[_abi.T_ROOT_STATE].cpu().numpy(), root, err_msg=f"generic root step {it}")
        np.testing.assert_array_equal(T[_abi.T_CONTACT].cpu().numpy(), contact, err_msg=f"generic contact step {it}")
    assert np.isfinite(root).all() and (np.abs(contact).sum(1) > 0).any()


def test_velocity_level_solve_other_contact_parameters_match_oracle_bitwise(oracle):
    """The other physx / material fields away from their defaults -- rest_offset, contact_offset, bounce_threshold_velocity with
    a restitution > 0 (robots dropped from 0.5 m bounce), max_depenetration_velocity, the Baumgarte factor -- on the fused A1
    step and on gym.simulate of the 13-body variant, against the oracle."""
    _need_gpu()
    n = 64
    kw = dict(solver="pgs", rest_offset=0.002, contact_offset=0.02, bounce_threshold=0.2, restitution=0.5, erp=0.3, max_depen_vel=0.5)
    cm, sp, tp, terr, hs, bufs, sim, task, rng = _a1_setup(n, True, group="chain32", env_off=9, **kw)
    assert abs(sp.restitution - 0.5) < 1e-7 and abs(sp.rest_offset - 0.002) < 1e-9
    for it in range(60):
        raw = (2 * rng.random((n, cm.blob.nd)) - 1).astype(np.float32) * 1.5
        task.step(torch.from_numpy(raw).cuda())
        oracle.a1_step(cm.blob, sp, tp, n, 9, bufs, raw, terrain=terr, heights=hs)
        if it % 6 == 5 or it < 2:
            _compare(sim, task, bufs, f"a1 step {it}")
    assert np.isfinite(bufs["obs"]).all()
    from shifu_amd.model import asset_path, compile_urdf
    cg = compile_urdf(asset_path("a1.urdf"), default_dof_drive_mode=_abi.DOF_MODE_EFFORT, honour_dont_collapse=False)
    m = cg.blob
    spg = H.sim_params(angular_damping=0.5, **kw)
    rng = np.random.default_rng(18)
    dof, root = _random_states(m, n, rng, z_lo=0.4, z_hi=0.6)
    simg = _make_sim(cg, spg, n, group=32)
    T = simg.tensors
    T[_abi.T_SIM_DOF].copy_(torch.from_numpy(dof))
    T[_abi.T_SIM_ROOT].copy_(torch.from_numpy(root))
    up = False
    for it in range(120):
        simg.step()
        simg.refresh(_abi.REFRESH_ALL)
        contact, bstate = oracle.step(m, spg, n, dof, root, want_contact=True, want_body_state=True)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(T[_abi.T_DOF_STATE].cpu().numpy(), dof, err_msg=f"generic dof step {it}")
        np.testing.assert_array_equal(T[_abi.T_ROOT_STATE].cpu().numpy(), root, err_msg=f"generic root step {it}")
        np.testing.assert_array_equal(T[_abi.T_CONTACT].cpu().numpy(), contact, err_msg=f"generic contact step {it}")
        up |= bool(((np.abs(contact).reshape(n, m.nb, 3).sum((1, 2)) == 0) & (root[:, 9] > 0.2)).any() and it > 40)
    assert np.isfinite(root).all() and up, "some robot left the ground again after its first impact (restitution)"


def test_hard_contact_kats_on_the_gpu(oracle):
    """tests/test_hard_contact.py's known answers on the HIP path -- a free block / sphere without joints: the run-time-shaped
    kernel with the generic solve on an articulation of ONE body and zero dofs (zero-sized dof tensors) -- bit for bit against
    the float oracle, and the closed forms asserted on what the GPU produced: no creep below the friction angle, Coulomb
    acceleration g (sin - mu cos) above it, mu g deceleration to a dead stop, a dropped sphere that stops at the surface."""
    _need_gpu()
    from tests import kat_models as K
    blk = K.block_model()
    nodof = np.zeros((0, 2), np.float32)

    def block(theta, mu_shape, steps, lin=(0, 0, 0)):
        sp = H.sim_params(gravity=(K.G * np.sin(theta), 0.0, -K.G * np.cos(theta)), solver="pgs")
        return _kat_run(oracle, blk, sp, nodof, K.root_row((0, 0, 0.05), lin=lin), steps, mu_shape, group=32)[0]
    tr = block(np.arctan(0.3), 0.2, 400)                       # mu = (0.2 + 1) / 2 = 0.6 > tan(theta): holds, no creep
    assert np.abs(tr[-100:, 7:13]).max() < 2e-5 and abs(tr[-1, 0]) < 400 * K.DT * 2e-5 + 1e-6
    th = np.arctan(1.0)
    tr = block(th, 0.2, 200)
    a = K.G * (np.sin(th) - 0.6 * np.cos(th))
    assert abs((tr[-1, 7] - tr[99, 7]) / (100 * K.DT) - a) < 0.01 * a
    tr = block(0.0, 0.6, 200, lin=(1.0, 0, 0))                 # mu = 0.8: stops after v0^2 / (2 mu g), then stays
    d = 1.0 / (2 * 0.8 * K.G)
    assert abs(tr[-1, 0] - d) < 1.0 * K.DT and np.abs(tr[-1, 7:10]).max() < 1e-5     # (within one step's travel of the continuous answer)
    tr = _kat_run(oracle, K.ball_model(), H.sim_params(solver="pgs"), nodof, K.root_row((0, 0, 0.55)), 400, group=32)[0]
    z, vz = tr[:, 2] - 0.05, tr[:, 9]
    assert abs(z[-1]) < 1e-4 and np.abs(vz[-50:]).max() < 1e-5 and z.min() > -2e-3       # rests ON the surface: no sag


@pytest.mark.parametrize("kmax", [8, 16])
@pytest.mark.parametrize("rough", [False, True])
def test_fused_a1_step_under_tgs_matches_oracle_bitwise(oracle, rough, kmax):
    """ShfSimParams.solver = SHF_SOLVER_TGS (physx.solver_type = 1, the reference's value: shifu/configs/env_config.py:50): the
    sweeps as sub-iterations of dt / 8 -- targets from the gaps as they stand, gaps advanced after every sweep, the poses moved by
    the mean of the impulses -- on the chain-mapped A1 kernels (8 and 16 constraints per env) against the oracle: every tensor,
    120 vec-steps with falls and resets."""
    _need_gpu()
    n = 96
    cm, sp, tp, terr, hs, bufs, sim, task, rng = _a1_setup(n, rough, group="chain32", env_off=2000, solver="tgs", max_contacts=kmax)
    assert sp.solver == _abi.SOLVER_TGS and task.kernel_symbol().startswith("_Z16k_a1_chain_tgs16" if kmax > 8 else "_Z14k_a1_chain_tgsI")
    resets = 0
    for it in range(120):
        raw = (2 * rng.random((n, cm.blob.nd)) - 1).astype(np.float32) * 1.5
        slot = task.step(torch.from_numpy(raw).cuda())
        oracle.a1_step(cm.blob, sp, tp, n, 2000, bufs, raw, terrain=terr, heights=hs)
        _compare(sim, task, bufs, f"step {it}")
        np.testing.assert_array_equal(task.tensors[_abi.A1_STATS][slot].cpu().numpy(), oracle.a1_stats(tp, n, bufs["done_sums"]), err_msg=f"stats step {it}")
        resets += int(bufs["reset"].sum())
    assert resets > n // 4 and np.isfinite(bufs["obs"]).all() and np.abs(bufs["contact"]).max() > 10.0
    # ... and it is a different computation from SHF_SOLVER_PGS on the same inputs
    cm2, sp2, tp2, terr2, hs2, bufs2, sim2, task2, rng2 = _a1_setup(n, rough, group="chain32", env_off=2000, solver="pgs", max_contacts=kmax)
    raw = (2 * rng2.random((n, cm.blob.nd)) - 1).astype(np.float32)
    b_t, b_p = {k: v.copy() for k, v in bufs.items()}, {k: v.copy() for k, v in bufs.items()}      # (from the run's end state: feet on the ground)
    oracle.a1_step(cm.blob, sp, tp, n, 2000, b_t, raw, terrain=terr, heights=hs)
    oracle.a1_step(cm.blob, sp2, tp, n, 2000, b_p, raw, terrain=terr, heights=hs)
    assert not np.array_equal(b_t["root_state"], b_p["root_state"])


def test_fused_a1_step_under_tgs_at_full_size(oracle):
    """... at BASELINE's env count: 4096 envs x 30 vec-steps under SHF_SOLVER_TGS, every tensor bit for bit."""
    _need_gpu()
    n = 4096
    cm, sp, tp, terr, hs, bufs, sim, task, rng = _a1_setup(n, True, seed=91, group="chain32", env_off=8192, solver="tgs")
    bufs["ep_len"][:] = rng.integers(900, 1001, n)
    _upload(sim, task, bufs)
    resets = 0
    for it in range(30):
        raw = (2 * rng.random((n, cm.blob.nd)) - 1).astype(np.float32)
        slot = task.step(torch.from_numpy(raw).cuda())
        oracle.a1_step(cm.blob, sp, tp, n, 8192, bufs, raw, terrain=terr, heights=hs)
        if it % 10 == 9:
            _compare(sim, task, bufs, f"step {it}")
            np.testing.assert_array_equal(task.tensors[_abi.A1_STATS][slot].cpu().numpy(), oracle.a1_stats(tp, n, bufs["done_sums"]))
        resets += int(bufs["reset"].sum())
    assert resets > 100


@pytest.mark.parametrize("mapping", ["split", "body"])
@pytest.mark.parametrize("link", [False, True])
def test_fused_abb_step_under_tgs_matches_oracle_bitwise(oracle, link, mapping):
    """... and the generic solve of the body-per-lane kernels (config 5: box actors as solver bodies, link contacts, a fixed base)
    under SHF_SOLVER_TGS: FusedAbbEnv(solver='tgs'), 256 envs x 60 vec-steps with re-spawns, every tensor."""
    _need_gpu()
    from shifu_amd.gym.abb_fused import FusedAbbEnv
    n = 256
    env = FusedAbbEnv(num_envs=n, seed=12, link_contacts=link, solver="tgs", **({"mapping": "body"} if mapping == "body" else {}))
    assert env.sim_params.solver == _abi.SOLVER_TGS and env.solver == "tgs" and env.mapping == mapping
    assert ("ws_hard" in env.task.kernel_symbol()) == (mapping == "split")
    env.task.tensors[_abi.ABB_EP_LEN].copy_(torch.randint(150, 201, (n,)))
    torch.cuda.synchronize()
    bufs = {k: env.sim.tensors[t].cpu().numpy().copy() for k, t in _ABB_SIM_T.items()}
    bufs.update({k: env.task.tensors[t].cpu().numpy().copy() for k, t in _ABB_T.items()})
    rng = np.random.default_rng(6)
    resets = 0
    for it in range(60):
        raw = (2 * rng.random((n, 3)) - 1).astype(np.float32)
        env.task.step(torch.from_numpy(raw).cuda())
        oracle.abb_step(env.cm.blob, env.sim_params, env.boxes, env.task_params, n, 0, bufs, raw)
        resets += int(bufs["reset"].sum())
    torch.cuda.synchronize()
    for k, t in list(_ABB_SIM_T.items()) + list(_ABB_T.items()):
        got = (env.sim.tensors if k in _ABB_SIM_T else env.task.tensors)[t].cpu().numpy().reshape(bufs[k].shape)
        np.testing.assert_array_equal(got, bufs[k], err_msg=k)
    assert resets > 20
