"""Known-answer tests of the velocity-level contact solve (ShfSimParams.solver = SHF_SOLVER_PGS; oracle/shf_oracle.c
hard_solve) -- the solver class the reference configures (shifu/configs/env_config.py:50-58: solver_type 1,
num_position_iterations 8, num_velocity_iterations 1, contact_offset 0.01, rest_offset 0, bounce_threshold_velocity 0.5,
max_depenetration_velocity 1) and every gym.simulate runs under (shifu/units/robot.py:69,
examples/a1_conditional/a1_conditional.py:69, shifu/gym/isaac_gym.py:140).  PhysX is absent, so what pins the solver is
(i) Coulomb's law and rigid impact in closed form, and (ii) an independently written solver of the same class
(oracle/hard_contact_ref.py: joint-space inertia matrix + projected Gauss-Seidel), which it must reproduce to rounding.
tests/test_gpu_parity.py runs the same scenarios on the HIP kernels (bit-equal to the float oracle)."""
import os
import sys

import numpy as np
import pytest

from tests import kat_models as K
from tests.helpers import sim_params

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
PREC = [pytest.param(True, id="f64"), pytest.param(False, id="f32")]
# "pgs": physx.solver_type = 0; "tgs": solver_type = 1, the reference's value (env_config.py:50) -- the same sweeps as sub-iterations
# of dt / 8 whose constraint errors follow the advanced motion and whose mean impulses move the poses (SHF_SOLVER_TGS).  Without
# warm starting (the arithmetic keeps no state between steps) the first sub-iterations of every step are far from converged, and
# what they let through moves the pose: TGS meets the same laws with looser numbers -- a held block creeps at ~2 mm/s (the
# compliant law: 1 mm/s; PGS: none), stated per test.
SOLVERS = ["pgs", "tgs"]


def run_block(oracle, f64, theta, mu_shape, steps, lin=(0, 0, 0), dt_sim=K.DT, z0=0.05, solver="pgs", **kw):
    dt = np.float64 if f64 else np.float32
    cm = K.block_model()
    sp = sim_params(dt=dt_sim, gravity=(K.G * np.sin(theta), 0.0, -K.G * np.cos(theta)), solver=solver, **kw)
    root = K.root_row((0, 0, z0), lin=lin, dtype=dt)
    dof = np.zeros((0, 2), dt)
    fr = np.full(1, mu_shape, np.float32)
    traj, force = [], []
    for k in range(steps):
        c, _ = oracle.step(cm.blob, sp, 1, dof, root, friction=fr, f64=f64, want_contact=True)
        traj.append(root[0].copy()); force.append(c[0].copy())
    return np.array(traj), np.array(force)


@pytest.mark.parametrize("solver", SOLVERS)
@pytest.mark.parametrize("f64", PREC)
def test_block_sticks_below_the_friction_angle_without_creep(oracle, f64, solver):
    """tan(theta) = 0.3 < mu = 0.6: a rigid Coulomb contact holds the block -- no regularisation creep (the compliant law
    creeps at v_eps tan(theta) / mu = 1 mm/s here), no sag; the contact force balances gravity."""
    tr, f = run_block(oracle, f64, np.arctan(0.3), 0.2, 400, solver=solver)
    tol = 1e-6 if f64 else 2e-5          # (nine sweeps over four corners: converged to ~1e-7)
    if solver == "tgs":
        tol = 2.5e-3                     # (no warm start: the cold first sub-iterations of every step leak 2 mm/s into the pose)
    assert np.abs(tr[-100:, 7:13]).max() < tol, np.abs(tr[-100:, 7:13]).max()
    assert abs(tr[-1, 0]) < 400 * K.DT * tol + 1e-7 and abs(tr[-1, 2] - 0.05) < 1e-5
    th = np.arctan(0.3)
    assert abs(f[-1, 2] - 2.0 * K.G * np.cos(th)) < 1e-3 * 2.0 * K.G and abs(f[-1, 0] + 2.0 * K.G * np.sin(th)) < 1e-3 * 2.0 * K.G, f[-1]


@pytest.mark.parametrize("solver", SOLVERS)
@pytest.mark.parametrize("f64", PREC)
def test_block_slides_above_the_friction_angle_with_coulomb_acceleration(oracle, f64, solver):
    """tan(theta) = 1 > mu = 0.6: a = g (sin(theta) - mu cos(theta)) from the first step on (the compliant law needs a
    start-up transient of 0.17 m/s)."""
    th, mu = np.arctan(1.0), 0.6
    a = K.G * (np.sin(th) - mu * np.cos(th))
    tr, f = run_block(oracle, f64, th, 0.2, 200, solver=solver)
    assert abs(tr[-1, 7] - a * 200 * K.DT) < (1e-4 if f64 else 2e-3) * a, (tr[-1, 7], a)
    acc = np.diff(tr[:, 7]) / K.DT
    assert np.abs(acc[5:] - a).max() < ((1e-5 if solver == "pgs" else 2e-4) if f64 else 5e-3) * a        # (the first steps: the corners' loads settle, 1e-3)
    assert np.abs(acc - a).max() < 5e-3 * a
    assert np.abs(tr[-1, 10:13]).max() < 1e-3 and abs(tr[-1, 2] - 0.05) < 1e-4      # flat on the slope, no tumbling
    assert abs(np.hypot(f[-1, 0], f[-1, 1]) - mu * f[-1, 2]) < 1e-4 * f[-1, 2]      # on the cone


@pytest.mark.parametrize("solver", SOLVERS)
@pytest.mark.parametrize("f64", PREC)
def test_sliding_block_decelerates_at_mu_g_and_stops_dead(oracle, f64, solver):
    """Level ground, v0 = 1 m/s, mu = 0.8: v(t) = v0 - mu g t step for step, stopping distance v0^2 / (2 mu g) to within
    half a step's travel, then exactly at rest."""
    mu, v0 = 0.8, 1.0
    tr, _ = run_block(oracle, f64, 0.0, 0.6, 100, lin=(v0, 0, 0), solver=solver)
    n = np.arange(1, 21)
    assert np.abs(tr[:20, 7] - (v0 - mu * K.G * K.DT * n)).max() < ((1e-6 if f64 else 2e-5) if solver == "pgs" else 5e-4)
    d = v0 * v0 / (2 * mu * K.G)
    assert abs(tr[-1, 0] - d) < 0.5 * v0 * K.DT, (tr[-1, 0], d)
    assert np.abs(tr[-40:, 7:13]).max() < ((5e-6 if f64 else 2e-5) if solver == "pgs" else 4e-3)         # (what nine sweeps leave)


@pytest.mark.parametrize("solver", SOLVERS)
@pytest.mark.parametrize("f64", PREC)
def test_dropped_sphere_stops_at_the_surface_and_rests_on_it(oracle, f64, solver):
    """restitution 0 (env_config.py:84): a 1 kg sphere dropped from 0.5 m arrives at 3.1 m/s; the speculative constraint
    (contact_offset, env_config.py:54) lets it close the gap and no more, so it stops ON the surface in the step that would
    have crossed it -- no penetration, no rebound, no sag."""
    dt = np.float64 if f64 else np.float32
    cm = K.ball_model()
    sp = sim_params(solver=solver)
    root = K.root_row((0, 0, 0.55), dtype=dt)
    dof = np.zeros((0, 2), dt)
    z, vz = [], []
    for k in range(200):
        oracle.step(cm.blob, sp, 1, dof, root, friction=np.ones(1, np.float32), f64=f64)
        z.append(root[0, 2] - 0.05); vz.append(root[0, 9])
    z, vz = np.array(z), np.array(vz)
    hit = int(np.argmin(vz)) + 1
    assert 3.0 < -vz[:hit].min() < 3.2
    tol = 1e-6 if f64 else 3e-6
    assert np.abs(z[hit:]).max() < tol and np.abs(vz[hit + 1:]).max() < 1e2 * tol, (np.abs(z[hit:]).max(), np.abs(vz[hit + 1:]).max())


@pytest.mark.parametrize("solver", SOLVERS)
def test_penetration_is_removed_at_the_baumgarte_rate_capped_by_max_depenetration_velocity(oracle, solver):
    """A sphere started 5 cm inside the ground is pushed out at min(erp pen / dt, max_depenetration_velocity) = 1 m/s
    (env_config.py:57), the last stretch decays geometrically (factor 1 - erp per step) -- and it is not shot out: the
    push-out moves the pose (position iterations) but is not in the velocity the step hands on (the velocity iteration runs
    without the bias, env_config.py:52), so with gravity off the sphere ends at rest on the surface."""
    cm = K.ball_model()
    sp = sim_params(solver=solver, gravity=(0, 0, 0))
    root = K.root_row((0, 0, 0.0), dtype=np.float64)
    dof = np.zeros((0, 2))
    z = []
    for k in range(60):
        oracle.step(cm.blob, sp, 1, dof, root, friction=np.ones(1, np.float32), f64=True)
        z.append(root[0, 2] - 0.05)
    z = np.array(z)
    assert abs((z[5] - z[4]) / K.DT - 1.0) < 2e-5                       # capped (regularisation 1e-6 trace(W) = 8e-6 of n.W n here)
    tail = z[(z > -0.02) & (z < -1e-4)]
    if solver == "pgs":
        assert len(tail) > 5 and np.abs(tail[1:] / tail[:-1] - 0.8).max() < 1e-5, tail   # erp 0.2
    else:       # every sub-iteration removes erp of what is left (capped): (1 - erp)^8 = 0.17 per step once the cap lets go
        assert 2 <= len(tail) <= 6 and (tail[1:] / tail[:-1] < 0.7).all() and abs(tail[-1] / tail[-2] - 0.8 ** 8) < 1e-3, tail
    assert abs(root[0, 9]) < 1e-9 and z[-1] < 1e-6                      # no momentum left behind, never above the surface


@pytest.mark.parametrize("solver", SOLVERS)
def test_restitution_above_the_bounce_threshold(oracle, solver):
    """physx.bounce_threshold_velocity (env_config.py:56): with restitution 0.5 an impact at 3.1 m/s rebounds at half its
    speed; one slower than the threshold (0.5 m/s) does not rebound at all."""
    cm = K.ball_model()
    out = {}
    for h0 in (0.5, 0.005):
        sp = sim_params(solver=solver, restitution=0.5)
        root = K.root_row((0, 0, 0.05 + h0), dtype=np.float64)
        vz = []
        for k in range(120):
            oracle.step(cm.blob, sp, 1, np.zeros((0, 2)), root, friction=np.ones(1, np.float32), f64=True)
            vz.append(root[0, 9])
        vz = np.array(vz)
        out[h0] = (vz.min(), vz.max())
    assert abs(out[0.5][1] + 0.5 * out[0.5][0]) < 0.03, out          # up at half the arrival speed (within one step of gravity)
    assert out[0.005][0] > -0.5 and out[0.005][1] < 1e-5, out


@pytest.mark.parametrize("solver", SOLVERS)
@pytest.mark.parametrize("f64", PREC)
def test_standing_a1_carries_its_weight_without_creep(oracle, f64, solver):
    """The A1 on explicit PD (kp 20, kd 0.5, task_config.py:22-23) settles on its four feet: the contact impulses carry
    12.454 kg g; the feet do not creep (<= 0.05 mm/s; the compliant law's splayed stance creeps at 1.7 mm/s,
    profiles/r04_model_gap.md)."""
    import model_gap as G
    cm, sp = G.a1_setup(solver)
    m = cm.blob
    dt = np.float64 if f64 else np.float32
    dof = np.zeros((m.nd, 2), dt); dof[:, 0] = G.A1_Q0
    root = np.zeros((1, 13), dt); root[0, 2] = 0.33; root[0, 6] = 1.0
    zs, fz = [], []
    for k in range(2400):
        tau = (G.KP * (G.A1_Q0 - dof[:, 0]) - G.KD * dof[:, 1]).astype(dt)
        c, _ = oracle.step(m, sp, 1, dof, root, effort=tau, friction=np.ones(1, np.float32), f64=f64, want_contact=True)
        zs.append(root[0, :3].copy()); fz.append(c[:, 2].sum())
    zs = np.array(zs)
    mass = sum(m.mass[b] for b in range(m.nb))
    assert abs(np.mean(fz[-200:]) / (mass * K.G) - 1.0) < 2e-3
    # the soft PD (kp 20) lets the trunk sway on its rigid feet for seconds after the drop (nothing in a rigid contact damps it):
    # creep = drift of the mean position between two windows 4 s apart
    creep = np.linalg.norm(zs[-200:].mean(0) - zs[-1000:-800].mean(0)) / (800 * sp.dt)
    assert creep < (5e-5 if solver == "pgs" else 5e-4), creep            # (TGS without warm start: 0.3 mm/s; the compliant law: 1.7 mm/s)
    assert np.abs(dof[:, 1]).max() < (5e-3 if solver == "pgs" else 1e-2)


@pytest.mark.parametrize("solver", SOLVERS)
def test_reproduces_the_independent_joint_space_solver(solver):
    """oracle/hard_contact_ref.py -- same algorithm, different building blocks (M(q) from unit-acceleration inverse
    dynamics, dense Delassus matrix J M^-1 J^T, a tangent basis per contact) -- and the oracle's articulated-body
    formulation in world axes: the A1 agrees to rounding, per sub-step and over 300 open-loop sub-steps of trotting with
    foot strikes (the compliant law: 1e-2 rad per sub-step at a foot strike, 9e-2 rad after 100; north-star tolerance 1e-4)."""
    import model_gap as G
    stand = G.run_a1("stand", 150, solver)
    assert stand["local_dq_max"] < 1e-9 and stand["local_droot_max"] < 1e-10, stand
    assert abs(stand["hard_contact_normal_force_over_weight"] - 1.0) < 5e-3
    trot = G.run_a1("trot", 300, solver)
    assert trot["local_dq_max"] < 1e-9 and trot["local_droot_max"] < 1e-9, trot
    assert trot["accum_dq"]["300"] < 1e-8 and trot["accum_droot"]["300"] < 1e-8, trot
    # the ABB scene: the two differ in collision GEOMETRY (finite table and vertex / edge / line-contact detection against a
    # plane and a sampled closest point), so they part where a contact switches on in one and not yet in the other
    abb = G.run_abb(150, solver)
    assert abb["local_dq_max"] < 1e-5 and abb["local_dq_mean"] < 5e-7 and abb["local_dcube_mean"] < 2e-5, abb
    assert abb["cube_travel_shipped"] > 0.05 and abs(abb["cube_travel_hard"] / abb["cube_travel_shipped"] - 1.0) < 0.15, abb


def test_the_deepest_contacts_are_kept_and_the_rest_counted(oracle):
    """max_contacts = 3 for a block on four corners: the three deepest are constrained, the fourth is counted as dropped
    (SHF_T_DROPPED on the GPU) -- and becomes one of the deepest as soon as it sinks."""
    cm = K.block_model()
    sp = sim_params(solver="pgs", max_contacts=3)
    root = K.root_row((0, 0, 0.05))
    oracle.dropped(reset=True)
    for k in range(50):
        oracle.step(cm.blob, sp, 1, np.zeros((0, 2), np.float32), root, friction=np.ones(1, np.float32))
    assert oracle.dropped() >= 50            # one of the four bottom corners every step
    assert abs(root[0, 2] - 0.05) < 2e-3 and np.abs(root[0, 3:5]).max() < 0.02      # it wobbles on three corners but does not sink
