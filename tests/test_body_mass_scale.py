"""Per-env link masses of an articulation: gym.set_actor_rigid_body_properties(env, actor, props, recomputeInertia=True)
(/root/reference/shifu/units/units.py:104-110) as a factor per env and body on the asset's mass and inertia tensor
(SHF_T_BODY_MASS_SCALE).  CPU: the oracle against closed forms and against a model compiled with the edited masses; the GPU
twins are in tests/test_gpu_parity.py."""
import copy

import numpy as np
import pytest

from tests import helpers as H


def _blob_scaled(blob, scale):
    """The model with mass and inertia tensor of every body multiplied in float32, as the oracle / kernels multiply them."""
    b = copy.deepcopy(blob)
    for k in range(b.nb):
        s = np.float32(scale[k])
        b.mass[k] = float(np.float32(b.mass[k]) * s)
        for j in range(6):
            b.inertia[k][j] = float(np.float32(b.inertia[k][j]) * s)
    return b


def test_uniform_factor_divides_the_response_to_joint_efforts(oracle):
    """M(q) is linear in the masses: with every body s times heavier, qdd(s; tau) = qdd(1; tau / s) -- gravity and velocity
    terms scale with the masses too and cancel (float64 oracle, ABA without contacts)."""
    cm = H.a1_model()
    m = copy.deepcopy(cm.blob)
    for d in range(m.nd):          # (passive damping and drive gains are implicit in the joint inertia and do not scale: off)
        m.damping[d] = 0.0; m.kp[d] = 0.0; m.kd[d] = 0.0
    sp = H.sim_params()
    rng = np.random.default_rng(0)
    dof = np.zeros((m.nd, 2))
    dof[:, 0] = np.array([0.0, 0.8, -1.6] * 4) + rng.uniform(-0.2, 0.2, m.nd)      # (inside the joint limits: their springs do not scale)
    dof[:, 1] = rng.uniform(-1.0, 1.0, m.nd)
    root = np.zeros(13); root[2] = 0.5; root[6] = 1.0; root[10:13] = rng.uniform(-1, 1, 3)
    tau = rng.uniform(-8, 8, m.nd)
    assert all(m.armature[d] == 0.0 for d in range(m.nd)), "the identity needs joint inertias that scale with the masses"
    s = 1.7
    with oracle.body_mass_scale(np.full((1, m.nb), s, np.float32)):
        qdd_s, ra_s = oracle.accel(m, sp, dof, root, tau)
    qdd_1, ra_1 = oracle.accel(m, sp, dof, root, tau / np.float32(s).astype(np.float64))
    np.testing.assert_allclose(qdd_s, qdd_1, rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(ra_s, ra_1, rtol=1e-9, atol=1e-9)
    qdd_0, _ = oracle.accel(m, sp, dof, root, tau)
    assert np.abs(qdd_0 - qdd_s).max() > 1.0, "the factor must matter"


@pytest.mark.parametrize("solver", ["compliant", "pgs"])
def test_per_body_factors_equal_a_model_with_the_edited_masses(oracle, solver):
    """Each env its own factors: stepping env e with row e of the tensor is, bit for bit (float32 oracle), stepping a model whose
    mass[b] and inertia[b] were multiplied beforehand -- standing and falling A1s on the plane, contacts and self-masses in."""
    cm = H.a1_model()
    m = cm.blob
    sp = H.sim_params(solver=solver) if solver == "pgs" else H.sim_params(solver="compliant")
    n = 6
    rng = np.random.default_rng(3)
    scale = rng.uniform(0.6, 1.5, (n, m.nb)).astype(np.float32)
    scale[0] = 1.0
    dof = np.zeros((n * m.nd, 2), np.float32)
    dof[:, 0] = np.tile(np.array([0.0, 0.8, -1.6] * 4, np.float32), n) + rng.uniform(-0.1, 0.1, n * m.nd).astype(np.float32)
    root = np.zeros((n, 13), np.float32); root[:, 2] = rng.uniform(0.27, 0.4, n); root[:, 6] = 1.0
    tgt = dof[:, 0].copy()
    d1, r1 = dof.copy(), root.copy()
    with oracle.body_mass_scale(scale):
        c1, b1 = oracle.step(m, sp, n, d1, r1, nsteps=150, pos_target=tgt, want_contact=True, want_body_state=True)
    for e in range(n):
        d2, r2 = dof[e * m.nd:(e + 1) * m.nd].copy(), root[e:e + 1].copy()
        c2, b2 = oracle.step(_blob_scaled(m, scale[e]), sp, 1, d2, r2, nsteps=150, pos_target=tgt[e * m.nd:(e + 1) * m.nd].copy(),
                             want_contact=True, want_body_state=True)
        np.testing.assert_array_equal(d1[e * m.nd:(e + 1) * m.nd], d2, err_msg=f"env {e}")
        np.testing.assert_array_equal(r1[e:e + 1], r2)
        np.testing.assert_array_equal(c1[e * m.nb:(e + 1) * m.nb], c2)
    # and it is the weight that the feet carry
    w = np.array([sum(float(np.float32(m.mass[b]) * scale[e, b]) for b in range(m.nb)) for e in range(n)]) * 9.81
    fz = c1.reshape(n, m.nb, 3)[:, :, 2].sum(1)
    assert np.isfinite(d1).all()
    standing = np.abs(r1[:, 7:10]).max(1) < 0.05
    assert standing.sum() >= 2
    np.testing.assert_allclose(fz[standing], w[standing], rtol=0.05)


def test_facade_turns_mass_edits_into_factors():
    """set_actor_rigid_body_properties on an articulation: accepted (it was refused before), stored as factors, read back by
    get_actor_rigid_body_properties; a body welded to its parent has no mass of its own to edit."""
    from shifu_amd.isaacgym import gymapi
    import os
    gym = gymapi.acquire_gym()
    sp = gymapi.SimParams()
    sim = gym.create_sim(0, 0, gymapi.SIM_PHYSX, sp)
    opt = gymapi.AssetOptions()
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "shifu_amd", "assets")
    urdf = [os.path.join(dp, f) for dp, _, fs in os.walk(root) for f in fs if f == "a1.urdf"]
    assert urdf, "a1.urdf ships with the package"
    asset = gym.load_asset(sim, os.path.dirname(urdf[0]), "a1.urdf", opt)
    env = gym.create_env(sim, gymapi.Vec3(0, 0, 0), gymapi.Vec3(1, 1, 1), 1)
    h = gym.create_actor(env, asset, gymapi.Transform(), "a1", 0, 0)
    props = gym.get_actor_rigid_body_properties(env, h)
    m0 = props[0].mass
    props[0].mass = m0 + 2.0
    assert gym.set_actor_rigid_body_properties(env, h, props, recomputeInertia=True)
    a = env.actors[h]
    assert a.mass_scale is not None and abs(a.mass_scale[0] - (m0 + 2.0) / m0) < 1e-6 and (a.mass_scale[1:] == 1.0).all()
    assert abs(gym.get_actor_rigid_body_properties(env, h)[0].mass - (m0 + 2.0)) < 1e-5
    welded = [b for b, p in enumerate(props) if p.mass == 0.0]
    if welded:
        props[welded[0]].mass = 0.1
        with pytest.raises(NotImplementedError):
            gym.set_actor_rigid_body_properties(env, h, props, recomputeInertia=True)
