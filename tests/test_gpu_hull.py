"""HIP twins of the convex narrow phase (csrc/shf_hull.h against oracle/shf_oracle.c: convex_manifold): mesh colliders as convex
hulls against box actors (family H of the link contacts), the clipped face manifold for box pairs that touch without a vertex or
an edge crossing (ShfScene.flags), under both contact solvers and at 16 / 32 / 64 lanes per env -- every tensor bit for bit.
SURVEY 8f f3; reference: asset/urdf/abb_rod_description/urdf/abb_rod_isaac.urdf:38-113, shifu/units/units.py:68."""
import numpy as np
import pytest
import torch

from shifu_amd import _abi
from tests import helpers as H
from tests import kat_models as K

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def oracle():
    from oracle import pyoracle
    pyoracle.build()
    return pyoracle


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")


def _params(solver):
    from shifu_amd.backend import default_sim_params
    return default_sim_params(solver=solver) if solver == "pgs" else H.sim_params(angular_damping=0.5)


def _scene_on_gpu(cm, sp, boxes, roots, n, group, flags=0, quats=None):
    from shifu_amd.backend import Sim
    m = cm.blob
    A = 1 + len(boxes)
    dof = np.zeros((n * m.nd, 2), np.float32)
    root = np.zeros((n * A, 13), np.float32)
    root[:, 6] = 1.0
    for k, p in enumerate(roots):
        root[1 + k::A, :3] = p
        if quats is not None:
            root[1 + k::A, 3:7] = quats[k]
    sim = Sim(sp, "cuda:0")
    sim.set_plane(1.0)
    sim.set_articulation(m)
    if cm.hulls is not None:
        sim.set_hulls(cm.hulls)
    if flags:
        sim.set_scene_flags(flags)
    for b in boxes:
        sim.add_box(b)
    sim.finalize(n, 0, group=group)
    sim.tensors[_abi.T_SIM_DOF].copy_(torch.from_numpy(dof))
    sim.tensors[_abi.T_SIM_ROOT].copy_(torch.from_numpy(root))
    return sim, dof, root


def _quat(axis, ang):
    a = np.asarray(axis, float) / np.linalg.norm(axis)
    return tuple(np.sin(ang / 2) * a) + (np.cos(ang / 2),)


def _twin(oracle, name, cm, sp, boxes, roots, n, group, v, steps, flags=0, quats=None, load_body=None, seed=3, jitter=0.004, tilt=0.0):
    m = cm.blob
    rng = np.random.default_rng(seed)
    sim, dof, root = _scene_on_gpu(cm, sp, boxes, roots, n, group, flags, quats)
    A = 1 + len(boxes)
    # the envs of a wavefront must differ: shift / tilt the boxes a little, vary the drive speed
    for k in range(len(boxes)):
        if not boxes[k].fixed:
            root[1 + k::A, 0] += rng.uniform(-jitter, jitter, n).astype(np.float32)
            root[1 + k::A, 1] += rng.uniform(-jitter, jitter, n).astype(np.float32)
            if tilt:
                for e in range(n):
                    root[(1 + k) + e * A, 3:7] = _quat(rng.normal(size=3), rng.uniform(0, tilt))
    vt = (v * rng.uniform(0.6, 1.0, n * m.nd)).astype(np.float32)
    sim.tensors[_abi.T_SIM_ROOT].copy_(torch.from_numpy(root))
    oracle.dropped(reset=True)
    sim.tensors[_abi.T_DROPPED].zero_()
    seen = 0.0
    with oracle.scene_extras(hulls=cm.hulls, flags=flags):
        for it in range(steps):
            sim.set_dof_command(_abi.T_VEL_TARGET, torch.from_numpy(vt).cuda())
            sim.step()
            sim.refresh(_abi.REFRESH_ALL)
            contact, bstate, _ = oracle.scene_step(m, sp, boxes, n, dof, root, vel_target=vt, friction=np.ones(n, np.float32))
            torch.cuda.synchronize()
            np.testing.assert_array_equal(sim.tensors[_abi.T_DOF_STATE].cpu().numpy(), dof, err_msg=f"{name}: dof step {it}")
            np.testing.assert_array_equal(sim.tensors[_abi.T_ROOT_STATE].cpu().numpy(), root, err_msg=f"{name}: root step {it}")
            np.testing.assert_array_equal(sim.tensors[_abi.T_CONTACT].cpu().numpy(), contact, err_msg=f"{name}: contact step {it}")
            seen += float(np.abs(contact.reshape(n, -1, 3)[:, load_body if load_body is not None else m.nb - 1]).sum())
    assert seen > 0, f"{name}: never touched"
    assert int(sim.tensors[_abi.T_DROPPED].sum()) == oracle.dropped(reset=True), name
    sim.destroy() if hasattr(sim, "destroy") else None


@pytest.mark.parametrize("solver,group", [("compliant", 16), ("compliant", 32), ("compliant", 64), ("pgs", 32)])
def test_hull_contacts_match_oracle_bitwise(oracle, solver, group):
    """Family (H): a mesh collider (a frustum; a 24-vertex reduced point cloud) on a rail (1) driven down flat onto a fixed table,
    (2) onto the table's edge, tilted boxes, (3) pushing a free cube, (4) a free cube tumbling onto it -- shf_sim_step against the
    oracle, the envs of a wavefront perturbed."""
    _need_gpu()
    from shifu_amd.abb_task import box_desc
    sp = _params(solver)
    rng = np.random.default_rng(8)
    cloud = rng.normal(size=(300, 3)) * [0.08, 0.05, 0.04] + [0.0, 0.0, 0.45]
    low = [[v[0], v[1], v[2] - 0.4 + 0.01] for v in K.prism_verts(a=0.05, b=0.05, top=0.6, h=0.1)]
    scenes = [
        ("frustum on table", K.hull_pusher_model(K.prism_verts()), [box_desc((0.6, 0.6, 0.1), 0.0, 0.5, True, (0.0, 0.0, 0.3))], [(0.0, 0.0, 0.3)], None, 0.5, 130, 0.0),
        ("frustum on the table's edge", K.hull_pusher_model(K.prism_verts()), [box_desc((0.6, 0.6, 0.1), 0.0, 0.5, True, (0.36, 0.0, 0.3))], [(0.36, 0.0, 0.3)], [_quat((0, 0, 1), 0.3)], 0.5, 130, 0.0),
        ("cloud hull on a tilted slab", K.hull_pusher_model(cloud), [box_desc((0.5, 0.5, 0.06), 0.0, 0.5, True, (0.0, 0.0, 0.3))], [(0.0, 0.0, 0.3)], [_quat((1, 0.3, 0), 0.25)], 0.5, 110, 0.0),
        ("frustum pushes a cube", K.hull_pusher_model(low, axis="1 0 0"), [box_desc((0.1, 0.1, 0.1), 0.5, 0.6, False, (0.2, 0.0, 0.05))], [(0.2, 0.0, 0.0499)], None, 0.4, 170, 0.0),
        ("cubes tumble onto the hull", K.hull_pusher_model(cloud, axis="1 0 0"), [box_desc((0.06, 0.05, 0.04), 0.3, 0.6, False, (0.0, 0.0, 0.6))], [(0.01, 0.0, 0.6)], None, 0.02, 150, 1.5),
    ]
    for name, cm, boxes, roots, quats, v, steps, tilt in scenes:
        assert cm.blob.nhull == 1 and cm.hulls is not None
        _twin(oracle, name, cm, sp, boxes, roots, 9, group, v, steps, quats=quats, tilt=tilt)


@pytest.mark.parametrize("solver,group", [("compliant", 16), ("compliant", 32), ("compliant", 64), ("pgs", 32)])
def test_face_manifolds_match_oracle_bitwise(oracle, solver, group):
    """ShfScene.flags = SHF_SCENE_FACE_MANIFOLD: (1) free bars set down across a fixed ridge (no vertex of either inside the other),
    slightly turned and shifted per env; (2) the ram's box volume lying across a fixed slab's corner region (family F)."""
    _need_gpu()
    from shifu_amd.abb_task import box_desc
    sp = _params(solver)
    far = K.box_pusher_model(size=(0.02, 0.02, 0.02), centre=(1.5, 0.0, 0.5))
    ridge = box_desc((0.06, 0.8, 0.1), 0.0, 0.6, True, (0.0, 0.0, 0.05))
    bar = box_desc((0.6, 0.04, 0.04), 1.0, 0.6, False, (0.0, 0.0, 0.1 + 0.02 + 0.002))
    _twin(oracle, "bars across a ridge", far, sp, [ridge, bar], [(0.0, 0.0, 0.05), (0.01, 0.02, 0.122)], 9, group, 0.0, 140,
          flags=_abi.SCENE_FACE_MANIFOLD, load_body=far.blob.nb + 1, tilt=0.02)
    # family F: a long flat ram (box volume 0.5 x 0.04 x 0.04) driven down across a narrow fixed slab
    ram = K.box_pusher_model(size=(0.5, 0.04, 0.04), centre=(0.0, 0.0, 0.5), axis="0 0 -1")
    slab = box_desc((0.05, 0.6, 0.1), 0.0, 0.5, True, (0.0, 0.0, 0.3))
    _twin(oracle, "ram across a slab", ram, sp, [slab], [(0.0, 0.0, 0.3)], 9, group, 0.5, 120, flags=_abi.SCENE_FACE_MANIFOLD)


@pytest.mark.parametrize("solver", ["pgs", "compliant"])
def test_fused_abb_step_with_hull_links_matches_oracle_bitwise(oracle, solver):
    """FusedAbbEnv(link_shapes='hull'): config 5 with the links as the (reduced) convex hulls of the reference's collision meshes
    (abb_rod_isaac.urdf:38-113) and the face manifold on -- 512 envs x 40 vec-steps, every tensor against the oracle.  The arm is
    started folded down onto the table in a third of the envs so that hull contacts carry load."""
    _need_gpu()
    from shifu_amd.gym.abb_fused import FusedAbbEnv
    from tests.test_gpu_parity import _ABB_SIM_T, _ABB_T
    n = 512
    env = FusedAbbEnv(num_envs=n, seed=5, link_shapes="hull", solver=solver, **({} if solver == "pgs" else {"group": 16}))
    assert env.cm.blob.nhull == 7 and env.face_manifold and env.mapping == "body"
    assert "DynDims8DynSceneLb1ELi0ELb%dELb1EE" % int(solver == "pgs") in env.task.kernel_symbol()
    # fold some arms down: joint 2 / 3 / 5 towards the table
    dof = env.sim.tensors[_abi.T_DOF_STATE].view(n, -1, 2)
    g = torch.Generator().manual_seed(1)
    bend = torch.rand(n, generator=g)
    dof[::3, 1, 0] += (0.5 + 0.4 * bend[::3]).cuda()
    dof[::3, 2, 0] += (0.3 * bend[::3]).cuda()
    env.sim.tensors[_abi.T_SIM_DOF].copy_(env.sim.tensors[_abi.T_DOF_STATE])
    env.task.tensors[_abi.ABB_EP_LEN].copy_(torch.randint(150, 201, (n,)))
    torch.cuda.synchronize()
    bufs = {k: env.sim.tensors[t].cpu().numpy().copy() for k, t in _ABB_SIM_T.items()}
    bufs.update({k: env.task.tensors[t].cpu().numpy().copy() for k, t in _ABB_T.items()})
    rng = np.random.default_rng(4)
    resets, load = 0, 0.0
    with oracle.scene_extras(hulls=env.cm.hulls, flags=_abi.SCENE_FACE_MANIFOLD):
        for it in range(40):
            raw = (2 * rng.random((n, 3)) - 1).astype(np.float32)
            env.task.step(torch.from_numpy(raw).cuda())
            oracle.abb_step(env.cm.blob, env.sim_params, env.boxes, env.task_params, n, 0, bufs, raw)
            resets += int(bufs["reset"].sum())
            load += float(np.abs(bufs["contact"].reshape(n, -1, 3)[:, 1:6]).sum())
    torch.cuda.synchronize()
    for k, t in list(_ABB_SIM_T.items()) + list(_ABB_T.items()):
        got = (env.sim.tensors if k in _ABB_SIM_T else env.task.tensors)[t].cpu().numpy().reshape(bufs[k].shape)
        np.testing.assert_array_equal(got, bufs[k], err_msg=k)
    assert resets > 20 and load > 0.0, (resets, load)


@pytest.mark.parametrize("lanes", [16, 32, 64])
def test_the_narrow_phase_itself_matches_the_oracle_on_random_pairs(oracle, lanes):
    """shf_convex_manifold (the device narrow phase on its own) against the oracle's convex_manifold, float32, bit for bit: 3000
    hull-box and 3000 box-box pairs placed near touching -- thin plates, deep overlaps and separated pairs included."""
    _need_gpu()
    import ctypes as C
    from shifu_amd import _lib
    from shifu_amd.model import hull_record, reduce_hull
    from tests.test_convex import box_poly, brute_sat, hull_poly, rot
    rng = np.random.default_rng(100 + lanes)
    h = reduce_hull(rng.normal(size=(60, 3)) * [0.1, 0.07, 0.05])
    rec = hull_record(h, 0, np.zeros(3), np.eye(3))
    for kind in ("hull", "box"):
        n = 3000
        rows = np.zeros((n, 30), np.float32)
        for i in range(n):
            Ra, Rb = rot(rng.normal(size=3), rng.uniform(0, np.pi)), rot(rng.normal(size=3), rng.uniform(0, np.pi))
            ha = rng.uniform(0.02, 0.12, 3) * (np.array([1, 1, 0.03]) if i % 7 == 0 else 1)      # every seventh a thin plate
            hb = rng.uniform(0.02, 0.15, 3) * (np.array([1, 1, 0.02]) if i % 5 == 0 else 1)
            A = hull_poly(h, Ra, np.zeros(3)) if kind == "hull" else box_poly(Ra, np.zeros(3), ha)
            d = rng.normal(size=3)
            d /= np.linalg.norm(d)
            pb = d * 0.3
            for _ in range(5):
                pb = pb - d * (brute_sat(A, box_poly(Rb, pb, hb)) - rng.uniform(-0.03, 0.013))
            off = rng.normal(size=3) * 0.3                       # both bodies away from the origin, like link poses about O
            rows[i] = np.concatenate([Ra.reshape(9), off, ha, Rb.reshape(9), pb + off, hb])
        want = np.zeros((n, 20), np.float32)
        for i in range(n):
            nn, cs = oracle.convex_manifold(rows[i, :9].reshape(3, 3), rows[i, 9:12], rows[i, 15:24].reshape(3, 3), rows[i, 24:27],
                                            ha=rows[i, 12:15], hb=rows[i, 27:30], hull_a=rec if kind == "hull" else None, f64=False)
            want[i, 0] = len(cs)
            want[i, 1:4] = nn
            for q, (r, phi) in enumerate(cs):
                want[i, 4 + 4 * q:7 + 4 * q] = r
                want[i, 7 + 4 * q] = phi
        din = torch.from_numpy(rows).cuda()
        dout = torch.zeros(n, 20, device="cuda")
        dh = torch.frombuffer(bytearray(bytes(rec)), dtype=torch.uint8).cuda() if kind == "hull" else None
        _lib.check(_lib.lib().shf_convex_manifold(n, C.c_void_p(din.data_ptr()), C.c_void_p(dh.data_ptr()) if dh is not None else None,
                                                  C.c_float(0.01), lanes, C.c_void_p(dout.data_ptr()), None))
        torch.cuda.synchronize()
        got = dout.cpu().numpy()
        bad = np.nonzero((got != want).any(1))[0]
        assert len(bad) == 0, (kind, len(bad), bad[:5], got[bad[:2]], want[bad[:2]])
        assert (want[:, 0] == 0).sum() > 50 and (want[:, 0] == 1).sum() > 50 and (want[:, 0] == 4).sum() > 200
