"""Self-collision (reference shifu/units/units.py:68 creates every actor with collision filter 0 =
self-collide): capsule pairs of the articulation's own links.  CPU: the oracle's segment-segment geometry against a
brute-force search, the model compiler's capsules / pair list for the A1, and what the contact does physically
(legs that are driven through each other stop at the surface, with equal and opposite forces; off = they pass).
tests/test_gpu_parity.py::test_self_collision_matches_oracle_bitwise holds the HIP path to the oracle."""
import numpy as np
import pytest

from shifu_amd import _abi
from shifu_amd.model import asset_path, compile_urdf
from tests.helpers import sim_params


def _brute(p1, q1, p2, q2, n=400):
    s = np.linspace(0, 1, n)
    A = p1[None] + s[:, None] * (q1 - p1)[None]
    B = p2[None] + s[:, None] * (q2 - p2)[None]
    d = np.linalg.norm(A[:, None] - B[None], axis=-1)
    return d.min()


def test_segment_closest_points_against_brute_force(oracle):
    rng = np.random.default_rng(0)
    segs = rng.uniform(-1, 1, (300, 4, 3))
    segs[:20, 1] = segs[:20, 0]                         # first segment degenerate (a sphere)
    segs[20:40, 3] = segs[20:40, 2]                     # second degenerate
    segs[40:50, 1] = segs[40:50, 0]; segs[40:50, 3] = segs[40:50, 2]   # both
    d = segs[50:80, 1] - segs[50:80, 0]
    segs[50:80, 3] = segs[50:80, 2] + d * rng.uniform(0.2, 2.0, (30, 1))   # parallel
    segs[80:100, 2] = 0.5 * (segs[80:100, 0] + segs[80:100, 1]) + rng.normal(0, 1e-3, (20, 3))   # (nearly) intersecting
    c = oracle.segment_closest(segs)
    for i in range(len(segs)):
        p1, q1, p2, q2 = segs[i]
        d_oracle = np.linalg.norm(c[i, 0] - c[i, 1])
        d_brute = _brute(p1, q1, p2, q2)
        assert d_oracle <= d_brute + 1e-9, (i, d_oracle, d_brute)          # the true minimum is never above a sampled one
        assert d_brute - d_oracle < 2e-2 * max(np.linalg.norm(q1 - p1), np.linalg.norm(q2 - p2), 1e-3) + 1e-9
        for pt, a, b in ((c[i, 0], p1, q1), (c[i, 1], p2, q2)):             # each point lies on its segment
            ab = b - a
            t = 0.0 if ab @ ab < 1e-20 else (pt - a) @ ab / (ab @ ab)
            assert -1e-9 <= t <= 1 + 1e-9 and np.linalg.norm(a + t * ab - pt) < 1e-9
    c32 = oracle.segment_closest(segs, f64=False)
    d64 = np.linalg.norm(c[:, 0] - c[:, 1], axis=1)
    d32 = np.linalg.norm(c32[:, 0] - c32[:, 1], axis=1)
    assert np.abs(d64 - d32).max() < 1e-5


def test_a1_capsules_and_pairs():
    cm = compile_urdf(asset_path("a1.urdf"), default_dof_drive_mode=_abi.DOF_MODE_EFFORT, self_collision=True)
    m, names = cm.blob, cm.body_names
    assert m.self_collide == 1 and m.ncap == 14 and m.npair == 78
    caps = [(names[m.cap_body[i]], m.cap_radius[i]) for i in range(m.ncap)]
    assert [c[0] for c in caps[:2]] == ["base", "base"] and abs(caps[0][1] - 0.057) < 1e-6   # trunk box: two side-by-side capsules
    assert sum(1 for n, _ in caps if n.endswith("_foot")) == 4 and sum(1 for n, _ in caps if n.endswith("_calf")) == 4
    pairs = {(names[m.cap_body[m.pair_a[k]]], names[m.cap_body[m.pair_b[k]]]) for k in range(m.npair)}
    assert ("FL_calf", "FR_calf") in pairs and ("base", "FL_thigh") in pairs and ("FL_foot", "RL_foot") in pairs
    for a, b in pairs:      # never two shapes of one rigid body, never bodies joined by a joint
        assert a != b and {a[3:], b[3:]} != {"thigh", "calf"} or a[:2] != b[:2]
        assert not (a[:2] == b[:2] and {a[3:], b[3:]} in ({"calf", "foot"}, {"thigh", "calf"}, {"thigh", "foot"}))
    off = compile_urdf(asset_path("a1.urdf"), default_dof_drive_mode=_abi.DOF_MODE_EFFORT)
    assert off.blob.self_collide == 0 and off.blob.npair == 78      # pairs are always listed, the flag decides


def _cross_legs(oracle, self_collision, steps=240):
    """A1 floating in zero gravity; the left front thigh is swung back and the left rear thigh forward (soft PD), so
    that the two left shanks are driven through each other under the trunk."""
    cm = compile_urdf(asset_path("a1.urdf"), default_dof_drive_mode=_abi.DOF_MODE_EFFORT, self_collision=self_collision)
    m = cm.blob
    for d in range(m.nd):
        m.damping[d] = 0.5
    sp = sim_params(gravity=(0.0, 0.0, 0.0))
    names = cm.dof_names
    q0 = np.array([0.1, 0.8, -1.5, 0.1, 0.8, -1.5, -0.1, 0.8, -1.5, -0.1, 0.8, -1.5], np.float64)
    tgt = q0.copy()
    for k, v in (("FL_thigh_joint", 2.2), ("FL_calf_joint", -1.0), ("RL_thigh_joint", -0.9), ("RL_calf_joint", -1.0)):
        tgt[names.index(k)] = v
    dof = np.zeros((m.nd, 2), np.float64); dof[:, 0] = q0
    root = np.zeros((1, 13), np.float64); root[0, 2] = 1.0; root[0, 6] = 1.0
    lim = np.array(m.effort[:m.nd], np.float64)
    fr = np.ones(1, np.float32)
    hist = []
    for k in range(steps):
        tau = np.clip(12.0 * (tgt - dof[:, 0]) - 1.0 * dof[:, 1], -lim, lim)
        contact, bs = oracle.step(m, sp, 1, dof, root, effort=tau, friction=fr, want_contact=True, want_body_state=True, f64=True)
        hist.append((contact.copy(), bs.copy(), dof[:, 0].copy()))
    return cm, hist


def _min_leg_gap(cm, bs):
    """Smallest surface distance between any left-front and left-rear leg capsule, from rigid_body_state."""
    from tests.helpers import quat_to_mat
    m = cm.blob
    best = 1e9
    caps = [(i, cm.body_names[m.cap_body[i]]) for i in range(m.ncap)]
    for i, ni in caps:
        for j, nj in caps:
            if ni.startswith("FL_") and nj.startswith("RL_"):
                seg = []
                for c, b in ((i, m.cap_body[i]), (j, m.cap_body[j])):
                    R, p = quat_to_mat(bs[b, 3:7]), bs[b, :3]
                    seg += [p + R @ np.array(m.cap_a[c]), p + R @ np.array(m.cap_b[c])]
                best = min(best, _brute(*seg, n=120) - m.cap_radius[i] - m.cap_radius[j])
    return best


def test_crossing_legs_stop_at_each_other_and_forces_balance(oracle):
    cm, on = _cross_legs(oracle, True)
    _, off = _cross_legs(oracle, False)
    gap_on = min(_min_leg_gap(cm, h[1]) for h in on[::4])
    gap_off = min(_min_leg_gap(cm, h[1]) for h in off[::4])
    assert gap_off < -0.02, gap_off                      # without self-collision the legs pass through each other
    assert gap_on > -0.003, gap_on                       # with it they stop at the surface (penalty contact: < 3 mm)
    # Only internal contacts act here (no ground), so the reported contact forces of all bodies should sum to zero.  Each
    # side of a self-contact is solved implicitly in its own body's acceleration (mass-ratio scaled, oracle
    # self_scales), which makes the pair's forces equal and opposite up to what the other forces on the two bodies
    # add within the step: 2 % of the force magnitude in the median, more only in the few steps of an impact.
    rel = []
    for contact, _, _ in on:
        mag = np.abs(contact).sum()
        if mag > 1.0:
            rel.append(np.linalg.norm(contact.sum(0)) / mag)
    rel = np.sort(np.array(rel))
    assert len(rel) > 50 and np.median(rel) < 0.05 and rel[int(0.9 * len(rel))] < 0.3, (len(rel), np.median(rel), rel[-5:])
    assert all(np.abs(c).sum() == 0 for c, _, _ in off)  # no ground, no self-collision: no contact force at all
