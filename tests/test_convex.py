"""The convex narrow phase of the oracle (oracle/shf_oracle.c: convex_manifold -- mesh colliders as convex hulls, SURVEY 8f f3;
the reference collides every link through a <mesh> collider, asset/urdf/abb_rod_description/urdf/abb_rod_isaac.urdf:38-113,
shifu/units/units.py:68) against known answers and against a brute-force separating-axis search in NumPy.  CPU only."""
import itertools

import numpy as np
import pytest

from oracle import pyoracle
from shifu_amd import _abi
from shifu_amd.model import hull_record, reduce_hull

I3 = np.eye(3)


def rot(axis, ang):
    axis = np.asarray(axis, float) / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return I3 + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K


def box_poly(R, p, h):
    v = np.array([p + R @ (np.array([sx, sy, sz]) * h) for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)])
    n = np.array([s * R[:, a] for a in range(3) for s in (1, -1)])
    ed = np.array([R[:, a] for a in range(3)])
    return v, n, ed


def hull_poly(h, R, p):
    v = p + h["verts"] @ R.T
    n = h["planes"][:, :3] @ R.T
    ed = np.array([v[b] - v[a] for a, b, _, _ in h["edges"]])
    return v, n, ed


def brute_sat(A, B):
    """Largest separation over every face normal of both polytopes and every pair of edge directions, with the full support
    functions (no Gauss-map filtering): the exact separation (> 0) / penetration depth (< 0) of two convex polytopes."""
    (va, na, ea), (vb, nb, eb) = A, B
    axes = [n for n in na] + [n for n in nb]
    for a in ea:
        for b in eb:
            c = np.cross(a, b)
            l = np.linalg.norm(c)
            if l > 1e-9 * np.linalg.norm(a) * np.linalg.norm(b):
                axes.append(c / l)
    axes = np.array(axes)
    pa, pb = va @ axes.T, vb @ axes.T                  # (nv, naxes)
    s1 = pb.min(0) - pa.max(0)                         # B beyond A along +axis
    s2 = pa.min(0) - pb.max(0)                         # ... along -axis
    return float(np.maximum(s1, s2).max())


def random_hull(rng, npts=40, scale=0.1):
    pts = rng.normal(size=(npts, 3)) * scale * rng.uniform(0.5, 1.5, 3)
    return reduce_hull(pts)


def test_a_box_resting_flat_on_a_bigger_box_is_held_at_its_four_bottom_corners():
    hA, hB = np.array([0.05, 0.04, 0.03]), np.array([0.3, 0.3, 0.05])
    pA, pB = np.array([0.02, -0.01, 0.05 + 0.03 + 0.002]), np.zeros(3)
    n, cs = pyoracle.convex_manifold(I3, pA, I3, pB, ha=hA, hb=hB)
    assert len(cs) == 4 and np.allclose(n, [0, 0, 1])
    got = sorted((round(float(r[0]), 6), round(float(r[1]), 6)) for r, _ in cs)
    want = sorted((round(pA[0] + sx * hA[0], 6), round(pA[1] + sy * hA[1], 6)) for sx in (-1, 1) for sy in (-1, 1))
    assert got == want
    for r, phi in cs:
        assert phi == pytest.approx(0.002, abs=1e-12) and r[2] == pytest.approx(0.05 + 0.001, abs=1e-12)   # midway between the surfaces


def test_a_bar_lying_across_a_ridge_with_no_vertex_inside_either_gets_the_overlap_rectangle():
    """The case rounds 1-5 could not see: an edge / face lying flat on a face, no corner of either box inside the other."""
    hA, hB = np.array([0.3, 0.02, 0.02]), np.array([0.03, 0.4, 0.05])          # bar along x, ridge along y
    pA, pB = np.array([0.0, 0.0, 0.05 + 0.02 - 0.001]), np.zeros(3)           # 1 mm deep
    n, cs = pyoracle.convex_manifold(I3, pA, I3, pB, ha=hA, hb=hB)
    assert len(cs) == 4 and np.allclose(n, [0, 0, 1])
    got = sorted((round(float(r[0]), 6), round(float(r[1]), 6)) for r, _ in cs)
    assert got == sorted((sx * 0.03, sy * 0.02) for sx in (-1, 1) for sy in (-1, 1))
    assert all(phi == pytest.approx(-0.001, abs=1e-12) for _, phi in cs)
    # neither the vertex families nor the edge test of rounds 1-5 produce anything here
    assert not pyoracle.box_box_edge(I3, pA, hA, I3, pB, hB)[0]
    # the same bar turned by 30 degrees about z: still four points, on the ridge's two long edges and the bar's two
    n, cs = pyoracle.convex_manifold(rot([0, 0, 1], 0.5), pA, I3, pB, ha=hA, hb=hB)
    assert len(cs) == 4 and np.allclose(n, [0, 0, 1])
    for r, phi in cs:
        assert abs(abs(r[0]) - 0.03) < 1e-9 and phi == pytest.approx(-0.001, abs=1e-12)


def test_two_bars_crossing_edge_over_edge_touch_in_one_point_like_the_box_edge_test():
    hA = hB = np.array([0.3, 0.02, 0.02])
    RA, RB = rot([1, 0, 0], np.pi / 4), rot([0, 0, 1], np.pi / 2) @ rot([1, 0, 0], np.pi / 4)     # both on edge, crossed
    d = 2 * 0.02 * np.sqrt(2.0)
    pA, pB = np.array([0.0, 0.0, d - 0.0015]), np.zeros(3)
    n, cs = pyoracle.convex_manifold(RA, pA, RB, pB, ha=hA, hb=hB)
    hit, phi_e, n_e, r_e = pyoracle.box_box_edge(RA, pA, hA, RB, pB, hB)
    assert hit and len(cs) == 1
    assert np.allclose(n, n_e, atol=1e-9) and np.allclose(cs[0][0], r_e, atol=1e-9) and cs[0][1] == pytest.approx(phi_e, abs=1e-6)     # (box_box_edge pads its radii by 1e-6)
    assert np.allclose(n, [0, 0, 1]) and cs[0][1] == pytest.approx(-0.0015, abs=1e-9)


def test_nothing_beyond_the_contact_offset_and_something_just_inside_it():
    hA, hB = np.array([0.05, 0.05, 0.05]), np.array([0.3, 0.3, 0.05])
    for gap, want in ((0.0101, 0), (0.0099, 4), (-0.004, 4)):
        n, cs = pyoracle.convex_manifold(I3, [0, 0, 0.1 + gap], I3, [0, 0, 0], ha=hA, hb=hB, offset=0.01)
        assert len(cs) == want, gap
        assert all(phi == pytest.approx(gap, abs=1e-9) for _, phi in cs)


def test_a_hull_resting_on_the_edge_of_a_box_is_held_only_where_the_box_is():
    """A prism (a reduced mesh collider) whose flat bottom overhangs the table's edge: the manifold is the part of the bottom
    face over the table, clipped at the table's edge."""
    pts = np.array([[sx * 0.1, sy * 0.06, 0.0] for sx in (-1, 1) for sy in (-1, 1)] + [[sx * 0.05, sy * 0.03, 0.08] for sx in (-1, 1) for sy in (-1, 1)])
    h = reduce_hull(pts)
    assert len(h["verts"]) == 8 and len(h["loops"]) == 6
    rec = hull_record(h, 0, np.zeros(3), I3)
    hB = np.array([0.3, 0.3, 0.05])
    pA = np.array([0.3 + 0.04, 0.0, 0.05 - 0.0005])               # the table ends at x = 0.3; 60 mm of the prism's bottom lie on it
    n, cs = pyoracle.convex_manifold(I3, pA, I3, np.zeros(3), hull_a=rec, hb=hB)
    assert len(cs) == 4 and np.allclose(n, [0, 0, 1])
    xs = sorted(round(float(r[0]), 6) for r, _ in cs)
    assert xs == [0.24, 0.24, 0.3, 0.3]                          # the prism's own corners at x = 0.24, the clip at the table's edge
    assert all(abs(abs(r[1]) - 0.06) < 1e-7 and phi == pytest.approx(-0.0005, abs=1e-7) for r, phi in cs)     # (the hull's vertices are float32)
    # standing on the table's edge with one bottom EDGE only (tilted about y): two points along that edge
    R = rot([0, 1, 0], 0.3)
    low = (pts @ R.T)[:, 2].min()
    n, cs = pyoracle.convex_manifold(R, [0.0, 0.0, 0.05 - low - 0.0005], I3, np.zeros(3), hull_a=rec, hb=hB)
    assert len(cs) == 2 and np.allclose(n, [0, 0, 1])
    assert all(phi == pytest.approx(-0.0005, abs=1e-6) for _, phi in cs)
    assert sorted(round(float(r[1]), 6) for r, _ in cs) == [-0.06, 0.06]


@pytest.mark.parametrize("kind", ["hull-box", "box-box"])
def test_4000_random_pairs_against_the_brute_force_separating_axis_search(kind):
    """2000 hull-box + 2000 box-box pairs placed near touching."""
    rng = np.random.default_rng(11 if kind == "hull-box" else 12)
    hulls = [random_hull(rng) for _ in range(16)]
    recs = [hull_record(h, 0, np.zeros(3), I3) for h in hulls]
    offset, n_contact, n_edge, n_apart, n_missed = 0.01, 0, 0, 0, 0
    tol = 3e-7 if kind == "hull-box" else 1e-9          # a hull's planes are stored in float32
    for trial in range(2000):
        Ra, Rb = rot(rng.normal(size=3), rng.uniform(0, np.pi)), rot(rng.normal(size=3), rng.uniform(0, np.pi))
        hb = rng.uniform(0.03, 0.15, 3)
        if kind == "hull-box":
            k = trial % len(hulls)
            A = hull_poly(hulls[k], Ra, np.zeros(3))
            reach = np.linalg.norm(hulls[k]["verts"], axis=1).max()
            kw = dict(hull_a=recs[k])
        else:
            ha = rng.uniform(0.03, 0.15, 3)
            A = box_poly(Ra, np.zeros(3), ha)
            reach = np.linalg.norm(ha)
            kw = dict(ha=ha)
        # place B so that the pair is near touching: along a random direction at about the sum of the two reaches, scaled
        d = rng.normal(size=3)
        d /= np.linalg.norm(d)
        pb = d * (reach + np.linalg.norm(hb)) * rng.uniform(0.3, 1.05)
        # ... then slide it along d until the exact separation is within +-15 mm of touching (most pairs would otherwise be
        # far apart or deeply interpenetrating)
        for _ in range(6):
            s = brute_sat(A, box_poly(Rb, pb, hb))
            pb = pb - d * (s - rng.uniform(-0.008, 0.013))
        B = box_poly(Rb, pb, hb)
        s_true = brute_sat(A, B)
        n, cs, seps = pyoracle.convex_manifold(Ra, np.zeros(3), Rb, pb, hb=hb, offset=offset, seps=True, **kw)
        # (1) overlapping: the oracle's largest separation over its Gauss-map filtered axes is the exact penetration depth.  Apart:
        # the closest features may be vertices, then no face / edge-pair axis measures the distance; both numbers are lower bounds of
        # it, the brute-force one (full support along every axis) the tighter, and both are positive
        s_or = max(seps)
        if s_true < -tol:
            assert s_or == pytest.approx(s_true, abs=tol), trial
        else:
            assert s_or <= s_true + tol and (s_true < tol or s_or > 0), (trial, s_or, s_true)
        # (2) contacts exactly when that separation is inside the contact offset (never for a pair further apart than the offset
        # by the exact measure ... in the other direction: a reported contact's gap never exceeds the true one)
        if s_or >= offset:
            assert len(cs) == 0 and s_true >= offset - tol, trial
            n_apart += 1
            continue
        if -0.02 < s_or < -tol:          # overlapping: always a contact (apart but inside the offset, vertex against edge: none is fine)
            assert len(cs) >= 1, (trial, s_true, seps)
        n_missed += len(cs) == 0
        if not cs:
            continue
        n_contact += 1
        n_edge += len(cs) == 1 and seps[2] > max(seps[0], seps[1])
        assert abs(np.linalg.norm(n) - 1) < 1e-6
        # (3) the reported normal is a (near-)optimal separating direction, pointing from B to A
        sep_n = (A[0] @ n).min() - (B[0] @ n).max()
        sface = max(seps[0], seps[1])
        fallback = len(cs) == 1 and abs(cs[0][1] - seps[2]) < 1e-12 and not seps[2] > sface + 0.05 * abs(sface) + 5e-4
        if fallback:
            # the preferred face's manifold was empty (the bodies meet beside that face): the best edge pair's crossing instead
            assert sep_n == pytest.approx(seps[2], abs=10 * tol + 1e-7), (trial, sep_n, seps)
        else:
            # within the two preference margins of the best axis -- or, when the best axis is an edge pair whose edges pass beside
            # each other (no crossing point inside both), of the best face
            assert sep_n >= min(s_or, sface) - 0.05 * abs(s_or) - 1.1e-3 - tol, (trial, sep_n, seps)
        for r, phi in cs:
            # (4) every gap is inside the offset and no deeper than the exact penetration; (5) the point sits between the two
            # surfaces: within |gap| / 2 (+ rounding) of both polytopes along the normal
            assert phi < offset and phi >= min(max(seps[0], seps[1]), seps[2]) - tol, (trial, phi, seps)       # (a face contact measures from the reference face's plane)
            sd_a = ((A[1] @ r) - (A[1] @ A[0].T).max(1)).max() if kind == "box-box" else (hulls[k]["planes"][:, :3] @ (Ra.T @ r) - hulls[k]["planes"][:, 3]).max()
            sd_b = ((B[1] @ r) - (B[1] @ B[0].T).max(1)).max()
            assert abs(sd_a) <= abs(phi) + 1e-4 and abs(sd_b) <= abs(phi) + 1e-4, (trial, sd_a, sd_b, phi)
    assert n_contact > 750 and n_apart > 150 and n_edge > 25 and n_missed < 0.08 * n_contact, (n_contact, n_apart, n_edge, n_missed)


def test_float_build_agrees_with_the_double_build():
    rng = np.random.default_rng(5)
    h = random_hull(rng)
    rec = hull_record(h, 0, np.zeros(3), I3)
    worst = 0.0
    for _ in range(300):
        Ra, Rb = rot(rng.normal(size=3), rng.uniform(0, np.pi)), rot(rng.normal(size=3), rng.uniform(0, np.pi))
        hb = rng.uniform(0.05, 0.15, 3)
        d = rng.normal(size=3)
        d /= np.linalg.norm(d)
        pb = d * 0.2
        for _ in range(5):
            pb = pb - d * (brute_sat(hull_poly(h, Ra, np.zeros(3)), box_poly(Rb, pb, hb)) + 0.002)
        n64, c64 = pyoracle.convex_manifold(Ra, np.zeros(3), Rb, pb, hull_a=rec, hb=hb)
        n32, c32 = pyoracle.convex_manifold(Ra, np.zeros(3), Rb, pb, hull_a=rec, hb=hb, f64=False)
        if len(c64) == len(c32) and len(c64) and np.allclose(n64, n32, atol=1e-4):
            worst = max(worst, max(abs(a[1] - b[1]) for a, b in zip(c64, c32)))
    assert worst < 2e-6


def test_reduce_hull_respects_the_limits_and_is_inscribed():
    rng = np.random.default_rng(3)
    pts = rng.normal(size=(2000, 3)) * [0.1, 0.2, 0.05]
    h = reduce_hull(pts)
    assert len(h["verts"]) <= _abi.HULL_MAX_VERTS and len(h["loops"]) <= _abi.HULL_MAX_FACES and len(h["edges"]) <= _abi.HULL_MAX_EDGES
    assert max(map(len, h["loops"])) <= _abi.HULL_MAX_FACE_VERTS
    # Euler: V - E + F = 2; every vertex of the polytope is one of the cloud's points; the planes hold all vertices
    assert len(h["verts"]) - len(h["edges"]) + len(h["loops"]) == 2
    assert all(np.min(np.linalg.norm(pts.astype(np.float32) - v, axis=1)) < 1e-6 for v in h["verts"])
    assert ((h["verts"] @ h["planes"][:, :3].T) <= h["planes"][:, 3] + 1e-7).all()
    # loops are counter-clockwise seen from outside
    for pl, loop in zip(h["planes"], h["loops"]):
        a, b, c = h["verts"][loop[0]], h["verts"][loop[1]], h["verts"][loop[2]]
        assert np.dot(np.cross(b - a, c - b), pl[:3]) > 0
