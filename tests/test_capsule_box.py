"""Capsule-vs-box geometry (capsules as a native shape, asset_config.py:32-46 replace_cylinder_with_capsule;
the ABB rod, abb_task.ROD_CAPSULE): the oracle's closed-form closest-point parameter against dense sampling."""
import numpy as np
import pytest

from oracle import pyoracle as O


def _rand_rot(rng, n):
    q = rng.normal(size=(n, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    x, y, z, w = q.T
    return np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                     2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                     2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], axis=1)


def _dist(bR, bpos, h, pts):
    """distance of world points (n,m,3) to the boxes (n)"""
    R = bR.reshape(-1, 3, 3)
    loc = np.einsum("nji,nmj->nmi", R, pts - bpos[:, None, :])            # R^T (p - bpos)
    e = loc - np.clip(loc, -h[:, None, :], h[:, None, :])
    return np.linalg.norm(e, axis=2)


def _cases(rng, n):
    bR = _rand_rot(rng, n)
    bpos = rng.uniform(-0.3, 0.3, (n, 3))
    h = rng.uniform(0.02, 0.2, (n, 3))
    c0 = bpos + rng.uniform(-0.5, 0.5, (n, 3))
    s = rng.uniform(-0.6, 0.6, (n, 3))
    return bR, bpos, h, c0, s


def test_closest_point_parameter_matches_dense_sampling():
    rng = np.random.default_rng(0)
    n = 4000
    bR, bpos, h, c0, s = _cases(rng, n)
    t = O.segment_box_param(bR, bpos, h, c0, s, f64=True)
    assert ((t >= 0) & (t <= 1)).all()
    ts = np.linspace(0, 1, 4001)
    pts = c0[:, None, :] + ts[None, :, None] * s[:, None, :]
    dmin = _dist(bR, bpos, h, pts).min(1)
    dstar = _dist(bR, bpos, h, (c0 + t[:, None] * s)[:, None, :])[:, 0]
    # never worse than the best sample -- beyond the flat-sample tolerance (5e-4 rad: up to 5e-4 |s| along a nearly parallel stretch)
    assert (dstar <= dmin + 5e-4 * np.linalg.norm(s, axis=1) + 1e-12).all(), float((dstar - dmin).max())
    assert np.mean(dstar <= dmin + 1e-12) > 0.995
    assert (dmin - dstar).max() < 2e-4                                      # and the samples come that close (sanity of the checker)
    assert (dmin == 0).sum() > 100 and (t == 0).sum() > 100 and (t == 1).sum() > 100 and ((t > 0) & (t < 1)).sum() > 1000


def test_axis_aligned_and_degenerate_segments():
    """parallel to a face (a stretch of exact zeros of g gives its midpoint), through the box, zero length"""
    I = np.eye(3).reshape(1, 9)
    z3 = np.zeros((1, 3)); h = np.array([[0.1, 0.1, 0.1]])
    par = O.segment_box_param(I, z3, h, np.array([[-0.05, 0.0, 0.3]]), np.array([[0.1, 0.0, 0.0]]))     # above the top face, inside its footprint
    assert par[0] == 0.5
    long = O.segment_box_param(I, z3, h, np.array([[-0.5, 0.0, 0.3]]), np.array([[1.0, 0.0, 0.0]]))     # overhanging both sides: middle of the face stretch
    assert abs(long[0] - 0.5) < 1e-12
    thru = O.segment_box_param(I, z3, h, np.array([[-0.5, 0.01, 0.02]]), np.array([[1.0, 0.0, 0.0]]))   # through the box
    assert 0.4 <= thru[0] <= 0.6
    pt = O.segment_box_param(I, z3, h, np.array([[0.3, 0.2, 0.1]]), np.zeros((1, 3)))                  # a sphere
    assert 0.0 <= pt[0] <= 1.0


def test_float_build_tracks_the_double_build():
    rng = np.random.default_rng(1)
    bR, bpos, h, c0, s = _cases(rng, 2000)
    t64 = O.segment_box_param(bR, bpos, h, c0, s, f64=True)
    t32 = O.segment_box_param(bR, bpos, h, c0, s, f64=False)
    d64 = _dist(bR, bpos, h, (c0 + t64[:, None] * s)[:, None, :])[:, 0]
    d32 = _dist(bR, bpos, h, (c0 + t32[:, None].astype(np.float64) * s)[:, None, :])[:, 0]
    assert np.abs(d32 - d64).max() < 2e-6                    # the distance is what enters the contact law; t itself is ill-conditioned when parallel


def test_line_contact_has_two_points_the_ends_of_the_flat_stretch():
    """ShfModel.sph_part (oracle segment_box_contact): a capsule lying along a face is a LINE contact held at both ends of
    the stretch over which the segment is equally close -- the face's edges or the segment's own ends, whichever come
    first; anything else is the single closest point (part 0) and part 1 does not exist."""
    I = np.eye(3).reshape(1, 9)
    z3 = np.zeros((1, 3)); h = np.array([[0.1, 0.1, 0.1]])
    inside = (np.array([[-0.05, 0.0, 0.3]]), np.array([[0.1, 0.0, 0.0]]))        # above the top face, within its footprint
    over = (np.array([[-0.5, 0.0, 0.3]]), np.array([[1.0, 0.0, 0.0]]))           # overhanging both edges
    half = (np.array([[0.0, 0.0, 0.3]]), np.array([[0.4, 0.0, 0.0]]))            # from the middle out over one edge
    for (c0, s), want in ((inside, (0.0, 1.0)), (over, (0.4, 0.6)), (half, (0.0, 0.25))):
        for part in (0, 1):
            ok, t = O.segment_box_contact(I, z3, h, c0, s, part)
            assert ok[0] and abs(t[0] - want[part]) < 1e-12, (c0, s, part, t)
    # tilted by 1e-2 rad: a point contact at the low end, no second point
    c0, s = np.array([[-0.05, 0.0, 0.3]]), np.array([[0.1, 0.0, 1e-3]])
    ok0, t0 = O.segment_box_contact(I, z3, h, c0, s, 0)
    ok1, _ = O.segment_box_contact(I, z3, h, c0, s, 1)
    assert ok0[0] and t0[0] == 0.0 and not ok1[0]
    # generic skew segments: part 0 is segment_box_param's point and there is no part 1; where a stretch exists its ends
    # bracket the classic midpoint
    rng = np.random.default_rng(5)
    bR, bpos, hh, cc, ss = _cases(rng, 500)
    ok0, t0 = O.segment_box_contact(bR, bpos, hh, cc, ss, 0)
    ok1, t1 = O.segment_box_contact(bR, bpos, hh, cc, ss, 1)
    tp = O.segment_box_param(bR, bpos, hh, cc, ss)
    assert ok0.all()
    single = ~ok1
    assert single.sum() > 400 and np.array_equal(t0[single], tp[single])
    both = ok1
    assert (t0[both] <= tp[both]).all() and (tp[both] <= t1[both]).all() and np.allclose(0.5 * (t0[both] + t1[both]), tp[both])
