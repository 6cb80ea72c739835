"""Trimesh terrain contact: the warped-grid query of the oracle against the explicit triangle mesh
convert_heightfield_to_trimesh builds (isaac_gym.py:369-385 hands that mesh to gym.add_triangle_mesh)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pyoracle
from shifu_amd import _abi
from shifu_amd.isaacgym import terrain_utils as tu

HS, VS = 0.1, 0.005


def _terrain(hf, warped=1, border=0.0):
    t = _abi.ShfTerrain()
    t.rows, t.cols = hf.shape
    t.hscale, t.vscale, t.border, t.friction, t.warped = HS, VS, border, 1.0, warped
    return t


def _brute_force(vertices, triangles, pts):
    """Highest triangle of the explicit mesh over each point (float64, every triangle)."""
    out = np.full(len(pts), -np.inf)
    V = vertices.astype(np.float64)
    for tri in triangles:
        p0, p1, p2 = V[tri[0]], V[tri[1]], V[tri[2]]
        d = (p1[0] - p0[0]) * (p2[1] - p0[1]) - (p2[0] - p0[0]) * (p1[1] - p0[1])
        if abs(d) < 1e-12:
            continue
        for k, (x, y) in enumerate(pts):
            l1 = ((x - p0[0]) * (p2[1] - p0[1]) - (p2[0] - p0[0]) * (y - p0[1])) / d
            l2 = ((p1[0] - p0[0]) * (y - p0[1]) - (x - p0[0]) * (p1[1] - p0[1])) / d
            if l1 >= -1e-9 and l2 >= -1e-9 and l1 + l2 <= 1 + 1e-9:
                out[k] = max(out[k], p0[2] + l1 * (p1[2] - p0[2]) + l2 * (p2[2] - p0[2]))
    return out


def test_warp_map_reproduces_the_mesh_vertices():
    rng = np.random.default_rng(0)
    hf = (rng.integers(0, 4, (14, 11)) * 25).astype(np.int16)
    v, _ = tu.convert_heightfield_to_trimesh(hf, HS, VS, 0.75)
    w = tu.trimesh_warp_map(hf, HS, VS, 0.75)
    ii, jj = np.meshgrid(np.arange(14), np.arange(11), indexing="ij")
    dx, dy = (w & 3).astype(int) - 1, ((w >> 2) & 3).astype(int) - 1
    np.testing.assert_allclose(v[:, 0].reshape(14, 11), (ii + dx) * HS, atol=1e-6)
    np.testing.assert_allclose(v[:, 1].reshape(14, 11), (jj + dy) * HS, atol=1e-6)
    assert (dx != 0).any() and (dy != 0).any()
    # bits 4-7: the rows / columns of neighbouring cells a query inside cell (i, j) has to search.  Conservative: every
    # neighbour triangle that covers a point strictly inside the cell has its row and column flagged; tight: nothing is
    # flagged where no vertex of the 4 x 4 neighbourhood moved.
    moved = (dx != 0) | (dy != 0)
    X, Y = ii + dx, jj + dy
    fr = np.linspace(0.04, 0.96, 9)
    for i in range(13):
        for j in range(10):
            hint = int(w[i, j]) >> 4
            if not moved[max(i - 1, 0):i + 3, max(j - 1, 0):j + 3].any():
                assert hint == 0, (i, j, hint)
            px, py = np.meshgrid(i + fr, j + fr, indexing="ij")
            for a in (-1, 0, 1):
                for b in (-1, 0, 1):
                    ci, cj = i + a, j + b
                    if (a == 0 and b == 0) or not (0 <= ci < 13 and 0 <= cj < 10):
                        continue
                    covers = False
                    for vs in (((ci, cj), (ci + 1, cj + 1), (ci, cj + 1)), ((ci, cj), (ci + 1, cj), (ci + 1, cj + 1))):
                        (x0, y0), (x1, y1), (x2, y2) = [(X[v], Y[v]) for v in vs]
                        d = (x1 - x0) * (y2 - y0) - (x2 - x0) * (y1 - y0)
                        if abs(d) < 1e-12:
                            continue
                        l1 = ((px - x0) * (y2 - y0) - (x2 - x0) * (py - y0)) / d
                        l2 = ((x1 - x0) * (py - y0) - (px - x0) * (y1 - y0)) / d
                        covers |= bool(((l1 >= 0) & (l2 >= 0) & (l1 + l2 <= 1)).any())
                    if covers:
                        need = (1 if a == -1 else 2 if a == 1 else 0) | (4 if b == -1 else 8 if b == 1 else 0)
                        assert hint & need == need, (i, j, a, b, hint)


@pytest.mark.parametrize("threshold", [None, 0.75])
def test_warped_query_equals_the_explicit_mesh(threshold):
    rng = np.random.default_rng(1 if threshold else 2)
    hf = (rng.integers(0, 5, (13, 12)) * 18).astype(np.int16)
    v, tris = tu.convert_heightfield_to_trimesh(hf, HS, VS, threshold)
    packed = tu.pack_trimesh_samples(hf, tu.trimesh_warp_map(hf, HS, VS, threshold))
    pts = np.stack([rng.uniform(0.2, 1.0, 6000), rng.uniform(0.2, 0.9, 6000)], 1)   # border vertices may move inwards
    want = _brute_force(v, tris, pts)
    h64, n64 = pyoracle.terrain_query(_terrain(hf), packed, pts, f64=True)
    h32, n32 = pyoracle.terrain_query(_terrain(hf), packed, pts)
    ok = np.isfinite(want)
    assert ok.all()
    np.testing.assert_allclose(h64[ok], want[ok], atol=1e-6)       # hscale / vscale and the mesh vertices are float32
    np.testing.assert_allclose(h32[ok], want[ok], atol=5e-6)
    np.testing.assert_allclose(np.linalg.norm(n64, axis=1), 1.0, atol=1e-12)
    assert (n64[:, 2] > 0).all()


def test_steep_step_becomes_a_vertical_riser():
    hf = np.zeros((12, 8), np.int16)
    hf[5:, :] = 40                                          # 0.2 m step between rows 4 and 5
    packed = tu.pack_trimesh_samples(hf, tu.trimesh_warp_map(hf, HS, VS, 0.75))
    xs = np.array([0.30, 0.45, 0.49, 0.499, 0.501, 0.55, 0.80])
    pts = np.stack([xs, np.full_like(xs, 0.35)], 1)
    h, n = pyoracle.terrain_query(_terrain(hf), packed, pts, f64=True)
    np.testing.assert_allclose(h, [0, 0, 0, 0, 0.2, 0.2, 0.2], atol=1e-12)
    np.testing.assert_allclose(n, np.tile([0, 0, 1.0], (7, 1)), atol=1e-12)
    # the height field of the same samples has a 63-degree ramp there instead
    hh, _ = pyoracle.terrain_query(_terrain(hf, warped=0), hf, pts, f64=True)
    assert 0.05 < hh[1] < 0.15 and hh[0] == 0 and hh[-1] == pytest.approx(0.2)


def test_heightfield_query_equals_its_explicit_triangulation():
    """gym.add_heightfield's surface (isaac_gym.py:350-367): cells split along (i+1, j)-(i, j+1) -- the oracle's
    height-field query (heights and unit normals) against a brute-force evaluation of that explicit triangle mesh."""
    rng = np.random.default_rng(5)
    hf = (rng.integers(-6, 7, (15, 13)) * 9).astype(np.int16)
    rows, cols = hf.shape
    X, Y = np.meshgrid(np.arange(rows) * HS, np.arange(cols) * HS, indexing="ij")
    V = np.stack([X, Y, hf * VS], -1).reshape(-1, 3)
    idx = lambda i, j: i * cols + j
    tris = []
    for i in range(rows - 1):
        for j in range(cols - 1):
            tris.append((idx(i, j), idx(i + 1, j), idx(i, j + 1)))             # lower: u + v <= 1
            tris.append((idx(i + 1, j + 1), idx(i, j + 1), idx(i + 1, j)))     # upper
    pts = np.stack([rng.uniform(0.02, (rows - 1) * HS - 0.02, 5000), rng.uniform(0.02, (cols - 1) * HS - 0.02, 5000)], 1)
    want = _brute_force(V, np.array(tris), pts)
    h, n = pyoracle.terrain_query(_terrain(hf, warped=0), hf, pts, f64=True)
    np.testing.assert_allclose(h, want, atol=1e-6)          # (hscale / vscale are float32 in ShfTerrain)
    # normals: of the triangle under the point
    i = np.floor(pts[:, 0] / HS).astype(int); j = np.floor(pts[:, 1] / HS).astype(int)
    u, v = pts[:, 0] / HS - i, pts[:, 1] / HS - j
    lo = u + v <= 1
    z = hf.astype(float) * VS
    gx = np.where(lo, z[i + 1, j] - z[i, j], z[i + 1, j + 1] - z[i, j + 1]) / HS
    gy = np.where(lo, z[i, j + 1] - z[i, j], z[i + 1, j + 1] - z[i + 1, j]) / HS
    nn = np.stack([-gx, -gy, np.ones_like(gx)], 1); nn /= np.linalg.norm(nn, axis=1, keepdims=True)
    edge = np.abs(u + v - 1) < 1e-3                          # on the diagonal either triangle may answer
    np.testing.assert_allclose(n[~edge], nn[~edge], atol=2e-6)
