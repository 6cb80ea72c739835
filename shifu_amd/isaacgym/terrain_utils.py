"""Sub-terrain generators with the call signatures of `isaacgym.terrain_utils`
[EXT -- not part of the reference tree], as used by shifu/utils/terrain.py:76,
106-148.  Restated from the behaviour summarised in SURVEY.md appendix C: all
generators edit `terrain.height_field_raw` (int16, units of vertical_scale) in
place.  Host/NumPy, init-time only; bit-parity with Isaac Gym's generators is
neither possible nor required (the height samples are an *input* of the step
path), they only have to be deterministic under `np.random.seed`.
"""
import numpy as np


class SubTerrain:
    def __init__(self, terrain_name="terrain", width=256, length=256, vertical_scale=1.0, horizontal_scale=1.0):
        self.terrain_name = terrain_name
        self.vertical_scale = vertical_scale
        self.horizontal_scale = horizontal_scale
        self.width = width
        self.length = length
        self.height_field_raw = np.zeros((self.width, self.length), dtype=np.int16)


def pyramid_sloped_terrain(terrain, slope=1, platform_size=1.):
    """Tent h = max_h * xx * yy with a flat centre platform."""
    w, l = terrain.width, terrain.length
    cx, cy = w / 2.0, l / 2.0
    xx = ((cx - np.abs(cx - np.arange(w))) / cx).reshape(w, 1)
    yy = ((cy - np.abs(cy - np.arange(l))) / cy).reshape(1, l)
    max_height = int(slope * (terrain.horizontal_scale / terrain.vertical_scale) * (w / 2))
    terrain.height_field_raw += (max_height * xx * yy).astype(terrain.height_field_raw.dtype)
    half = int(platform_size / terrain.horizontal_scale / 2)
    x1, y1 = w // 2 - half, l // 2 - half
    corner = terrain.height_field_raw[x1, y1]
    terrain.height_field_raw = np.clip(terrain.height_field_raw, min(corner, 0), max(corner, 0))
    return terrain


def _bilinear_upsample(coarse, out_rows, out_cols):
    r = np.linspace(0, coarse.shape[0] - 1, out_rows)
    c = np.linspace(0, coarse.shape[1] - 1, out_cols)
    r0 = np.clip(np.floor(r).astype(int), 0, coarse.shape[0] - 2)
    c0 = np.clip(np.floor(c).astype(int), 0, coarse.shape[1] - 2)
    fr, fc = (r - r0)[:, None], (c - c0)[None, :]
    a = coarse[r0][:, c0]; b = coarse[r0 + 1][:, c0]; cc = coarse[r0][:, c0 + 1]; d = coarse[r0 + 1][:, c0 + 1]
    return a * (1 - fr) * (1 - fc) + b * fr * (1 - fc) + cc * (1 - fr) * fc + d * fr * fc


def random_uniform_terrain(terrain, min_height, max_height, step=1, downsampled_scale=None):
    """Coarse grid of heights drawn from {min..max step} upsampled linearly and ADDED."""
    if downsampled_scale is None:
        downsampled_scale = terrain.horizontal_scale
    lo, hi = int(min_height / terrain.vertical_scale), int(max_height / terrain.vertical_scale)
    st = max(int(step / terrain.vertical_scale), 1)
    heights_range = np.arange(lo, hi + st, st)
    rows = max(int(terrain.width * terrain.horizontal_scale / downsampled_scale), 2)
    cols = max(int(terrain.length * terrain.horizontal_scale / downsampled_scale), 2)
    coarse = np.random.choice(heights_range, (rows, cols)).astype(np.float64)
    up = np.rint(_bilinear_upsample(coarse, terrain.width, terrain.length))
    terrain.height_field_raw += up.astype(np.int16)
    return terrain


def pyramid_stairs_terrain(terrain, step_width, step_height, platform_size=1.):
    """Concentric square steps rising (or descending) towards the centre platform."""
    sw = int(step_width / terrain.horizontal_scale)
    sh = int(step_height / terrain.vertical_scale)
    plat = int(platform_size / terrain.horizontal_scale)
    height = 0
    x0, x1, y0, y1 = 0, terrain.width, 0, terrain.length
    while (x1 - x0) > plat and (y1 - y0) > plat:
        x0 += sw; x1 -= sw; y0 += sw; y1 -= sw
        height += sh
        terrain.height_field_raw[x0:x1, y0:y1] = height
    return terrain


def discrete_obstacles_terrain(terrain, max_height, min_size, max_size, num_rects, platform_size=1.):
    """Random rectangles of height in {-h, -h/2, h/2, h}; centre platform cleared."""
    mh = int(max_height / terrain.vertical_scale)
    mn, mx = int(min_size / terrain.horizontal_scale), int(max_size / terrain.horizontal_scale)
    plat = int(platform_size / terrain.horizontal_scale)
    rows, cols = terrain.height_field_raw.shape
    height_range = [-mh, -mh // 2, mh // 2, mh]
    size_range = range(mn, mx, 4)
    for _ in range(num_rects):
        w = np.random.choice(size_range)
        l = np.random.choice(size_range)
        i = np.random.choice(range(0, rows - w, 4))
        j = np.random.choice(range(0, cols - l, 4))
        terrain.height_field_raw[i:i + w, j:j + l] = np.random.choice(height_range)
    x1, x2 = (rows - plat) // 2, (rows + plat) // 2
    y1, y2 = (cols - plat) // 2, (cols + plat) // 2
    terrain.height_field_raw[x1:x2, y1:y2] = 0
    return terrain


def stepping_stones_terrain(terrain, stone_size, stone_distance, max_height, platform_size=1., depth=-10):
    """Square stones on a `depth` m deep floor (unreachable with the reference's
    5-entry terrain_proportions, terrain.py:146-152; provided for completeness)."""
    ss = max(int(stone_size / terrain.horizontal_scale), 1)
    sd = max(int(stone_distance / terrain.horizontal_scale), 1)
    mh = int(max_height / terrain.vertical_scale)
    plat = int(platform_size / terrain.horizontal_scale)
    height_range = np.arange(-mh - 1, mh, 1)
    terrain.height_field_raw[:, :] = int(depth / terrain.vertical_scale)
    rows, cols = terrain.height_field_raw.shape
    for i in range(0, rows, ss + sd):
        for j in range(0, cols, ss + sd):
            terrain.height_field_raw[i:i + ss, j:j + ss] = np.random.choice(height_range)
    x1, x2 = (rows - plat) // 2, (rows + plat) // 2
    y1, y2 = (cols - plat) // 2, (cols + plat) // 2
    terrain.height_field_raw[x1:x2, y1:y2] = 0
    return terrain


def vertex_shifts(height_field_raw, horizontal_scale, vertical_scale, slope_threshold):
    """(dx, dy) in cells, each in {-1, 0, 1}, that convert_heightfield_to_trimesh applies to every vertex: next to a
    step steeper than the threshold the vertex is pulled over its neighbour so that the riser becomes vertical."""
    hf = height_field_raw
    rows, cols = hf.shape
    thr = slope_threshold * horizontal_scale / vertical_scale
    move_x = np.zeros((rows, cols)); move_y = np.zeros((rows, cols)); move_c = np.zeros((rows, cols))
    move_x[:rows - 1, :] += (hf[1:, :] - hf[:rows - 1, :] > thr)
    move_x[1:, :] -= (hf[:rows - 1, :] - hf[1:, :] > thr)
    move_y[:, :cols - 1] += (hf[:, 1:] - hf[:, :cols - 1] > thr)
    move_y[:, 1:] -= (hf[:, :cols - 1] - hf[:, 1:] > thr)
    move_c[:rows - 1, :cols - 1] += (hf[1:, 1:] - hf[:rows - 1, :cols - 1] > thr)
    move_c[1:, 1:] -= (hf[:rows - 1, :cols - 1] - hf[1:, 1:] > thr)
    return move_x + move_c * (move_x == 0), move_y + move_c * (move_y == 0)


def trimesh_warp_map(height_field_raw, horizontal_scale, vertical_scale, slope_threshold=None):
    """The per-vertex byte the backend collides a trimesh terrain with (include/shifu_amd.h, ShfTerrain.warped):
    bits 0-1 dx+1, bits 2-3 dy+1, bits 4-7 = which neighbouring rows / columns of cells a query has to search
    (warp_map_from_shifts)."""
    hf = np.asarray(height_field_raw)
    rows, cols = hf.shape
    if slope_threshold is None:
        dx = dy = np.zeros((rows, cols))
    else:
        dx, dy = vertex_shifts(hf.astype(np.int64), horizontal_scale, vertical_scale, slope_threshold)
    return warp_map_from_shifts(dx, dy)


def warp_map_from_shifts(dx, dy):
    """Per-vertex bytes (see trimesh_warp_map) from explicit vertex shifts in cells: bits 0-1 dx+1, bits 2-3 dy+1, and
    for the cell whose lower corner the vertex is, which neighbouring rows / columns of cells a query inside that cell has
    to search besides the cell itself: bit 4 row i-1, bit 5 row i+1, bit 6 column j-1, bit 7 column j+1.  A neighbour row
    (column) is needed when a triangle of one of its three cells next to this one reaches into the cell's square -- its
    bounding box overlaps the open square, or touches the closed one with a vertex that was moved (a riser standing
    exactly on the cell boundary).  Unmoved neighbours only touch along the shared edge, where the surface is the cell's
    own."""
    dx, dy = np.asarray(dx).astype(np.int64), np.asarray(dy).astype(np.int64)
    rows, cols = dx.shape
    R, C = rows - 1, cols - 1
    X = np.arange(rows)[:, None] + dx
    Y = np.arange(cols)[None, :] + dy
    moved = (dx != 0) | (dy != 0)
    s00, s10 = (slice(0, R), slice(0, C)), (slice(1, R + 1), slice(0, C))
    s01, s11 = (slice(0, R), slice(1, C + 1)), (slice(1, R + 1), slice(1, C + 1))
    tris = []
    for vs in ((s00, s11, s01), (s00, s10, s11)):             # the two triangles of a cell
        xs, ys = np.stack([X[v] for v in vs]), np.stack([Y[v] for v in vs])
        tris.append((xs.min(0), xs.max(0), ys.min(0), ys.max(0), np.stack([moved[v] for v in vs]).any(0)))
    I, J = np.arange(R)[:, None], np.arange(C)[None, :]
    need = np.zeros((3, 3, R, C), bool)
    for a in (-1, 0, 1):
        for b in (-1, 0, 1):
            if a == 0 and b == 0:
                continue
            ci, cj = np.clip(I + a, 0, R - 1), np.clip(J + b, 0, C - 1)
            valid = (I + a >= 0) & (I + a < R) & (J + b >= 0) & (J + b < C)
            ov = np.zeros((R, C), bool)
            for x0, x1, y0, y1, mv in tris:
                x0, x1, y0, y1, mv = x0[ci, cj], x1[ci, cj], y0[ci, cj], y1[ci, cj], mv[ci, cj]
                ov |= (x0 < I + 1) & (x1 > I) & (y0 < J + 1) & (y1 > J)
                ov |= mv & (x0 <= I + 1) & (x1 >= I) & (y0 <= J + 1) & (y1 >= J)
            need[a + 1, b + 1] = valid & ov
    hint = np.zeros((rows, cols), np.int64)
    hint[:R, :C] = (need[0].any(0).astype(np.int64) << 4) | (need[2].any(0).astype(np.int64) << 5) | \
                   (need[:, 0].any(0).astype(np.int64) << 6) | (need[:, 2].any(0).astype(np.int64) << 7)
    w = (dx + 1) | ((dy + 1) << 2) | hint
    return w.astype(np.uint8)


def heightfield_from_trimesh(vertices, triangles, horizontal_scale=None, vertical_scale=None):
    """Inverse of convert_heightfield_to_trimesh: (int16 samples, horizontal_scale, vertical_scale, warp bytes) of a
    mesh that function made -- exact, because it keeps the vertices in grid order and only shifts some of them by one
    cell in x / y.  Anything else (another triangulation, off-grid vertices, heights that are not multiples of one
    vertical scale) raises: the backend collides grid terrains only.  The scales are taken from the caller when given
    (TriangleMeshParams carries none in the reference); otherwise hs is the median spacing of neighbouring vertices
    (a shift moves a few of them by a whole cell, never the median) and vs the largest step for which every height is
    an integer multiple -- the smallest gap between levels divided by 1..4, so levels such as {0, 2, 5} x vs resolve."""
    v = np.asarray(vertices, dtype=np.float64).reshape(-1, 3)
    t = np.asarray(triangles).reshape(-1, 3).astype(np.int64)
    n = v.shape[0]
    bad = NotImplementedError("add_triangle_mesh: only meshes made by convert_heightfield_to_trimesh (a grid of height "
                              "samples, cells split along (i,j)-(i+1,j+1)) can be collided by this backend")
    if t.shape[0] < 2 or t[0, 0] != 0 or t[0, 2] != 1:
        raise bad
    cols = int(t[0, 1]) - 1
    if cols < 2 or n % cols:
        raise bad
    rows = n // cols
    if t.shape[0] != 2 * (rows - 1) * (cols - 1):
        raise bad
    ind0 = (np.arange(rows - 1)[:, None] * cols + np.arange(cols - 1)[None, :]).reshape(-1)
    exp = np.empty((2 * ind0.size, 3), np.int64)
    exp[0::2] = np.stack([ind0, ind0 + cols + 1, ind0 + 1], 1)
    exp[1::2] = np.stack([ind0, ind0 + cols, ind0 + cols + 1], 1)
    if not np.array_equal(t, exp):
        raise bad
    x, y, z = v[:, 0].reshape(rows, cols), v[:, 1].reshape(rows, cols), v[:, 2].reshape(rows, cols)
    if horizontal_scale is not None:
        hs = float(horizontal_scale)
    else:
        hs = round(float(np.median(np.diff(x, axis=0))), 6)
        if not hs > 0 or abs(float(np.median(np.diff(y, axis=1))) - hs) > 1e-4 * hs:
            raise bad
    # grid origin: the un-shifted vertices sit on origin + i * hs; the median offset is theirs
    x0 = float(np.median(x - np.arange(rows)[:, None] * hs))
    y0 = float(np.median(y - np.arange(cols)[None, :] * hs))
    dx = (x - x0 - np.arange(rows)[:, None] * hs) / hs
    dy = (y - y0 - np.arange(cols)[None, :] * hs) / hs
    if np.abs(dx - np.rint(dx)).max() > 1e-3 or np.abs(dy - np.rint(dy)).max() > 1e-3 or \
            np.abs(np.rint(dx)).max() > 1 or np.abs(np.rint(dy)).max() > 1:
        raise bad
    if vertical_scale is not None:
        cands = [float(vertical_scale)]
    else:
        u = np.unique(np.abs(z))
        steps = np.diff(u)
        steps = steps[steps > 1e-7]
        gap = float(steps.min()) if steps.size else 0.005
        cands = [round(gap / d, 6) for d in (1, 2, 3, 4)]
    vs = None
    for c in cands:
        k = z / c
        if c > 0 and np.abs(k - np.rint(k)).max() <= 2e-2 and np.abs(k).max() <= 32767:
            vs = c
            break
    if vs is None:
        raise bad
    k = z / vs
    return (np.ascontiguousarray(np.rint(k).astype(np.int16)), hs, vs,
            warp_map_from_shifts(np.rint(dx).astype(np.int64), np.rint(dy).astype(np.int64)))


def pack_trimesh_samples(height_field_raw, warp):
    """int16 samples followed by the per-vertex bytes, as one flat int16 array (the SHF_T_HEIGHTS payload of a
    warped terrain)."""
    hf = np.ascontiguousarray(height_field_raw, np.int16).reshape(-1)
    wb = np.ascontiguousarray(warp, np.uint8).reshape(-1)
    if wb.size % 2:
        wb = np.concatenate([wb, np.zeros(1, np.uint8)])
    return np.concatenate([hf, wb.view(np.int16)])


def convert_heightfield_to_trimesh(height_field_raw, horizontal_scale, vertical_scale, slope_threshold=None):
    """(vertices (R*C,3) f32, triangles (2(R-1)(C-1),3) u32).  Where the slope between
    neighbouring samples exceeds the threshold the upper vertex is pulled over the
    lower one so steps become vertical walls."""
    hf = height_field_raw
    rows, cols = hf.shape
    y = np.linspace(0, (cols - 1) * horizontal_scale, cols)
    x = np.linspace(0, (rows - 1) * horizontal_scale, rows)
    yy, xx = np.meshgrid(y, x)
    if slope_threshold is not None:
        dx, dy = vertex_shifts(hf, horizontal_scale, vertical_scale, slope_threshold)
        xx = xx + dx * horizontal_scale
        yy = yy + dy * horizontal_scale
    vertices = np.zeros((rows * cols, 3), dtype=np.float32)
    vertices[:, 0] = xx.flatten()
    vertices[:, 1] = yy.flatten()
    vertices[:, 2] = hf.flatten() * vertical_scale
    triangles = -np.ones((2 * (rows - 1) * (cols - 1), 3), dtype=np.uint32)
    for i in range(rows - 1):
        ind0 = np.arange(0, cols - 1) + i * cols
        ind1, ind2, ind3 = ind0 + 1, ind0 + cols, ind0 + cols + 1
        s, e = 2 * i * (cols - 1), 2 * i * (cols - 1) + 2 * (cols - 1)
        triangles[s:e:2, 0] = ind0; triangles[s:e:2, 1] = ind3; triangles[s:e:2, 2] = ind1
        triangles[s + 1:e:2, 0] = ind0; triangles[s + 1:e:2, 1] = ind2; triangles[s + 1:e:2, 2] = ind3
    return vertices, triangles
