"""URDF -> flattened articulation (ShfModel).

Replaces what `gym.load_asset(sim, root, file, AssetOptions)` does inside Isaac
Gym for the reference (shifu/units/units.py:73-89, asset options
shifu/configs/asset_config.py:32-46): fixed-joint collapse honouring
`dont_collapse`, inertial merge, body/DOF ordering, limits, collision
primitives.  [EXT] Isaac Gym orders bodies depth-first with siblings sorted by
link name (A1 -> FL, FR, RL, RR); the order chosen here
is recorded in `CompiledModel.body_names / dof_names` and user code indexes by
name through `rigid_body_dict` exactly as in the reference (robot.py:198).
"""
from __future__ import annotations

import dataclasses
import os
import xml.etree.ElementTree as ET
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _abi

HULL_POINTS = 8          # contact sample points per convex-hull (mesh) collider, like a box's eight corners

ASSET_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets")


def _rpy(r: float, p: float, y: float) -> np.ndarray:
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    return np.array([[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
                     [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
                     [-sp, cp * sr, cp * cr]])


def _vec(s: Optional[str], n=3) -> np.ndarray:
    return np.zeros(n) if s is None else np.array([float(t) for t in s.split()])


def _origin(el) -> Tuple[np.ndarray, np.ndarray]:
    o = None if el is None else el.find("origin")
    if o is None:
        return np.zeros(3), np.eye(3)
    return _vec(o.get("xyz")), _rpy(*_vec(o.get("rpy")))


@dataclasses.dataclass
class _Inertial:
    mass: float = 0.0
    com: np.ndarray = dataclasses.field(default_factory=lambda: np.zeros(3))
    I: np.ndarray = dataclasses.field(default_factory=lambda: np.zeros((3, 3)))  # about com, link axes

    def transformed(self, p, R) -> "_Inertial":
        return _Inertial(self.mass, p + R @ self.com, R @ self.I @ R.T)

    def merged(self, o: "_Inertial") -> "_Inertial":
        m = self.mass + o.mass
        if m <= 0.0:
            return _Inertial()
        c = (self.mass * self.com + o.mass * o.com) / m
        I = np.zeros((3, 3))
        for t in (self, o):
            d = t.com - c
            I += t.I + t.mass * (d @ d * np.eye(3) - np.outer(d, d))
        return _Inertial(m, c, I)


@dataclasses.dataclass
class _Shape:
    kind: str                    # "box" | "sphere" | "capsule" | "hull" (convex hull of a <mesh>)
    size: np.ndarray
    pos: np.ndarray
    rot: np.ndarray
    verts: Optional[np.ndarray] = None    # hull: vertices (n,3) in the shape frame, scale applied
    faces: Optional[np.ndarray] = None    # hull: triangles (m,3) into verts, outward orientation


@dataclasses.dataclass
class _Link:
    name: str
    inertial: _Inertial
    shapes: List[_Shape]


@dataclasses.dataclass
class _Joint:
    name: str
    type: str
    parent: str
    child: str
    pos: np.ndarray
    rot: np.ndarray
    axis: np.ndarray
    lower: float
    upper: float
    effort: float
    velocity: float
    damping: float
    dont_collapse: bool


@dataclasses.dataclass
class CompiledModel:
    blob: _abi.ShfModel
    body_names: List[str]
    dof_names: List[str]
    total_mass: float
    hulls: Optional["_abi.ShfHullSet"] = None     # the articulation's convex hulls (blob.nhull of them), for Sim.set_hulls

    @property
    def rigid_body_dict(self) -> Dict[str, int]:
        return {n: i for i, n in enumerate(self.body_names)}

    @property
    def num_bodies(self) -> int:
        return self.blob.nb

    @property
    def num_dof(self) -> int:
        return self.blob.nd

    def dof_properties(self) -> np.ndarray:
        """NumPy structured array with the fields shifu touches (robot.py:35-42)."""
        dt = np.dtype([("hasLimits", "?"), ("lower", "f4"), ("upper", "f4"), ("driveMode", "i4"),
                       ("velocity", "f4"), ("effort", "f4"), ("stiffness", "f4"), ("damping", "f4"),
                       ("friction", "f4"), ("armature", "f4")])
        out = np.zeros(self.blob.nd, dtype=dt)
        for i in range(self.blob.nd):
            out[i] = (True, self.blob.lower[i], self.blob.upper[i], self.blob.drive_mode[i],
                      self.blob.vel_limit[i], self.blob.effort[i], self.blob.kp[i], self.blob.kd[i], 0.0,
                      self.blob.armature[i])
        return out


def _resolve_mesh(urdf_path: str, filename: str) -> Optional[str]:
    """<mesh filename>: a path relative to the URDF, or package://<pkg>/<rest> looked up as <ancestor dir>/<pkg>/<rest>
    and <ancestor dir>/<rest> for the URDF's ancestors (how the reference's asset tree lays its packages out)."""
    if not filename:
        return None
    base = os.path.dirname(os.path.abspath(urdf_path))
    if filename.startswith("package://"):
        rest = filename[len("package://"):]
        tail = rest.split("/", 1)[1] if "/" in rest else rest
        d = base
        for _ in range(6):
            for cand in (os.path.join(d, rest), os.path.join(d, tail)):
                if os.path.isfile(cand):
                    return cand
            d = os.path.dirname(d)
        return None
    if filename.startswith("file://"):
        filename = filename[len("file://"):]
    cand = filename if os.path.isabs(filename) else os.path.join(base, filename)
    return cand if os.path.isfile(cand) else None


def _load_stl(path: str) -> np.ndarray:
    """Vertices (n,3) of a collision mesh file: STL (binary or ASCII) or Wavefront OBJ (only the convex hull is used)."""
    raw = open(path, "rb").read()
    if path.lower().endswith(".obj"):
        pts = [list(map(float, ln.split()[1:4])) for ln in raw.decode("utf-8", "ignore").splitlines() if ln.startswith("v ")]
        if not pts:
            raise ValueError(f"{path}: no vertices")
        return np.asarray(pts, dtype=float)
    if len(raw) >= 84:
        ntri = int(np.frombuffer(raw, dtype="<u4", count=1, offset=80)[0])
        if 84 + 50 * ntri == len(raw):
            rec = np.frombuffer(raw, dtype=np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("a", "<u2")]), count=ntri, offset=84)
            return rec["v"].reshape(-1, 3).astype(float)
    pts = [list(map(float, ln.split()[1:4])) for ln in raw.decode("ascii", "ignore").splitlines() if ln.strip().startswith("vertex")]
    if not pts:
        raise ValueError(f"{path}: neither a binary nor an ASCII STL")
    return np.asarray(pts, dtype=float)


def _convex_hull(pts: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Convex hull (vertices, outward triangles) -- what PhysX collides a dynamic triangle mesh as (convex decomposition
    off, asset_config.py:32-46 vhacd_enabled = False)."""
    from scipy.spatial import ConvexHull
    pts = np.unique(np.round(pts, 9), axis=0)
    if len(pts) < 4:
        raise ValueError("collision mesh with fewer than four distinct vertices")
    h = ConvexHull(pts)
    idx = np.unique(h.simplices)
    remap = -np.ones(len(pts), dtype=int)
    remap[idx] = np.arange(len(idx))
    v, t = pts[idx], remap[h.simplices]
    c = v.mean(0)
    for k in range(len(t)):                          # orient every triangle outward
        a, b, d = v[t[k]]
        if np.dot(np.cross(b - a, d - a), a - c) < 0:
            t[k] = t[k][::-1]
    return v, t


def _hull_sample(v: np.ndarray, n: int) -> np.ndarray:
    """At most n hull vertices as contact sample points: the extreme vertices along +-x, +-y, +-z first (a plane contact
    of an axis-aligned resting pose is then exact), then farthest-point sampling."""
    if len(v) <= n:
        return v.copy()
    pick = []
    for ax in range(3):
        for sgn in (-1, 1):
            k = int(np.argmax(sgn * v[:, ax]))
            if k not in pick:
                pick.append(k)
    pick = pick[:n]
    d = np.min(np.linalg.norm(v[:, None, :] - v[pick][None, :, :], axis=2), axis=1)
    while len(pick) < n:
        k = int(np.argmax(d))
        pick.append(k)
        d = np.minimum(d, np.linalg.norm(v - v[k], axis=1))
    return v[pick]


def _hull_inertial(v: np.ndarray, t: np.ndarray, density: float) -> "_Inertial":
    """Uniform-density mass properties of a closed triangulated convex surface (signed tetrahedra from the origin)."""
    vol, first, second = 0.0, np.zeros(3), np.zeros((3, 3))
    canon = (np.ones((3, 3)) + np.eye(3)) / 120.0             # integral of x_i x_j over the unit tetrahedron
    for tri in t:
        A = v[tri].T                                          # columns a, b, c
        det = float(np.linalg.det(A))
        vol += det / 6.0
        first += det / 24.0 * A.sum(1)
        second += det * (A @ canon @ A.T)
    if vol <= 0:
        raise ValueError("collision mesh hull has no volume")
    com = first / vol
    C = second - vol * np.outer(com, com)                     # covariance about the centre of mass
    I = density * (np.trace(C) * np.eye(3) - C)
    return _Inertial(density * vol, com, I)


def _sphere_directions(n: int) -> np.ndarray:
    """A fixed set of unit directions: the 26 axis / face-diagonal / cube-diagonal directions first (a box-like hull keeps its
    eight corners and a resting face stays flat), then a Fibonacci spiral."""
    d = [np.array(v, float) for v in
         [(1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)] +
         [(sx, sy, sz) for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)] +
         [(sx, sy, 0) for sx in (-1, 1) for sy in (-1, 1)] + [(sx, 0, sz) for sx in (-1, 1) for sz in (-1, 1)] +
         [(0, sy, sz) for sy in (-1, 1) for sz in (-1, 1)]]
    k = 0
    while len(d) < n:
        z = 1.0 - (2.0 * k + 1.0) / max(n, 1)
        r, ph = np.sqrt(max(0.0, 1.0 - z * z)), k * np.pi * (3.0 - np.sqrt(5.0))
        d.append(np.array([r * np.cos(ph), r * np.sin(ph), z]))
        k += 1
    return np.array([v / np.linalg.norm(v) for v in d[:n]])


def reduce_hull(pts: np.ndarray, max_verts: int = _abi.HULL_MAX_VERTS) -> dict:
    """A mesh collider as the convex polytope the narrow phase collides (include/shifu_amd.h: ShfHull): the convex hull of
    `pts`, reduced to at most `max_verts` of its vertices -- the support points along a fixed set of directions, so the result
    is inscribed in the true hull -- with coplanar triangles merged into polygonal faces; fewer vertices are taken until the
    face / edge / loop limits hold too ([EXT] PhysX cooks a convex mesh of at most 64 vertices and polygons the same way).
    Returns dict(verts (nv,3), planes (nf,4), loops [list of vertex indices, counter-clockwise from outside], edges (ne,4)
    = v0, v1, face, face, centroid)."""
    from scipy.spatial import ConvexHull
    v0, _ = _convex_hull(np.asarray(pts, float))
    n = min(max_verts, len(v0))
    while n >= 4:
        if len(v0) > n:
            idx = []
            for d in _sphere_directions(4 * n):
                k = int(np.argmax(v0 @ d))
                if k not in idx:
                    idx.append(k)
                if len(idx) >= n:
                    break
            v = v0[idx]
        else:
            v = v0.copy()
        v = np.asarray(v, np.float32).astype(float)                 # the planes are those of the stored (float32) vertices
        v, tri = _convex_hull(v)
        c = v.mean(0)
        # triangle normals; merge coplanar neighbours
        nrm = np.cross(v[tri[:, 1]] - v[tri[:, 0]], v[tri[:, 2]] - v[tri[:, 0]])
        nrm /= np.linalg.norm(nrm, axis=1)[:, None]
        off = np.einsum("ij,ij->i", nrm, v[tri[:, 0]])
        group = -np.ones(len(tri), int)
        edge_tris = {}
        for t, (a, b, d) in enumerate(tri):
            for e in ((a, b), (b, d), (d, a)):
                edge_tris.setdefault(tuple(sorted(e)), []).append(t)
        ng = 0
        for t in range(len(tri)):
            if group[t] >= 0:
                continue
            group[t] = ng
            stack = [t]
            while stack:
                u = stack.pop()
                for e in ((tri[u][0], tri[u][1]), (tri[u][1], tri[u][2]), (tri[u][2], tri[u][0])):
                    for w in edge_tris[tuple(sorted(e))]:
                        if group[w] < 0 and np.dot(nrm[w], nrm[t]) > 1.0 - 1e-7 and abs(off[w] - off[t]) < 1e-6:
                            group[w] = ng
                            stack.append(w)
            ng += 1
        loops, planes, ok = [], [], True
        for g in range(ng):
            ts = np.nonzero(group == g)[0]
            directed = {}
            for t in ts:
                a, b, d = tri[t]
                for e in ((a, b), (b, d), (d, a)):
                    if len([w for w in edge_tris[tuple(sorted(e))] if group[w] == g]) == 1:      # a boundary edge of the face
                        directed[int(e[0])] = int(e[1])
            start = min(directed)
            loop, cur = [start], directed[start]
            while cur != start and len(loop) <= len(directed):
                loop.append(cur)
                cur = directed[cur]
            if len(loop) != len(directed) or len(loop) > _abi.HULL_MAX_FACE_VERTS:
                ok = False
                break
            nn = nrm[ts].mean(0)
            nn /= np.linalg.norm(nn)
            planes.append(list(nn) + [float(np.max(v[loop] @ nn))])
            loops.append(loop)
        edges = []
        if ok:
            face_of = {}
            for f, loop in enumerate(loops):
                for i in range(len(loop)):
                    face_of.setdefault(tuple(sorted((loop[i], loop[(i + 1) % len(loop)]))), []).append(f)
            ok = all(len(fs) == 2 for fs in face_of.values())
            edges = [[a, b, fs[0], fs[1]] for (a, b), fs in sorted(face_of.items())] if ok else []
        if ok and len(loops) <= _abi.HULL_MAX_FACES and len(edges) <= _abi.HULL_MAX_EDGES and sum(map(len, loops)) <= _abi.HULL_MAX_LOOP:
            return dict(verts=v, planes=np.array(planes), loops=loops, edges=np.array(edges, int), centroid=c)
        n -= 2
    raise ValueError("collision mesh cannot be reduced to a convex polytope within the SHF_HULL_* limits")


def hull_record(h: dict, body: int, p: np.ndarray, R: np.ndarray) -> "_abi.ShfHull":
    """ShfHull of reduce_hull()'s polytope fixed to reported body `body` at (p, R) in the body frame."""
    o = _abi.ShfHull()
    o.body, o.nv, o.nf, o.ne = body, len(h["verts"]), len(h["loops"]), len(h["edges"])
    cw = p + R @ h["centroid"]
    for k in range(3):
        o.centroid[k] = cw[k]
    for i, q in enumerate(h["verts"]):
        w = p + R @ q
        for k in range(3):
            o.vert[i][k] = w[k]
    k0 = 0
    for f, (pl, loop) in enumerate(zip(h["planes"], h["loops"])):
        nw = R @ pl[:3]
        for k in range(3):
            o.plane[f][k] = nw[k]
        o.plane[f][3] = pl[3] + float(nw @ p)
        o.face_start[f], o.face_count[f] = k0, len(loop)
        for i in loop:
            o.face_loop[k0] = i
            k0 += 1
    for e, rec in enumerate(h["edges"]):
        for k in range(4):
            o.edge[e][k] = int(rec[k])
    return o


def parse_urdf(path: str, meshes: str = "error", **_unused) -> Tuple[Dict[str, _Link], List[_Joint]]:
    """meshes: "error" -- <mesh> collisions become convex hulls, a missing file is refused; "auto" -- hulls where the file
    is found, silently none where it is not; "drop" -- mesh collisions are ignored."""
    root = ET.parse(path).getroot()
    links: Dict[str, _Link] = {}
    for l in root.findall("link"):
        ine = l.find("inertial")
        inertial = _Inertial()
        if ine is not None:
            p, R = _origin(ine)
            I = ine.find("inertia")          # URDFs in the wild omit <inertia> or <mass>: zero, like the importer
            g = lambda k: float(I.get(k, 0.0)) if I is not None else 0.0
            It = np.array([[g("ixx"), g("ixy"), g("ixz")], [g("ixy"), g("iyy"), g("iyz")],
                           [g("ixz"), g("iyz"), g("izz")]])
            mass_el = ine.find("mass")
            inertial = _Inertial(float(mass_el.get("value")) if mass_el is not None else 0.0, p, R @ It @ R.T)
        shapes = []
        for c in l.findall("collision"):
            p, R = _origin(c)
            g = c.find("geometry")
            if g is None:
                continue
            if g.find("box") is not None:
                shapes.append(_Shape("box", _vec(g.find("box").get("size")), p, R))
            elif g.find("sphere") is not None:
                shapes.append(_Shape("sphere", np.array([float(g.find("sphere").get("radius"))]), p, R))
            elif g.find("cylinder") is not None or g.find("capsule") is not None:
                e = g.find("cylinder") if g.find("cylinder") is not None else g.find("capsule")
                shapes.append(_Shape("capsule", np.array([float(e.get("radius")), float(e.get("length"))]), p, R))
            elif g.find("mesh") is not None:
                e = g.find("mesh")
                f = _resolve_mesh(path, e.get("filename", ""))
                if f is None or meshes == "drop":
                    if f is None and meshes == "error":
                        raise FileNotFoundError(f"link '{l.get('name')}': collision mesh '{e.get('filename')}' not found "
                                                f"next to {path} (pass meshes='drop' to ignore mesh colliders)")
                    continue
                sc = _vec(e.get("scale")) if e.get("scale") else np.ones(3)
                v, t = _convex_hull(_load_stl(f) * sc)
                shapes.append(_Shape("hull", sc, p, R, v, t))
        links[l.get("name")] = _Link(l.get("name"), inertial, shapes)
    joints = []
    for j in root.findall("joint"):
        p, R = _origin(j)
        a = j.find("axis")
        lim = j.find("limit")
        dyn = j.find("dynamics")
        jt = j.get("type")
        lo = float(lim.get("lower", 0.0)) if lim is not None else 0.0
        up = float(lim.get("upper", 0.0)) if lim is not None else 0.0
        if jt == "continuous":
            lo, up = -1e9, 1e9
        joints.append(_Joint(j.get("name"), jt, j.find("parent").get("link"), j.find("child").get("link"), p, R,
                             _vec(a.get("xyz")) if a is not None else np.array([1.0, 0, 0]), lo, up,
                             float(lim.get("effort", 0.0)) if lim is not None else 0.0,
                             float(lim.get("velocity", 0.0)) if lim is not None else 0.0,
                             float(dyn.get("damping", 0.0)) if dyn is not None else 0.0,
                             j.get("dont_collapse", "false").lower() == "true"))
    return links, joints


def _shape_points(shape: _Shape) -> List[Tuple[np.ndarray, float]]:
    """Contact sample points of a primitive in its link frame: box -> 8 corners
    (radius 0), sphere -> centre, capsule -> 2 end spheres + midpoint."""
    pts = []
    if shape.kind == "box":
        h = 0.5 * shape.size
        for sx in (-1, 1):
            for sy in (-1, 1):
                for sz in (-1, 1):
                    pts.append((shape.pos + shape.rot @ (h * np.array([sx, sy, sz])), 0.0))
    elif shape.kind == "sphere":
        pts.append((shape.pos.copy(), float(shape.size[0])))
    elif shape.kind == "capsule":
        r, L = float(shape.size[0]), float(shape.size[1])
        for z in (-0.5 * L, 0.0, 0.5 * L):
            pts.append((shape.pos + shape.rot @ np.array([0, 0, z]), r))
    elif shape.kind == "hull":
        for q in _hull_sample(shape.verts, HULL_POINTS):
            pts.append((shape.pos + shape.rot @ q, 0.0))
    return pts


def _shape_inertial(shape: _Shape, density: float) -> _Inertial:
    """Uniform-density inertial of a collision primitive, in the link frame."""
    if shape.kind == "hull":
        i = _hull_inertial(shape.verts, shape.faces, density)
        return _Inertial(i.mass, shape.pos + shape.rot @ i.com, shape.rot @ i.I @ shape.rot.T)
    if shape.kind == "box":
        x, y, z = shape.size
        m = density * x * y * z
        I = np.diag([m * (y * y + z * z) / 12.0, m * (x * x + z * z) / 12.0, m * (x * x + y * y) / 12.0])
    elif shape.kind == "sphere":
        r = float(shape.size[0])
        m = density * 4.0 / 3.0 * np.pi * r ** 3
        I = np.eye(3) * 0.4 * m * r * r
    else:                                            # capsule along z: cylinder + two half spheres
        r, L = float(shape.size[0]), float(shape.size[1])
        mc, ms = density * np.pi * r * r * L, density * 4.0 / 3.0 * np.pi * r ** 3
        m = mc + ms
        ixx = mc * (3 * r * r + L * L) / 12.0 + ms * (0.4 * r * r + 0.25 * L * L + 0.375 * r * L)
        I = np.diag([ixx, ixx, 0.5 * mc * r * r + 0.4 * ms * r * r])
    return _Inertial(m, shape.pos.copy(), shape.rot @ I @ shape.rot.T)


def _fill_missing_inertials(links: Dict[str, _Link], density: float):
    """[EXT] Isaac Gym derives the mass properties of a link that has none from its collision geometry and
    AssetOptions.density (default 1000 kg/m^3; asset_config.py:47-51 leaves it alone).  Same here for the collision
    primitives this backend knows; a link given a mass but no inertia tensor keeps its mass and gets the shape of the
    geometry's tensor.  Mesh-only links stay as parsed (and are refused by _check_inertia where that matters)."""
    for l in links.values():
        if not l.shapes:
            continue
        has_I = np.abs(l.inertial.I).max() > 0.0
        if l.inertial.mass > 0.0 and has_I:
            continue
        geo = _Inertial()
        for sh in l.shapes:
            geo = geo.merged(_shape_inertial(sh, density))
        if geo.mass <= 0.0:
            continue
        if l.inertial.mass > 0.0:
            l.inertial = _Inertial(l.inertial.mass, l.inertial.com, geo.I * (l.inertial.mass / geo.mass))
        else:
            l.inertial = geo


def _check_inertia(names, parent, jtype, dyn, tpos, trot, axis, merged, armature, fixed_base):
    """Refuse a model whose forward dynamics would divide by zero.  The articulated-body recursion inverts, for every
    joint, D = S^T I_A S + armature, where I_A contains the inertia of the whole subtree below the joint, and for a
    floating base the 6x6 articulated inertia of the root.  URDFs may omit <mass>/<inertia> (parsed as zero, like
    Isaac Gym's importer -- which then derives them from the collision geometry and a density; this compiler does
    not), so a massless subtree behind a moving joint, or a massless floating root, is rejected here instead of
    producing inf/NaN for the whole env on the GPU (and in the oracle alike)."""
    nb = len(names)
    Rw, pw = [np.eye(3)] * nb, [np.zeros(3)] * nb
    for b in range(1, nb):                       # zero-configuration world frames
        pb = parent[b]
        Rw[b], pw[b] = Rw[pb] @ trot[b], pw[pb] + Rw[pb] @ tpos[b]
    sub = [_Inertial() for _ in range(nb)]       # composite inertia of the subtree, world frame, about its own com
    for b in range(nb):
        if dyn[b] == b:
            sub[b] = merged[b].transformed(pw[b], Rw[b])
    for b in range(nb - 1, 0, -1):
        if dyn[b] == b:
            a = dyn[parent[b]]
            sub[a] = sub[a].merged(sub[b])
    for b in range(1, nb):
        if dyn[b] != b:
            continue
        if jtype[b] == _abi.JOINT_PRISMATIC:
            d = sub[b].mass
        else:
            aw = Rw[b] @ axis[b]
            c = sub[b].com - pw[b]
            d = aw @ (sub[b].I + sub[b].mass * (c @ c * np.eye(3) - np.outer(c, c))) @ aw
        if not (d + armature > 1e-12):
            raise ValueError(f"link '{names[b]}': the subtree behind its joint has no mass/inertia about the joint axis "
                             f"(URDF without <mass>/<inertia>?) and armature is 0 -- forward dynamics would divide by zero; "
                             f"give the link an <inertial> or pass armature > 0")
    if not fixed_base:
        if not (sub[0].mass > 1e-12) or np.linalg.eigvalsh(sub[0].I).min() <= 1e-14:
            raise ValueError(f"floating root '{names[0]}': total mass {sub[0].mass:g} / rotational inertia is not positive "
                             f"definite -- the 6x6 root solve would divide by zero; give the links <inertial> data")


def _shape_capsules(shape: _Shape) -> List[Tuple[np.ndarray, np.ndarray, float]]:
    """Self-collision capsules (end points a, b in the link frame, radius) of a collision primitive: a sphere is a
    capsule of zero length; a capsule is itself; a box L >= W >= H gets capsules of radius H/2 along its longest axis
    -- one (radius (W + H)/4) when the cross-section is nearly square, else as many side by side as cover W."""
    if shape.kind == "sphere":
        return [(shape.pos.copy(), shape.pos.copy(), float(shape.size[0]))]
    if shape.kind == "hull":
        # one capsule along the hull's longest principal axis: radius = RMS distance of the vertices from that axis
        c = shape.verts.mean(0)
        _, _, vt = np.linalg.svd(shape.verts - c, full_matrices=False)
        ax = vt[0]
        proj = (shape.verts - c) @ ax
        perp = np.linalg.norm((shape.verts - c) - np.outer(proj, ax), axis=1)
        r = float(np.sqrt(np.mean(perp ** 2)))
        lo, hi = min(float(proj.min()) + r, 0.0), max(float(proj.max()) - r, 0.0)
        return [(shape.pos + shape.rot @ (c + lo * ax), shape.pos + shape.rot @ (c + hi * ax), r)]
    if shape.kind == "capsule":
        r, L = float(shape.size[0]), float(shape.size[1])
        z = shape.rot @ np.array([0.0, 0.0, 0.5 * L])
        return [(shape.pos - z, shape.pos + z, r)]
    order = np.argsort(-shape.size)                     # longest, middle, shortest axis
    Ld, Wd, Hd = (float(shape.size[k]) for k in order)
    ex, ew = np.eye(3)[order[0]], np.eye(3)[order[1]]
    if Wd - Hd < 0.5 * Hd:
        r, offs = 0.25 * (Wd + Hd), [0.0]
    else:
        r = 0.5 * Hd
        n = int(np.ceil((Wd - Hd) / Hd)) + 1
        offs = list(np.linspace(-0.5 * (Wd - Hd), 0.5 * (Wd - Hd), n))
    half = max(0.5 * Ld - r, 0.0)
    return [(shape.pos + shape.rot @ (ew * o - ex * half), shape.pos + shape.rot @ (ew * o + ex * half), r) for o in offs]


def compile_urdf(path: str, *, fix_base_link: bool = False, disable_gravity: bool = False,
                 collapse_fixed_joints: bool = True, default_dof_drive_mode: int = _abi.DOF_MODE_NONE,
                 armature: float = 0.0, honour_dont_collapse: bool = True,
                 extra_spheres: Sequence[tuple] = (), density: float = 1000.0,
                 self_collision: bool = False, meshes: str = "error", link_contacts: bool = False,
                 extra_boxes: Sequence[tuple] = (), extra_hulls: Sequence[tuple] = (), hull_contacts: bool = False) -> CompiledModel:
    """Compile `path` with the AssetOptions the reference passes (asset_config.py:32-46).

    meshes: <mesh> collision geometry becomes the convex hull of the STL file ("error": a missing file is refused;
    "auto": a missing file is skipped; "drop": meshes are ignored, what strip_urdf.py did to the vendored assets); a hull
    contributes HULL_POINTS of its vertices
    as ground-contact sample points, its uniform-density mass properties when the link has no <inertial>, and one
    self-collision capsule along its longest principal axis.
    extra_spheres: rounded collision shapes tested against box actors -- the substitute for mesh colliders (ABB rod):
    (link name, xyz in that link's frame, radius) spheres, or (link name, xyz_a, xyz_b, radius) capsules.
    extra_boxes: (link name, xyz, rpy, size) box collision shapes added to links before compilation -- the box-shaped
    stand-ins for mesh colliders whose files do not ship (shifu_amd/assets/abb_link_boxes.json: the bounding boxes of the
    reference's ABB link hulls).
    extra_hulls: (link name, vertices (n, 3) in that link's frame) convex mesh colliders added to links before compilation --
    the reduced hulls of mesh files that do not ship (shifu_amd/assets/abb_link_hulls.json, tools/make_link_hulls.py).
    hull_contacts: mesh colliders meet the box actors as convex polytopes (include/shifu_amd.h: ShfHull; needs link_contacts):
    each hull shape becomes a record of CompiledModel.hulls (ShfModel.nhull) for the convex narrow phase -- separating-axis
    test and clipped face manifold -- instead of eight sample vertices and a bounding-box volume; its sample points stay for
    ground contact but are left out of the link contacts' vertex family.
    link_contacts: ShfModel.link_collide -- every collision shape of the articulation also meets the box actors of its env
    (SURVEY 8f f3): sample points against boxes, box corners against the articulation's box volumes (abox_*: the <box>
    primitives and, for hulls, their bounding boxes in the shape frame)."""
    links, joints = parse_urdf(path, meshes=meshes)
    for (lname, verts) in extra_hulls:
        if lname not in links:
            raise ValueError(f"extra_hulls: no link '{lname}'")
        hv, ht = _convex_hull(np.asarray(verts, float))
        links[lname].shapes.append(_Shape("hull", np.ones(3), np.zeros(3), np.eye(3), hv, ht))
    if hull_contacts and not link_contacts:
        raise ValueError("hull_contacts needs link_contacts=True")
    hull_recs: List = []          # (body, reduced polytope, p, R)
    for (lname, xyz, rpy, size) in extra_boxes:
        if lname not in links:
            raise ValueError(f"extra_boxes: no link '{lname}'")
        links[lname].shapes.append(_Shape("box", np.asarray(size, float), np.asarray(xyz, float), _rpy(*rpy)))
    _fill_missing_inertials(links, density)
    children: Dict[str, List[_Joint]] = {n: [] for n in links}
    is_child = set()
    for j in joints:
        children[j.parent].append(j)
        is_child.add(j.child)
    roots = [n for n in links if n not in is_child]
    assert len(roots) == 1, f"URDF must have exactly one root link, got {roots}"

    # reported bodies, depth first, siblings by child link name
    names: List[str] = []
    parent: List[int] = []
    jtype: List[int] = []
    tpos: List[np.ndarray] = []
    trot: List[np.ndarray] = []
    axis: List[np.ndarray] = []
    jref: List[Optional[_Joint]] = []
    inert: List[_Inertial] = []  # own (collapsed-in) inertia, body frame
    points: List[Tuple[int, np.ndarray, float]] = []
    capsules: List[Tuple[int, np.ndarray, np.ndarray, float]] = []
    aboxes: List[Tuple[int, np.ndarray, np.ndarray, np.ndarray]] = []   # (body, centre, rot, half extents) in the body frame
    link_frame: Dict[str, Tuple[int, np.ndarray, np.ndarray]] = {}  # urdf link -> (body, p, R) in body frame

    def absorb(body: int, link: _Link, p: np.ndarray, R: np.ndarray):
        inert[body] = inert[body].merged(link.inertial.transformed(p, R))
        link_frame[link.name] = (body, p.copy(), R.copy())
        for s in link.shapes:
            if s.kind == "hull" and hull_contacts:
                # the convex narrow phase takes this shape against the box actors (no sample points: they would meet the boxes a
                # second time as family A; a fixed-base arm has no ground to touch)
                hull_recs.append((body, reduce_hull(s.verts), p + R @ s.pos, R @ s.rot))
                for ca, cb, rad in _shape_capsules(s):
                    capsules.append((body, p + R @ ca, p + R @ cb, rad))
                continue
            for q, rad in _shape_points(s):
                points.append((body, p + R @ q, rad))
            for ca, cb, rad in _shape_capsules(s):
                capsules.append((body, p + R @ ca, p + R @ cb, rad))
            if s.kind == "box":
                aboxes.append((body, p + R @ s.pos, R @ s.rot, 0.5 * np.asarray(s.size, float)))
            elif s.kind == "hull":
                lo, hi = s.verts.min(0), s.verts.max(0)
                aboxes.append((body, p + R @ (s.pos + s.rot @ (0.5 * (lo + hi))), R @ s.rot, 0.5 * (hi - lo)))

    def visit(link_name: str, body: int, p: np.ndarray, R: np.ndarray):
        """link_name's frame sits at (p, R) inside reported body `body`."""
        for j in sorted(children[link_name], key=lambda jj: jj.child):
            cp, cR = p + R @ j.pos, R @ j.rot
            if j.type == "fixed" and collapse_fixed_joints and not (j.dont_collapse and honour_dont_collapse):
                absorb(body, links[j.child], cp, cR)
                visit(j.child, body, cp, cR)
                continue
            names.append(j.child)
            parent.append(body)
            if j.type in ("revolute", "continuous"):
                jtype.append(_abi.JOINT_REVOLUTE)
            elif j.type == "prismatic":
                jtype.append(_abi.JOINT_PRISMATIC)
            elif j.type == "fixed":
                jtype.append(_abi.JOINT_WELD)
            else:
                raise NotImplementedError(f"joint type {j.type}")
            tpos.append(cp)
            trot.append(cR)
            a = j.axis / max(np.linalg.norm(j.axis), 1e-12)
            axis.append(a)
            jref.append(j)
            inert.append(_Inertial())
            me = len(names) - 1
            absorb(me, links[j.child], np.zeros(3), np.eye(3))
            visit(j.child, me, np.zeros(3), np.eye(3))

    names.append(roots[0]); parent.append(-1); jtype.append(_abi.JOINT_ROOT)
    tpos.append(np.zeros(3)); trot.append(np.eye(3)); axis.append(np.array([0.0, 0, 1])); jref.append(None)
    inert.append(_Inertial())
    absorb(0, links[roots[0]], np.zeros(3), np.eye(3))
    visit(roots[0], 0, np.zeros(3), np.eye(3))

    nb = len(names)
    assert nb <= _abi.MAX_BODIES, f"{nb} bodies > SHF_MAX_BODIES"

    # moving-body bookkeeping; welded bodies hand their inertia to dyn[b]
    dyn = list(range(nb))
    dyn_T: List[Tuple[np.ndarray, np.ndarray]] = [(np.zeros(3), np.eye(3)) for _ in range(nb)]
    level = [0] * nb
    for b in range(1, nb):
        if jtype[b] == _abi.JOINT_WELD:
            pb = parent[b]
            pp, pR = dyn_T[pb]
            dyn[b] = dyn[pb]
            dyn_T[b] = (pp + pR @ tpos[b], pR @ trot[b])
            level[b] = level[pb]
        else:
            level[b] = level[dyn[parent[b]]] + 1
    merged = [dataclasses.replace(i) for i in inert]
    for b in range(1, nb):
        if dyn[b] != b:
            p, R = dyn_T[b]
            merged[dyn[b]] = merged[dyn[b]].merged(inert[b].transformed(p, R))
            merged[b] = _Inertial()

    _check_inertia(names, parent, jtype, dyn, tpos, trot, axis, merged, armature, bool(fix_base_link))

    m = _abi.ShfModel()
    m.nb = nb
    m.fixed_base = int(fix_base_link)
    m.gravity_on = int(not disable_gravity)
    m.nlevels = max(level)
    klevel = [0] * nb
    for b in range(1, nb):
        klevel[b] = klevel[parent[b]] + 1
    m.nklevels = max(klevel)
    dof_names: List[str] = []
    for b in range(nb):
        m.parent[b] = parent[b]
        m.jtype[b] = jtype[b]
        m.level[b] = level[b]
        m.klevel[b] = klevel[b]
        m.dyn[b] = dyn[b]
        m.dof[b] = -1
        for k in range(3):
            m.tpos[b][k] = tpos[b][k]
            m.axis[b][k] = axis[b][k]
            m.com[b][k] = merged[b].com[k]
        for k in range(9):
            m.trot[b][k] = trot[b].reshape(-1)[k]
        m.mass[b] = merged[b].mass
        I = merged[b].I
        for k, (r, c) in enumerate(((0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2))):
            m.inertia[b][k] = I[r, c]
        if jtype[b] in (_abi.JOINT_REVOLUTE, _abi.JOINT_PRISMATIC):
            d = len(dof_names)
            assert d < _abi.MAX_DOFS
            j = jref[b]
            m.dof[b] = d
            m.dof_body[d] = b
            m.lower[d], m.upper[d] = j.lower, j.upper
            m.vel_limit[d] = j.velocity if j.velocity > 0 else 1e9
            m.effort[d] = j.effort
            m.damping[d] = j.damping
            m.armature[d] = armature
            m.drive_mode[d] = default_dof_drive_mode
            dof_names.append(j.name)
    m.nd = len(dof_names)

    # children of moving bodies (the moving parent of b is dyn[parent[b]])
    kids: List[List[int]] = [[] for _ in range(nb)]
    for b in range(1, nb):
        if dyn[b] == b:
            kids[dyn[parent[b]]].append(b)
    k = 0
    for b in range(nb):
        m.child_start[b] = k
        m.child_count[b] = len(kids[b])
        for c in kids[b]:
            m.child_list[k] = c
            k += 1

    # contact points grouped by moving body, distal groups first: the kernels evaluate the points in rounds of
    # one per lane, and the links that usually touch the ground (feet, shanks) then share the early rounds, so
    # the later rounds skip the contact response wave-wide.  Per-body order is unchanged (same arithmetic).
    points.sort(key=lambda t: (-m.level[dyn[t[0]]], dyn[t[0]], t[0]))
    assert len(points) <= _abi.MAX_POINTS, f"{len(points)} contact points > SHF_MAX_POINTS"
    m.np = len(points)
    for b in range(nb):
        m.pt_start[b] = 0
        m.pt_count[b] = 0
    for i, (b, p, rad) in enumerate(points):
        m.pt_body[i] = b
        m.pt_radius[i] = rad
        for kk in range(3):
            m.pt_pos[i][kk] = p[kk]
        d = dyn[b]
        if m.pt_count[d] == 0:
            m.pt_start[d] = i
        m.pt_count[d] += 1

    # evaluation slots (ShfModel.pt_eval / pt_slot / neval): the points of a moving body in consecutive slots, in point order,
    # never across a multiple of 32; bodies whose rest pose reaches lowest first (they are the ones that usually touch the
    # ground: the later rounds can then skip the contact response); first-fit into the 32-slot blocks
    R0 = [np.eye(3) for _ in range(nb)]
    p0 = [np.zeros(3) for _ in range(nb)]
    for b in range(1, nb):
        pb = parent[b]
        R0[b] = R0[pb] @ np.array([[m.trot[b][3 * r + c] for c in range(3)] for r in range(3)], dtype=float)
        p0[b] = p0[pb] + R0[pb] @ np.array([m.tpos[b][kk] for kk in range(3)], dtype=float)
    zs = [float((p0[bb] + R0[bb] @ np.asarray(pp, dtype=float))[2] - rr) for (bb, pp, rr) in points]
    groups = [(int(m.pt_start[d]), int(m.pt_count[d])) for d in range(nb) if m.dyn[d] == d and m.pt_count[d] > 0]
    groups.sort(key=lambda g_: (round(min(zs[g_[0]:g_[0] + g_[1]]), 4), g_[0]))
    for s_ in range(_abi.MAX_POINTS):
        m.pt_eval[s_] = -1
    blocks = []                                     # free slots left in each 32-slot block
    packed = all(c <= 32 for _, c in groups)
    if packed:
        for i0, c in groups:
            k = next((k for k, free in enumerate(blocks) if free >= c), None)
            if k is None:
                blocks.append(32)
                k = len(blocks) - 1
            s0 = 32 * k + (32 - blocks[k])
            blocks[k] -= c
            for j in range(c):
                m.pt_eval[s0 + j] = i0 + j
                m.pt_slot[i0 + j] = s0 + j
        neval = max((s_ + 1 for s_ in range(32 * len(blocks)) if m.pt_eval[s_] >= 0), default=0)
        packed = neval <= _abi.MAX_POINTS
    if not packed:                                  # a body with more than 32 points, or no room for the padding: the identity
        neval = len(points)
        for i in range(len(points)):
            m.pt_eval[i] = i
            m.pt_slot[i] = i
    m.neval = neval

    # rounded shapes vs box actors: a sphere is one record, a capsule two consecutive ones (ShfModel.sph_part: the closest
    # point / first end of a line contact, and the second end)
    recs = []
    for rec in extra_spheres:
        lname, xyz, rad = rec[0], rec[1], rec[-1]
        body, p, R = link_frame[lname]
        q = p + R @ np.asarray(xyz, dtype=float)
        seg = R @ (np.asarray(rec[2], dtype=float) - np.asarray(xyz, dtype=float)) if len(rec) == 4 else np.zeros(3)
        for part in ((0, 1) if np.any(seg != 0.0) else (0,)):
            recs.append((body, q, seg, rad, part))
    assert len(recs) <= _abi.MAX_SPHERES, f"{len(recs)} rounded-shape records > SHF_MAX_SPHERES"
    m.nsph = len(recs)
    for i, (body, q, seg, rad, part) in enumerate(recs):
        m.sph_body[i] = body
        m.sph_radius[i] = rad
        m.sph_part[i] = part
        for kk in range(3):
            m.sph_pos[i][kk] = q[kk]
            m.sph_seg[i][kk] = seg[kk]

    # link contacts: the articulation's box volumes (vertex-in-box against the corners of box actors)
    m.link_collide = int(bool(link_contacts))
    if len(aboxes) > _abi.MAX_ABOX:
        if link_contacts:
            # not a reason to refuse the asset (gym.load_asset asks for link contacts on every URDF, and a scene without box
            # actors never reads these records): the vertex-in-volume family (B) is dropped, the articulation's own sample
            # points and rounded shapes still meet the box actors (families A and C)
            import warnings
            warnings.warn(f"{os.path.basename(str(path))}: {len(aboxes)} box / hull collision volumes > SHF_MAX_ABOX = "
                          f"{_abi.MAX_ABOX}; box-actor corners are not tested against this articulation's volumes "
                          f"(its own vertices and rounded shapes still collide with box actors)", stacklevel=2)
        aboxes = []
    m.nabox = len(aboxes)
    for j, (b, c, Rb, hh) in enumerate(aboxes):
        m.abox_body[j] = b
        for kk in range(3):
            m.abox_pos[j][kk] = c[kk]
            m.abox_half[j][kk] = hh[kk]
        for kk in range(9):
            m.abox_rot[j][kk] = Rb[kk // 3, kk % 3]

    # self-collision: capsules of every shape, and the pairs to test -- all but capsules of one rigid body (same moving
    # body: welded links included) and of moving bodies joined by a joint ([EXT] PhysX filters those the same way)
    m.self_collide = int(bool(self_collision))
    if len(capsules) > _abi.MAX_CAPSULES:
        if self_collision:
            raise AssertionError(f"{len(capsules)} self-collision capsules > SHF_MAX_CAPSULES")
        capsules = []
    m.ncap = len(capsules)
    for i, (b, ca, cb, rad) in enumerate(capsules):
        m.cap_body[i] = b
        m.cap_radius[i] = rad
        for kk in range(3):
            m.cap_a[i][kk], m.cap_b[i][kk] = ca[kk], cb[kk]
    pairs = []
    for i in range(m.ncap):
        for j in range(i + 1, m.ncap):
            da, db = dyn[capsules[i][0]], dyn[capsules[j][0]]
            if da == db:
                continue
            # the pair law scales each side's implicit part by 1 + m_own / m_other (csrc/shf_boxes.h self_scales): a moving
            # body without mass of its own (legal when its subtree has inertia, _check_inertia) would turn that into
            # inf / NaN for the whole env on first touch -- such pairs are not tested
            if merged[da].mass <= 0.0 or merged[db].mass <= 0.0:
                continue
            if (db > 0 and dyn[parent[db]] == da) or (da > 0 and dyn[parent[da]] == db):
                continue
            pairs.append((i, j))
    # pairs that are close in the rest pose (q = 0) first: the kernels test the pairs in rounds of one per lane and skip a
    # round's closest-point search wave-wide when every pair in it is far apart (csrc/shf_boxes.h self_contacts_eval) --
    # with the neighbours up front, the later rounds of a walking robot are all far.  (The order is part of the model: active
    # pairs are folded in pair order by the oracle and the kernels alike.)
    def _rest_gap(ij):
        (ba, a0, a1, ra), (bb, b0, b1, rb) = capsules[ij[0]], capsules[ij[1]]
        ca = p0[ba] + R0[ba] @ (0.5 * (np.asarray(a0, float) + np.asarray(a1, float)))
        cb = p0[bb] + R0[bb] @ (0.5 * (np.asarray(b0, float) + np.asarray(b1, float)))
        reach = 0.5 * (np.linalg.norm(np.asarray(a1, float) - np.asarray(a0, float)) + np.linalg.norm(np.asarray(b1, float) - np.asarray(b0, float))) + ra + rb
        return round(float(np.linalg.norm(ca - cb) - reach), 4)
    pairs.sort(key=lambda ij: (_rest_gap(ij), ij))
    if len(pairs) > _abi.MAX_PAIRS:
        if self_collision:
            raise AssertionError(f"{len(pairs)} self-collision pairs > SHF_MAX_PAIRS")
        pairs = []
    m.npair = len(pairs)
    for k, (i, j) in enumerate(pairs):
        m.pair_a[k], m.pair_b[k] = i, j

    hulls = None
    m.nhull = 0
    if hull_recs:
        if len(hull_recs) > _abi.MAX_HULLS:
            raise ValueError(f"{len(hull_recs)} convex hulls > SHF_MAX_HULLS = {_abi.MAX_HULLS}")
        hulls = _abi.ShfHullSet()
        hulls.nhull = m.nhull = len(hull_recs)
        for j, (b, h, hp, hR) in enumerate(sorted(hull_recs, key=lambda t: t[0])):
            hulls.hull[j] = hull_record(h, b, hp, hR)
    return CompiledModel(m, names, dof_names, float(sum(t.mass for t in merged)), hulls)


def asset_path(name: str) -> str:
    return os.path.join(ASSET_DIR, name)
