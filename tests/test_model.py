"""Model compiler robustness: every URDF the reference ships (asset/urdf/**, present only in the build container)
flattens into a ShfModel -- missing <inertial> children, capsules, fixed-joint chains, mesh-only links -- or is
refused with a capacity message; the two vendored physics-only URDFs are checked for their known shape."""
import glob
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shifu_amd import _abi
from shifu_amd.model import asset_path, compile_urdf

REF = "/root/reference/asset/urdf"


def test_vendored_robots():
    a1 = compile_urdf(asset_path("a1.urdf"), default_dof_drive_mode=_abi.DOF_MODE_EFFORT)
    assert (a1.blob.nb, a1.blob.nd, a1.blob.np, a1.blob.nlevels, a1.blob.nklevels) == (17, 12, 76, 3, 4)
    assert abs(a1.total_mass - 12.454) < 1e-3 and a1.body_names[0] == "base" and "FL_foot" in a1.body_names
    from shifu_amd.abb_task import abb_model
    abb = abb_model()
    assert abb.blob.nd == 6 and abb.blob.fixed_base == 1 and abb.blob.nsph == 2 and sum(x * x for x in abb.blob.sph_seg[0]) ** 0.5 > 0.15
    assert [abb.blob.sph_part[i] for i in range(2)] == [0, 1] and list(abb.blob.sph_seg[0]) == list(abb.blob.sph_seg[1])   # the rod capsule: two contact parts


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is only present in the build container")
def test_every_reference_urdf_compiles_or_is_refused_with_a_reason():
    paths = sorted(glob.glob(os.path.join(REF, "**", "*.urdf"), recursive=True))
    assert len(paths) > 20
    refused, massless = [], []
    for p in paths:
        try:
            cm = compile_urdf(p, meshes="auto")
        except AssertionError as e:
            assert "SHF_MAX" in str(e), (p, e)
            refused.append(os.path.basename(p))
            continue
        except ValueError as e:                # a massless subtree behind a moving joint / massless floating root
            assert "divide by zero" in str(e), (p, e)
            massless.append(os.path.basename(p))
            continue
        m = cm.blob
        assert 1 <= m.nb <= _abi.MAX_BODIES and 0 <= m.nd <= 32 and len(cm.body_names) == m.nb and len(cm.dof_names) == m.nd
        assert all(m.parent[b] < b for b in range(1, m.nb)), "bodies are numbered parents-first"
        assert sum(m.pt_count[b] for b in range(m.nb)) == m.np
    assert refused == []          # anymal (143 sample points) and sektion_cabinet_2 (164) fit SHF_MAX_POINTS = 176
    print("refused for missing inertia:", massless)


MASSLESS = """<robot name="m">
 <link name="base"><inertial><mass value="1"/><inertia ixx="0.01" ixy="0" ixz="0" iyy="0.01" iyz="0" izz="0.01"/></inertial></link>
 <link name="arm">{inertial}</link>
 <joint name="j" type="revolute"><parent link="base"/><child link="arm"/><axis xyz="0 1 0"/>
  <limit effort="10" lower="-1" upper="1" velocity="10"/></joint>
</robot>"""


def test_massless_moving_body_is_refused_or_steps_finite(tmp_path, oracle):
    """A link without <mass>/<inertia> behind a revolute joint makes the joint-space inertia D = S^T I S exactly zero:
    the compiler refuses it (message names the link), accepts it with armature > 0, and a model it accepts steps to
    finite state."""
    import numpy as np
    from tests.helpers import sim_params
    path = tmp_path / "m.urdf"
    path.write_text(MASSLESS.format(inertial=""))
    with pytest.raises(ValueError, match="arm"):
        compile_urdf(str(path), default_dof_drive_mode=_abi.DOF_MODE_EFFORT)
    path.write_text(MASSLESS.format(inertial='<inertial><mass value="0"/></inertial>'))
    with pytest.raises(ValueError, match="divide by zero"):
        compile_urdf(str(path), default_dof_drive_mode=_abi.DOF_MODE_EFFORT)
    for kw, text in (({"armature": 0.01}, ""),
                     ({}, '<inertial><origin xyz="0 0 -0.1"/><mass value="0.2"/><inertia ixx="1e-4" ixy="0" ixz="0" iyy="1e-4" iyz="0" izz="1e-4"/></inertial>')):
        path.write_text(MASSLESS.format(inertial=text))
        cm = compile_urdf(str(path), default_dof_drive_mode=_abi.DOF_MODE_EFFORT, **kw)
        dof = np.zeros((1, 2), np.float32)
        root = np.zeros((1, 13), np.float32); root[0, 2] = 1.0; root[0, 6] = 1.0
        oracle.step(cm.blob, sim_params(), 1, dof, root, nsteps=50, effort=np.full(1, 0.5, np.float32))
        assert np.isfinite(dof).all() and np.isfinite(root).all()
    # a massless floating root is refused as well
    path.write_text('<robot name="r"><link name="only"/></robot>')
    with pytest.raises(ValueError, match="floating root"):
        compile_urdf(str(path))
    assert compile_urdf(str(path), fix_base_link=True).blob.nb == 1


# ---- mesh colliders: convex hull of the STL / OBJ (the reference's ABB links collide as STL hulls) ----
def _write_box_stl(path, size, binary=True):
    import struct
    h = 0.5 * np.asarray(size, float)
    v = np.array([[sx, sy, sz] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)], float) * h
    quads = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
    tris = [(q[0], q[1], q[2]) for q in quads] + [(q[0], q[2], q[3]) for q in quads]
    if binary:
        with open(path, "wb") as f:
            f.write(b"box".ljust(80, b" ") + struct.pack("<I", len(tris)))
            for t in tris:
                f.write(struct.pack("<12fH", 0, 0, 0, *v[t[0]], *v[t[1]], *v[t[2]], 0))
    else:
        with open(path, "w") as f:
            f.write("solid box\n")
            for t in tris:
                f.write(" facet normal 0 0 0\n  outer loop\n" + "".join(f"   vertex {v[k][0]} {v[k][1]} {v[k][2]}\n" for k in t) + "  endloop\n endfacet\n")
            f.write("endsolid box\n")


MESH_URDF = """<robot name="m"><link name="base">{inertial}
 <collision><origin xyz="0.1 0 0" rpy="0 0 0.3"/><geometry>{geom}</geometry></collision></link></robot>"""


@pytest.mark.parametrize("binary", [True, False])
def test_mesh_collider_is_the_convex_hull_of_the_stl(tmp_path, binary):
    size = (0.4, 0.2, 0.1)
    _write_box_stl(tmp_path / "box.stl", size, binary)
    (tmp_path / "mesh.urdf").write_text(MESH_URDF.format(inertial="", geom='<mesh filename="box.stl"/>'))
    (tmp_path / "prim.urdf").write_text(MESH_URDF.format(inertial="", geom=f'<box size="{size[0]} {size[1]} {size[2]}"/>'))
    a, b = compile_urdf(str(tmp_path / "mesh.urdf")), compile_urdf(str(tmp_path / "prim.urdf"))
    assert a.blob.np == 8 and b.blob.np == 8
    pa = sorted(tuple(np.round(a.blob.pt_pos[i][:], 6)) for i in range(8))
    pb = sorted(tuple(np.round(b.blob.pt_pos[i][:], 6)) for i in range(8))
    assert pa == pb                                                     # the hull's vertices are the box's corners
    # no <inertial>: uniform-density mass properties of the hull = those of the box primitive
    assert abs(a.total_mass - 1000.0 * 0.4 * 0.2 * 0.1) < 1e-6 and abs(a.total_mass - b.total_mass) < 1e-6
    np.testing.assert_allclose(np.array(a.blob.inertia[0][:]), np.array(b.blob.inertia[0][:]), rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(np.array(a.blob.com[0][:]), np.array(b.blob.com[0][:]), atol=1e-7)


def test_mesh_collider_scale_obj_missing_file_and_sampling(tmp_path):
    from shifu_amd.model import HULL_POINTS, _convex_hull, _hull_inertial, _hull_sample
    # OBJ, scaled: a unit tetrahedron scaled by 2 -> volume 8 / 6
    (tmp_path / "tet.obj").write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nv 0 0 1\nv 0.2 0.2 0.2\nf 1 2 3\n")
    (tmp_path / "t.urdf").write_text(MESH_URDF.format(inertial="", geom='<mesh filename="tet.obj" scale="2 2 2"/>'))
    cm = compile_urdf(str(tmp_path / "t.urdf"))
    assert cm.blob.np == 4 and abs(cm.total_mass - 1000.0 * 8.0 / 6.0) < 1e-6          # the interior vertex is not on the hull
    # a missing file is refused unless asked otherwise
    (tmp_path / "gone.urdf").write_text(MESH_URDF.format(inertial='<inertial><mass value="1"/><inertia ixx="1" iyy="1" izz="1"/></inertial>',
                                                          geom='<mesh filename="package://pkg/meshes/none.stl"/>'))
    with pytest.raises(FileNotFoundError, match="not found"):
        compile_urdf(str(tmp_path / "gone.urdf"))
    assert compile_urdf(str(tmp_path / "gone.urdf"), meshes="auto").blob.np == 0
    assert compile_urdf(str(tmp_path / "gone.urdf"), meshes="drop").blob.np == 0
    # many-vertex hull: HULL_POINTS samples, the six axis extremes among them; mass properties of a sphere
    rng = np.random.default_rng(0)
    p = rng.normal(size=(4000, 3)); p /= np.linalg.norm(p, axis=1, keepdims=True)
    v, t = _convex_hull(0.3 * p)
    sm = _hull_sample(v, HULL_POINTS)
    assert len(sm) == HULL_POINTS
    for ax in range(3):
        assert sm[:, ax].max() == v[:, ax].max() and sm[:, ax].min() == v[:, ax].min()
    ine = _hull_inertial(v, t, 1000.0)
    m = 1000.0 * 4.0 / 3.0 * np.pi * 0.3 ** 3
    assert abs(ine.mass - m) < 0.01 * m and np.abs(ine.com).max() < 1e-3
    np.testing.assert_allclose(np.diag(ine.I), 0.4 * m * 0.09 * np.ones(3), rtol=0.02)


def test_mesh_box_rests_on_the_plane_like_the_primitive_box(tmp_path):
    """the hull collider in the physics: a mesh box dropped on the ground comes to rest where the <box> does"""
    from oracle import pyoracle as O
    from tests import helpers as H
    size = (0.3, 0.2, 0.1)
    _write_box_stl(tmp_path / "box.stl", size)
    geo = {"mesh": '<mesh filename="box.stl"/>', "prim": f'<box size="{size[0]} {size[1]} {size[2]}"/>'}
    z = {}
    for k, g in geo.items():
        (tmp_path / f"{k}.urdf").write_text(MESH_URDF.replace('xyz="0.1 0 0" rpy="0 0 0.3"', 'xyz="0 0 0" rpy="0 0 0"').format(inertial="", geom=g))
        cm = compile_urdf(str(tmp_path / f"{k}.urdf"))
        root = np.zeros((1, 13)); root[0, 2] = 0.08; root[0, 6] = 1.0
        dof = np.zeros((0, 2))
        O.step(cm.blob, H.sim_params(), 1, dof, root, nsteps=400, f64=True)
        z[k] = root[0, 2]
        assert abs(root[0, 9]) < 1e-4                                    # at rest
    assert 0.045 < z["mesh"] < 0.0501 and abs(z["mesh"] - z["prim"]) < 1e-7


MASSLESS_MID = """<robot name="r">
 <link name="base"><inertial><mass value="2"/><inertia ixx="0.01" ixy="0" ixz="0" iyy="0.01" iyz="0" izz="0.01"/></inertial>
  <collision><geometry><box size="0.3 0.2 0.1"/></geometry></collision></link>
 <link name="mid"><collision><origin xyz="0 0 -0.1"/><geometry><sphere radius="0.03"/></geometry></collision></link>
 <link name="tip"><inertial><origin xyz="0 0 -0.1"/><mass value="0.3"/><inertia ixx="1e-3" ixy="0" ixz="0" iyy="1e-3" iyz="0" izz="1e-3"/></inertial>
  <collision><origin xyz="0 0 -0.2"/><geometry><sphere radius="0.03"/></geometry></collision></link>
 <joint name="j1" type="revolute"><parent link="base"/><child link="mid"/><origin xyz="0.2 0 0"/><axis xyz="0 1 0"/><limit lower="-1" upper="1" effort="10" velocity="10"/></joint>
 <joint name="j2" type="revolute"><parent link="mid"/><child link="tip"/><origin xyz="0 0 -0.2"/><axis xyz="0 1 0"/><limit lower="-1" upper="1" effort="10" velocity="10"/></joint>
</robot>"""


def test_self_collision_pairs_skip_massless_moving_bodies(tmp_path):
    """the self-collision pair law divides by each body's own mass (1 + m_own / m_other).  A massless
    intermediate link with a collision shape compiles (its subtree has inertia) -- its pairs must not be listed."""
    path = tmp_path / "mid.urdf"
    path.write_text(MASSLESS_MID)
    cm = compile_urdf(str(path), default_dof_drive_mode=_abi.DOF_MODE_EFFORT, self_collision=True,
                      inertia_from_geometry=False) if "inertia_from_geometry" in compile_urdf.__code__.co_varnames else None
    if cm is None:
        cm = compile_urdf(str(path), default_dof_drive_mode=_abi.DOF_MODE_EFFORT, self_collision=True)
    m = cm.blob
    mid = cm.body_names.index("mid")
    if m.mass[mid] > 0.0:
        pytest.skip("this compiler derives the link's mass from its collision geometry: nothing massless to guard")
    for k in range(m.npair):
        for c in (m.pair_a[k], m.pair_b[k]):
            assert m.mass[m.dyn[m.cap_body[c]]] > 0.0, "pair with a massless moving body"


def test_link_contacts_on_a_robot_with_many_box_volumes_loads_with_a_warning(tmp_path):
    """gym.load_asset asks for link contacts on every URDF (the reference's collision filter 0, units.py:68): a robot with
    more box collision volumes than SHF_MAX_ABOX has to load -- family (B), box-actor corners against its volumes, is
    dropped with a warning; its own vertices and rounded shapes keep colliding."""
    n = _abi.MAX_ABOX + 3
    links = ["<link name='base'><inertial><mass value='1'/><inertia ixx='0.01' iyy='0.01' izz='0.01' ixy='0' ixz='0' iyz='0'/></inertial>"
             + "".join(f"<collision><origin xyz='{0.05 * k} 0 0'/><geometry><box size='0.04 0.04 0.04'/></geometry></collision>" for k in range(n))
             + "</link>"]
    p = tmp_path / "many_boxes.urdf"
    p.write_text("<robot name='r'>" + "".join(links) + "</robot>")
    with pytest.warns(UserWarning, match="SHF_MAX_ABOX"):
        cm = compile_urdf(str(p), link_contacts=True)
    assert cm.blob.link_collide == 1 and cm.blob.nabox == 0 and cm.blob.np == 8 * n
    few = compile_urdf(str(p), link_contacts=False)
    assert few.blob.link_collide == 0 and few.blob.nabox == 0


def test_model_bounds_enclose_every_link_contact_shape():
    """shf_model_bounds (the kernels' link-contact broad phase, ShfModel.bbox): every sample point and rounded shape, grown by
    its radius, and every vertex of every box volume lies inside its body's box; bodies without shapes are marked."""
    import ctypes as C
    from shifu_amd import _lib
    from shifu_amd.abb_task import abb_model
    for cm in (abb_model(link_contacts=True), abb_model(link_contacts=False),
               compile_urdf(asset_path("a1.urdf"), link_contacts=True)):
        m = cm.blob
        m.bounds_ok = 0
        assert _lib.lib().shf_model_bounds(C.byref(m)) == 0
        assert m.bounds_ok == 0x42534831
        balls = [(m.pt_body[i], np.array(m.pt_pos[i][:]), m.pt_radius[i]) for i in range(m.np)]
        for i in range(m.nsph):
            for e in (0.0, 1.0):
                balls.append((m.sph_body[i], np.array(m.sph_pos[i][:]) + e * np.array(m.sph_seg[i][:]), m.sph_radius[i]))
        for j in range(m.nabox):
            Rj = np.array(m.abox_rot[j][:]).reshape(3, 3)
            for c in range(8):
                s = np.array([1.0 if c & 4 else -1.0, 1.0 if c & 2 else -1.0, 1.0 if c & 1 else -1.0])
                balls.append((m.abox_body[j], np.array(m.abox_pos[j][:]) + Rj @ (s * np.array(m.abox_half[j][:])), 0.0))
        have = set()
        for b, c, r in balls:
            bb = np.array(m.bbox[b][:])
            have.add(b)
            assert bb[3] >= 0.0
            assert np.all(np.abs(c - bb[:3]) + r <= bb[3:] + 1e-7), (b, c, r, bb)
        for b in range(_abi.MAX_BODIES):
            if b not in have:
                assert m.bbox[b][3] < 0.0
        # tight: the box is the bounding box of the shapes, not an arbitrary superset
        for b in have:
            pts = np.array([c for bb_, c, r in balls if bb_ == b])
            rad = np.array([r for bb_, c, r in balls if bb_ == b])
            ext = 0.5 * ((pts + rad[:, None]).max(0) - (pts - rad[:, None]).min(0))
            assert np.all(np.array(m.bbox[b][3:]) <= ext * 1.001 + 1e-5)
