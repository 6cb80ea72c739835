"""Tiny single-purpose models for the contact known-answer tests (tests/test_contact_kats.py and its GPU twin)."""
import os
import tempfile

import numpy as np

from shifu_amd import _abi
from shifu_amd.model import compile_urdf

BOX_URDF = """<robot name="block"><link name="block">
 <inertial><mass value="{m}"/><inertia ixx="{ixx}" ixy="0" ixz="0" iyy="{iyy}" iyz="0" izz="{izz}"/></inertial>
 <collision><geometry><box size="{x} {y} {z}"/></geometry></collision></link></robot>"""
SPHERE_URDF = """<robot name="ball"><link name="ball">
 <inertial><mass value="{m}"/><inertia ixx="{i}" ixy="0" ixz="0" iyy="{i}" iyz="0" izz="{i}"/></inertial>
 <collision><geometry><sphere radius="{r}"/></geometry></collision></link></robot>"""
LIMIT_URDF = """<robot name="lim"><link name="world_link"/>
 <link name="bar"><inertial><origin xyz="0 0 -0.2"/><mass value="1.0"/>
  <inertia ixx="0.01" ixy="0" ixz="0" iyy="0.01" iyz="0" izz="0.01"/></inertial></link>
 <joint name="hinge" type="revolute"><parent link="world_link"/><child link="bar"/><axis xyz="0 1 0"/>
  <limit effort="{effort}" lower="{lo}" upper="{up}" velocity="100"/></joint></robot>"""


def _compile(text, **kw):
    with tempfile.NamedTemporaryFile("w", suffix=".urdf", delete=False) as f:
        f.write(text)
        path = f.name
    try:
        return compile_urdf(path, **kw)
    finally:
        os.unlink(path)


def block_model(x=0.2, y=0.2, z=0.1, m=2.0):
    return _compile(BOX_URDF.format(m=m, x=x, y=y, z=z, ixx=m * (y * y + z * z) / 12, iyy=m * (x * x + z * z) / 12,
                                    izz=m * (x * x + y * y) / 12))


def ball_model(r=0.05, m=1.0):
    return _compile(SPHERE_URDF.format(m=m, r=r, i=0.4 * m * r * r))


def limit_model(lo=-0.5, up=0.5, effort=30.0):
    return _compile(LIMIT_URDF.format(lo=lo, up=up, effort=effort), fix_base_link=True, disable_gravity=True,
                    default_dof_drive_mode=_abi.DOF_MODE_EFFORT)


def root_row(pos, quat=(0, 0, 0, 1), lin=(0, 0, 0), ang=(0, 0, 0), dtype=np.float32):
    r = np.zeros((1, 13), dtype)
    r[0, :3], r[0, 3:7], r[0, 7:10], r[0, 10:13] = pos, quat, lin, ang
    return r


G = 9.81
# contact parameters of shifu_amd.backend.default_sim_params (what every env in this repo runs with)
K_N, D_N, V_EPS, DT = 5e4, 300.0, 0.002, 0.005


PUSHER_URDF = """<robot name="pusher"><link name="rail"/>
 <link name="pusher"><inertial><mass value="5.0"/><inertia ixx="0.05" ixy="0" ixz="0" iyy="0.05" iyz="0" izz="0.05"/></inertial></link>
 <joint name="slide" type="prismatic"><parent link="rail"/><child link="pusher"/><axis xyz="1 0 0"/>
  <limit effort="200" lower="-1" upper="1" velocity="10"/></joint></robot>"""


def pusher_model(yaw=0.0, z=0.05, half=0.1, r=0.02, kd=2000.0):
    """A rail-mounted slider carrying one horizontal capsule (axis along y, turned by `yaw` about z) at height z: the
    capsule-vs-box slot in its simplest setting.  Velocity drive on the slide."""
    c, s = np.cos(yaw), np.sin(yaw)
    a, b = (half * s, -half * c, z), (-half * s, half * c, z)
    cm = _compile(PUSHER_URDF, fix_base_link=True, disable_gravity=True, default_dof_drive_mode=_abi.DOF_MODE_VEL,
                  extra_spheres=[("pusher", a, b, r)])
    cm.blob.kd[0] = kd
    return cm


BOX_PUSHER_URDF = """<robot name="boxpusher"><link name="rail"/>
 <link name="ram"><inertial><mass value="5.0"/><inertia ixx="0.05" ixy="0" ixz="0" iyy="0.05" iyz="0" izz="0.05"/></inertial>
  {shapes}</link>
 <joint name="slide" type="prismatic"><parent link="rail"/><child link="ram"/><axis xyz="{axis}"/>
  <limit effort="200" lower="-1" upper="1" velocity="10"/></joint></robot>"""


def box_pusher_model(size=(0.06, 0.06, 0.06), centre=(0.0, 0.0, 0.05), axis="1 0 0", kd=2000.0, extra_shapes="", capsule=None,
                     shape_rpy=(0.0, 0.0, 0.0)):
    """A rail-mounted ram carrying box collision shape(s) (link contacts on: ShfModel.link_collide) -- the box-vs-box
    cases of SURVEY 8f f3 in their simplest setting.  Velocity drive on the slide."""
    shapes = ('<collision><origin xyz="%g %g %g" rpy="%.17g %.17g %.17g"/><geometry><box size="%g %g %g"/></geometry></collision>'
              % (tuple(centre) + tuple(shape_rpy) + tuple(size))) if size is not None else ""
    cm = _compile(BOX_PUSHER_URDF.format(shapes=shapes + extra_shapes, axis=axis), fix_base_link=True, disable_gravity=True,
                  default_dof_drive_mode=_abi.DOF_MODE_VEL, link_contacts=True,
                  extra_spheres=[("ram",) + tuple(capsule)] if capsule else ())
    cm.blob.kd[0] = kd
    return cm


def hull_pusher_model(verts, axis="0 0 -1", kd=2000.0):
    """The rail-mounted ram carrying ONE convex mesh collider (vertices in the ram's frame): the convex narrow phase (hulls against
    box actors: ShfModel.nhull, include/shifu_amd.h ShfHull) in its simplest setting.  Velocity drive on the slide."""
    cm = _compile(BOX_PUSHER_URDF.format(shapes="", axis=axis), fix_base_link=True, disable_gravity=True,
                  default_dof_drive_mode=_abi.DOF_MODE_VEL, link_contacts=True, extra_hulls=[("ram", np.asarray(verts, float))],
                  hull_contacts=True)
    cm.blob.kd[0] = kd
    return cm


def prism_verts(a=0.1, b=0.06, top=0.5, h=0.08, z0=0.4):
    """A frustum: bottom rectangle 2a x 2b at z0, top rectangle scaled by `top` at z0 + h (eight vertices, six faces)."""
    return [[sx * a, sy * b, z0] for sx in (-1, 1) for sy in (-1, 1)] + [[sx * a * top, sy * b * top, z0 + h] for sx in (-1, 1) for sy in (-1, 1)]
