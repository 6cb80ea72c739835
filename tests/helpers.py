"""Shared test helpers: default sim params, an independent Newton-Euler inverse
dynamics in NumPy float64 (classical vectors, no spatial algebra) used to pin
the oracle's ABA, and small model builders."""
import os
import tempfile

import numpy as np

from shifu_amd import _abi
from shifu_amd.model import asset_path, compile_urdf


def sim_params(dt=0.005, gravity=(0.0, 0.0, -9.81), **kw):
    p = _abi.ShfSimParams()
    p.dt = dt
    p.gravity[:] = gravity
    p.contact_k = kw.get("contact_k", 5e4)
    p.contact_d = kw.get("contact_d", 300.0)
    p.friction_vel = kw.get("friction_vel", 0.002)
    p.limit_k = kw.get("limit_k", 2000.0)
    p.limit_d = kw.get("limit_d", 20.0)
    p.angular_damping = kw.get("angular_damping", 0.0)
    p.max_ang_vel = kw.get("max_ang_vel", 64.0)
    p.max_depen_vel = kw.get("max_depen_vel", 1.0)
    p.contact_offset = kw.get("contact_offset", 0.01)
    # contact solver (ABI v12): "compliant" = rounds 1-4's law (what the known answers of tests/test_contact_kats.py are about);
    # "pgs" = the velocity-level solve with the reference's PhysX settings (env_config.py:50-58)
    if kw.get("solver", "compliant") in ("pgs", "tgs"):
        p.solver = _abi.SOLVER_TGS if kw.get("solver") == "tgs" else _abi.SOLVER_PGS
        p.pos_iters, p.vel_iters = kw.get("pos_iters", 8), kw.get("vel_iters", 1)
        p.max_contacts = kw.get("max_contacts", 8)
        p.rest_offset = kw.get("rest_offset", 0.0)
        p.bounce_threshold = kw.get("bounce_threshold", 0.5)
        p.restitution = kw.get("restitution", 0.0)
        p.erp = kw.get("erp", 0.2)
    return p


def a1_model(**kw):
    kw.setdefault("default_dof_drive_mode", _abi.DOF_MODE_EFFORT)
    return compile_urdf(asset_path("a1.urdf"), **kw)


def quat_to_mat(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def rodrigues(a, q):
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(q) * K + (1 - np.cos(q)) * (K @ K)


def sym(i6):
    xx, xy, xz, yy, yz, zz = i6
    return np.array([[xx, xy, xz], [xy, yy, yz], [xz, yz, zz]])


def forward_kinematics(m, q, root_pos, root_quat):
    """World pose of every reported body: (R list, p list, world joint axes)."""
    R = [None] * m.nb
    p = [None] * m.nb
    aw = [None] * m.nb
    for b in range(m.nb):
        if m.jtype[b] == _abi.JOINT_ROOT:
            R[b], p[b] = quat_to_mat(root_quat), np.array(root_pos, float)
            continue
        par = m.parent[b]
        Rj = R[par] @ np.array(m.trot[b]).reshape(3, 3)
        p[b] = p[par] + R[par] @ np.array(m.tpos[b])
        ax = np.array(m.axis[b])
        aw[b] = Rj @ ax
        if m.jtype[b] == _abi.JOINT_REVOLUTE:
            R[b] = Rj @ rodrigues(ax, q[m.dof[b]])
        elif m.jtype[b] == _abi.JOINT_PRISMATIC:
            R[b] = Rj
            p[b] = p[b] + aw[b] * q[m.dof[b]]
        else:
            R[b] = Rj
    return R, p, aw


def newton_euler(m, q, qd, qdd, root_pos, root_quat, root_lin, root_ang, root_lin_acc, root_ang_acc, gravity):
    """Classical recursive Newton-Euler.  Returns (tau[nd], root force, root moment
    about the root origin).  root_lin_acc is the classical acceleration of the root
    origin.  Revolute joints only (the two shipped robots)."""
    R, p, aw = forward_kinematics(m, q, root_pos, root_quat)
    nb = m.nb
    w = [None] * nb; al = [None] * nb; a = [None] * nb
    for b in range(nb):
        if m.jtype[b] == _abi.JOINT_ROOT:
            w[b], al[b], a[b] = np.array(root_ang, float), np.array(root_ang_acc, float), \
                np.array(root_lin_acc, float)
            continue
        par = m.parent[b]
        d = p[b] - p[par]
        a_o = a[par] + np.cross(al[par], d) + np.cross(w[par], np.cross(w[par], d))
        if m.jtype[b] == _abi.JOINT_REVOLUTE:
            j = m.dof[b]
            w[b] = w[par] + aw[b] * qd[j]
            al[b] = al[par] + aw[b] * qdd[j] + np.cross(w[par], aw[b] * qd[j])
            a[b] = a_o
        else:
            w[b], al[b], a[b] = w[par], al[par], a_o
    f = [np.zeros(3) for _ in range(nb)]
    n = [np.zeros(3) for _ in range(nb)]
    g = np.array(gravity, float) * m.gravity_on
    for b in reversed(range(nb)):
        mass = m.mass[b]
        rc = R[b] @ np.array(m.com[b])
        Iw = R[b] @ sym(m.inertia[b]) @ R[b].T
        ac = a[b] + np.cross(al[b], rc) + np.cross(w[b], np.cross(w[b], rc))
        F = mass * (ac - g)
        N = Iw @ al[b] + np.cross(w[b], Iw @ w[b])
        f[b] += F
        n[b] += N + np.cross(rc, F)
        par = m.parent[b]
        if par >= 0:
            f[par] += f[b]
            n[par] += n[b] + np.cross(p[b] - p[par], f[b])
    tau = np.zeros(m.nd)
    for b in range(nb):
        if m.jtype[b] == _abi.JOINT_REVOLUTE:
            tau[m.dof[b]] = aw[b] @ n[b]
    return tau, f[0], n[0]


def mechanical_state(m, q, qd, root_pos, root_quat, root_lin, root_ang, gravity):
    """(kinetic+potential energy, linear momentum, angular momentum about the
    world origin, centre of mass)."""
    R, p, aw = forward_kinematics(m, q, root_pos, root_quat)
    nb = m.nb
    w = [None] * nb; v = [None] * nb
    E = 0.0; P = np.zeros(3); L = np.zeros(3); C = np.zeros(3); M = 0.0
    g = np.array(gravity, float) * m.gravity_on
    for b in range(nb):
        if m.jtype[b] == _abi.JOINT_ROOT:
            w[b], v[b] = np.array(root_ang, float), np.array(root_lin, float)
        else:
            par = m.parent[b]
            v[b] = v[par] + np.cross(w[par], p[b] - p[par])
            w[b] = w[par] + (aw[b] * qd[m.dof[b]] if m.jtype[b] == _abi.JOINT_REVOLUTE else 0.0)
        rc = R[b] @ np.array(m.com[b])
        vc = v[b] + np.cross(w[b], rc)
        Iw = R[b] @ sym(m.inertia[b]) @ R[b].T
        mass = m.mass[b]
        E += 0.5 * mass * vc @ vc + 0.5 * w[b] @ Iw @ w[b] - mass * g @ (p[b] + rc)
        P += mass * vc
        L += np.cross(p[b] + rc, mass * vc) + Iw @ w[b]
        C += mass * (p[b] + rc); M += mass
    return E, P, L, C / M


PENDULUM_URDF = """<robot name="pend">
 <link name="world_link"/>
 <link name="bob"><inertial><origin xyz="0 0 -{L}" rpy="0 0 0"/><mass value="{m}"/>
  <inertia ixx="{I}" ixy="0" ixz="0" iyy="{I}" iyz="0" izz="{I}"/></inertial></link>
 <joint name="hinge" type="revolute"><origin xyz="0 0 0" rpy="0 0 0"/><parent link="world_link"/><child link="bob"/>
  <axis xyz="0 1 0"/><limit effort="100" lower="-10" upper="10" velocity="1000"/></joint>
</robot>"""


def pendulum_model(L=0.5, mass=2.0, I=0.01):
    with tempfile.NamedTemporaryFile("w", suffix=".urdf", delete=False) as f:
        f.write(PENDULUM_URDF.format(L=L, m=mass, I=I))
        path = f.name
    try:
        return compile_urdf(path, fix_base_link=True, default_dof_drive_mode=_abi.DOF_MODE_EFFORT)
    finally:
        os.unlink(path)
