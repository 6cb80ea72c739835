"""The oracle's articulated-body dynamics against an independent derivation: the manipulator equation
M(q) qdd + c(q, qd) + g(q) = tau obtained with computer algebra (sympy) from the Lagrangian of a four-joint spatial chain
-- forward kinematics from the URDF's own joint frames, M = sum m Jv^T Jv + Jw^T I Jw, Christoffel symbols of M, gradient
of the potential.  No spatial algebra, no recursion: nothing shared with Featherstone's algorithm (oracle/shf_oracle.c)
or with the Newton-Euler checker of tests/helpers.py.  Also exercises the URDF compiler's frame conventions (origin xyz /
rpy, skew joint axes, a prismatic joint in the middle of the chain, off-diagonal inertia tensors, com offsets)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shifu_amd import _abi
from tests import helpers as H

sympy = pytest.importorskip("sympy")

JOINTS = [  # type, origin xyz, origin rpy, axis
    ("revolute", (0.0, 0.0, 0.1), (0.0, 0.0, 0.0), (0.0, 0.0, 1.0)),
    ("revolute", (0.1, 0.0, 0.2), (0.3, 0.0, 0.0), (0.0, 1.0, 0.0)),
    ("prismatic", (0.2, 0.05, 0.0), (0.0, 0.2, 0.4), (1.0, 0.0, 0.0)),
    ("revolute", (0.1, 0.0, 0.1), (0.1, -0.2, 0.3), (0.3, 0.5, 0.8)),
]
LINKS = [  # mass, com xyz, inertia (ixx, ixy, ixz, iyy, iyz, izz) about the com in link axes
    (2.0, (0.02, -0.01, 0.05), (0.020, 0.002, -0.001, 0.030, 0.003, 0.025)),
    (1.5, (0.10, 0.00, 0.02), (0.010, -0.001, 0.002, 0.020, 0.001, 0.015)),
    (1.0, (0.05, 0.03, -0.02), (0.008, 0.001, 0.000, 0.009, -0.002, 0.012)),
    (0.7, (-0.02, 0.04, 0.06), (0.004, 0.0005, 0.0010, 0.005, 0.0007, 0.006)),
]


def _urdf():
    out = ['<robot name="chain"><link name="base"/>']
    for k, ((jt, xyz, rpy, ax), (m, com, I)) in enumerate(zip(JOINTS, LINKS)):
        n = np.array(ax) / np.linalg.norm(ax)
        out.append(f'<link name="l{k}"><inertial><origin xyz="{com[0]} {com[1]} {com[2]}"/><mass value="{m}"/>'
                   f'<inertia ixx="{I[0]}" ixy="{I[1]}" ixz="{I[2]}" iyy="{I[3]}" iyz="{I[4]}" izz="{I[5]}"/></inertial></link>')
        out.append(f'<joint name="j{k}" type="{jt}"><parent link="{"base" if k == 0 else "l%d" % (k - 1)}"/><child link="l{k}"/>'
                   f'<origin xyz="{xyz[0]} {xyz[1]} {xyz[2]}" rpy="{rpy[0]} {rpy[1]} {rpy[2]}"/>'
                   f'<axis xyz="{float(n[0])!r} {float(n[1])!r} {float(n[2])!r}"/><limit effort="1000" lower="-10" upper="10" velocity="100"/></joint>')
    out.append("</robot>")
    return "\n".join(out)


def _manipulator_equation():
    """lambdified (M, c + g) of the chain, gravity (0, 0, -9.81), from the Lagrangian"""
    sp = sympy
    q = sp.symbols("q0:4"); qd = sp.symbols("v0:4")

    def rot_axis(n, th):                                   # Rodrigues
        K = sp.Matrix([[0, -n[2], n[1]], [n[2], 0, -n[0]], [-n[1], n[0], 0]])
        return sp.eye(3) + sp.sin(th) * K + (1 - sp.cos(th)) * K * K

    def rot_rpy(r, p, y):                                  # URDF: fixed-axis roll, pitch, yaw = Rz(y) Ry(p) Rx(r)
        return rot_axis((0, 0, 1), y) * rot_axis((0, 1, 0), p) * rot_axis((1, 0, 0), r)

    R, p = sp.eye(3), sp.zeros(3, 1)
    Mq, V = sp.zeros(4, 4), 0
    axes = []                                              # world rotation axes of the revolute joints so far (else None)
    for k, ((jt, xyz, rpy, ax), (m, com, I)) in enumerate(zip(JOINTS, LINKS)):
        n = sp.Matrix([sp.Float(float(v)) for v in np.array(ax) / np.linalg.norm(ax)])
        p = p + R * sp.Matrix([sp.Float(v) for v in xyz])     # <origin xyz>: in the parent link's axes
        R = R * rot_rpy(*[sp.Float(v) for v in rpy])           # <origin rpy>: the joint frame
        if jt == "revolute":
            axes.append(R * n)
            R = R * rot_axis(n, q[k])
        else:
            axes.append(None)
            p = p + R * n * q[k]
        pc = p + R * sp.Matrix([sp.Float(v) for v in com])
        Jv = pc.jacobian(sp.Matrix(q))
        Jw = sp.zeros(3, 4)
        for j in range(k + 1):
            if axes[j] is not None:
                Jw[:, j] = axes[j]
        Ib = sp.Matrix([[I[0], I[1], I[2]], [I[1], I[3], I[4]], [I[2], I[4], I[5]]])
        Mq += m * Jv.T * Jv + Jw.T * (R * Ib * R.T) * Jw
        V += m * 9.81 * pc[2]
    bias = sp.zeros(4, 1)
    for i in range(4):
        ci = 0
        for j in range(4):
            for kk in range(4):
                ci += (sp.diff(Mq[i, j], q[kk]) - sp.Rational(1, 2) * sp.diff(Mq[j, kk], q[i])) * qd[j] * qd[kk]
        bias[i] = ci + sp.diff(V, q[i])
    return sp.lambdify([q, qd], [Mq, bias], modules="numpy", cse=True)


def test_aba_matches_the_lagrangian_manipulator_equation(oracle, tmp_path):
    from shifu_amd.model import compile_urdf
    (tmp_path / "chain.urdf").write_text(_urdf())
    cm = compile_urdf(str(tmp_path / "chain.urdf"), fix_base_link=True, default_dof_drive_mode=_abi.DOF_MODE_EFFORT)
    m = cm.blob
    assert m.nd == 4 and cm.dof_names == ["j0", "j1", "j2", "j3"]
    f = _manipulator_equation()
    sp_ = H.sim_params()
    rng = np.random.default_rng(0)
    root = np.array([0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0], float)
    for _ in range(12):
        q, qd, tau = rng.uniform(-1.2, 1.2, 4), rng.uniform(-3, 3, 4), rng.uniform(-20, 20, 4)
        M, b = f(q, qd)
        want = np.linalg.solve(np.array(M, float), tau - np.array(b, float).reshape(-1))
        got, _ = oracle.accel(m, sp_, np.stack([q, qd], 1).reshape(-1), root, tau, f64=True)
        # ShfModel holds its constants (frames, masses, inertia tensors) as float32: agreement to that precision,
        # relative to the largest acceleration of the state (the light last link reaches thousands of rad/s^2)
        assert np.abs(got - want).max() <= 2e-7 * (1.0 + np.abs(want).max()), (got, want)
