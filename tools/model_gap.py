#!/usr/bin/env python3
"""How far the shipped compliant contact model sits from a hard-contact solve.

Runs the shipped model (oracle/shf_oracle.c, float64 build -- what the HIP kernels reproduce bit for bit in float32) and the
independently written rigid-contact reference (oracle/hard_contact_ref.py: joint-space inertia matrix + projected
Gauss-Seidel, 8 + 1 sweeps) on the same scenes with the same actuation, and reports

  * the LOCAL deviation: every sub-step both models start from the shipped model's state; |dq|, |d root| after the one step
    (what "per-step state within 1e-4 of the reference" can mean for two different contact models);
  * the ACCUMULATED deviation: both run open loop from the same initial state.

Test infrastructure (uses oracle/); CPU only.   python tools/model_gap.py [--steps 1000] [--out profiles/r04_model_gap]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

A1_Q0 = np.array([0.1, 0.8, -1.5, 0.1, 0.8, -1.5, -0.1, 0.8, -1.5, -0.1, 0.8, -1.5])   # shifu_amd/a1_task.py: task_config.py:17-20 in dof order
KP, KD = 20.0, 0.5                                                                      # task_config.py:22-23


def set_solver(sp, solver):
    """solver: 'pgs' = the velocity-level solve the reference's PhysX fields configure (env_config.py:50-58: 8 + 1 sweeps), the
    default since round 5; 'compliant' = rounds 1-4's spring-damper law"""
    from shifu_amd import _abi
    if solver in ("pgs", "tgs"):      # "tgs": physx.solver_type = 1, the sub-stepped sweeps (include/shifu_amd.h: SHF_SOLVER_TGS)
        sp.solver, sp.pos_iters, sp.vel_iters, sp.max_contacts, sp.erp = (_abi.SOLVER_TGS if solver == "tgs" else _abi.SOLVER_PGS), 8, 1, 16, 0.2
    else:
        sp.solver = _abi.SOLVER_COMPLIANT
    return sp


def a1_setup(solver="compliant"):
    from shifu_amd import _abi
    from shifu_amd.model import asset_path, compile_urdf
    from tests.helpers import sim_params
    cm = compile_urdf(asset_path("a1.urdf"), default_dof_drive_mode=_abi.DOF_MODE_EFFORT)
    for d in range(cm.blob.nd):
        cm.blob.damping[d] = 0.5                       # dof_props['damping'] (robot.py:35-37), as FusedA1Env sets it
    return cm, set_solver(sim_params(angular_damping=0.0), solver)


def a1_targets(kind, k, dt):
    if kind == "stand":
        return A1_Q0
    # "trot": diagonal pairs in phase, 2 Hz, thigh and calf swinging a quarter period apart (open loop, same for both models)
    t = k * dt
    tgt = A1_Q0.copy()
    for leg, ph in enumerate((0.0, np.pi, np.pi, 0.0)):
        s, c = np.sin(2 * np.pi * 2.0 * t + ph), np.cos(2 * np.pi * 2.0 * t + ph)
        tgt[3 * leg + 1] += 0.25 * s
        tgt[3 * leg + 2] += 0.35 * max(c, 0.0) - 0.1
    return tgt


def shipped_step(oracle, m, sp, dof, root, tau):
    oracle.step(m, sp, 1, dof, root, effort=np.ascontiguousarray(tau, np.float64), friction=np.ones(1, np.float32), f64=True)


def run_a1(kind, steps, solver="compliant"):
    from oracle import pyoracle as oracle
    from oracle.hard_contact_ref import HardContactStepper
    oracle.build()
    cm, sp = a1_setup(solver)
    m = cm.blob
    dt = sp.dt
    ref = HardContactStepper(m, sp, mu=1.0, tgs=solver == "tgs")
    # compliant model: the LOCAL comparison starts from its state, whose feet sit m g / 4 k = 0.6 mm in the ground: a stabilised
    # hard solver would spend its step pushing them out (0.2 x 0.6 mm / 5 ms = 24 mm/s), so the one-step comparison is made
    # at the velocity level (no Baumgarte term); the accumulated one uses the stabilised solver on its own trajectory.
    # pgs: both sides are the stabilised solve.
    ref_local = HardContactStepper(m, sp, mu=1.0, baumgarte=0.0 if solver == "compliant" else 0.2, tgs=solver == "tgs")
    # settle the shipped model on its feet first (400 sub-steps of PD hold), so that both start from a state at rest
    dof = np.zeros((m.nd, 2)); dof[:, 0] = A1_Q0
    root = np.zeros((1, 13)); root[0, 2] = 0.33; root[0, 6] = 1.0
    for k in range(400):
        tau = KP * (A1_Q0 - dof[:, 0]) - KD * dof[:, 1]
        shipped_step(oracle, m, sp, dof, root, tau)
    dof0, root0 = dof.copy(), root.copy()
    # accumulated: open loop from the same state
    q, qd, rr = dof0[:, 0].copy(), dof0[:, 1].copy(), root0[0].copy()
    acc = []
    loc = []
    weight = []
    for k in range(steps):
        tgt = a1_targets(kind, k, dt)
        # local: the reference advances one step from the shipped model's current state
        ql, qdl, rl = dof[:, 0].copy(), dof[:, 1].copy(), root[0].copy()
        tau_s = KP * (tgt - dof[:, 0]) - KD * dof[:, 1]
        ref_local.step(ql, qdl, rl, tau_s.copy())
        shipped_step(oracle, m, sp, dof, root, tau_s)
        loc.append((np.abs(ql - dof[:, 0]).max(), np.linalg.norm(rl[:3] - root[0, :3]), np.abs(qdl - dof[:, 1]).max(),
                    np.linalg.norm(rl[7:10] - root[0, 7:10])))
        # accumulated: the reference on its own trajectory
        tau_r = KP * (tgt - q) - KD * qd
        fz = ref.step(q, qd, rr, tau_r)
        acc.append((np.abs(q - dof[:, 0]).max(), np.linalg.norm(rr[:3] - root[0, :3]), abs(rr[2] - root[0, 2])))
        weight.append(fz)
    loc, acc = np.array(loc), np.array(acc)
    mass = float(sum(m.mass[b] for b in range(m.nb)))
    # creep of the shipped model's base over the second half of the run (standing: the feet should stay where they are)
    return {"scene": f"A1 {kind}", "solver": solver, "steps": steps, "dt": dt,
            "local_dq_max": float(loc[:, 0].max()), "local_dq_mean": float(loc[:, 0].mean()),
            "local_droot_max": float(loc[:, 1].max()), "local_droot_mean": float(loc[:, 1].mean()),
            "local_dqd_max": float(loc[:, 2].max()), "local_dvroot_max": float(loc[:, 3].max()),
            "accum_dq": {str(n): float(acc[n - 1, 0]) for n in (1, 10, 100, steps) if n <= steps},
            "accum_droot": {str(n): float(acc[n - 1, 1]) for n in (1, 10, 100, steps) if n <= steps},
            "accum_dz_final": float(acc[-1, 2]),
            "hard_contact_normal_force_over_weight": float(np.mean(weight[steps // 2:]) / (mass * 9.81)),
            "root_z_shipped": float(root[0, 2]), "root_z_hard": float(rr[2])}


def run_abb(steps, solver="compliant", ensemble=False):
    """The ABB arm sweeps its rod sideways into the cube on the table (joint 1 turns at 0.2 rad/s): implicit POS drives in both."""
    from oracle import pyoracle as oracle
    from oracle.hard_contact_ref import HardContactStepper
    from shifu_amd.abb_task import ABB_BASE_POS, ABB_DEFAULT_DOF_POS, abb_boxes, abb_model
    from shifu_amd.backend import default_sim_params
    oracle.build()
    cm = abb_model(link_contacts=False)
    m = cm.blob
    sp = set_solver(default_sim_params(dt=0.02), solver)
    dt = sp.dt
    boxes = abb_boxes()
    q0 = np.array(ABB_DEFAULT_DOF_POS)
    kp, kd = np.array(m.kp[:m.nd]), np.array(m.kd[:m.nd])
    dof = np.zeros((m.nd, 2)); dof[:, 0] = q0
    root = np.zeros((4, 13)); root[:, 6] = 1.0
    root[0, :3] = ABB_BASE_POS
    root[1, :3] = (0, 0, 0.05); root[3, :3] = (0, 0, 0.1)
    # lower the arm (joint 2) until the rod's lower end hangs 1.2 cm above the table -- the height the task works at
    # (min_ee_pos z = 0.11, task_config.py:63) -- and put the cube 3 cm beside the rod (in +y), resting on the table
    from oracle.hard_contact_ref import Articulation
    A = Articulation(m)
    b, lp, seg, rad = A.spheres[0]

    def low_end(qv):
        R, p, aw = A.fk(qv, root[0, :3], root[0, 3:7])
        e0, e1 = p[b] + R[b] @ lp, p[b] + R[b] @ (lp + seg)
        return e0 if e0[2] < e1[2] else e1
    lo, hi = 0.0, 0.6
    for _ in range(50):
        mid = 0.5 * (lo + hi)
        qv = q0.copy(); qv[1] += mid
        if low_end(qv)[2] - rad > 0.112: lo = mid
        else: hi = mid
    q0[1] += 0.5 * (lo + hi)
    dof[:, 0] = q0
    tip = low_end(q0)
    root[2, :3] = (tip[0], tip[1] + 0.025 + rad + 0.03, 0.125)
    ref = HardContactStepper(m, sp, mu=1.0, box={"dim": [0.05, 0.05, 0.05], "mass": 0.1}, box_plane_z=0.1, box_mu=0.5, tgs=solver == "tgs")
    ref_local = HardContactStepper(m, sp, mu=1.0, box={"dim": [0.05, 0.05, 0.05], "mass": 0.1}, box_plane_z=0.1, box_mu=0.5, tgs=solver == "tgs",
                                   baumgarte=0.0 if solver == "compliant" else 0.2)
    # implicit POS drive in the reference: M~ += dt (kd + dt kp) on the diagonal, exactly the shipped joint law (the -kd qd part
    # of the torque rides on the damping term)
    for r_ in (ref, ref_local):
        r_.A.damping = r_.A.damping + kd + dt * kp

    # The sweep ends with the cube slipping off the turning rod after a rocking, yawing push: WHEN it slips is a bifurcation
    # (a 10 um shift of the cube's start moves the reference's own answer from 15.7 to 17.5 cm), so the distance travelled is
    # compared as an ensemble over such shifts, next to the single run
    start_xy = root[2, :2].copy()
    ens_ref, ens_shipped = [], []
    for shift in ((0.0, 1e-5, -1e-5, 2e-5, -2e-5) if ensemble else ()):
        d2, r2 = dof.copy(), root.copy()
        r2[2, 1] += shift
        q, qd, rr, bx = d2[:, 0].copy(), d2[:, 1].copy(), r2[0].copy(), r2[2].copy()
        for k in range(steps):
            tgt = q0.copy(); tgt[0] += 0.2 * dt * (k + 1)
            oracle.scene_step(m, sp, boxes, 1, d2, r2, pos_target=np.ascontiguousarray(tgt, np.float64), friction=np.ones(1, np.float32), f64=True)
            ref.step(q, qd, rr, kp * (tgt - q), box_state=bx)
        ens_shipped.append(float(np.linalg.norm(r2[2, :2] - start_xy)))
        ens_ref.append(float(np.linalg.norm(bx[:2] - start_xy)))
    q, qd, rr, bx = dof[:, 0].copy(), dof[:, 1].copy(), root[0].copy(), root[2].copy()
    loc, acc = [], []
    for k in range(steps):
        tgt = q0.copy(); tgt[0] += 0.2 * dt * (k + 1)
        ql, qdl, rl, bl = dof[:, 0].copy(), dof[:, 1].copy(), root[0].copy(), root[2].copy()
        ref_local.step(ql, qdl, rl, kp * (tgt - ql), box_state=bl)
        oracle.scene_step(m, sp, boxes, 1, dof, root, pos_target=np.ascontiguousarray(tgt, np.float64), friction=np.ones(1, np.float32), f64=True)
        loc.append((np.abs(ql - dof[:, 0]).max(), np.linalg.norm(bl[:3] - root[2, :3])))
        ref.step(q, qd, rr, kp * (tgt - q), box_state=bx)
        acc.append((np.abs(q - dof[:, 0]).max(), np.linalg.norm(bx[:3] - root[2, :3])))
    loc, acc = np.array(loc), np.array(acc)
    return {"scene": "ABB rod pushes the cube", "solver": solver, "steps": steps, "dt": dt,
            "local_dq_max": float(loc[:, 0].max()), "local_dq_mean": float(loc[:, 0].mean()),
            "local_dcube_max": float(loc[:, 1].max()), "local_dcube_mean": float(loc[:, 1].mean()),
            "accum_dq": {str(n): float(acc[n - 1, 0]) for n in (1, 10, 100, steps) if n <= steps},
            "accum_dcube": {str(n): float(acc[n - 1, 1]) for n in (1, 10, 100, steps) if n <= steps},
            "cube_travel_ensemble_shipped": ens_shipped, "cube_travel_ensemble_hard": ens_ref,
            "cube_travel_shipped": float(np.linalg.norm(root[2, :2] - np.array([tip[0], tip[1] + 0.025 + rad + 0.03]))),
            "cube_travel_hard": float(np.linalg.norm(bx[:2] - np.array([tip[0], tip[1] + 0.025 + rad + 0.03])))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--out", default=None)
    ap.add_argument("--solver", default="both", choices=("pgs", "tgs", "compliant", "both"))
    args = ap.parse_args()
    res = []
    for solver in (("pgs", "compliant") if args.solver == "both" else (args.solver,)):
        res += [run_a1("stand", args.steps, solver), run_a1("trot", args.steps, solver), run_abb(min(args.steps, 250), solver, ensemble=True)]
    for r in res:
        print(json.dumps(r))
    if args.out:
        with open(args.out + ".json", "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
