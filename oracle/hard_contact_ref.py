"""hard_contact_ref.py -- a SECOND, independently structured reference for the step path: rigid (hard) contacts.

TEST INFRASTRUCTURE ONLY, like everything under oracle/: nothing under shifu_amd/ imports it.  It exists to put numbers on
the one parity that cannot be pinned: the reference's physics is PhysX 5 -- closed, absent -- a
hard-contact velocity-level solver (TGS: `num_position_iterations = 8`, `num_velocity_iterations = 1`,
reference shifu/configs/env_config.py:50-52), while the shipped model (oracle/shf_oracle.c, the HIP kernels) is a compliant
contact law folded linearly-implicitly into one articulated-body solve.  This file is neither: it is a textbook
time-stepping scheme written from different building blocks --

  * dynamics in generalised coordinates, classical 3-vectors: the joint-space inertia matrix M(q) assembled column by column
    from a recursive Newton-Euler inverse dynamics (unit accelerations), the bias from the same routine (the shipped path
    never forms M: it is Featherstone's O(n) articulated-body algorithm in world-aligned Pluecker coordinates);
  * contacts as unilateral velocity constraints with Coulomb friction, solved by projected block Gauss-Seidel on the
    Delassus operator J M^-1 J^T, 8 + 1 sweeps (J. J. Moreau / D. Stewart & J. Trinkle time stepping; E. Catto, "Iterative
    dynamics with temporal coherence", 2005, for the sweep), Baumgarte stabilisation of penetration capped at
    `max_depenetration_velocity`, speculative margin `contact_offset` -- no stiffness, no damping, no sag;
  * the same explicit joint torques, passive joint damping (implicit), armature and semi-implicit Euler as the shipped step,
    so that what differs is the contact model and the arithmetic, not the actuation.

tools/model_gap.py runs both on the same scenes and writes the table DESIGN.md 3 quotes; tests/test_model_gap.py fails
if the measured deviations grow.  Scope: one floating- or fixed-base articulation with revolute joints, its sample points
against the plane z = 0 (what BASELINE config 2 needs), optionally one free box whose corners rest on a horizontal plane and
which a sphere of the articulation pushes (config 5's cube on the table).
"""
import numpy as np

JOINT_ROOT, JOINT_REVOLUTE, JOINT_PRISMATIC, JOINT_WELD = 0, 1, 2, 3


def _skew(r):
    return np.array([[0.0, -r[2], r[1]], [r[2], 0.0, -r[0]], [-r[1], r[0], 0.0]])


def _quat_to_mat(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def _rodrigues(a, q):
    K = _skew(a)
    return np.eye(3) + np.sin(q) * K + (1.0 - np.cos(q)) * (K @ K)


def _integrate_quat(q, w, dt):
    x, y, z, ww = q
    hx, hy, hz = 0.5 * dt * w
    n = np.array([x + hx * ww + hy * z - hz * y, y + hy * ww + hz * x - hx * z, z + hz * ww + hx * y - hy * x,
                  ww - hx * x - hy * y - hz * z])
    return n / np.linalg.norm(n)


class Articulation:
    """The tree of a compiled ShfModel (shifu_amd/model.py) read into plain NumPy arrays."""

    def __init__(self, m):
        self.nb, self.nd, self.fixed = int(m.nb), int(m.nd), bool(m.fixed_base)
        nb = self.nb
        self.parent = [int(m.parent[b]) for b in range(nb)]
        self.jtype = [int(m.jtype[b]) for b in range(nb)]
        self.dof = [int(m.dof[b]) for b in range(nb)]
        self.tpos = [np.array(m.tpos[b][:], float) for b in range(nb)]
        self.trot = [np.array(m.trot[b][:], float).reshape(3, 3) for b in range(nb)]
        self.axis = [np.array(m.axis[b][:], float) for b in range(nb)]
        self.mass = [float(m.mass[b]) for b in range(nb)]
        self.com = [np.array(m.com[b][:], float) for b in range(nb)]
        I6 = [np.array(m.inertia[b][:], float) for b in range(nb)]
        self.inertia = [np.array([[i[0], i[1], i[2]], [i[1], i[3], i[4]], [i[2], i[4], i[5]]]) for i in I6]
        self.gravity_on = float(m.gravity_on)
        self.armature = np.array(m.armature[:self.nd], float)
        self.damping = np.array(m.damping[:self.nd], float)
        self.points = [(int(m.pt_body[i]), np.array(m.pt_pos[i][:], float), float(m.pt_radius[i])) for i in range(int(m.np))]
        self.spheres = [(int(m.sph_body[i]), np.array(m.sph_pos[i][:], float), np.array(m.sph_seg[i][:], float), float(m.sph_radius[i]))
                        for i in range(int(m.nsph))]
        for b in range(nb):
            assert self.jtype[b] in (JOINT_ROOT, JOINT_REVOLUTE, JOINT_WELD), "revolute trees only"
        self.nv = self.nd + (0 if self.fixed else 6)

    # -- kinematics -------------------------------------------------------------------------------------------------
    def fk(self, q, root_pos, root_quat):
        R, p, aw = [None] * self.nb, [None] * self.nb, [None] * self.nb
        for b in range(self.nb):
            if self.jtype[b] == JOINT_ROOT:
                R[b], p[b] = _quat_to_mat(root_quat), np.array(root_pos, float)
                continue
            par = self.parent[b]
            Rj = R[par] @ self.trot[b]
            p[b] = p[par] + R[par] @ self.tpos[b]
            aw[b] = Rj @ self.axis[b]
            R[b] = Rj @ _rodrigues(self.axis[b], q[self.dof[b]]) if self.jtype[b] == JOINT_REVOLUTE else Rj
        return R, p, aw

    def inverse_dynamics(self, kin, qd, qdd, w0, al0, a0, gravity):
        """Classical recursive Newton-Euler: joint torques, root force and root moment (about the root origin) that produce
        the accelerations (al0, a0 = classical acceleration of the root origin, qdd) at velocities (w0, qd)."""
        R, p, aw = kin
        nb = self.nb
        w, al, a = [None] * nb, [None] * nb, [None] * nb
        for b in range(nb):
            if self.jtype[b] == JOINT_ROOT:
                w[b], al[b], a[b] = w0, al0, a0
                continue
            par = self.parent[b]
            d = p[b] - p[par]
            a[b] = a[par] + np.cross(al[par], d) + np.cross(w[par], np.cross(w[par], d))
            if self.jtype[b] == JOINT_REVOLUTE:
                j = self.dof[b]
                w[b] = w[par] + aw[b] * qd[j]
                al[b] = al[par] + aw[b] * qdd[j] + np.cross(w[par], aw[b] * qd[j])
            else:
                w[b], al[b] = w[par], al[par]
        f = [np.zeros(3) for _ in range(nb)]
        n = [np.zeros(3) for _ in range(nb)]
        for b in reversed(range(nb)):
            rc = R[b] @ self.com[b]
            Iw = R[b] @ self.inertia[b] @ R[b].T
            ac = a[b] + np.cross(al[b], rc) + np.cross(w[b], np.cross(w[b], rc))
            F = self.mass[b] * (ac - gravity * self.gravity_on)
            f[b] = f[b] + F
            n[b] = n[b] + Iw @ al[b] + np.cross(w[b], Iw @ w[b]) + np.cross(rc, F)
            par = self.parent[b]
            if par >= 0:
                f[par] = f[par] + f[b]
                n[par] = n[par] + n[b] + np.cross(p[b] - p[par], f[b])
        tau = np.zeros(self.nd)
        for b in range(nb):
            if self.jtype[b] == JOINT_REVOLUTE:
                tau[self.dof[b]] = aw[b] @ n[b]
        return np.concatenate([tau] if self.fixed else [n[0], f[0], tau])

    def mass_and_bias(self, kin, v, gravity):
        """M (nv x nv) and b with M a + b = generalised force; v = [w_root, v_root_origin, qd] (qd alone for a fixed base)."""
        z3 = np.zeros(3)
        w0, qd = (z3, v) if self.fixed else (v[:3], v[6:])
        bias = self.inverse_dynamics(kin, qd, np.zeros(self.nd), w0, z3, z3, gravity)
        M = np.zeros((self.nv, self.nv))
        zero_qd = np.zeros(self.nd)
        for k in range(self.nv):
            e = np.zeros(self.nv); e[k] = 1.0
            al0, a0, qdd = (z3, z3, e) if self.fixed else (e[:3], e[3:6], e[6:])
            M[:, k] = self.inverse_dynamics(kin, zero_qd, qdd, z3, al0, a0, z3)
        M = 0.5 * (M + M.T)
        off = 0 if self.fixed else 6
        M[np.arange(off, self.nv), np.arange(off, self.nv)] += self.armature
        return M, bias

    def point_jacobian(self, kin, body, x):
        """3 x nv: velocity of the material point of `body` that sits at world position x."""
        R, p, aw = kin
        J = np.zeros((3, self.nv))
        off = 0 if self.fixed else 6
        if not self.fixed:
            J[:, :3] = -_skew(x - p[0])
            J[:, 3:6] = np.eye(3)
        b = body
        while b >= 0:
            if self.jtype[b] == JOINT_REVOLUTE:
                J[:, off + self.dof[b]] = np.cross(aw[b], x - p[b])
            b = self.parent[b]
        return J


class HardContactStepper:
    """One articulation (+ optionally one free box) under rigid contacts; step() advances dt."""

    def __init__(self, model, params, mu=1.0, sweeps=(8, 1), baumgarte=0.2, box=None, box_plane_z=None, box_mu=0.5, tgs=False):
        self.A = Articulation(model)
        self.dt = float(params.dt)
        self.g = np.array(params.gravity[:], float)
        self.offset = float(params.contact_offset)
        self.vdep = float(params.max_depen_vel)
        self.ang_damp = float(params.angular_damping)
        self.mu, self.npos, self.nvel, self.beta = mu, int(sweeps[0]), int(sweeps[1]), baumgarte
        # tgs: physx.solver_type = 1 (shifu/configs/env_config.py:50) as sub-stepped sweeps -- the position iterations are
        # sub-iterations of h = dt / npos: targets from the gaps as they stand (horizon h), gaps advanced by the normal velocity
        # each sweep leaves, the poses moved by the mean of the impulses after the sweeps; velocity iterations against what is left
        self.tgs = bool(tgs)
        self.box = box          # dict(dim, mass) or None
        self.box_plane_z, self.box_mu = box_plane_z, box_mu

    def step(self, q, qd, root, tau, box_state=None):
        """q, qd (nd), root (13: pos quat lin ang, world), tau (nd) explicit joint torques; box_state (13) for the free box.
        Updated in place.  Returns the normal impulses / dt of the articulation's active contacts (their sum ~ weight)."""
        A, dt = self.A, self.dt
        kin = A.fk(q, root[:3], root[3:7])
        v = qd.copy() if A.fixed else np.concatenate([root[10:13], root[7:10], qd])
        M, bias = A.mass_and_bias(kin, v, self.g)
        off = 0 if A.fixed else 6
        gen = np.zeros(A.nv)
        gen[off:] = tau
        Mt = M.copy()
        Mt[np.arange(off, A.nv), np.arange(off, A.nv)] += dt * A.damping        # passive joint damping, implicit
        gen[off:] -= A.damping * qd
        nbx = 6 if self.box is not None else 0
        nv = A.nv + nbx
        Mfull = np.zeros((nv, nv)); Mfull[:A.nv, :A.nv] = Mt
        rhs = np.zeros(nv); rhs[:A.nv] = gen - bias
        vfull = np.zeros(nv); vfull[:A.nv] = v
        if self.box is not None:
            bm = self.box["mass"]; dx, dy, dz = self.box["dim"]
            Rb = _quat_to_mat(box_state[3:7])
            Ib = Rb @ np.diag([bm * (dy * dy + dz * dz) / 12, bm * (dx * dx + dz * dz) / 12, bm * (dx * dx + dy * dy) / 12]) @ Rb.T
            wb = box_state[10:13]
            Mfull[A.nv:A.nv + 3, A.nv:A.nv + 3] = Ib                # [w_box, v_box (centre of mass)]
            Mfull[A.nv + 3:, A.nv + 3:] = bm * np.eye(3)
            rhs[A.nv:A.nv + 3] = -np.cross(wb, Ib @ wb)
            rhs[A.nv + 3:] = bm * self.g
            vfull[A.nv:A.nv + 3] = wb; vfull[A.nv + 3:] = box_state[7:10]
        Minv = np.linalg.inv(Mfull)
        vfree = vfull + dt * (Minv @ rhs)
        # contacts: rows (normal, t1, t2) per contact, normal pointing at the first body
        Js, gaps, mus = [], [], []
        R, p, aw = kin
        for (b, lp, rad) in A.points:
            x = p[b] + R[b] @ lp
            phi = x[2] - rad
            if phi < self.offset:
                J = np.zeros((3, nv)); J[:, :A.nv] = A.point_jacobian(kin, b, x - np.array([0.0, 0.0, rad]))
                Js.append(J[[2, 0, 1]]); gaps.append(phi); mus.append(self.mu)
        narm = len(Js)
        if self.box is not None:
            Rb = _quat_to_mat(box_state[3:7]); cb = box_state[:3]
            hd = 0.5 * np.array(self.box["dim"])
            for c in range(8):
                lc = np.array([hd[0] if c & 4 else -hd[0], hd[1] if c & 2 else -hd[1], hd[2] if c & 1 else -hd[2]])
                x = cb + Rb @ lc
                phi = x[2] - self.box_plane_z
                if phi < self.offset:
                    J = np.zeros((3, nv)); J[:, A.nv:A.nv + 3] = -_skew(x - cb); J[:, A.nv + 3:] = np.eye(3)
                    Js.append(J[[2, 0, 1]]); gaps.append(phi); mus.append(self.box_mu)
            for (b, lp, seg, rad) in A.spheres[:1]:                      # the rounded shape that pushes the box
                c0 = p[b] + R[b] @ lp; s = R[b] @ seg
                ts = np.linspace(0.0, 1.0, 401)
                pts = c0[None, :] + ts[:, None] * s[None, :]
                dl = (pts - cb) @ Rb                                      # samples of the centre line in the box frame
                ql = np.clip(dl, -hd, hd)
                dists = np.linalg.norm(dl - ql, axis=1)
                imin = int(np.argmin(dists))
                # a capsule lying along a face is a LINE contact, held at both ends of the stretch that is equally close (within
                # 5e-4 rad of parallel, the shipped model's flat-sample tolerance); otherwise the closest point alone
                flat = np.nonzero(dists <= dists[imin] + 5e-4 * np.linalg.norm(s) * np.abs(ts - ts[imin]) + 1e-12)[0]
                picks = [imin] if ts[flat[-1]] - ts[flat[0]] < 1e-2 else [int(flat[0]), int(flat[-1])]
                for ip in picks:
                    dist, d, qc = dists[ip], dl[ip], ql[ip]
                    if dist > 1e-12 and dist - rad < self.offset:
                        n = Rb @ ((d - qc) / dist)                             # from the box towards the sphere
                        xc = cb + Rb @ qc
                        t1 = np.cross(n, [0.0, 0.0, 1.0]); t1 = t1 / np.linalg.norm(t1) if np.linalg.norm(t1) > 1e-9 else np.array([1.0, 0, 0])
                        t2 = np.cross(n, t1)
                        Jr = np.zeros((3, nv)); Jr[:, :A.nv] = A.point_jacobian(kin, b, xc)
                        Jr[:, A.nv:A.nv + 3] -= -_skew(xc - cb); Jr[:, A.nv + 3:] -= np.eye(3)
                        Js.append(np.vstack([n @ Jr, t1 @ Jr, t2 @ Jr])); gaps.append(dist - rad); mus.append(0.5 * (self.mu + self.box_mu))
        imp_n = 0.0
        vnew = vpos = vfree
        if Js:
            J = np.vstack(Js)
            W = J @ Minv @ J.T
            u0 = J @ vfree
            k = len(Js)
            target = np.zeros(3 * k)
            for c in range(k):
                phi = gaps[c]
                target[3 * c] = -phi / dt if phi > 0.0 else min(self.beta * (-phi) / dt, self.vdep)
            # position iterations against the biased targets -> the velocities the poses advance with; velocity iterations
            # (physx.num_velocity_iterations) from there against the targets without the penetration bias -> the velocities
            # the step hands on
            target_v = target.copy()
            for c in range(k):
                if gaps[c] <= 0.0:
                    target_v[3 * c] = 0.0
            pimp = np.zeros(3 * k)
            tgs = self.tgs and self.npos > 0
            h = dt / self.npos if tgs else dt
            gap_now = np.array(gaps, float)
            psum = np.zeros(3 * k)
            for phase, (nsweep, tg) in enumerate(((self.npos, target), (self.nvel, target_v))):
                # Sequential impulses with a friction cone (E. Catto 2005): per contact the normal impulse first (clamped at
                # zero), then the tangential impulse that stops the sliding under it if that lies inside the cone mu p_n, else a
                # projected step against the sliding velocity.  (Round 4's sweep solved the 3x3 block for sticking and
                # projected THAT onto the cone: in steady sliding its friction came out at 0.44 N for mu = 0.6 on a block on an
                # incline -- tests/test_hard_contact.py now holds both solvers to Coulomb's law.)
                if tgs and phase == 1:
                    tg = np.zeros(3 * k)
                    for c in range(k):
                        tg[3 * c] = -gap_now[c] / dt if gap_now[c] >= 0.0 else 0.0
                for it in range(nsweep):
                    if tgs and phase == 0:
                        tg = np.zeros(3 * k)
                        for c in range(k):
                            tg[3 * c] = -gap_now[c] / h if gap_now[c] >= 0.0 else min(self.beta * (-gap_now[c]) / h, self.vdep)
                    for c in range(k):
                        sl = slice(3 * c, 3 * c + 3)
                        Wcc = W[sl, sl] + 1e-6 * np.trace(W[sl, sl]) * np.eye(3)     # the same regularisation as oracle/shf_oracle.c hard_solve
                        u = u0[sl] + W[sl] @ pimp + (Wcc - W[sl, sl]) @ pimp[sl]
                        pn = max(pimp[3 * c] - (u[0] - tg[3 * c]) / Wcc[0, 0], 0.0)
                        u = u + Wcc[:, 0] * (pn - pimp[3 * c])
                        pt = pimp[3 * c + 1:3 * c + 3] - np.linalg.solve(Wcc[1:, 1:], u[1:])
                        lim = mus[c] * pn
                        if np.hypot(pt[0], pt[1]) > lim:
                            # sliding: a projected step against the sliding velocity with a scalar gain (fixed point: friction
                            # opposite to u_t, Coulomb's law; the block inverse as gain leaves it opposite to W_tt^-1 u_t)
                            pt = pimp[3 * c + 1:3 * c + 3] - u[1:] / np.trace(Wcc[1:, 1:])
                            nt = np.hypot(pt[0], pt[1])
                            if nt > lim:
                                pt = pt * (lim / nt)
                        pimp[sl] = [pn, pt[0], pt[1]]
                    if tgs and phase == 0:
                        un = (u0 + W @ pimp)[0::3]
                        un = un + 1e-6 * np.array([np.trace(W[3 * c:3 * c + 3, 3 * c:3 * c + 3]) for c in range(k)]) * pimp[0::3]
                        gap_now = gap_now + h * un
                        psum += pimp
                if phase == 0:
                    vpos = vfree + Minv @ (J.T @ (psum / self.npos if tgs else pimp))
            vnew = vfree + Minv @ (J.T @ pimp)
            imp_n = float(sum(pimp[3 * c] for c in range(narm))) / dt
        # semi-implicit Euler: poses with the velocities of the position iterations, state velocities from the velocity iterations
        if A.fixed:
            qd[:] = vnew[:A.nd]
            q += dt * vpos[:A.nd]
        else:
            wn = vnew[:3] / (1.0 + dt * self.ang_damp)
            wp = vpos[:3] / (1.0 + dt * self.ang_damp)
            root[10:13] = wn
            root[7:10] = vnew[3:6]
            root[:3] += dt * vpos[3:6]
            root[3:7] = _integrate_quat(root[3:7], wp, dt)
            qd[:] = vnew[6:A.nv]
            q += dt * vpos[6:A.nv]
        if self.box is not None:
            box_state[10:13] = vnew[A.nv:A.nv + 3] / (1.0 + dt * self.ang_damp)
            box_state[7:10] = vnew[A.nv + 3:]
            box_state[:3] += dt * vpos[A.nv + 3:]
            box_state[3:7] = _integrate_quat(box_state[3:7], vpos[A.nv:A.nv + 3] / (1.0 + dt * self.ang_damp), dt)
        return imp_n
