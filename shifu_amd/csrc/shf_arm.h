// shf_arm.h -- gym.simulate() for a fixed-base serial arm (the ABB IRB1200 of config 5: base + NL revolute links in one
// chain) next to the box actors of its scene, with the arm's recursions on ONE lane.
//
// In a serial chain every kinematic / ABA level holds exactly one body, so the body-per-lane sub-step of shf_device.h
// gains nothing from its lanes there and pays an LDS hand-off plus a group synchronisation per level, 3 x NL per
// sub-step.  Here (tools/valu_microbench: a wave issues one VALU instruction per ~4.6 clocks whatever its active lanes,
// an LDS write -> read hand-off costs 70-110 clocks):
//   lane b in 1..NL   joint b's local rotation, link b's rigid inertia and its contact / pair folds
//   lane d < NL       dof d's drive effort and integration
//   lane 0            the chain: poses root -> tip, the ABA inward pass tip -> root and the outward pass, link after
//                     link; per-link (S, c, U, 1/D, u) parked in an LDS record (krec) instead of 20 x NL registers
//   lanes nb..        the box actors (shf_boxes.h, unchanged)
// Five group synchronisations per sub-step besides the box code's own.
//
// ARITHMETIC: the same operations in the same order as substep<> (shf_device.h) and oracle/shf_oracle.c; only the lane
// that executes them changes.  tests/test_gpu_parity.py holds this path and the body-per-lane path to the oracle bit
// for bit (the run-time-shaped kernel is still the body-per-lane one).
// Reference call site replaced: gym.simulate inside IsaacGymEnv.step (shifu/gym/isaac_gym.py:140) for
// examples/abb_pushbox_vision (AbbRobot, shifu/units/robot.py:96-151).
#pragma once
#include "shf_link.h"

template <int NL_>
struct ArmChain {
  static constexpr int NL = NL_, NB = NL_ + 1, ND = NL_;
  static bool matches(const ShfModel& m) {
    if (m.nb != NB || m.nd != ND || !m.fixed_base || m.jtype[0] != SHF_JOINT_ROOT || m.nlevels != NL) return false;
    for (int b = 1; b < NB; b++)
      if (m.jtype[b] != SHF_JOINT_REVOLUTE || m.parent[b] != b - 1 || m.dof[b] != b - 1 || m.dyn[b] != b || m.level[b] != b)
        return false;
    return true;
  }
};
// What substep_hard_finish<.., RECORDS = false, ARMNL = NL> asks of its lane's model view (csrc/shf_hard.h) for such an arm: no
// tree indices, no children -- lane_model_load's fifteen dependent LDS reads per sub-step shrink to one
struct ArmSolveLane {
  bool isbody, isdyn, moving;
  int lev;
  float vel_limit;
  DEV int level() const { return lev; }
  template <int NL>
  DEV static ArmSolveLane load(const ShfModel* m, int l) {
    ArmSolveLane M;
    M.isbody = l <= NL; M.isdyn = M.isbody; M.moving = l >= 1 && l <= NL;     // (ArmChain<NL>::matches: dyn[b] == b, the root fixed)
    M.lev = M.isbody ? l : -1;
    M.vel_limit = m->vel_limit[l < NL ? l : 0];
    return M;
  }
};
#define KREC_STRIDE 20   /* S[6] c[6] U[6] invD u */
#define ARM_KREC_WORDS(nl) ((nl) * KREC_STRIDE)

// One lane's view of its env's arm during a sub-step: the phases of the sub-step as members, so that one wave can run
// them in sequence (arm_substep) or an "arm" wave can run them while a "box" wave of the same workgroup handles the box
// actors (k_abb_step_ws in shf_api.hip).  krec: ARM_KREC_WORDS(NL) floats of this env's LDS, 16-byte aligned.
template <int G, class DM, int NL, class LM = LaneModel>      // LM: the lane's model constants in registers (LaneModel) or read from the LDS model at use
struct ArmLane {
  static_assert(DM::NPC > 0 && G < 64, "compile-time point count; lane groups inside one wavefront");
  static constexpr int nb = NL + 1, nd = NL, NR = LANE_ROUNDS(G, DM);   // NR rounds of one sample point per lane
  const StepCtx& C;
  const EnvLds& L;
  float* krec;
  const int l;
  const LM& M;
  const LanePoints<LANE_ROUNDS(G, DM)>& P;
  float g[3];
  bool islink;
  unsigned long long active[LANE_ROUNDS(G, DM)];   // ballots of this sub-step's terrain contacts (round k, bit j = sample point k G + j)

  DEV ArmLane(const StepCtx& C_, const EnvLds& L_, float* krec_, int l_, const LM& M_, const LanePoints<LANE_ROUNDS(G, DM)>& P_)
      : C(C_), L(L_), krec(krec_), l(l_), M(M_), P(P_) {
    const float gon = (float)C.m->gravity_on;
    g[0] = C.sp.gravity[0] * gon; g[1] = C.sp.gravity[1] * gon; g[2] = C.sp.gravity[2] * gon;
    islink = l >= 1 && l <= NL;
  }

  // A. joint lanes: local rotation -> the link's exchange slot (free until its (IA, pA) go there); dof lanes: drive
  //    effort.  pos_tgt: LDS, POS-drive targets of this env step.  A group sync has to follow.
  DEV void joints() const {
    if (islink) {
      const int d = l - 1;
      float* rec = L.xch + l * XCH_STRIDE;
      float Rl[9];
      joint_local_rotation(M.tr, M.ax, L.dofb[d * DOF_STRIDE], Rl);
#pragma unroll
      for (int k = 0; k < 9; k++) rec[k] = Rl[k];
      rec[9] = L.dofb[d * DOF_STRIDE + 1];
    }
  }
  DEV void joints_and_drives(const float* pos_tgt) const {
    const float dt = C.sp.dt;
    joints();
    if (l < nd) {
      float* D = L.dofb + l * DOF_STRIDE;
      const float q = D[0], qd = D[1];
      float t0 = 0.0f, de = M.armature;
      const int mode = M.mode;
      if (mode == SHF_DOF_MODE_EFFORT) {
        t0 = D[5];
      } else if (mode == SHF_DOF_MODE_POS || mode == SHF_DOF_MODE_VEL) {
        float kp = mode == SHF_DOF_MODE_POS ? M.kp : 0.0f, kd = M.kd;
        const float tq = pos_tgt ? pos_tgt[l] : 0.0f, tv = 0.0f;
        const float est = fmaf(kp, tq - q, kd * (tv - qd));
        const float lim = M.effort;
        if (lim > 0.0f && fabsf(est) > lim) { const float sc = lim / fabsf(est); kp *= sc; kd *= sc; }
        const float bj = fmaf(dt, kp, kd);
        t0 = fmaf(kp, tq - q, fmaf(kd, tv, -(bj * qd)));
        de = fmaf(dt, bj, de);
      }
      const float jd = M.damping;
      if (jd > 0.0f) { t0 = fmaf(-jd, qd, t0); de = fmaf(dt, jd, de); }
      const float lo = M.lower, up = M.upper;
      const float viol = q < lo ? lo - q : (q > up ? up - q : 0.0f);
      if (viol != 0.0f) {
        const float bl = fmaf(dt, C.sp.limit_k, C.sp.limit_d);
        t0 = fmaf(C.sp.limit_k, viol, fmaf(-bl, qd, t0));
        de = fmaf(dt, bl, de);
      }
      D[2] = t0; D[3] = de;
    }
  }

  // B. chain lane: poses, velocities, motion subspaces and bias accelerations root -> tip.  A group sync has to follow.
  DEV void compose() const {
    const ShfModel* m = C.m;
    if (l == 0) {
      float Rc[9], pc[3] = {0.0f, 0.0f, 0.0f}, vc[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
      quat_to_mat(L.root + 3, Rc);
      pose_store(L.pose, Rc, pc, vc);
#pragma unroll
      for (int b = 1; b <= NL; b++) {
        const float* rec = L.xch + b * XCH_STRIDE;
        float* kr = krec + (b - 1) * KREC_STRIDE;
        float Rl[9], S[6], c[6];
#pragma unroll
        for (int j = 0; j < 9; j++) Rl[j] = rec[j];
        chain_compose_link(m->tpos[b], m->axis[b], Rl, rec[9], Rc, pc, vc, S, c);
        pose_store(L.pose + b * POSE_STRIDE, Rc, pc, vc);
#pragma unroll
        for (int j = 0; j < 6; j++) { kr[j] = S[j]; kr[6 + j] = c[j]; }
      }
    }
  }

  // C. link lanes: rigid inertia and bias force; the arm's terrain sample points (lane per point) and their fold
  DEV void inertia_and_points(BodyRegs& B, float mu_shape) {
    const float dt = C.sp.dt;
    PHASE_BEGIN();
    if (islink) {
      const float* pb = L.pose + l * POSE_STRIDE;
#pragma unroll
      for (int k = 0; k < 9; k++) B.Rw[k] = pb[k];
#pragma unroll
      for (int k = 0; k < 3; k++) B.p[k] = pb[9 + k];
#pragma unroll
      for (int k = 0; k < 6; k++) B.v[k] = pb[12 + k];
      body_inertia(M, B, C.mscale, l);
    }
    PHASE_MARK(2);
    const float kc = C.sp.contact_k, dc = C.sp.contact_d, veps = C.sp.friction_vel;
    const float beta = fmaf(kc, dt, dc);
    const float mu = 0.5f * (mu_shape + C.terr.t.friction);
    const ContactConsts K = {dt, {g[0], g[1], g[2]}, kc, beta, mu, veps, C.sp.max_depen_vel, C.sp.contact_offset};
    const int lane0 = (int)(threadIdx.x & 63u) - l;
#pragma unroll
    for (int k = 0; k < NR; k++) {
      const float* pb = L.pose + P.body(k) * POSE_STRIDE;
      float Rb[9], r[3], n[3], h, on = 0.0f;
#pragma unroll
      for (int j = 0; j < 9; j++) Rb[j] = pb[j];
      mv3(Rb, P.pos[k], r);
#pragma unroll
      for (int j = 0; j < 3; j++) r[j] += pb[9 + j];
      terrain_query<false>(C.terr, L.root[0] + r[0], L.root[1] + r[1], &h, n);
      const float phi = fmaf(L.root[2] + r[2] - h, n[2], -P.rad[k]);
      if (l + k * G < DM::NPC) {
        float* o = L.pt + (l + k * G) * PT_STRIDE;
        if (phi < K.offset) on = contact_point_response(K, pb, r, n, P.rad[k], phi, o);
        o[PT_ON] = on;
      }
      active[k] = (__ballot(on != 0.0f) >> lane0) & ((1ull << (G & 63)) - 1ull);
    }
    GROUP_SYNC();
    PHASE_MARK(3);
    if (islink) {
      const int i0 = M.pt0, i1 = i0 + M.npt;     // the link's points, ascending -- whatever round evaluated them
#pragma unroll
      for (int k = 0; k < NR; k++) {
        const int a0 = (i0 > k * G ? i0 : k * G) - k * G, a1 = (i1 < (k + 1) * G ? i1 : (k + 1) * G) - k * G;
        unsigned long long bits = a1 > a0 ? (active[k] >> a0) & ((1ull << (a1 - a0)) - 1ull) : 0ull;
        while (bits) {
          const int j = __builtin_ctzll(bits);
          bits &= bits - 1ull;
          contact_accumulate(L.pt + (k * G + a0 + j) * PT_STRIDE, dt, B);
        }
      }
    }
  }

  // C'. the same under the velocity-level solve (ShfSimParams.solver != SHF_SOLVER_COMPLIANT): the links' rigid inertia, and
  //     the sample points as candidate constraints only (substep<.., HARD>'s point loop: gap from rest_offset inside the contact
  //     offset) -- nothing is folded, the articulated-body solve runs free
  DEV void inertia_and_candidates(BodyRegs& B, float mu_shape) const {
    if (islink) {
      const float* pb = L.pose + l * POSE_STRIDE;
#pragma unroll
      for (int k = 0; k < 9; k++) B.Rw[k] = pb[k];
#pragma unroll
      for (int k = 0; k < 3; k++) B.p[k] = pb[9 + k];
#pragma unroll
      for (int k = 0; k < 6; k++) B.v[k] = pb[12 + k];
      body_inertia(M, B, C.mscale, l);
    }
    const float mu = 0.5f * (mu_shape + C.terr.t.friction);
    const float reach = C.sp.contact_offset + C.sp.rest_offset;
#pragma unroll
    for (int k = 0; k < NR; k++) {
      const float* pb = L.pose + P.body(k) * POSE_STRIDE;
      float Rb[9], r[3], n[3], h;
#pragma unroll
      for (int j = 0; j < 9; j++) Rb[j] = pb[j];
      mv3(Rb, P.pos[k], r);
#pragma unroll
      for (int j = 0; j < 3; j++) r[j] += pb[9 + j];
      terrain_query<false>(C.terr, L.root[0] + r[0], L.root[1] + r[1], &h, n);
      const float rad = P.rad[k];
      const float phi = fmaf(L.root[2] + r[2] - h, n[2], -rad);
      if (l + k * G < DM::NPC) {
        float* o = L.pt + (l + k * G) * PT_STRIDE;
        float on = 0.0f;
        if (phi < reach) {
          on = 1.0f;
#pragma unroll
          for (int j = 0; j < 3; j++) { o[PT_R + j] = fmaf(-rad, n[j], r[j]); o[PT_N + j] = n[j]; }
          o[PT_F] = phi - C.sp.rest_offset; o[PT_F + 1] = mu; o[PT_F + 2] = 0.0f; o[PT_CT] = 0.0f; o[PT_BN] = 0.0f;
        }
        o[PT_ON] = on;
      }
    }
  }

  // link lanes -> chain lane: (IA, pA) with every contact folded in.  A group sync has to follow.
  DEV void hand_over(const BodyRegs& B) const {
    if (islink) {
      float* o = L.xch + l * XCH_STRIDE;
#pragma unroll
      for (int k = 0; k < 21; k++) o[k] = B.IA[k];
#pragma unroll
      for (int k = 0; k < 6; k++) o[21 + k] = B.pA[k];
    }
  }

  // D. chain lane: ABA inward tip -> root, then outward root -> tip (the base does not move: a0 = -g).  A group sync
  //    has to follow.
  DEV void recursions() const {
    if (l == 0) {
      PHASE_BEGIN();
      float IAc[21], pAc[6];
#pragma unroll
      for (int b = NL; b >= 1; b--) {
        const float* o = L.xch + b * XCH_STRIDE;
        float* kr = krec + (b - 1) * KREC_STRIDE;
        ChainLink Kb;
        float IA[21], pA[6], pa[6];
#pragma unroll
        for (int k = 0; k < 21; k++) IA[k] = b == NL ? o[k] : o[k] + IAc[k];
#pragma unroll
        for (int k = 0; k < 6; k++) pA[k] = b == NL ? o[21 + k] : o[21 + k] + pAc[k];
#pragma unroll
        for (int k = 0; k < 6; k++) { Kb.S[k] = kr[k]; Kb.c[k] = kr[6 + k]; }
        chain_inward_link(Kb, IA, pA, L.dofb[(b - 1) * DOF_STRIDE + 3], L.dofb[(b - 1) * DOF_STRIDE + 2], pa);
#pragma unroll
        for (int k = 0; k < 6; k++) kr[12 + k] = Kb.U[k];
        kr[18] = Kb.invD; kr[19] = Kb.u;
#pragma unroll
        for (int k = 0; k < 21; k++) IAc[k] = IA[k];
#pragma unroll
        for (int k = 0; k < 6; k++) pAc[k] = pa[k];
      }
      PHASE_MARK(6);
      float a[6] = {0.0f, 0.0f, 0.0f, -g[0], -g[1], -g[2]};
#pragma unroll
      for (int k = 0; k < 6; k++) L.acc[k] = a[k];
#pragma unroll
      for (int b = 1; b <= NL; b++) {
        const float* kr = krec + (b - 1) * KREC_STRIDE;
        float ap[6];
#pragma unroll
        for (int i = 0; i < 6; i++) ap[i] = a[i] + kr[6 + i];
        float ua = kr[12] * ap[0];
#pragma unroll
        for (int j = 1; j < 6; j++) ua = fmaf(kr[12 + j], ap[j], ua);
        const float qdd = (kr[19] - ua) * kr[18];
        L.dofb[(b - 1) * DOF_STRIDE + 4] = qdd;
#pragma unroll
        for (int i = 0; i < 6; i++) { a[i] = fmaf(kr[i], qdd, ap[i]); L.acc[b * 6 + i] = a[i]; }
      }
    }
  }

  // E. net terrain-contact force per reported body (contact_out: LDS, rows of 3).  A group sync has to follow.
  DEV void point_forces(float* contact_out) const {
    const ShfModel* m = C.m;
#pragma unroll
    for (int k = 0; k < NR; k++)
      if ((active[k] >> l) & 1ull) contact_force_final(L.pt + (l + k * G) * PT_STRIDE, L.acc + m->dyn[P.body(k)] * 6, C.sp.dt);
    GROUP_SYNC();
    if (l < nb) {
      float f[3] = {0.0f, 0.0f, 0.0f};
      const int dl = m->dyn[l];
      const int i0 = m->pt_start[dl], i1 = i0 + m->pt_count[dl];
#pragma unroll
      for (int k = 0; k < NR; k++) {
        const int a0 = (i0 > k * G ? i0 : k * G) - k * G, a1 = (i1 < (k + 1) * G ? i1 : (k + 1) * G) - k * G;
        unsigned long long bits = a1 > a0 ? (active[k] >> a0) & ((1ull << (a1 - a0)) - 1ull) : 0ull;
        while (bits) {
          const int i = k * G + a0 + __builtin_ctzll(bits);
          bits &= bits - 1ull;
          if (m->pt_body[i] != l) continue;
          const float* o = L.pt + i * PT_STRIDE;
          f[0] += o[PT_F]; f[1] += o[PT_F + 1]; f[2] += o[PT_F + 2];
        }
      }
      contact_out[3 * l] = f[0]; contact_out[3 * l + 1] = f[1]; contact_out[3 * l + 2] = f[2];
    }
  }

  // semi-implicit Euler of the joints.  A group sync has to follow.
  DEV void integrate() const {
    if (l < nd) {
      float* D = L.dofb + l * DOF_STRIDE;
      const float vl = M.vel_limit;
      const float qd = rclampf(fmaf(C.sp.dt, D[4], D[1]), -vl, vl);
      D[1] = qd;
      D[0] = fmaf(C.sp.dt, qd, D[0]);
    }
  }
};

// One gym.simulate() for one env, all of it on the G lanes of its group (one wave).
template <int G, class DM, class SC, int NL>
DEV void arm_substep(const StepCtx& C, const EnvLds& L, float* krec, int l, const LaneModel& M, const LanePoints<LANE_ROUNDS(G, DM)>& P,
                     const float* pos_tgt, float mu_shape, float* contact_out, const BoxLane& BL) {
  static_assert(SC::NBX > 0, "serial-arm sub-step: compiled for a fixed scene");
  ArmLane<G, DM, NL> A(C, L, krec, l, M, P);
  BodyRegs B;
  PHASE_BEGIN();
  A.joints_and_drives(pos_tgt);
  boxes_pose<G>(C, L, l, B);          // box lanes: pose, velocity, inertia; ends with the group sync
  PHASE_MARK(0);
  A.compose();
  GROUP_SYNC();
  PHASE_MARK(1);
  A.inertia_and_points(B, mu_shape);
  PHASE_RESET();
  BoxMasks BM;
  boxes_contacts<G, SC, false>(C, L, l, B, mu_shape, A.g, BL, BM, 0);
  A.hand_over(B);
  GROUP_SYNC();
  PHASE_MARK(4);
  A.recursions();
  GROUP_SYNC();
  PHASE_MARK(8);
  if (contact_out) A.point_forces(contact_out);
  GROUP_SYNC();
  boxes_finish<G, SC>(C, L, l, B, contact_out, BL, BM, 0);
  PHASE_MARK(9);
  A.integrate();
  GROUP_SYNC();
  PHASE_MARK(10);
}
