// shf_chain_hard.h -- the chain-mapped fused A1 step with the velocity-level contact solve
// (ShfSimParams.solver == SHF_SOLVER_PGS): what gymapi.SimParams.physx configures in the reference
// (shifu/configs/env_config.py:50-58: solver_type, num_position_iterations = 8, num_velocity_iterations = 1, contact_offset,
// rest_offset, bounce_threshold_velocity, max_depenetration_velocity) and every gym.simulate runs under
// (examples/a1_conditional/a1_conditional.py:69, shifu/gym/isaac_gym.py:140).
//
// One sub-step at 32 lanes per env (two envs per wavefront):
//   A-C   as shf_chain.h: joint records, chain composition, rigid inertias            (dof / chain / body lanes)
//   P     sample points -> candidate constraints; the K <= 8 with the smallest gap      (point lanes, ballots)
//   E-G   the FREE articulated-body solve (contacts left out); the factors U, 1/D per link and the root's LDL^T stay in LDS
//   H1    response matrix W = J M^-1 J^T, one lane per column (contact j, axis k): impulse up the chain, root solve,
//         down every constrained chain                                                  (<= 24 column lanes)
//   H2    per contact: free velocity, targets, regularised diagonal block and its normal / tangential inverses   (owner lanes)
//   H3    projected Gauss-Seidel: contact after contact; the owner lane of contact c computes its new impulse, every
//         owner lane moves its contact's velocity by W[i][c] dp (one 3x3 block from LDS)
//   H4    the impulses as forces: one vector pass inward / root / outward               (body, chain, root lanes)
//         -- after the position iterations (poses) and after the velocity iterations (velocities)
//   H5    integration
//
// ARITHMETIC: the operations of oracle/shf_oracle.c (substep with hard = 1, hard_solve, hc_apply) in the same order on the
// same values -- only which lane executes them differs; tests/test_gpu_parity.py holds the kernel to the oracle bit for bit.
#pragma once
#include "shf_chain.h"

#define HCK 8           /* constraints one env's solve holds on this kernel (ShfSimParams.max_contacts <= HCK) */
#define HC_STRIDE 16    /* r[3] n[3] phi mu body rep p[3] bodyb repb .  (bodyb / repb: the other side of a self-contact, else -1) */
#define HC_R 0
#define HC_N 3
#define HC_PHI 6
#define HC_MU 7
#define HC_BODY 8
#define HC_REP 9
#define HC_P 10
#define HC_BODYB 13
#define HC_REPB 14
#define UF_STRIDE 8     /* U[6] invD . per link */
template <class CD>
struct HardTail {
  static constexpr int HC = 0, W = HCK * HC_STRIDE, UF = W + HCK * HCK * 9, PHI = UF + CD::ND * UF_STRIDE, NEVP = (CD::NEV + 3) & ~3,
                       END = PHI + NEVP + SHF_MAX_SELF_CONTACTS;
};

// point velocity of the spatial velocity v6 (about O) at r
DEV void hard_point(const float* v6, const float* r, float* o) {
  float t[3];
  cross3(v6, r, t);
#pragma unroll
  for (int k = 0; k < 3; k++) o[k] = v6[3 + k] + t[k];
}
DEV void mat3_inv_spd(const float* A, float* Ai) {
  const float c00 = fmaf(A[4], A[8], -(A[5] * A[7])), c01 = fmaf(A[5], A[6], -(A[3] * A[8])), c02 = fmaf(A[3], A[7], -(A[4] * A[6]));
  const float id = rcp_spec(fmaf(A[0], c00, fmaf(A[1], c01, A[2] * c02)));
  Ai[0] = c00 * id; Ai[1] = fmaf(A[2], A[7], -(A[1] * A[8])) * id; Ai[2] = fmaf(A[1], A[5], -(A[2] * A[4])) * id;
  Ai[3] = c01 * id; Ai[4] = fmaf(A[0], A[8], -(A[2] * A[6])) * id; Ai[5] = fmaf(A[2], A[3], -(A[0] * A[5])) * id;
  Ai[6] = c02 * id; Ai[7] = fmaf(A[1], A[6], -(A[0] * A[7])) * id; Ai[8] = fmaf(A[0], A[4], -(A[1] * A[3])) * id;
}
// the root's LDL^T factors in LDS: L below the diagonal row by row (15), then 1/D (6)
DEV void root_factors_store(const Ldlt6& F, float* o) {
  int q = 0;
#pragma unroll
  for (int i = 1; i < 6; i++)
#pragma unroll
    for (int j = 0; j < i; j++) o[q++] = F.Lm[i][j];
#pragma unroll
  for (int j = 0; j < 6; j++) o[15 + j] = F.iD[j];
}
DEV void root_factors_apply(const float* o, const float* pA, float* x) {   // ldlt_substitute6 from the LDS copy
  float y[6];
#pragma unroll
  for (int i = 0; i < 6; i++) {
    float v = -pA[i];
#pragma unroll
    for (int k = 0; k < i; k++) v = fmaf(-o[i * (i - 1) / 2 + k], y[k], v);
    y[i] = v;
  }
#pragma unroll
  for (int i = 0; i < 6; i++) y[i] = y[i] * o[15 + i];
#pragma unroll
  for (int i = 5; i >= 0; i--) {
    float v = y[i];
#pragma unroll
    for (int k = i + 1; k < 6; k++) v = fmaf(-o[k * (k - 1) / 2 + i], x[k], v);
    x[i] = v;
  }
}

// Response of the articulation to the impulse e at r on its moving body bs (0: the root): the root's velocity change and the
// joint terms of the links on bs's chain up to bs (oracle: hc_impulse) ...
template <class CD>
struct HardResp { int cs, ks; float ub[CD::NLK], dv0[6]; };
template <class CD>
DEV void hard_impulse(const ChainLds& L, const float* tail, int bs, const float* r, const float* e, HardResp<CD>& q) {
  constexpr int NLK = CD::NLK;
  typedef HardTail<CD> T;
  q.cs = bs > 0 ? (bs - 1) / (NLK + 1) : 0;
  q.ks = bs > 0 ? (bs - 1) % (NLK + 1) : -1;
  float p6[6], t[3];
  cross3(r, e, t);
#pragma unroll
  for (int k = 0; k < 3; k++) { p6[k] = -t[k]; p6[3 + k] = -e[k]; }
#pragma unroll
  for (int k = NLK - 1; k >= 0; k--) {
    q.ub[k] = 0.0f;
    if (k <= q.ks) {
      const int li = q.cs * NLK + k;
      const float* rec = L.jrec + li * JREC_STRIDE;
      const float* uf = tail + T::UF + li * UF_STRIDE;
      float sp = rec[JREC_S] * p6[0];
#pragma unroll
      for (int j = 1; j < 6; j++) sp = fmaf(rec[JREC_S + j], p6[j], sp);
      q.ub[k] = -sp;
      const float tt = q.ub[k] * uf[6];
#pragma unroll
      for (int j = 0; j < 6; j++) p6[j] = fmaf(uf[j], tt, p6[j]);
    }
  }
  root_factors_apply(L.xroot, p6, q.dv0);
}
// ... and the velocity change of the point r of moving body bt under it (oracle: hc_velocity); bt < 0: nothing moves
template <class CD>
DEV void hard_velocity(const ChainLds& L, const float* tail, const HardResp<CD>& q, int bt, const float* r, float* vel) {
  constexpr int NLK = CD::NLK;
  typedef HardTail<CD> T;
  if (bt < 0) { vel[0] = 0.0f; vel[1] = 0.0f; vel[2] = 0.0f; return; }
  const int ct = bt > 0 ? (bt - 1) / (NLK + 1) : 0, kt = bt > 0 ? (bt - 1) % (NLK + 1) : -1;
  float dv[6];
#pragma unroll
  for (int j = 0; j < 6; j++) dv[j] = q.dv0[j];
#pragma unroll
  for (int k = 0; k < NLK; k++) {
    if (k <= kt) {
      const int li = ct * NLK + k;
      const float* rec = L.jrec + li * JREC_STRIDE;
      const float* uf = tail + T::UF + li * UF_STRIDE;
      const float ubk = (ct == q.cs && k <= q.ks) ? q.ub[k] : 0.0f;
      float ua = uf[0] * dv[0];
#pragma unroll
      for (int j = 1; j < 6; j++) ua = fmaf(uf[j], dv[j], ua);
      const float dq = (ubk - ua) * uf[6];
#pragma unroll
      for (int j = 0; j < 6; j++) dv[j] = fmaf(rec[JREC_S + j], dq, dv[j]);
    }
  }
  hard_point(dv, r, vel);
}

// Per-lane state of the solve that outlives a phase.
struct HardOwner {     // the owner lane of contact c
  float n[3], mu, u[3], p[3], tgt, tgt_v, Wn[3], iwnn, Ti[9], rt;
};

// H4: the impulses in the constraint records as forces on their bodies -> what they add to the accelerations: joint
// accelerations to dofb[.][slot], the root's to `ac0` (root lane).  oracle: hc_apply.
template <class CD>
DEV void chain_hard_apply(const ShfModel* m, const ChainLds& L, float* tail, int l, int K, float idt, bool isbody_h0, bool islink, bool isroot,
                          bool ischain, int ci, int lb, int myb, int slot, float* ac0) {
  constexpr int NCH = CD::NCH, NLK = CD::NLK;
  typedef HardTail<CD> T;
  float pcr[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
  if (isbody_h0) {
    for (int c = 0; c < K; c++) {
      const float* h = tail + T::HC + c * HC_STRIDE;
      const bool ona = __float_as_int(h[HC_BODY]) == myb, onb = __float_as_int(h[HC_BODYB]) == myb;
      if (!ona && !onb) continue;
      const float r[3] = {h[HC_R], h[HC_R + 1], h[HC_R + 2]};
      const float f[3] = {h[HC_P] * idt, h[HC_P + 1] * idt, h[HC_P + 2] * idt};
      float t[3];
      cross3(r, f, t);
      if (ona) {
#pragma unroll
        for (int k = 0; k < 3; k++) { pcr[k] -= t[k]; pcr[3 + k] -= f[k]; }
      } else {
#pragma unroll
        for (int k = 0; k < 3; k++) { pcr[k] += t[k]; pcr[3 + k] += f[k]; }
      }
    }
    if (islink) {
      float* o = L.xch + lb * XCH_STRIDE + 21;
#pragma unroll
      for (int j = 0; j < 6; j++) o[j] = pcr[j];
    }
  }
  GROUP_SYNC();
  float ucl[NLK];
  if (ischain) {
    float pl[6];
#pragma unroll
    for (int k = NLK - 1; k >= 0; k--) {
      const int li = ci * NLK + k;
      const float* o = L.xch + li * XCH_STRIDE + 21;
      const float* rec = L.jrec + li * JREC_STRIDE;
      const float* uf = tail + T::UF + li * UF_STRIDE;
      if (k == NLK - 1) {
#pragma unroll
        for (int j = 0; j < 6; j++) pl[j] = o[j];
      } else {
#pragma unroll
        for (int j = 0; j < 6; j++) pl[j] = o[j] + pl[j];
      }
      float sp = rec[JREC_S] * pl[0];
#pragma unroll
      for (int j = 1; j < 6; j++) sp = fmaf(rec[JREC_S + j], pl[j], sp);
      ucl[k] = -sp;
      const float tt = ucl[k] * uf[6];
#pragma unroll
      for (int j = 0; j < 6; j++) pl[j] = fmaf(uf[j], tt, pl[j]);
    }
    float* o = L.xch + (ci * NLK) * XCH_STRIDE + 21;
#pragma unroll
    for (int j = 0; j < 6; j++) o[j] = pl[j];
  }
  GROUP_SYNC();
  if (isroot) {
#pragma unroll
    for (int j = 0; j < 6; j++) {
      float v = pcr[j];
#pragma unroll
      for (int c = 0; c < NCH; c++) v += L.xch[(c * NLK) * XCH_STRIDE + 21 + j];
      pcr[j] = v;
    }
    root_factors_apply(L.xroot, pcr, ac0);
#pragma unroll
    for (int j = 0; j < 6; j++) L.acc[j] = ac0[j];
  }
  GROUP_SYNC();
  if (ischain) {
    float ac[6];
#pragma unroll
    for (int j = 0; j < 6; j++) ac[j] = L.acc[j];
#pragma unroll
    for (int k = 0; k < NLK; k++) {
      const int li = ci * NLK + k;
      const float* rec = L.jrec + li * JREC_STRIDE;
      const float* uf = tail + T::UF + li * UF_STRIDE;
      float ua = uf[0] * ac[0];
#pragma unroll
      for (int j = 1; j < 6; j++) ua = fmaf(uf[j], ac[j], ua);
      const float qc = (ucl[k] - ua) * uf[6];
#pragma unroll
      for (int j = 0; j < 6; j++) ac[j] = fmaf(rec[JREC_S + j], qc, ac[j]);
      L.dofb[li * DOF_STRIDE + slot] = qc;
    }
  }
  GROUP_SYNC();
}

// One gym.simulate() for one env under the velocity-level contact solve; lane roles as chain_substep at 32 lanes per env.
template <int G, class CD, bool TW, bool SELF>
DEV void chain_substep_hard(const StepCtx& C, const ChainLds& L, int l, DofLane& X, const ChainPoints<(CD::NEV + G - 1) / G>& P,
                            const RowLane& RL, const float* fext, float mu_shape, float* contact_out) {
  static_assert(G == 32, "the velocity-level solve is written for two envs per wavefront");
  constexpr int NCH = CD::NCH, NLK = CD::NLK, NB = CD::NB, ND = CD::ND, NR = (CD::NEV + G - 1) / G;
  typedef HardTail<CD> T;
  static_assert(T::END <= CD::NPC * PT_STRIDE, "the solve's LDS fits the contact-slot region");
  static_assert(3 * HCK <= G && ND < 16, "a lane per column of the response matrix");
  const ShfModel* m = C.m;
  const float dt = C.sp.dt, idt = 1.0f / dt;
  const float gon = (float)m->gravity_on;
  const float g[3] = {C.sp.gravity[0] * gon, C.sp.gravity[1] * gon, C.sp.gravity[2] * gon};
  float* tail = L.pt;
  const int lb = l & 15;
  const int half = (l >> 4) & 1;
  const bool isdof = l < ND, isroot = l == ND;
  const bool ischain = (l & 7) == 0 && (l >> 3) < NCH;
  const int ci = l >> 3;
  const bool isbody = lb <= ND;
  const bool islink = isbody && lb < ND;
  const int myb = lb < ND ? CD::body(lb / NLK, lb % NLK) : 0;
  const int lane0 = (int)(threadIdx.x & 63u) - l;
  PHASE_BEGIN();

  // ---- A. dof lanes: drive effort and the joint's local rotation -> joint record
  if (isdof) {
    float* rec = L.jrec + l * JREC_STRIDE;
    float Rl[9], t0, de;
    chain_dof_effort(C, l, X.q, X.qd, X.tau, &t0, &de, X.tq, X.tv);
    joint_local_rotation(m->trot[myb], m->axis[myb], X.q, Rl);
#pragma unroll
    for (int k = 0; k < 9; k++) rec[k] = Rl[k];
    rec[JREC_QD] = X.qd; rec[JREC_T0] = t0; rec[JREC_DE] = de;
  }
  GROUP_SYNC();
  PHASE_MARK(0);

  // ---- B. chain lanes: poses, velocities, motion subspaces root -> tip
  if (ischain || isroot) {
    float Rc[9], pc[3] = {0.0f, 0.0f, 0.0f}, vc[6];
    quat_to_mat(L.root + 3, Rc);
#pragma unroll
    for (int k = 0; k < 3; k++) { vc[k] = L.root[10 + k]; vc[3 + k] = L.root[7 + k]; }
    if (isroot && !ischain) {
      pose_store(L.pose, Rc, pc, vc);
    } else {
      const int b0 = CD::body(ci, 0);
#pragma unroll
      for (int k = 0; k < NLK; k++) {
        const int b = b0 + k;
        float* rec = L.jrec + (ci * NLK + k) * JREC_STRIDE;
        float Rl[9], Sx[6], cx[6];
#pragma unroll
        for (int j = 0; j < 9; j++) Rl[j] = rec[j];
        chain_compose_link(m->tpos[b], m->axis[b], Rl, rec[JREC_QD], Rc, pc, vc, Sx, cx);
#pragma unroll
        for (int j = 0; j < 6; j++) { rec[JREC_S + j] = Sx[j]; rec[JREC_C + j] = cx[j]; }
        pose_store(L.pose + b * POSE_STRIDE, Rc, pc, vc);
      }
      const int b = b0 + NLK;
      chain_kin_weld(m->tpos[b], m->trot[b], Rc, pc);
      pose_store(L.pose + b * POSE_STRIDE, Rc, pc, vc);
    }
  }
  GROUP_SYNC();
  PHASE_MARK(1);

  // ---- C. body lanes: rigid inertia and bias force of their moving body, external forces -> exchange slots (no contact folds)
  if (isbody) {
    float IA[21], pA[6];
    const float* pb = L.pose + myb * POSE_STRIDE;
    float Rb[9], pp[3], vb[6];
#pragma unroll
    for (int k = 0; k < 9; k++) Rb[k] = pb[k];
#pragma unroll
    for (int k = 0; k < 3; k++) pp[k] = pb[9 + k];
#pragma unroll
    for (int k = 0; k < 6; k++) vb[k] = pb[12 + k];
    rigid_inertia_p(m->mass[myb], m->com[myb], m->inertia[myb], Rb, pp, vb, IA, pA);
    if (fext) {
      const float F[3] = {fext[3 * myb], fext[3 * myb + 1], fext[3 * myb + 2]};
      chain_ext_force(F, m->com[myb], Rb, pp, pA);
      if (islink && (lb % NLK) == NLK - 1) {
        const int bw = myb + 1;
        const float* pw = L.pose + bw * POSE_STRIDE;
        float Rw[9], pq[3];
#pragma unroll
        for (int k = 0; k < 9; k++) Rw[k] = pw[k];
#pragma unroll
        for (int k = 0; k < 3; k++) pq[k] = pw[9 + k];
        const float Fw[3] = {fext[3 * bw], fext[3 * bw + 1], fext[3 * bw + 2]};
        chain_ext_force(Fw, m->com[bw], Rw, pq, pA);
      }
    }
    float* o = (islink ? L.xch + lb * XCH_STRIDE : L.xroot);
    if (half == 0) {
#pragma unroll
      for (int j = 0; j < 11; j++) o[j] = IA[j];
    } else {
#pragma unroll
      for (int j = 11; j < 21; j++) o[j] = IA[j];
#pragma unroll
      for (int j = 0; j < 6; j++) o[21 + j] = pA[j];
    }
  }
  PHASE_MARK(2);

  // ---- P. sample points -> candidate constraints (gap from rest_offset inside the contact offset), evaluation-slot order
  const float rest = C.sp.rest_offset, offs = C.sp.contact_offset + rest;
  const float mu = 0.5f * (mu_shape + C.terr.t.friction);
  const int kmax = C.sp.max_contacts > 0 ? (C.sp.max_contacts < HCK ? C.sp.max_contacts : HCK) : HCK;
  SlotBits act = {{0ull, 0ull}};
  int K;
  {
    float r[NR][3], nn[NR][3], ph[NR], gx[NR], gy[NR], dz[NR];
    bool near[NR], cand[NR];
    const float nzmin = TW ? 0.0f : C.terr.t.nz_min;
#pragma unroll
    for (int k = 0; k < NR; k++) {
      const float* pb = L.pose + P.body[k] * POSE_STRIDE;
      float Rb[9], h = 0.0f;
#pragma unroll
      for (int j = 0; j < 9; j++) Rb[j] = pb[j];
      mv3(Rb, P.pos[k], r[k]);
#pragma unroll
      for (int j = 0; j < 3; j++) r[k][j] += pb[9 + j];
      if constexpr (TW) {
        gx[k] = 0.0f; gy[k] = 0.0f; dz[k] = 0.0f;
        near[k] = P.idx[k] >= 0;
      } else {
        terrain_height_gradient(C.terr, L.root[0] + r[k][0], L.root[1] + r[k][1], &h, &gx[k], &gy[k]);
        dz[k] = L.root[2] + r[k][2] - h;
        near[k] = P.idx[k] >= 0 && !(dz[k] * nzmin >= P.thr[k]);
      }
    }
#pragma unroll
    for (int k = 0; k < NR; k++) {
      cand[k] = false; ph[k] = 0.0f;
      if (__ballot(near[k]) == 0ull) continue;
      if (near[k]) {
        float phi;
        if constexpr (TW) {
          float h;
          terrain_query<true>(C.terr, L.root[0] + r[k][0], L.root[1] + r[k][1], &h, nn[k]);
          phi = fmaf(L.root[2] + r[k][2] - h, nn[k][2], -P.rad[k]);
        } else {
          terrain_normal_from_gradient(C.terr, gx[k], gy[k], nn[k]);
          phi = fmaf(dz[k], nn[k][2], -P.rad[k]);
        }
        if (phi < offs) {
          cand[k] = true;
          ph[k] = phi - rest;
#pragma unroll
          for (int j = 0; j < 3; j++) r[k][j] = fmaf(-P.rad[k], nn[k][j], r[k][j]);
        }
      }
      const unsigned long long bits = (__ballot(cand[k]) >> lane0) & ((1ull << G) - 1ull);
      if (k * G < 64) act.w[0] |= bits << ((k * G) & 63); else act.w[1] |= bits << ((k * G - 64) & 63);
    }
    // self-collision (ShfModel.self_collide): capsule pairs, the active ones (<= SHF_MAX_SELF_CONTACTS, pair order) in the slots
    // behind the contact-slot region; candidates after the sample points', in that order.  Lane k < nself speaks for self-contact k.
    int nself = 0;
    if constexpr (SELF) nself = self_contacts_eval<G>(C, L, l, CD::NPC, mu_shape);
    if constexpr (SELF) GROUP_SYNC();
    const float* sslot = L.pt + (CD::NPC + (l < nself ? l : 0)) * PT_STRIDE;
    bool scand = SELF && l < nself;
    const float sph = scand ? sslot[PT_F] : 0.0f;
    const int npts = __popcll(act.w[0]) + __popcll(act.w[1]);
    int total = npts + nself;
    unsigned smask = nself >= 32 ? ~0u : ((1u << nself) - 1u);      // selected self-contacts of this env
    if (__ballot(total > kmax) != 0ull) {
      // more candidates than the solve holds (somewhere in this wavefront): keep the kmax with the smallest gap, ties by
      // candidate order (slots, then self-contacts)
      float* phis = tail + T::PHI;
#pragma unroll
      for (int k = 0; k < NR; k++)
        if (cand[k]) phis[l + k * G] = ph[k];
      if (scand) phis[T::NEVP + l] = sph;
      GROUP_SYNC();
      if (total > kmax) {
#pragma unroll
        for (int k = 0; k < NR; k++) {
          if (!cand[k]) continue;
          const int s = l + k * G;
          int rank = 0;
          for (int wd = 0; wd < 2; wd++) {
            unsigned long long bits = act.w[wd];
            while (bits) {
              const int j = __builtin_ctzll(bits) + 64 * wd;
              bits &= bits - 1ull;
              const float pj = phis[j];
              rank += (pj < ph[k] || (pj == ph[k] && j < s)) ? 1 : 0;
            }
          }
          for (int j = 0; j < nself; j++) rank += phis[T::NEVP + j] < ph[k] ? 1 : 0;
          cand[k] = rank < kmax;
        }
        if (scand) {
          int rank = 0;
          for (int wd = 0; wd < 2; wd++) {
            unsigned long long bits = act.w[wd];
            while (bits) {
              const int j = __builtin_ctzll(bits) + 64 * wd;
              bits &= bits - 1ull;
              rank += phis[j] <= sph ? 1 : 0;
            }
          }
          for (int j = 0; j < nself; j++) { const float pj = phis[T::NEVP + j]; rank += (pj < sph || (pj == sph && j < l)) ? 1 : 0; }
          scand = rank < kmax;
        }
        if (l == 0 && C.dropped) *C.dropped += total - kmax;
      }
      act.w[0] = 0ull; act.w[1] = 0ull;
#pragma unroll
      for (int k = 0; k < NR; k++) {
        const unsigned long long bits = (__ballot(cand[k]) >> lane0) & ((1ull << G) - 1ull);
        if (k * G < 64) act.w[0] |= bits << ((k * G) & 63); else act.w[1] |= bits << ((k * G - 64) & 63);
      }
      smask = (unsigned)((__ballot(scand) >> lane0) & ((1ull << G) - 1ull));
      total = total > kmax ? kmax : total;
      GROUP_SYNC();
    }
    K = total;
    const int nps = __popcll(act.w[0]) + __popcll(act.w[1]);        // selected sample points: the self-contacts follow them
    if (SELF && scand) {
      float* h = tail + T::HC + (nps + __popc(smask & ((1u << l) - 1u))) * HC_STRIDE;
      const int pr = (int)sslot[PT_ON] - 1;
      const int ba = m->cap_body[m->pair_a[pr]], bb = m->cap_body[m->pair_b[pr]];
#pragma unroll
      for (int j = 0; j < 3; j++) { h[HC_R + j] = sslot[PT_R + j]; h[HC_N + j] = sslot[PT_N + j]; }
      h[HC_PHI] = sph; h[HC_MU] = sslot[PT_F + 1];
      h[HC_BODY] = __int_as_float(m->dyn[ba]); h[HC_REP] = __int_as_float(ba);
      h[HC_BODYB] = __int_as_float(m->dyn[bb]); h[HC_REPB] = __int_as_float(bb);
    }
#pragma unroll
    for (int k = 0; k < NR; k++) {
      if (!cand[k]) continue;
      const int s = l + k * G;
      const int idx = s < 64 ? __popcll(act.w[0] & ((1ull << s) - 1ull)) : __popcll(act.w[0]) + __popcll(act.w[1] & ((1ull << (s - 64)) - 1ull));
      float* h = tail + T::HC + idx * HC_STRIDE;
#pragma unroll
      for (int j = 0; j < 3; j++) { h[HC_R + j] = r[k][j]; h[HC_N + j] = nn[k][j]; }
      h[HC_PHI] = ph[k]; h[HC_MU] = mu;
      h[HC_BODY] = __int_as_float(m->dyn[P.body[k]]); h[HC_REP] = __int_as_float(P.body[k]);
      h[HC_BODYB] = __int_as_float(-1); h[HC_REPB] = __int_as_float(-1);
    }
  }
  GROUP_SYNC();
  PHASE_MARK(3);

  // ---- E. free inward pass tip -> root, row-parallel (chain_substep E); U and 1/D of every link stay in LDS for the solve
  float uk[NLK];   // the free solve's joint terms, for the outward pass (chain lanes)
  if (RL.on) {
    float Ic[6], pc = 0.0f;
    float* sx = L.acc + RL.c * 12;
#pragma unroll
    for (int k = NLK - 1; k >= 0; k--) {
      const float* o = L.xch + (RL.c * NLK + k) * XCH_STRIDE;
      const float* rec = L.jrec + (RL.c * NLK + k) * JREC_STRIDE;
      float row[6], S[6], cc[6], plg[6], Wg[6], Ug[6];
#pragma unroll
      for (int j = 0; j < 6; j++) row[j] = o[RL.off[j]];
      float pl = o[21 + RL.i];
#pragma unroll
      for (int j = 0; j < 6; j++) { S[j] = rec[JREC_S + j]; cc[j] = rec[JREC_C + j]; }
      const float dex = rec[JREC_DE], tau0 = rec[JREC_T0];
      if (k < NLK - 1) {
#pragma unroll
        for (int j = 0; j < 6; j++) row[j] += Ic[j];
        pl += pc;
      }
      float Ui = row[0] * S[0];
#pragma unroll
      for (int j = 1; j < 6; j++) Ui = fmaf(row[j], S[j], Ui);
      sx[RL.i] = Ui; sx[6 + RL.i] = pl;
      GROUP_SYNC();
#pragma unroll
      for (int j = 0; j < 6; j++) { Ug[j] = sx[j]; plg[j] = sx[6 + j]; }
      GROUP_SYNC();
      float D = S[0] * Ug[0];
#pragma unroll
      for (int j = 1; j < 6; j++) D = fmaf(S[j], Ug[j], D);
      D += dex;
      float sp = S[0] * plg[0];
#pragma unroll
      for (int j = 1; j < 6; j++) sp = fmaf(S[j], plg[j], sp);
      const float invD = rcp_spec(D);
      uk[k] = tau0 - sp;
#pragma unroll
      for (int j = 0; j < 6; j++) Wg[j] = Ug[j] * invD;
      const float Wi = Ui * invD;
#pragma unroll
      for (int j = 0; j < 6; j++) row[j] = j < RL.i ? fmaf(-Ug[j], Wi, row[j]) : fmaf(-Ui, Wg[j], row[j]);
      float acc = row[0] * cc[0];
#pragma unroll
      for (int j = 1; j < 6; j++) acc = fmaf(row[j], cc[j], acc);
      pc = fmaf(Wi, uk[k], pl + acc);
#pragma unroll
      for (int j = 0; j < 6; j++) Ic[j] = row[j];
      float* uf = tail + T::UF + (RL.c * NLK + k) * UF_STRIDE;
      uf[RL.i] = Ui;
      if (RL.i == 0) uf[6] = invD;
    }
    float* o = L.xch + (RL.c * NLK) * XCH_STRIDE;
#pragma unroll
    for (int j = 0; j < 6; j++)
      if (j >= RL.i) o[RL.off[j]] = Ic[j];
    o[21 + RL.i] = pc;
  }
  GROUP_SYNC();
  PHASE_MARK(6);

  // ---- F. root: sum with the chains, LDL^T, free acceleration; factors, a0 and the root's velocity rate to LDS
  for (int j = l; j < 27; j += G) {
    float v = L.xroot[j];
#pragma unroll
    for (int c = 0; c < NCH; c++) v += L.xch[(c * NLK) * XCH_STRIDE + j];
    L.xroot[j] = v;
  }
  GROUP_SYNC();
  float a0[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
  if (isroot) {
    float IA[21], pA[6];
#pragma unroll
    for (int j = 0; j < 21; j++) IA[j] = L.xroot[j];
#pragma unroll
    for (int j = 0; j < 6; j++) pA[j] = L.xroot[21 + j];
    Ldlt6 F;
    ldlt_factor6(IA, F);
    ldlt_substitute6(F, pA, a0);
    root_factors_store(F, L.xroot);
    const float ang[3] = {L.root[10], L.root[11], L.root[12]}, lin[3] = {L.root[7], L.root[8], L.root[9]};
    float wxv[3];
    cross3(ang, lin, wxv);
#pragma unroll
    for (int k = 0; k < 3; k++) { L.acc[k] = a0[k]; L.acc[3 + k] = (a0[3 + k] + g[k]) + wxv[k]; }
#pragma unroll
    for (int j = 0; j < 6; j++) L.xroot[21 + j] = a0[j];
  }
  GROUP_SYNC();
  PHASE_MARK(7);

  // ---- G. chain lanes: free outward pass; joint accelerations to the dof block, velocity rates Dl of the links to acc
  if (ischain) {
    float ap[6], dl[6];
#pragma unroll
    for (int j = 0; j < 6; j++) { ap[j] = L.xroot[21 + j]; dl[j] = L.acc[j]; }
    const int b0 = CD::body(ci, 0);
#pragma unroll
    for (int k = 0; k < NLK; k++) {
      const int li = ci * NLK + k;
      const float* rec = L.jrec + li * JREC_STRIDE;
      const float* uf = tail + T::UF + li * UF_STRIDE;
      float Sk[6];
#pragma unroll
      for (int j = 0; j < 6; j++) { Sk[j] = rec[JREC_S + j]; ap[j] = ap[j] + rec[JREC_C + j]; }
      float ua = uf[0] * ap[0];
#pragma unroll
      for (int j = 1; j < 6; j++) ua = fmaf(uf[j], ap[j], ua);
      const float qdd = (uk[k] - ua) * uf[6];
      float* o = L.acc + (b0 + k) * 6;
#pragma unroll
      for (int j = 0; j < 6; j++) { ap[j] = fmaf(Sk[j], qdd, ap[j]); dl[j] = fmaf(Sk[j], qdd, dl[j]); o[j] = dl[j]; }
      L.dofb[li * DOF_STRIDE + 4] = qdd;
    }
  }
  GROUP_SYNC();
  PHASE_MARK(8);

  // ---- H. the contact solve.  Loops over contacts run while any env of the wavefront has one left (wave-uniform bounds)
  float ac0p[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f}, ac0v[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
  const int npos = C.sp.pos_iters > 0 ? C.sp.pos_iters : 0, nvel = C.sp.vel_iters > 0 ? C.sp.vel_iters : 0;
  if (__ballot(K > 0) != 0ull) {
    // H1. column lanes: contact j = l / 3, axis l % 3
    {
      const int j = (l * 11) >> 5, ax = l - 3 * j;
      const bool col = l < 3 * K;
      const float* hj = tail + T::HC + (col ? j : 0) * HC_STRIDE;
      const float rj[3] = {hj[HC_R], hj[HC_R + 1], hj[HC_R + 2]};
      const int bsa = col ? __float_as_int(hj[HC_BODY]) : 0;
      const int bsb = (SELF && col) ? __float_as_int(hj[HC_BODYB]) : -1;
      const float e[3] = {ax == 0 ? 1.0f : 0.0f, ax == 1 ? 1.0f : 0.0f, ax == 2 ? 1.0f : 0.0f};
      HardResp<CD> qa, qb;
      hard_impulse<CD>(L, tail, bsa, rj, e, qa);
      if constexpr (SELF) {
        if (__ballot(bsb >= 0) != 0ull) hard_impulse<CD>(L, tail, bsb >= 0 ? bsb : 0, rj, e, qb);
      }
      for (int i = 0; i < HCK; i++) {
        if (__ballot(i < K) == 0ull) break;
        if (!(col && i < K)) continue;
        const float* hi = tail + T::HC + i * HC_STRIDE;
        const float ri[3] = {hi[HC_R], hi[HC_R + 1], hi[HC_R + 2]};
        const int bta = __float_as_int(hi[HC_BODY]);
        float aa[3], ab[3] = {0.0f, 0.0f, 0.0f}, ba[3] = {0.0f, 0.0f, 0.0f}, bb[3] = {0.0f, 0.0f, 0.0f};
        hard_velocity<CD>(L, tail, qa, bta, ri, aa);
        if constexpr (SELF) {
          const int btb = __float_as_int(hi[HC_BODYB]);
          hard_velocity<CD>(L, tail, qa, btb, ri, ab);
          if (bsb >= 0) {
            hard_velocity<CD>(L, tail, qb, bta, ri, ba);
            hard_velocity<CD>(L, tail, qb, btb, ri, bb);
          }
        }
        float* Wb = tail + T::W + (j * HCK + i) * 9 + ax;   // block (i, j), column ax
#pragma unroll
        for (int r = 0; r < 3; r++) Wb[3 * r] = (aa[r] - ab[r]) - (ba[r] - bb[r]);
      }
    }
    GROUP_SYNC();

    // H2. owner lanes: free velocity, targets, the regularised diagonal block and its inverses
    HardOwner O;
    const bool own = l < K;
    {
      const float* h = tail + T::HC + (own ? l : 0) * HC_STRIDE;
      const float r[3] = {h[HC_R], h[HC_R + 1], h[HC_R + 2]};
#pragma unroll
      for (int k = 0; k < 3; k++) { O.n[k] = h[HC_N + k]; O.p[k] = 0.0f; }
      O.mu = h[HC_MU];
      const int b = own ? __float_as_int(h[HC_BODY]) : 0;
      const float* pb = L.pose + b * POSE_STRIDE + 12;
      const float* D = L.acc + b * 6;
      float v[6], v6[6], vs[3], vf[3], vsb[3] = {0.0f, 0.0f, 0.0f}, vfb[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int k = 0; k < 6; k++) { v[k] = pb[k]; v6[k] = fmaf(dt, D[k], v[k]); }
      hard_point(v, r, vs);
      hard_point(v6, r, vf);
      if constexpr (SELF) {
        const int b2 = own ? __float_as_int(h[HC_BODYB]) : -1;
        if (b2 >= 0) {
          const float* pb2 = L.pose + b2 * POSE_STRIDE + 12;
          const float* D2 = L.acc + b2 * 6;
#pragma unroll
          for (int k = 0; k < 6; k++) { v[k] = pb2[k]; v6[k] = fmaf(dt, D2[k], v[k]); }
          hard_point(v, r, vsb);
          hard_point(v6, r, vfb);
        }
      }
#pragma unroll
      for (int k = 0; k < 3; k++) { O.u[k] = vf[k] - vfb[k]; vs[k] = vs[k] - vsb[k]; }
      const float phi = h[HC_PHI];
      const float erp = C.sp.erp > 0.0f ? C.sp.erp : 0.2f;
      float tg = phi >= 0.0f ? -(phi * idt) : rminf(erp * -(phi) * idt, C.sp.max_depen_vel);
      float tv = phi >= 0.0f ? tg : 0.0f;
      const float vn0 = dot3(O.n, vs);
      if (C.sp.restitution > 0.0f && vn0 < -C.sp.bounce_threshold) { tg = rmaxf(tg, -(C.sp.restitution * vn0)); tv = rmaxf(tv, -(C.sp.restitution * vn0)); }
      O.tgt = tg; O.tgt_v = tv;
      float* Wd = tail + T::W + ((own ? l : 0) * HCK + (own ? l : 0)) * 9;
      float A[9];
#pragma unroll
      for (int k = 0; k < 9; k++) A[k] = Wd[k];
      const float cfm = 1e-6f * ((A[0] + A[4]) + A[8]);
      A[0] += cfm; A[4] += cfm; A[8] += cfm;
      const float s01 = 0.5f * (A[1] + A[3]), s02 = 0.5f * (A[2] + A[6]), s12 = 0.5f * (A[5] + A[7]);
      A[1] = s01; A[3] = s01; A[2] = s02; A[6] = s02; A[5] = s12; A[7] = s12;
      if (own) {
#pragma unroll
        for (int k = 0; k < 9; k++) Wd[k] = A[k];
      }
      mv3(A, O.n, O.Wn);
      O.iwnn = rcp_spec(dot3(O.n, O.Wn));
      float PW[9], B[9];
#pragma unroll
      for (int rr = 0; rr < 3; rr++)
#pragma unroll
        for (int q = 0; q < 3; q++) PW[3 * rr + q] = fmaf(-O.n[rr], O.Wn[q], A[3 * rr + q]);
#pragma unroll
      for (int rr = 0; rr < 3; rr++) {
        const float pwn = dot3(PW + 3 * rr, O.n);
#pragma unroll
        for (int q = 0; q < 3; q++) B[3 * rr + q] = fmaf(O.n[rr], O.n[q], fmaf(-pwn, O.n[q], PW[3 * rr + q]));
      }
      const float b01 = 0.5f * (B[1] + B[3]), b02 = 0.5f * (B[2] + B[6]), b12 = 0.5f * (B[5] + B[7]);
      B[1] = b01; B[3] = b01; B[2] = b02; B[6] = b02; B[5] = b12; B[7] = b12;
      mat3_inv_spd(B, O.Ti);
      O.rt = rcp_spec(((B[0] + B[4]) + B[8]) - 1.0f);
    }
    GROUP_SYNC();

    // H3 / H4. position iterations -> poses; velocity iterations -> velocities
#pragma unroll 1
    for (int phase = 0; phase < 2; phase++) {
      const int sweeps = phase == 0 ? npos : nvel;
      const float tg = phase == 0 ? O.tgt : O.tgt_v;
#pragma unroll 1
      for (int it = 0; it < sweeps; it++) {
#pragma unroll 1
        for (int c = 0; c < HCK; c++) {
          if (__ballot(c < K) == 0ull) break;
          // every owner lane computes its own update; lane c's is the one that counts
          const float pn0 = dot3(O.n, O.p);
          const float pn = rmaxf(fmaf(-(dot3(O.n, O.u) - tg), O.iwnn, pn0), 0.0f);
          const float dn = pn - pn0;
          float un3[3], ut[3], pt[3], ps[3];
#pragma unroll
          for (int r = 0; r < 3; r++) un3[r] = fmaf(dn, O.Wn[r], O.u[r]);
          const float un = dot3(O.n, un3);
#pragma unroll
          for (int r = 0; r < 3; r++) { ut[r] = fmaf(-un, O.n[r], un3[r]); pt[r] = fmaf(-pn0, O.n[r], O.p[r]); }
#pragma unroll
          for (int r = 0; r < 3; r++) ps[r] = pt[r] - fmaf(O.Ti[3 * r + 2], ut[2], fmaf(O.Ti[3 * r + 1], ut[1], O.Ti[3 * r] * ut[0]));
          const float lim = O.mu * pn;
          if (dot3(ps, ps) > lim * lim) {
#pragma unroll
            for (int r = 0; r < 3; r++) ps[r] = fmaf(-O.rt, ut[r], pt[r]);
            const float nt2 = dot3(ps, ps);
            const float sc1 = nt2 > lim * lim ? lim * rsqrt_spec(nt2) : 1.0f;
#pragma unroll
            for (int r = 0; r < 3; r++) ps[r] *= sc1;
          }
          const bool commit = l == c && c < K;
          float dp[3];
#pragma unroll
          for (int r = 0; r < 3; r++) {
            const float pnew = fmaf(pn, O.n[r], ps[r]);
            dp[r] = commit ? pnew - O.p[r] : 0.0f;
            if (commit) O.p[r] = pnew;
          }
          // the change of contact c's impulse, from its owner lane to every lane of the env
#pragma unroll
          for (int r = 0; r < 3; r++) dp[r] = __shfl(dp[r], lane0 + c);
          if (own && c < K) {
            const float* Wb = tail + T::W + (c * HCK + l) * 9;   // block (l, c)
#pragma unroll
            for (int r = 0; r < 3; r++) O.u[r] = fmaf(Wb[3 * r + 2], dp[2], fmaf(Wb[3 * r + 1], dp[1], fmaf(Wb[3 * r], dp[0], O.u[r])));
          }
        }
      }
      if (phase == 1 && nvel == 0) break;
      if (own) {
        float* h = tail + T::HC + l * HC_STRIDE;
#pragma unroll
        for (int r = 0; r < 3; r++) h[HC_P + r] = O.p[r];
      }
      GROUP_SYNC();
      chain_hard_apply<CD>(m, L, tail, l, K, idt, isbody && half == 0, islink, isroot, ischain, ci, lb, myb, phase == 0 ? 2 : 3,
                           phase == 0 ? ac0p : ac0v);
    }
  } else {
    if (isdof) { L.dofb[l * DOF_STRIDE + 2] = 0.0f; L.dofb[l * DOF_STRIDE + 3] = 0.0f; }
  }
  PHASE_MARK(4);

  // ---- H5. integration: poses with the accelerations after the position iterations, velocities after the velocity iterations
  if (isdof) {
    const float vl = m->vel_limit[l];
    const float qf = L.dofb[l * DOF_STRIDE + 4];
    const float qcp = K > 0 ? L.dofb[l * DOF_STRIDE + 2] : 0.0f;
    const float qcv = K > 0 ? (nvel > 0 ? L.dofb[l * DOF_STRIDE + 3] : qcp) : 0.0f;
    const float qdv = K > 0 ? qf + qcv : qf, qdp = K > 0 ? qf + qcp : qf;
    const float qdn = rclampf(fmaf(dt, qdv, X.qd), -vl, vl);
    const float qpn = rclampf(fmaf(dt, qdp, X.qd), -vl, vl);
    X.q = fmaf(dt, qpn, X.q);
    X.qd = qdn;
  }
  if (isroot) {
    float av[6], apz[6];
#pragma unroll
    for (int j = 0; j < 6; j++) {
      const float cv = nvel > 0 ? ac0v[j] : ac0p[j];
      av[j] = K > 0 ? a0[j] + cv : a0[j];
      apz[j] = K > 0 ? a0[j] + ac0p[j] : a0[j];
    }
    float* Rt = L.root;
    float ang[3] = {Rt[10], Rt[11], Rt[12]}, lin[3] = {Rt[7], Rt[8], Rt[9]}, wxv[3];
    cross3(ang, lin, wxv);
    const float damp = 1.0f / fmaf(dt, C.sp.angular_damping, 1.0f);
    float wn[3], vn[3], wp[3], vp[3];
    const float wmax = C.sp.max_ang_vel;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      wn[k] = fmaf(dt, av[k], ang[k]) * damp;
      vn[k] = fmaf(dt, av[3 + k] + g[k] + wxv[k], lin[k]);
      wp[k] = fmaf(dt, apz[k], ang[k]) * damp;
      vp[k] = fmaf(dt, apz[3 + k] + g[k] + wxv[k], lin[k]);
    }
    const float w2 = dot3(wn, wn);
    if (w2 > wmax * wmax) {
      const float sc2 = wmax * rsqrt_spec(w2);
#pragma unroll
      for (int k = 0; k < 3; k++) wn[k] *= sc2;
    }
    const float p2 = dot3(wp, wp);
    if (p2 > wmax * wmax) {
      const float sc2 = wmax * rsqrt_spec(p2);
#pragma unroll
      for (int k = 0; k < 3; k++) wp[k] *= sc2;
    }
#pragma unroll
    for (int k = 0; k < 3; k++) { Rt[10 + k] = wn[k]; Rt[7 + k] = vn[k]; Rt[k] = fmaf(dt, vp[k], Rt[k]); }
    const float hx = 0.5f * dt * wp[0], hy = 0.5f * dt * wp[1], hz = 0.5f * dt * wp[2];
    const float x = Rt[3], y = Rt[4], z = Rt[5], ww = Rt[6];
    const float nx = x + fmaf(hx, ww, fmaf(hy, z, -(hz * y)));
    const float ny = y + fmaf(hy, ww, fmaf(hz, x, -(hx * z)));
    const float nz = z + fmaf(hz, ww, fmaf(hx, y, -(hy * x)));
    const float nw = ww - fmaf(hx, x, fmaf(hy, y, hz * z));
    const float inv = rsqrt_spec(fmaf(nw, nw, fmaf(nz, nz, fmaf(ny, ny, nx * nx))));
    Rt[3] = nx * inv; Rt[4] = ny * inv; Rt[5] = nz * inv; Rt[6] = nw * inv;
  }

  // ---- net contact force per reported body: the final impulses / dt, constraint order (oracle: hc_forces)
  if (contact_out) {
    GROUP_SYNC();
    if (isbody && half == 0) {
      const bool last = islink && (lb % NLK) == NLK - 1;
      float f[3] = {0.0f, 0.0f, 0.0f}, fw[3] = {0.0f, 0.0f, 0.0f};
      for (int c = 0; c < K; c++) {
        const float* h = tail + T::HC + c * HC_STRIDE;
        const int rep = __float_as_int(h[HC_REP]), repb = __float_as_int(h[HC_REPB]);
        if (rep == myb) { f[0] += h[HC_P] * idt; f[1] += h[HC_P + 1] * idt; f[2] += h[HC_P + 2] * idt; }
        else if (last && rep == myb + 1) { fw[0] += h[HC_P] * idt; fw[1] += h[HC_P + 1] * idt; fw[2] += h[HC_P + 2] * idt; }
        if (repb == myb) { f[0] -= h[HC_P] * idt; f[1] -= h[HC_P + 1] * idt; f[2] -= h[HC_P + 2] * idt; }
        else if (last && repb == myb + 1) { fw[0] -= h[HC_P] * idt; fw[1] -= h[HC_P + 1] * idt; fw[2] -= h[HC_P + 2] * idt; }
      }
      contact_out[3 * myb] = f[0]; contact_out[3 * myb + 1] = f[1]; contact_out[3 * myb + 2] = f[2];
      if (last) { contact_out[3 * (myb + 1)] = fw[0]; contact_out[3 * (myb + 1) + 1] = fw[1]; contact_out[3 * (myb + 1) + 2] = fw[2]; }
    }
  }
  GROUP_SYNC();
  PHASE_MARK(9);
}
