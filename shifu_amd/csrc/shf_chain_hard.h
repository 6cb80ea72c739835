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

#define UF_STRIDE 8     /* U[6] invD . per link */
// LDS of the solve, KC = the constraints it holds (ShfSimParams.max_contacts <= KC):
//   KC = 8  everything inside the contact-slot region: constraint records | W, KC x KC blocks of 9 words | U, 1/D per link
//   KC = 16 W is symmetric (block (j, i) = block (i, j)^T: oracle hard_solve), so only its upper triangle is kept -- KC (KC + 1) / 2
//           = 136 blocks, block (i, j), i <= j, at number j (j + 1) / 2 + i: the first NB0 behind the constraint records in the
//           contact-slot region, the others over the pose / velocity-rate / exchange slots (contiguous: ChainLds), which are dead
//           between the free outward pass and the impulse passes once the owners have read their points' velocities (H2a).  U, 1/D
//           and the gap list of the selection sit where the first blocks of W will be written (dead by then).
template <class CD, int KC>
struct HardTail {
  static constexpr int HC = 0, W = KC * HC_STRIDE, NEVP = (CD::NEV + 3) & ~3;
  static constexpr bool PACKED = KC > 8;
  static constexpr int WS = PACKED ? 9 : 10;       // words per block (KC = 8: padded to ten, so that a block is four 8-byte reads and one word)
  static constexpr int NBLK = PACKED ? KC * (KC + 1) / 2 : KC * KC;
  static constexpr int NB0 = PACKED ? (CD::NPC * PT_STRIDE - W) / 9 : NBLK;          // blocks of W in the contact-slot region
  // U, 1/D per link (phases E - G) and the selection's gap list (phase P) sit where W will be written (H1): both dead by then
  static constexpr int UF = W, PHI = W, END = W + NB0 * WS;
  static constexpr int SPARE = CD::NB * POSE_STRIDE + ((CD::NB * 6 + 3) & ~3) + CD::ND * XCH_STRIDE;   // pose | acc | xch words
  static_assert(NEVP + SHF_MAX_SELF_CONTACTS <= NB0 * WS && CD::ND * UF_STRIDE <= NB0 * WS, "the gap list / U, 1/D fit the response matrix's place");
  static_assert(!PACKED || (NBLK - NB0) * 9 <= SPARE, "the rest of the packed response matrix fits the pose / rate / exchange slots");
  static_assert(KC <= 16 && (W % 2) == 0, "owner lanes, 16-bit constraint masks; 8-byte aligned blocks");
};
// block number and address of W's block (i, j) as stored (PACKED: i <= j)
template <class CD, int KC>
DEV float* hard_wblock(const ChainLds& L, float* tail, int i, int j) {
  typedef HardTail<CD, KC> T;
  if constexpr (T::PACKED) {
    const int b = ((j * (j + 1)) >> 1) + i;
    return b < T::NB0 ? tail + T::W + b * 9 : L.pose + (b - T::NB0) * 9;
  } else {
    return tail + T::W + (j * KC + i) * T::WS;
  }
}

DEV void mat3_inv_spd(const float* A, float* Ai) {
  const float c00 = fmaf(A[4], A[8], -(A[5] * A[7])), c01 = fmaf(A[5], A[6], -(A[3] * A[8])), c02 = fmaf(A[3], A[7], -(A[4] * A[6]));
  const float id = rcp_spec(fmaf(A[0], c00, fmaf(A[1], c01, A[2] * c02)));
  Ai[0] = c00 * id; Ai[1] = fmaf(A[2], A[7], -(A[1] * A[8])) * id; Ai[2] = fmaf(A[1], A[5], -(A[2] * A[4])) * id;
  Ai[3] = c01 * id; Ai[4] = fmaf(A[0], A[8], -(A[2] * A[6])) * id; Ai[5] = fmaf(A[2], A[3], -(A[0] * A[5])) * id;
  Ai[6] = c02 * id; Ai[7] = fmaf(A[1], A[6], -(A[0] * A[7])) * id; Ai[8] = fmaf(A[0], A[4], -(A[1] * A[3])) * id;
}
// The link record of the solve (the joint record once the free outward pass is through with it, JREC_STRIDE words, 16-byte
// aligned): S[6] U[6] 1/D -- four 16-byte LDS reads.
#define LREC_U 6
#define LREC_INVD 12
struct HardLink { float S[6], U[6], invD; };
DEV HardLink hard_link_load(const float* rec) {
  const float4* r4 = reinterpret_cast<const float4*>(rec);
  const float4 a = r4[0], b = r4[1], c = r4[2], d = r4[3];
  HardLink K;
  K.S[0] = a.x; K.S[1] = a.y; K.S[2] = a.z; K.S[3] = a.w; K.S[4] = b.x; K.S[5] = b.y;
  K.U[0] = b.z; K.U[1] = b.w; K.U[2] = c.x; K.U[3] = c.y; K.U[4] = c.z; K.U[5] = c.w;
  K.invD = d.x;
  return K;
}
// Response of the articulation to the impulse e at r on its moving body bs (0: the root): the root's velocity change and the
// joint terms of the links on bs's chain up to bs (oracle: hc_impulse) ...
template <class CD>
struct HardResp { int cs, ks; float ub[CD::NLK], dv0[6]; };
template <class CD>
DEV void hard_impulse(const ChainLds& L, const float* tail, int bs, const float* r, const float* e, HardResp<CD>& q) {
  constexpr int NLK = CD::NLK;
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
      const HardLink Lk = hard_link_load(L.jrec + (q.cs * NLK + k) * JREC_STRIDE);
      float sp = Lk.S[0] * p6[0];
#pragma unroll
      for (int j = 1; j < 6; j++) sp = fmaf(Lk.S[j], p6[j], sp);
      q.ub[k] = -sp;
      const float tt = q.ub[k] * Lk.invD;
#pragma unroll
      for (int j = 0; j < 6; j++) p6[j] = fmaf(Lk.U[j], tt, p6[j]);
    }
  }
  root_factors_apply(L.xroot, p6, q.dv0);
}
// ... and the velocity change of the point r of moving body bt under it (oracle: hc_velocity); bt < 0: nothing moves
template <class CD>
DEV void hard_velocity(const ChainLds& L, const float* tail, const HardResp<CD>& q, int bt, const float* r, float* vel) {
  constexpr int NLK = CD::NLK;
  if (bt < 0) { vel[0] = 0.0f; vel[1] = 0.0f; vel[2] = 0.0f; return; }
  const int ct = bt > 0 ? (bt - 1) / (NLK + 1) : 0, kt = bt > 0 ? (bt - 1) % (NLK + 1) : -1;
  float dv[6];
#pragma unroll
  for (int j = 0; j < 6; j++) dv[j] = q.dv0[j];
#pragma unroll
  for (int k = 0; k < NLK; k++) {
    if (k <= kt) {
      const HardLink Lk = hard_link_load(L.jrec + (ct * NLK + k) * JREC_STRIDE);
      const float ubk = (ct == q.cs && k <= q.ks) ? q.ub[k] : 0.0f;
      float ua = Lk.U[0] * dv[0];
#pragma unroll
      for (int j = 1; j < 6; j++) ua = fmaf(Lk.U[j], dv[j], ua);
      const float dq = (ubk - ua) * Lk.invD;
#pragma unroll
      for (int j = 0; j < 6; j++) dv[j] = fmaf(Lk.S[j], dq, dv[j]);
    }
  }
  hard_point(dv, r, vel);
}

// H4: the impulses in the constraint records as forces on their bodies -> what they add to the accelerations.  Both
// impulse sets in one pass through the tree: q = 0 the impulses after the position iterations (HC_P), q = 1 after the
// velocity iterations (HC_PV; NQ = 1: there are none) -- joint accelerations to dofb[.][2 + q], the root's to ac0[q]
// (root lane).  Per set the operations of the oracle's hc_apply.
template <class CD, int NQ, int KC>
DEV void chain_hard_apply(const ShfModel* m, const ChainLds& L, float* tail, int l, int K, float idt, bool isbody_h0, bool islink, bool isroot,
                          bool ischain, int ci, int lb, int myb, float (*ac0)[6]) {
  constexpr int NCH = CD::NCH, NLK = CD::NLK;
  typedef HardTail<CD, KC> T;
  float pcr[NQ][6];
#pragma unroll
  for (int q = 0; q < NQ; q++)
#pragma unroll
    for (int j = 0; j < 6; j++) pcr[q][j] = 0.0f;
  if (isbody_h0) {
    // which constraints act on this body: the ids of all of them in flight at once (a load per iteration behind its own
    // branch would serialise eight LDS round trips)
    unsigned mine_a = 0u, mine_b = 0u;
#pragma unroll
    for (int c = 0; c < KC; c++) {
      const float* h = tail + T::HC + c * HC_STRIDE;
      const int ia = __float_as_int(h[HC_BODY]), ib = __float_as_int(h[HC_BODYB]);
      if (c < K && ia == myb) mine_a |= 1u << c;
      if (c < K && ib == myb) mine_b |= 1u << c;
    }
    for (unsigned bits = mine_a | mine_b; bits; bits &= bits - 1u) {
      const int c = __builtin_ctz(bits);
      const float* h = tail + T::HC + c * HC_STRIDE;
      const bool ona = (mine_a >> c) & 1u;
      const float r[3] = {h[HC_R], h[HC_R + 1], h[HC_R + 2]};
#pragma unroll
      for (int q = 0; q < NQ; q++) {
        const float f[3] = {h[q == 0 ? HC_P : HC_PV0] * idt, h[q == 0 ? HC_P + 1 : HC_PV1] * idt, h[q == 0 ? HC_P + 2 : HC_PV2] * idt};
        float t[3];
        cross3(r, f, t);
        if (ona) {
#pragma unroll
          for (int k = 0; k < 3; k++) { pcr[q][k] -= t[k]; pcr[q][3 + k] -= f[k]; }
        } else {
#pragma unroll
          for (int k = 0; k < 3; k++) { pcr[q][k] += t[k]; pcr[q][3 + k] += f[k]; }
        }
      }
    }
    if (islink) {
      // (the exchange slot's inertia words are free by now: set 0 in the bias words 21-26, set 1 in words 8-13)
      float* o = L.xch + lb * XCH_STRIDE;
#pragma unroll
      for (int q = 0; q < NQ; q++)
#pragma unroll
        for (int j = 0; j < 6; j++) o[(q == 0 ? 21 : 8) + j] = pcr[q][j];
    }
  }
  GROUP_SYNC();
  float ucl[NQ][NLK];
  if (ischain) {
    float pl[NQ][6];
#pragma unroll
    for (int k = NLK - 1; k >= 0; k--) {
      const int li = ci * NLK + k;
      const float* o = L.xch + li * XCH_STRIDE;
      const HardLink Lk = hard_link_load(L.jrec + li * JREC_STRIDE);
      const float* S = Lk.S; const float* U = Lk.U;
      const float invD = Lk.invD;
#pragma unroll
      for (int q = 0; q < NQ; q++) {
        if (k == NLK - 1) {
#pragma unroll
          for (int j = 0; j < 6; j++) pl[q][j] = o[(q == 0 ? 21 : 8) + j];
        } else {
#pragma unroll
          for (int j = 0; j < 6; j++) pl[q][j] = o[(q == 0 ? 21 : 8) + j] + pl[q][j];
        }
        float sp = S[0] * pl[q][0];
#pragma unroll
        for (int j = 1; j < 6; j++) sp = fmaf(S[j], pl[q][j], sp);
        ucl[q][k] = -sp;
        const float tt = ucl[q][k] * invD;
#pragma unroll
        for (int j = 0; j < 6; j++) pl[q][j] = fmaf(U[j], tt, pl[q][j]);
      }
    }
    float* o = L.xch + (ci * NLK) * XCH_STRIDE;
#pragma unroll
    for (int q = 0; q < NQ; q++)
#pragma unroll
      for (int j = 0; j < 6; j++) o[(q == 0 ? 21 : 8) + j] = pl[q][j];
  }
  GROUP_SYNC();
  if (isroot) {
#pragma unroll
    for (int q = 0; q < NQ; q++) {
#pragma unroll
      for (int j = 0; j < 6; j++) {
        float v = pcr[q][j];
#pragma unroll
        for (int c = 0; c < NCH; c++) v += L.xch[(c * NLK) * XCH_STRIDE + (q == 0 ? 21 : 8) + j];
        pcr[q][j] = v;
      }
      root_factors_apply(L.xroot, pcr[q], ac0[q]);
#pragma unroll
      for (int j = 0; j < 6; j++) L.acc[6 * q + j] = ac0[q][j];
    }
  }
  GROUP_SYNC();
  if (ischain) {
    float ac[NQ][6];
#pragma unroll
    for (int q = 0; q < NQ; q++)
#pragma unroll
      for (int j = 0; j < 6; j++) ac[q][j] = L.acc[6 * q + j];
#pragma unroll
    for (int k = 0; k < NLK; k++) {
      const int li = ci * NLK + k;
      const HardLink Lk = hard_link_load(L.jrec + li * JREC_STRIDE);
      const float* S = Lk.S; const float* U = Lk.U;
      const float invD = Lk.invD;
#pragma unroll
      for (int q = 0; q < NQ; q++) {
        float ua = U[0] * ac[q][0];
#pragma unroll
        for (int j = 1; j < 6; j++) ua = fmaf(U[j], ac[q][j], ua);
        const float qc = (ucl[q][k] - ua) * invD;
#pragma unroll
        for (int j = 0; j < 6; j++) ac[q][j] = fmaf(S[j], qc, ac[q][j]);
        L.dofb[li * DOF_STRIDE + 2 + q] = qc;
      }
    }
  }
  GROUP_SYNC();
}

// One gym.simulate() for one env under the velocity-level contact solve; lane roles as chain_substep at 32 lanes per env.
template <int G, class CD, bool TW, bool SELF, int KC, bool TGS>
DEV void chain_substep_hard(const StepCtx& C, const ChainLds& L, int l, DofLane& X, const ChainPoints<(CD::NEV + G - 1) / G>& P,
                            const RowLane& RL, const float* fext, float mu_shape, float* contact_out) {
  static_assert(G == 32, "the velocity-level solve is written for two envs per wavefront");
  constexpr int NCH = CD::NCH, NLK = CD::NLK, NB = CD::NB, ND = CD::ND, NR = (CD::NEV + G - 1) / G;
  typedef HardTail<CD, KC> T;
  static_assert(T::END <= CD::NPC * PT_STRIDE, "the solve's LDS fits the contact-slot region");
  static_assert((KC == 8 || KC == 16) && KC <= G / 2 + 8 && ND < 16, "owner lanes; a lane per column of the response matrix, in one or two passes");
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
    {
      float mass = m->mass[myb], I6[6];
#pragma unroll
      for (int k = 0; k < 6; k++) I6[k] = m->inertia[myb][k];
      if (C.mscale) {     // SHF_T_BODY_MASS_SCALE bound: this env's factor on the body's mass and inertia
        const float s = C.mscale[myb];
        mass *= s;
#pragma unroll
        for (int k = 0; k < 6; k++) I6[k] *= s;
      }
      rigid_inertia_p(mass, m->com[myb], I6, Rb, pp, vb, IA, pA);
    }
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
  const int kmax = hard_kmax_of(C.sp, KC);
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
    contact_hist_count(C, l, total);
    unsigned smask = nself >= 32 ? ~0u : ((1u << nself) - 1u);      // selected self-contacts of this env
    if (__ballot(total > kmax) != 0ull) {
      // more candidates than the solve holds (somewhere in this wavefront): keep the kmax with the smallest gap, ties by
      // candidate order (slots, then self-contacts)
      // The gaps as a compact list in candidate order (sample-point slots ascending, then the self-contacts): a candidate's rank
      // is the number of entries that come before it in (gap, position).  Read four entries at a time -- a list walked one LDS
      // round trip per entry was most of this phase for the wavefronts that overflow.
      float* lp = tail + T::PHI;
      int pos[NR];
#pragma unroll
      for (int k = 0; k < NR; k++) {
        const int sl = l + k * G;
        pos[k] = sl < 64 ? __popcll(act.w[0] & ((1ull << sl) - 1ull)) : __popcll(act.w[0]) + __popcll(act.w[1] & ((1ull << (sl - 64)) - 1ull));
        if (cand[k]) lp[pos[k]] = ph[k];
      }
      const int spos = npts + l;
      if (scand) lp[spos] = sph;
      if (l < 4) lp[total + l] = 3.0e38f;          // padding to a multiple of four: behind every gap
      GROUP_SYNC();
      if (total > kmax) {
        int rank[NR], srank = 0;
#pragma unroll
        for (int k = 0; k < NR; k++) rank[k] = 0;
        for (int j4 = 0; j4 < total; j4 += 4) {
          const float4 pj4 = *reinterpret_cast<const float4*>(lp + j4);
          const float pj[4] = {pj4.x, pj4.y, pj4.z, pj4.w};
#pragma unroll
          for (int e = 0; e < 4; e++) {
#pragma unroll
            for (int k = 0; k < NR; k++) rank[k] += (pj[e] < ph[k] || (pj[e] == ph[k] && j4 + e < pos[k])) ? 1 : 0;
            if constexpr (SELF) srank += (pj[e] < sph || (pj[e] == sph && j4 + e < spos)) ? 1 : 0;
          }
        }
#pragma unroll
        for (int k = 0; k < NR; k++) cand[k] = cand[k] && rank[k] < kmax;
        if constexpr (SELF) scand = scand && srank < kmax;
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
      const float ns[3] = {sslot[PT_N], sslot[PT_N + 1], sslot[PT_N + 2]};
      hard_frame(ns, h + HC_T1, h + HC_T2);
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
      hard_frame(nn[k], h + HC_T1, h + HC_T2);
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
      // the joint record becomes the solve's link record: U and 1/D take the place of the velocity-product term and the efforts
      float* recw = L.jrec + li * JREC_STRIDE;
#pragma unroll
      for (int j = 0; j < 6; j++) recw[LREC_U + j] = uf[j];
      recw[LREC_INVD] = uf[6];
    }
  }
  GROUP_SYNC();
  PHASE_MARK(8);

  // ---- H. the contact solve.  Loops over contacts run while any env of the wavefront has one left (wave-uniform bounds)
  float ac0[2][6] = {{0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f}};   // the contacts' share of the root's acceleration: poses / velocities
  const int npos = C.sp.pos_iters > 0 ? C.sp.pos_iters : 0, nvel = C.sp.vel_iters > 0 ? C.sp.vel_iters : 0;
  if (__ballot(K > 0) != 0ull) {
    // H2a. owner lanes: the contact's bodies' velocities at its point -- free velocity in the contact frame, targets.  (With the packed
    // response matrix before the columns: the poses and velocity rates read here are overwritten by it; KC = 8 keeps round 5's order.)
    HardOwner O;
    const bool own = l < K;
    auto owner_velocities = [&]() {
      const float* h = tail + T::HC + (own ? l : 0) * HC_STRIDE;
      const float r[3] = {h[HC_R], h[HC_R + 1], h[HC_R + 2]};
      const float n[3] = {h[HC_N], h[HC_N + 1], h[HC_N + 2]};
#pragma unroll
      for (int k = 0; k < 3; k++) O.p[k] = 0.0f;
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
      for (int k = 0; k < 3; k++) { vf[k] = vf[k] - vfb[k]; vs[k] = vs[k] - vsb[k]; }
      O.u[0] = dot3(n, vf); O.u[1] = dot3(h + HC_T1, vf); O.u[2] = dot3(h + HC_T2, vf);
      const float phi = h[HC_PHI];
      const float erp = C.sp.erp > 0.0f ? C.sp.erp : 0.2f;
      float tg = phi >= 0.0f ? -(phi * idt) : rminf(erp * -(phi) * idt, C.sp.max_depen_vel);
      float tv = phi >= 0.0f ? tg : 0.0f;
      const float vn0 = dot3(n, vs);
      if (C.sp.restitution > 0.0f && vn0 < -C.sp.bounce_threshold) { tg = rmaxf(tg, -(C.sp.restitution * vn0)); tv = rmaxf(tv, -(C.sp.restitution * vn0)); }
      O.tgt = tg; O.tgt_v = tv;
      O.gap = phi;
      O.rfloor = (C.sp.restitution > 0.0f && vn0 < -C.sp.bounce_threshold) ? -(C.sp.restitution * vn0) : -1e30f;
    };
    if constexpr (T::PACKED) owner_velocities();
    if constexpr (T::PACKED) GROUP_SYNC();     // (every owner has read its poses and rates: their place may take blocks of W)
    // H1. column lanes: column q = 3 j + axis of contact j's frame; the upper triangle of W (oracle: hard_solve, "columns"): block
    // (i, j), i <= j -- block (j, i) is its transpose, written as such (KC = 8) or left to the readers (packed)
#pragma unroll 1
    for (int q0 = 0; q0 < 3 * KC; q0 += G) {
      const int q = q0 + l;
      if (__ballot(q < 3 * K) == 0ull) break;
      const int j = (q * 43) >> 7, ax = q - 3 * j;
      const bool col = q < 3 * K;
      const float* hj = tail + T::HC + (col ? j : 0) * HC_STRIDE;
      const float rj[3] = {hj[HC_R], hj[HC_R + 1], hj[HC_R + 2]};
      const int bsa = col ? __float_as_int(hj[HC_BODY]) : 0;
      const int bsb = (SELF && col) ? __float_as_int(hj[HC_BODYB]) : -1;
      const float* ej = hj + (ax == 0 ? HC_N : (ax == 1 ? HC_T1 : HC_T2));     // axis ax of contact j's frame
      const float e[3] = {ej[0], ej[1], ej[2]};
      HardResp<CD> qa, qb;
      hard_impulse<CD>(L, tail, bsa, rj, e, qa);
      if constexpr (SELF) {
        if (__ballot(bsb >= 0) != 0ull) hard_impulse<CD>(L, tail, bsb >= 0 ? bsb : 0, rj, e, qb);
      }
      // One block of every symmetric pair (oracle: hard_solve, "columns"): block (i, j) from this column for i = j, j - 1, .. j - K / 2
      // (modulo K; for even K the pair at distance K / 2 belongs to the columns j >= K / 2): K / 2 + 1 targets per lane at most
#pragma unroll 1
      for (int d = 0; 2 * d <= KC; d++) {
        if (__ballot(2 * d <= K) == 0ull) break;
        const bool todo = col && 2 * d <= K && d < K && !(d > 0 && 2 * d == K && 2 * j < K);
        int i = j - d;
        if (i < 0) i += K;
        if (!todo) continue;
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
        float vw[3];
#pragma unroll
        for (int r = 0; r < 3; r++) vw[r] = (aa[r] - ab[r]) - (ba[r] - bb[r]);
        const float w0 = dot3(hi + HC_N, vw), w1 = dot3(hi + HC_T1, vw), w2 = dot3(hi + HC_T2, vw);    // column ax of block (i, j): the velocity in contact i's frame
        if constexpr (T::PACKED) {
          // the triangle i <= j is kept: block (i, j) as it is, or -- i > j -- as row ax of block (j, i), its transpose
          if (i <= j) { float* Wb = hard_wblock<CD, KC>(L, tail, i, j) + ax; Wb[0] = w0; Wb[3] = w1; Wb[6] = w2; }
          else { float* Wt = hard_wblock<CD, KC>(L, tail, j, i) + 3 * ax; Wt[0] = w0; Wt[1] = w1; Wt[2] = w2; }
        } else {
          float* Wb = hard_wblock<CD, KC>(L, tail, i, j) + ax;
          Wb[0] = w0; Wb[3] = w1; Wb[6] = w2;
          if (i != j) { float* Wt = hard_wblock<CD, KC>(L, tail, j, i) + 3 * ax; Wt[0] = w0; Wt[1] = w1; Wt[2] = w2; }     // row ax of block (j, i)
        }
      }
    }
    GROUP_SYNC();
    PHASE_MARK(5);

    // H2b. owner lanes: the regularised diagonal block and its inverses
    if constexpr (!T::PACKED) owner_velocities();
    {
      float* Wd = hard_wblock<CD, KC>(L, tail, own ? l : 0, own ? l : 0);
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
      O.iwnn = rcp_spec(A[0]);
      O.w10 = A[3]; O.w20 = A[6];
      const float id = rcp_spec(fmaf(A[4], A[8], -(A[5] * A[5])));
      O.Ti[0] = A[8] * id; O.Ti[1] = -(A[5] * id); O.Ti[2] = A[4] * id;
      O.rt = rcp_spec(A[4] + A[8]);
    }
    GROUP_SYNC();
    PHASE_MARK(10);

    // H3 / H4. position iterations -> poses; velocity iterations -> velocities
    int Kw = 0;      // the larger constraint count of the wavefront's envs (wave-uniform)
#pragma unroll
    for (int c = 0; c < KC; c++)
      if (__ballot(c < K) != 0ull) Kw = c + 1;
    const bool hi = lane0 != 0;
#ifdef SHF_PHASE_CLOCK
    if ((threadIdx.x & 63u) == 0u) atomicAdd(&g_phase_cycles[38 + (Kw < 8 ? Kw : 8)], 1ull);     // histogram of the wavefronts' constraint counts
#endif
    const float* Wcol = tail + T::W + (own ? l : 0) * T::WS;     // (KC = 8) block (l, c) sits at Wcol + c * KC * WS
    const int li = own ? l : 0;
    // one visit of the sweep: contact c (oracle: hard_solve, sweeps)
    auto visit = [&](int c, float tg) {
      // an open, unloaded contact whose normal velocity keeps it open asks for nothing (oracle: the same test): when that
      // is so for contact c of both envs of the wavefront the visit is skipped
      const bool act = !(O.p[0] == 0.0f && O.p[1] == 0.0f && O.p[2] == 0.0f && !(O.u[0] < tg));
      if (__ballot(l == c && c < K && act) == 0ull) return;
      // this lane's block of column c: in flight while the update is computed
      float Wb[9];
      if constexpr (T::PACKED) {
        // block (l, c) from the upper triangle: as stored when l <= c, else the transpose of block (c, l)
        const bool tr = li > c;
        const float* blk = hard_wblock<CD, KC>(L, tail, tr ? c : li, tr ? li : c);
        float Ws[9];
#pragma unroll
        for (int k = 0; k < 9; k++) Ws[k] = blk[k];
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
          for (int k = 0; k < 3; k++) Wb[3 * r + k] = tr ? Ws[3 * k + r] : Ws[3 * r + k];
      } else {
        const float2* b2 = reinterpret_cast<const float2*>(Wcol + c * KC * T::WS);      // (8-byte aligned: W and WS are even)
        const float2 w01 = b2[0], w23 = b2[1], w45 = b2[2], w67 = b2[3];
        Wb[0] = w01.x; Wb[1] = w01.y; Wb[2] = w23.x; Wb[3] = w23.y; Wb[4] = w45.x; Wb[5] = w45.y; Wb[6] = w67.x; Wb[7] = w67.y;
        Wb[8] = Wcol[c * KC * T::WS + 8];
      }
      // every owner lane computes its own update; lane c's is the one that counts
      const float pn0 = O.p[0];
      const float pn = rmaxf(fmaf(-(O.u[0] - tg), O.iwnn, pn0), 0.0f);
      const float dn = pn - pn0;
      const float ut1 = fmaf(dn, O.w10, O.u[1]), ut2 = fmaf(dn, O.w20, O.u[2]);
      float ps1 = O.p[1] - fmaf(O.Ti[1], ut2, O.Ti[0] * ut1), ps2 = O.p[2] - fmaf(O.Ti[2], ut2, O.Ti[1] * ut1);
      const float lim = O.mu * pn, lim2 = lim * lim;
      const bool commit = l == c && c < K && act;
      // (the sliding step only where it counts: the other owner lanes' updates are discarded, and the wavefront runs this
      // branch -- a third of the visit's instructions -- only when contact c of one of its envs really slides)
      if (commit && fmaf(ps2, ps2, ps1 * ps1) > lim2) {
        ps1 = fmaf(-O.rt, ut1, O.p[1]); ps2 = fmaf(-O.rt, ut2, O.p[2]);
        const float nt2 = fmaf(ps2, ps2, ps1 * ps1);
        const float sc1 = nt2 > rmaxf(lim2, 1e-30f) ? lim * rsqrt_spec(nt2) : 1.0f;   // (1e-30: a subnormal |p_t|^2 over a zero cone would make 0 * inf)
        ps1 *= sc1; ps2 *= sc1;
      }
      // (the committing lane takes the update by selects -- an exec-masked block of moves cost ten instructions per visit --; the
      // change is new - old: pn - pn0 = dn on that lane, exactly 0 on the others)
      const float n0 = commit ? pn : O.p[0], n1 = commit ? ps1 : O.p[1], n2 = commit ? ps2 : O.p[2];
      float dp0 = n0 - O.p[0], dp1 = n1 - O.p[1], dp2 = n2 - O.p[2];
      O.p[0] = n0; O.p[1] = n1; O.p[2] = n2;
      // the change of contact c's impulse, from its owner lane (lane c of each env: wave lanes c and 32 + c) to every lane
      {
        const float a0 = hard_readlane(dp0, c), a1 = hard_readlane(dp1, c), a2 = hard_readlane(dp2, c);
        const float b0 = hard_readlane(dp0, 32 + c), b1 = hard_readlane(dp1, 32 + c), b2 = hard_readlane(dp2, 32 + c);
        dp0 = hi ? b0 : a0; dp1 = hi ? b1 : a1; dp2 = hi ? b2 : a2;
      }
      if (own && c < K) {
#pragma unroll
        for (int r = 0; r < 3; r++) O.u[r] = fmaf(Wb[3 * r + 2], dp2, fmaf(Wb[3 * r + 1], dp1, fmaf(Wb[3 * r], dp0, O.u[r])));
      }
    };
    const HardTgs TG = hard_tgs(C.sp, npos);      // SHF_SOLVER_TGS: sub-stepped sweeps (csrc/shf_hard.h); compiled in when TGS
    float psum[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll 1
    for (int phase = 0; phase < 2; phase++) {
      const int sweeps = phase == 0 ? npos : nvel;
      float tg = phase == 0 ? O.tgt : O.tgt_v;
      if constexpr (TGS) { if (phase == 1) tg = hard_tgs_target_vel(TG, O); }
#pragma unroll 1
      for (int it = 0; it < sweeps; it++) {
        if constexpr (TGS) { if (phase == 0) tg = hard_tgs_target_pos(TG, O); }
        if constexpr (T::PACKED) {
#pragma unroll 1
          for (int c = 0; c < Kw; c++) visit(c, tg);       // (a loop: sixteen unrolled visits would not fit the instruction cache)
        } else {
#pragma unroll
          for (int c = 0; c < KC; c++) {
            if (c >= Kw) break;
            visit(c, tg);
          }
        }
        if constexpr (TGS) {
          if (phase == 0) {
            O.gap = fmaf(TG.hN, O.u[0], O.gap);
#pragma unroll
            for (int r = 0; r < 3; r++) psum[r] += O.p[r];
          }
        }
      }
      if (phase == 1 && nvel == 0) break;
      if (own) {
        // the impulses of this phase in world axes, for the pass through the tree
        float* h = tail + T::HC + l * HC_STRIDE;
        float pw[3];
        const bool mean = TGS && phase == 0;
        const float q0 = mean ? psum[0] * TG.inv_n : O.p[0], q1 = mean ? psum[1] * TG.inv_n : O.p[1], q2 = mean ? psum[2] * TG.inv_n : O.p[2];
#pragma unroll
        for (int r = 0; r < 3; r++) pw[r] = fmaf(q2, h[HC_T2 + r], fmaf(q1, h[HC_T1 + r], q0 * h[HC_N + r]));
        if (phase == 0) { h[HC_P] = pw[0]; h[HC_P + 1] = pw[1]; h[HC_P + 2] = pw[2]; }
        else { h[HC_PV0] = pw[0]; h[HC_PV1] = pw[1]; h[HC_PV2] = pw[2]; }
      }
    }
    PHASE_MARK(17);
    GROUP_SYNC();
    if (nvel > 0) chain_hard_apply<CD, 2, KC>(m, L, tail, l, K, idt, isbody && half == 0, islink, isroot, ischain, ci, lb, myb, ac0);
    else chain_hard_apply<CD, 1, KC>(m, L, tail, l, K, idt, isbody && half == 0, islink, isroot, ischain, ci, lb, myb, ac0);
    PHASE_MARK(18);
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
      const float cv = nvel > 0 ? ac0[1][j] : ac0[0][j];
      av[j] = K > 0 ? a0[j] + cv : a0[j];
      apz[j] = K > 0 ? a0[j] + ac0[0][j] : a0[j];
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
#pragma unroll
      for (int c = 0; c < KC; c++) {      // (unrolled: the records of all constraints in flight at once)
        const float* h = tail + T::HC + c * HC_STRIDE;
        const int rep = c < K ? __float_as_int(h[HC_REP]) : -9, repb = c < K ? __float_as_int(h[HC_REPB]) : -9;
        const float pf[3] = {(nvel > 0 ? h[HC_PV0] : h[HC_P]) * idt, (nvel > 0 ? h[HC_PV1] : h[HC_P + 1]) * idt, (nvel > 0 ? h[HC_PV2] : h[HC_P + 2]) * idt};
        if (rep == myb) { f[0] += pf[0]; f[1] += pf[1]; f[2] += pf[2]; }
        else if (last && rep == myb + 1) { fw[0] += pf[0]; fw[1] += pf[1]; fw[2] += pf[2]; }
        if (repb == myb) { f[0] -= pf[0]; f[1] -= pf[1]; f[2] -= pf[2]; }
        else if (last && repb == myb + 1) { fw[0] -= pf[0]; fw[1] -= pf[1]; fw[2] -= pf[2]; }
      }
      contact_out[3 * myb] = f[0]; contact_out[3 * myb + 1] = f[1]; contact_out[3 * myb + 2] = f[2];
      if (last) { contact_out[3 * (myb + 1)] = fw[0]; contact_out[3 * (myb + 1) + 1] = fw[1]; contact_out[3 * (myb + 1) + 2] = fw[2]; }
    }
  }
  GROUP_SYNC();
  PHASE_MARK(9);
}
