// shf_chain.h -- the fused A1 step with one lane per kinematic CHAIN.
//
// Tree shape served: a floating root with NCH serial chains of NLK revolute links, each chain ending in one welded
// reported body (the Unitree A1: trunk + 4 legs of hip / thigh / calf + foot).  The body-per-lane
// mapping of shf_device.h executes the tree level by level -- every level is one pass of the whole wavefront with a
// quarter of its lanes active and an LDS hand-off (write, fence, read: ~70 clocks, tools/valu_microbench) on either
// side -- so one env step is a dependent chain of ~100 LDS round trips.  Here lane c < NCH walks chain c from the
// root outwards and back with its links' state (S, c, IA, pA, U) in registers: forward kinematics, the three rigid
// inertias, the inward and outward passes of the articulated-body algorithm need no hand-off at all; lane NCH owns
// the root (its inertia, the 6x6 solve, the integration of the floating base).  Per sub-step the group synchronises
// five times: poses -> contact sample points (all G lanes, one point each per round) -> contact slots -> fold;
// hips -> root; root acceleration -> chains.  G = 16 packs four envs into a wavefront (4096 envs = 1024 waves =
// one per SIMD, with the whole 512-register file to itself).
//
// ARITHMETIC: the same operations in the same order as oracle/shf_oracle.c (and as the body-mapped kernel) -- only
// which lane executes them changes, so every result stays bit-identical; tests/test_gpu_parity.py holds both mappings to
// the oracle.  Reference call sites replaced: gym.simulate (examples/a1_conditional/a1_conditional.py:69,
// shifu/gym/isaac_gym.py:140) inside ShifuVecEnv.step (shifu/gym/env.py:85-106).
#pragma once
#include "shf_task.h"

template <int NCH_, int NLK_, int NP_>
struct ChainDims {
  static constexpr int NCH = NCH_, NLK = NLK_, NB = 1 + NCH_ * (NLK_ + 1), ND = NCH_ * NLK_, NPC = NP_;
  DEV static constexpr int body(int c, int k) { return 1 + c * (NLK_ + 1) + k; }   // k == NLK: the welded end body
  static bool matches(const ShfModel& m) {
    if (m.nb != NB || m.nd != ND || m.np != NPC || m.fixed_base || m.jtype[0] != SHF_JOINT_ROOT) return false;
    if (m.child_count[0] != NCH || m.nlevels != NLK) return false;
    for (int i = 0; i < NPC; i++)   // the evaluation order must be a permutation with its inverse
      if (m.pt_eval[i] < 0 || m.pt_eval[i] >= NPC || m.pt_slot[m.pt_eval[i]] != i) return false;
    for (int c = 0; c < NCH; c++) {
      if (m.child_list[m.child_start[0] + c] != 1 + c * (NLK + 1)) return false;
      for (int k = 0; k <= NLK; k++) {
        const int b = 1 + c * (NLK + 1) + k;
        if (m.parent[b] != (k == 0 ? 0 : b - 1)) return false;
        if (k < NLK) {
          if (m.jtype[b] != SHF_JOINT_REVOLUTE || m.dof[b] != c * NLK + k || m.dyn[b] != b || m.level[b] != k + 1) return false;
          if (m.child_count[b] != (k < NLK - 1 ? 1 : 0)) return false;
          if (k < NLK - 1 && m.child_list[m.child_start[b]] != b + 1) return false;
        } else {
          if (m.jtype[b] != SHF_JOINT_WELD || m.dyn[b] != b - 1 || m.dof[b] != -1) return false;
        }
      }
    }
    return true;
  }
};
typedef ChainDims<4, 3, 76> A1Chain;

// LDS of one env: pose records of all reported bodies, accelerations, NCH hip -> root hand-off slots (the region is
// also the net-contact-force staging, nb x 3 floats), the dof block (epilogue layout), the root state, contact slots.
template <class CD>
__host__ __device__ inline int chain_lds_words(int min_tail) {
  int tail = CD::NPC * PT_STRIDE;
  if (tail < min_tail) tail = min_tail;
  const int w = CD::NB * POSE_STRIDE + ((CD::NB * 6 + 3) & ~3) + CD::NCH * XCH_STRIDE + ((CD::ND * DOF_STRIDE + 3) & ~3) + root_words(1) + tail;
  return (w + 3) & ~3;
}
template <class CD>
DEV EnvLds chain_lds_carve(float* base) {
  static_assert(CD::NCH * XCH_STRIDE >= 3 * CD::NB, "the hand-off slots double as the contact-force staging");
  EnvLds L;
  L.pose = base;
  L.acc = L.pose + CD::NB * POSE_STRIDE;
  L.xch = L.acc + ((CD::NB * 6 + 3) & ~3);
  L.dofb = L.xch + CD::NCH * XCH_STRIDE;
  L.root = L.dofb + ((CD::ND * DOF_STRIDE + 3) & ~3);
  L.pt = L.root + root_words(1);
  return L;
}

// One link's share of the ABA, kept in the chain lane's registers across the sub-step.
struct ChainLink {
  float S[6], c[6], IA[21], pA[6], U[6], invD, u;
};

DEV void pose_store(float* o, const float* Rw, const float* p, const float* v) {
#pragma unroll
  for (int k = 0; k < 9; k++) o[k] = Rw[k];
#pragma unroll
  for (int k = 0; k < 3; k++) o[9 + k] = p[k];
#pragma unroll
  for (int k = 0; k < 6; k++) o[12 + k] = v[k];
}

// Forward kinematics of one revolute link below (Rp, pp, vp) -- the parent's pose and velocity, replaced by the
// link's own on return -- with its motion subspace and bias acceleration (oracle kinematics(), same operations).
DEV void chain_kin_link(const float* tp, const float* tr, const float* ax, float qv, float qdv, float* Rp, float* pp,
                        float* vp, float* S, float* c) {
  float sn, cs;
  sincos_spec(qv, &sn, &cs);
  const float oc = 1.0f - cs;
  const float Rq[9] = {fmaf(oc, ax[0] * ax[0], cs),           fmaf(oc, ax[0] * ax[1], -(sn * ax[2])), fmaf(oc, ax[0] * ax[2], sn * ax[1]),
                       fmaf(oc, ax[1] * ax[0], sn * ax[2]),    fmaf(oc, ax[1] * ax[1], cs),            fmaf(oc, ax[1] * ax[2], -(sn * ax[0])),
                       fmaf(oc, ax[2] * ax[0], -(sn * ax[1])), fmaf(oc, ax[2] * ax[1], sn * ax[0]),    fmaf(oc, ax[2] * ax[2], cs)};
  float Rl[9], t[3], Rn[9], pn[3], aw[3], t2[3], vJ[6], cc[6];
  mm3(tr, Rq, Rl);
  mv3(Rp, tp, t);
#pragma unroll
  for (int k = 0; k < 3; k++) pn[k] = pp[k] + t[k];
  mm3(Rp, Rl, Rn);
  mv3(Rn, ax, aw);
  cross3(pn, aw, t2);
#pragma unroll
  for (int k = 0; k < 3; k++) { S[k] = aw[k]; S[3 + k] = t2[k]; }
#pragma unroll
  for (int k = 0; k < 6; k++) vJ[k] = S[k] * qdv;
  crm(vp, vJ, cc);
#pragma unroll
  for (int k = 0; k < 6; k++) { c[k] = cc[k]; vp[k] = vp[k] + vJ[k]; }
#pragma unroll
  for (int k = 0; k < 9; k++) Rp[k] = Rn[k];
#pragma unroll
  for (int k = 0; k < 3; k++) pp[k] = pn[k];
}
// the welded end body: pose only (its velocity is the parent's)
DEV void chain_kin_weld(const float* tp, const float* tr, float* Rp, float* pp) {
  float t[3], Rn[9];
  mv3(Rp, tp, t);
#pragma unroll
  for (int k = 0; k < 3; k++) pp[k] = pp[k] + t[k];
  mm3(Rp, tr, Rn);
#pragma unroll
  for (int k = 0; k < 9; k++) Rp[k] = Rn[k];
}

// gym.apply_rigid_body_force_at_pos_tensors(force, None) on reported body b, folded into its moving body's bias force
DEV void chain_ext_force(const float* F, const float* com, const float* Rw, const float* p, float* pA) {
  if (F[0] == 0.0f && F[1] == 0.0f && F[2] == 0.0f) return;
  float cw[3], t[3];
  mv3(Rw, com, cw);
#pragma unroll
  for (int k = 0; k < 3; k++) cw[k] += p[k];
  cross3(cw, F, t);
#pragma unroll
  for (int k = 0; k < 3; k++) { pA[k] -= t[k]; pA[3 + k] -= F[k]; }
}

// joint-space effort of one dof: explicit part t0 and implicit diagonal de (oracle substep(), "joint-space efforts")
DEV void chain_dof_effort(const StepCtx& C, int d, float q, float qd, float tau_cmd, float* t0o, float* deo) {
  const ShfModel* m = C.m;
  const float dt = C.sp.dt;
  float t0 = 0.0f, de = m->armature[d];
  const int mode = m->drive_mode[d];
  if (mode == SHF_DOF_MODE_EFFORT) {
    t0 = tau_cmd;
  } else if (mode == SHF_DOF_MODE_POS || mode == SHF_DOF_MODE_VEL) {
    // the fused A1 step sets no drive targets (they read as zero, like the body-mapped kernel's null target pointers)
    float kp = mode == SHF_DOF_MODE_POS ? m->kp[d] : 0.0f, kd = m->kd[d];
    const float tq = 0.0f, tv = 0.0f;
    const float est = fmaf(kp, tq - q, kd * (tv - qd));
    const float lim = m->effort[d];
    if (lim > 0.0f && fabsf(est) > lim) { const float sc = lim / fabsf(est); kp *= sc; kd *= sc; }
    const float bj = fmaf(dt, kp, kd);
    t0 = fmaf(kp, tq - q, fmaf(kd, tv, -(bj * qd)));
    de = fmaf(dt, bj, de);
  }
  const float jd = m->damping[d];
  if (jd > 0.0f) { t0 = fmaf(-jd, qd, t0); de = fmaf(dt, jd, de); }
  const float lo = m->lower[d], up = m->upper[d];
  const float viol = q < lo ? lo - q : (q > up ? up - q : 0.0f);
  if (viol != 0.0f) {
    const float bl = fmaf(dt, C.sp.limit_k, C.sp.limit_d);
    t0 = fmaf(C.sp.limit_k, viol, fmaf(-bl, qd, t0));
    de = fmaf(dt, bl, de);
  }
  *t0o = t0; *deo = de;
}

// Inward step of one link: U = IA S, D, u, IA -= U U^T / D, pa = pA + IA c + U u / D  (pa returned, IA updated).
DEV void chain_inward_link(ChainLink& K, float dex, float tau0, float* pa) {
#pragma unroll
  for (int i = 0; i < 6; i++) {
    float acc = SYMG(K.IA, i, 0) * K.S[0];
#pragma unroll
    for (int j = 1; j < 6; j++) acc = fmaf(SYMG(K.IA, i, j), K.S[j], acc);
    K.U[i] = acc;
  }
  float D = K.S[0] * K.U[0];
#pragma unroll
  for (int j = 1; j < 6; j++) D = fmaf(K.S[j], K.U[j], D);
  D += dex;
  float sp = K.S[0] * K.pA[0];
#pragma unroll
  for (int j = 1; j < 6; j++) sp = fmaf(K.S[j], K.pA[j], sp);
  const float invD = 1.0f / D;
  K.invD = invD;
  K.u = tau0 - sp;
  float W[6];
#pragma unroll
  for (int i = 0; i < 6; i++) W[i] = K.U[i] * invD;
#pragma unroll
  for (int i = 0; i < 6; i++)
#pragma unroll
    for (int j = i; j < 6; j++) K.IA[SYM(i, j)] = fmaf(-K.U[i], W[j], K.IA[SYM(i, j)]);
#pragma unroll
  for (int i = 0; i < 6; i++) {
    float acc = SYMG(K.IA, i, 0) * K.c[0];
#pragma unroll
    for (int j = 1; j < 6; j++) acc = fmaf(SYMG(K.IA, i, j), K.c[j], acc);
    pa[i] = fmaf(W[i], K.u, K.pA[i] + acc);
  }
}

// Sample-point constants of a lane's evaluation slots (slot l + k*G evaluates point pt_eval[slot], ShfModel.pt_eval).
template <int NR>
struct ChainPoints {
  int idx[NR];         // point index, -1 for an empty slot
  int body[NR];        // reported body whose pose places the point
  float pos[NR][3], rad[NR], thr[NR];   // thr: clearance x nz_min above which the point cannot be within the contact offset
};
template <int G, int NR>
DEV void chain_points_load(const ShfModel* m, int np, int l, float offset, ChainPoints<NR>& P) {
#pragma unroll
  for (int k = 0; k < NR; k++) {
    const int s = l + k * G;
    const int i = s < np ? m->pt_eval[s] : 0;
    P.idx[k] = s < np ? i : -1;
    P.body[k] = m->pt_body[i];
    P.rad[k] = m->pt_radius[i];
    // 1 % and a micrometre of slack over the exact bound: float rounding in the exact test cannot bridge it
    P.thr[k] = (offset + P.rad[k]) * 1.01f + 1e-6f;
#pragma unroll
    for (int j = 0; j < 3; j++) P.pos[k][j] = m->pt_pos[i][j];
  }
}

// terrain_query_heightfield in two stages -- height and gradient first, the unit normal only where it is needed;
// the same operations on the same values as the one-stage query (and the oracle's).
DEV void terrain_height_gradient(const TerrainDev& T, float x, float y, float* h, float* gxo, float* gyo) {
  if (T.t.rows == 0) { *h = 0.0f; *gxo = 0.0f; *gyo = 0.0f; return; }
  float inv = 1.0f / T.t.hscale;
  float fx = (x + T.t.border) * inv, fy = (y + T.t.border) * inv;
  float fi = rclampf(floorf(fx), 0.0f, (float)(T.t.rows - 2)), fj = rclampf(floorf(fy), 0.0f, (float)(T.t.cols - 2));
  int i = (int)fi, j = (int)fj;
  float u = rclampf(fx - fi, 0.0f, 1.0f), v = rclampf(fy - fj, 0.0f, 1.0f);
  float vs = T.t.vscale;
  const int16_t* row0 = T.h + (size_t)i * T.t.cols + j;
  const int16_t* row1 = row0 + T.t.cols;
  float h00 = (float)row0[0] * vs, h10 = (float)row1[0] * vs, h01 = (float)row0[1] * vs, h11 = (float)row1[1] * vs;
  const bool lo = u + v <= 1.0f;
  float gx = (lo ? h10 : h11) - (lo ? h00 : h01);
  float gy = (lo ? h01 : h11) - (lo ? h00 : h10);
  *h = fmaf(lo ? v : 1.0f - v, lo ? gy : -gy, fmaf(lo ? u : 1.0f - u, lo ? gx : -gx, lo ? h00 : h11));
  *gxo = gx * inv; *gyo = gy * inv;
}
DEV void terrain_normal_from_gradient(const TerrainDev& T, float gx, float gy, float* n) {
  if (T.t.rows == 0) { n[0] = 0.0f; n[1] = 0.0f; n[2] = 1.0f; return; }
  const float nz = 1.0f / sqrtf(fmaf(gy, gy, fmaf(gx, gx, 1.0f)));
  n[0] = -gx * nz; n[1] = -gy * nz; n[2] = nz;
}

// Active evaluation slots of one env as two 64-bit words (slot s = bit s).
struct SlotBits {
  unsigned long long w[2];
  DEV bool test(int s) const { return ((s < 64 ? w[0] >> s : w[1] >> (s - 64)) & 1ull) != 0ull; }
};
// Slots that evaluate the points [i0, i1) of one body (constant per lane and body: which bits of SlotBits to look at)
DEV SlotBits body_slot_mask(const ShfModel* m, int i0, int i1) {
  SlotBits B = {{0ull, 0ull}};
  for (int i = i0; i < i1; i++) {
    const int s = m->pt_slot[i];
    if (s < 64) B.w[0] |= 1ull << s; else B.w[1] |= 1ull << (s - 64);
  }
  return B;
}

// The active contact slots of body range [i0, i1), ascending, picked out of the per-round ballots.
template <int G, int NR, class F>
DEV void for_active_points(const unsigned long long (&active)[NR], int i0, int i1, F f) {
#pragma unroll
  for (int k = 0; k < NR; k++) {
    const int a0 = (i0 > k * G ? i0 : k * G) - k * G, a1 = (i1 < (k + 1) * G ? i1 : (k + 1) * G) - k * G;
    unsigned long long bits = a1 > a0 ? (active[k] >> a0) & (a1 - a0 >= 64 ? ~0ull : ((1ull << (a1 - a0)) - 1ull)) : 0ull;
    while (bits) {
      const int j = __builtin_ctzll(bits);
      bits &= bits - 1ull;
      f(k * G + a0 + j);
    }
  }
}

// Per-lane state of a chain lane that lives across the sub-steps of one env step.
template <class CD>
struct ChainState {
  float q[CD::NLK], qd[CD::NLK], tau[CD::NLK];
};

// One gym.simulate() for one env on the chain mapping.  Lanes l < NCH: chain l; lane NCH: the root; all G lanes:
// contact sample points.  contact_out (LDS, nb x 3) is written when non-null.
template <int G, class CD, bool TW>
DEV void chain_substep(const StepCtx& C, const EnvLds& L, int l, ChainState<CD>& X,
                       const ChainPoints<(CD::NPC + G - 1) / G>& P, const SlotBits (&mine)[CD::NLK], const float* fext,
                       float mu_shape, float* contact_out) {
  static_assert(G > CD::NCH && G <= 64, "a lane per chain plus one for the root");
  constexpr int NCH = CD::NCH, NLK = CD::NLK, NB = CD::NB, NR = (CD::NPC + G - 1) / G;
  const ShfModel* m = C.m;
  const float dt = C.sp.dt;
  const float gon = (float)m->gravity_on;
  const float g[3] = {C.sp.gravity[0] * gon, C.sp.gravity[1] * gon, C.sp.gravity[2] * gon};
  const bool ischain = l < NCH, isroot = l == NCH;
  const int b0 = ischain ? CD::body(l, 0) : 0;   // first body of this lane (slot 0); slots 1.. exist on chain lanes only
  PHASE_BEGIN();

  // ---- kinematics, rigid inertias, external forces: chains walk outwards from the root
  ChainLink K[NLK];
  if (l <= NCH) {
    float Rc[9], pc[3] = {0.0f, 0.0f, 0.0f}, vc[6];
    quat_to_mat(L.root + 3, Rc);
#pragma unroll
    for (int k = 0; k < 3; k++) { vc[k] = L.root[10 + k]; vc[3 + k] = L.root[7 + k]; }
    if (isroot) pose_store(L.pose, Rc, pc, vc);
#pragma unroll
    for (int k = 0; k < NLK; k++) {
      const int b = b0 + k;
      if (ischain) {
        chain_kin_link(m->tpos[b], m->trot[b], m->axis[b], X.q[k], X.qd[k], Rc, pc, vc, K[k].S, K[k].c);
        pose_store(L.pose + b * POSE_STRIDE, Rc, pc, vc);
      }
      if (ischain || k == 0) {
        rigid_inertia_p(m->mass[b], m->com[b], m->inertia[b], Rc, pc, vc, K[k].IA, K[k].pA);
        if (fext) {
          const float F[3] = {fext[3 * b], fext[3 * b + 1], fext[3 * b + 2]};
          chain_ext_force(F, m->com[b], Rc, pc, K[k].pA);
        }
      }
    }
    if (ischain) {
      const int b = b0 + NLK;   // the welded end body: reported pose, contact points and forces; inertia merged into the last link
      chain_kin_weld(m->tpos[b], m->trot[b], Rc, pc);
      pose_store(L.pose + b * POSE_STRIDE, Rc, pc, vc);
      if (fext) {
        const float F[3] = {fext[3 * b], fext[3 * b + 1], fext[3 * b + 2]};
        chain_ext_force(F, m->com[b], Rc, pc, K[NLK - 1].pA);
      }
    }
  }
  GROUP_SYNC();
  PHASE_MARK(1);

  // ---- contact sample points: one lane per evaluation slot and round.  Pass 1 places every round's point and reads its
  // terrain height (all rounds' loads in flight together); a round none of whose points can be within the contact offset
  // (clearance x nz_min above offset + radius: exact, ShfTerrain.nz_min) ends there -- with the points evaluated lowest
  // first (ShfModel.pt_eval) that is every round but the first on a walking robot.  Pass 2: unit normal, gap, response.
  const float kc = C.sp.contact_k, dc = C.sp.contact_d, veps = C.sp.friction_vel;
  const float beta = fmaf(kc, dt, dc);
  const float mu = 0.5f * (mu_shape + C.terr.t.friction);
  const ContactConsts KC = {dt, {g[0], g[1], g[2]}, kc, beta, mu, veps, C.sp.max_depen_vel, C.sp.contact_offset};
  SlotBits act = {{0ull, 0ull}};
  {
    float r[NR][3], gx[NR], gy[NR], dz[NR], n[3];
    bool near[NR];
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
    const int lane0 = (int)(threadIdx.x & 63u) - l;
#pragma unroll
    for (int k = 0; k < NR; k++) {
      if (__ballot(near[k]) == 0ull) continue;      // wave-uniform: nobody in this round is near the ground
      float on = 0.0f;
      if (near[k]) {
        float phi;
        if constexpr (TW) {
          float h;
          terrain_query<true>(C.terr, L.root[0] + r[k][0], L.root[1] + r[k][1], &h, n);
          phi = fmaf(L.root[2] + r[k][2] - h, n[2], -P.rad[k]);
        } else {
          terrain_normal_from_gradient(C.terr, gx[k], gy[k], n);
          phi = fmaf(dz[k], n[2], -P.rad[k]);
        }
        if (phi < KC.offset)
          on = contact_point_response(KC, L.pose + P.body[k] * POSE_STRIDE, r[k], n, P.rad[k], phi, L.pt + P.idx[k] * PT_STRIDE);
      }
      const unsigned long long bits = (__ballot(on != 0.0f) >> lane0) & (G >= 64 ? ~0ull : ((1ull << (G & 63)) - 1ull));
      if (k * G < 64) act.w[0] |= bits << ((k * G) & 63); else act.w[1] |= bits << ((k * G - 64) & 63);
    }
  }
  GROUP_SYNC();
  PHASE_MARK(3);

  // ---- fold the active slots into their moving bodies, in ascending point order whatever slot evaluated them
  if (l <= NCH) {
#pragma unroll
    for (int k = 0; k < NLK; k++) {
      if ((ischain || k == 0) && ((act.w[0] & mine[k].w[0]) | (act.w[1] & mine[k].w[1])) != 0ull) {
        const int b = b0 + k, i0 = m->pt_start[b], i1 = i0 + m->pt_count[b];
        for (int i = i0; i < i1; i++)
          if (act.test(m->pt_slot[i])) contact_accumulate_p(L.pt + i * PT_STRIDE, dt, K[k].IA, K[k].pA);
      }
    }
  }
  PHASE_MARK(4);

  // ---- inward pass along the chain, in registers; the first link's result goes to the root through LDS
  if (ischain) {
#pragma unroll
    for (int k = NLK - 1; k >= 0; k--) {
      float t0, de, pa[6];
      chain_dof_effort(C, l * NLK + k, X.q[k], X.qd[k], X.tau[k], &t0, &de);
      chain_inward_link(K[k], de, t0, pa);
      if (k > 0) {
#pragma unroll
        for (int j = 0; j < 21; j++) K[k - 1].IA[j] += K[k].IA[j];
#pragma unroll
        for (int j = 0; j < 6; j++) K[k - 1].pA[j] += pa[j];
      } else {
        float* o = L.xch + l * XCH_STRIDE;
#pragma unroll
        for (int j = 0; j < 21; j++) o[j] = K[0].IA[j];
#pragma unroll
        for (int j = 0; j < 6; j++) o[21 + j] = pa[j];
      }
    }
  }
  GROUP_SYNC();
  PHASE_MARK(6);

  // ---- root: children in child_list order (= chain order, ChainDims::matches), 6x6 solve
  float a[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
  if (isroot) {
#pragma unroll
    for (int c = 0; c < NCH; c++) {
      const float* o = L.xch + c * XCH_STRIDE;
#pragma unroll
      for (int j = 0; j < 21; j++) K[0].IA[j] += o[j];
#pragma unroll
      for (int j = 0; j < 6; j++) K[0].pA[j] += o[21 + j];
    }
    ldlt_solve6(K[0].IA, K[0].pA, a);
#pragma unroll
    for (int j = 0; j < 6; j++) L.acc[j] = a[j];
  }
  GROUP_SYNC();
  PHASE_MARK(7);

  // ---- outward pass and integration of the joints
  if (ischain) {
    float ap[6];
#pragma unroll
    for (int j = 0; j < 6; j++) ap[j] = L.acc[j];
#pragma unroll
    for (int k = 0; k < NLK; k++) {
#pragma unroll
      for (int j = 0; j < 6; j++) ap[j] = ap[j] + K[k].c[j];
      float ua = K[k].U[0] * ap[0];
#pragma unroll
      for (int j = 1; j < 6; j++) ua = fmaf(K[k].U[j], ap[j], ua);
      const float qdd = (K[k].u - ua) * K[k].invD;
#pragma unroll
      for (int j = 0; j < 6; j++) ap[j] = fmaf(K[k].S[j], qdd, ap[j]);
      if (contact_out) {
        float* o = L.acc + (b0 + k) * 6;
#pragma unroll
        for (int j = 0; j < 6; j++) o[j] = ap[j];
      }
      const float vl = m->vel_limit[l * NLK + k];
      const float qdn = rclampf(fmaf(dt, qdd, X.qd[k]), -vl, vl);
      X.qd[k] = qdn;
      X.q[k] = fmaf(dt, qdn, X.q[k]);
    }
  }
  PHASE_MARK(8);

  // ---- net contact force per reported body (the sub-step whose forces the task reads)
  if (contact_out) {
    GROUP_SYNC();
#pragma unroll
    for (int k = 0; k < NR; k++) {
      if (P.idx[k] >= 0 && act.test(l + k * G)) contact_force_final(L.pt + P.idx[k] * PT_STRIDE, L.acc + m->dyn[P.body[k]] * 6, dt);
    }
    GROUP_SYNC();
    for (int b = l; b < NB; b += G) {
      float f[3] = {0.0f, 0.0f, 0.0f};
      const int dl = m->dyn[b];
      const int i0 = m->pt_start[dl], i1 = i0 + m->pt_count[dl];
      if ((act.w[0] | act.w[1]) != 0ull) {
        for (int i = i0; i < i1; i++) {
          if (m->pt_body[i] != b || !act.test(m->pt_slot[i])) continue;
          const float* o = L.pt + i * PT_STRIDE;
          f[0] += o[PT_F]; f[1] += o[PT_F + 1]; f[2] += o[PT_F + 2];
        }
      }
      contact_out[3 * b] = f[0]; contact_out[3 * b + 1] = f[1]; contact_out[3 * b + 2] = f[2];
    }
  }
  PHASE_MARK(9);

  // ---- floating base: semi-implicit Euler (oracle substep(), last block)
  if (isroot) {
    float* Rt = L.root;
    float ang[3] = {Rt[10], Rt[11], Rt[12]}, lin[3] = {Rt[7], Rt[8], Rt[9]}, wxv[3];
    cross3(ang, lin, wxv);
    const float damp = 1.0f / fmaf(dt, C.sp.angular_damping, 1.0f);
    float wn[3], vn[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
      wn[k] = fmaf(dt, a[k], ang[k]) * damp;
      vn[k] = fmaf(dt, a[3 + k] + g[k] + wxv[k], lin[k]);
    }
    const float wmag = sqrtf(dot3(wn, wn)), wmax = C.sp.max_ang_vel;
    if (wmag > wmax) {
      const float sc2 = wmax / wmag;
#pragma unroll
      for (int k = 0; k < 3; k++) wn[k] *= sc2;
    }
#pragma unroll
    for (int k = 0; k < 3; k++) { Rt[10 + k] = wn[k]; Rt[7 + k] = vn[k]; Rt[k] = fmaf(dt, vn[k], Rt[k]); }
    const float hx = 0.5f * dt * wn[0], hy = 0.5f * dt * wn[1], hz = 0.5f * dt * wn[2];
    const float x = Rt[3], y = Rt[4], z = Rt[5], ww = Rt[6];
    const float nx = x + fmaf(hx, ww, fmaf(hy, z, -(hz * y)));
    const float ny = y + fmaf(hy, ww, fmaf(hz, x, -(hx * z)));
    const float nz = z + fmaf(hz, ww, fmaf(hx, y, -(hy * x)));
    const float nw = ww - fmaf(hx, x, fmaf(hy, y, hz * z));
    const float inv = 1.0f / sqrtf(fmaf(nw, nw, fmaf(nz, nz, fmaf(ny, ny, nx * nx))));
    Rt[3] = nx * inv; Rt[4] = ny * inv; Rt[5] = nz * inv; Rt[6] = nw * inv;
  }
  GROUP_SYNC();
  PHASE_MARK(10);
}

// gym.refresh_rigid_body_state_tensor for one env on the chain mapping: rows -> `stage` (LDS, nb x 13)
template <int G, class CD>
DEV void chain_body_states(const ShfModel* m, const EnvLds& L, int l, const ChainState<CD>& X, float* stage) {
  constexpr int NCH = CD::NCH, NLK = CD::NLK;
  auto row = [&](int b, const float* Rw, const float* p, const float* v) {
    float* o = stage + 13 * b;
    float t[3], q[4];
#pragma unroll
    for (int k = 0; k < 3; k++) o[k] = L.root[k] + p[k];
    mat_to_quat(Rw, q);
#pragma unroll
    for (int k = 0; k < 4; k++) o[3 + k] = q[k];
    cross3(v, p, t);
#pragma unroll
    for (int k = 0; k < 3; k++) { o[7 + k] = v[3 + k] + t[k]; o[10 + k] = v[k]; }
  };
  if (l <= NCH) {
    float Rc[9], pc[3] = {0.0f, 0.0f, 0.0f}, vc[6], S[6], c[6];
    quat_to_mat(L.root + 3, Rc);
#pragma unroll
    for (int k = 0; k < 3; k++) { vc[k] = L.root[10 + k]; vc[3 + k] = L.root[7 + k]; }
    if (l == NCH) {
      row(0, Rc, pc, vc);
    } else {
      const int b0 = CD::body(l, 0);
#pragma unroll
      for (int k = 0; k < NLK; k++) {
        const int b = b0 + k;
        chain_kin_link(m->tpos[b], m->trot[b], m->axis[b], X.q[k], X.qd[k], Rc, pc, vc, S, c);
        row(b, Rc, pc, vc);
      }
      const int b = b0 + NLK;
      chain_kin_weld(m->tpos[b], m->trot[b], Rc, pc);
      row(b, Rc, pc, vc);
    }
  }
  GROUP_SYNC();
}

// ShifuVecEnv.step for the A1 task (env.py:85-106) on the chain mapping; the task glue after the physics is shared
// with the body-mapped kernel (a1_post_step, shf_task.h).
template <int G, class CD, bool TW>
DEV void a1_chain_step_body(const A1Args& A) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NLK = CD::NLK, NCH = CD::NCH, nb = CD::NB, nd = CD::ND, np = CD::NPC, NR = (CD::NPC + G - 1) / G;
  PHASE_BEGIN();
  float* stats_lds = smem + MODEL_WORDS + TASK_WORDS;
  stats_block_init(stats_lds);
  const unsigned long long stats_step = stats_step_load(A.stats);
  const int epb = 256 / G, es = threadIdx.x / G, l = threadIdx.x % G;
  const int e = blockIdx.x * epb + es;
  const int n = A.S.n;
  // the env's state and action loads go out before the model is staged: their round trip overlaps the staging copy
  constexpr int NW = (2 * nd + G - 1) / G;
  float pre_dof[NW], pre_root = 0.0f, pre_act = 0.0f;
#pragma unroll
  for (int k = 0; k < NW; k++) pre_dof[k] = 0.0f;
  static_assert(G >= 13 && G >= nd, "one root word and one action per lane");
  if (e < n) {
#pragma unroll
    for (int k = 0; k < NW; k++)
      if (l + k * G < 2 * nd) pre_dof[k] = A.S.dof[(size_t)e * nd * 2 + l + k * G];
    if (l < 13) pre_root = A.S.root[(size_t)e * 13 + l];
    if (l < nd) pre_act = A.raw_actions[(size_t)e * nd + l];
  }
  stage_block<(int)sizeof(ShfA1TaskParams)>(A.tp, smem + MODEL_WORDS);
  const ShfModel* m = stage_model(A.S.model, smem);
  const ShfA1TaskParams& tp = *reinterpret_cast<const ShfA1TaskParams*>(smem + MODEL_WORDS);
  if (e >= n) return;
  const int H = tp.num_history, P = tp.num_height_points;
  const int nobs = 12 + 2 * nd + nd * H + P;
  const int env_words = chain_lds_words<CD>(SCR_OBS + nobs);
  EnvLds L = chain_lds_carve<CD>(smem + MODEL_WORDS + TASK_WORDS + STATS_LDS_WORDS + es * env_words);
  float* scr = L.pt;

#pragma unroll
  for (int k = 0; k < NW; k++) {
    const int i = l + k * G;
    if (i < 2 * nd) L.dofb[(i >> 1) * DOF_STRIDE + (i & 1)] = pre_dof[k];
  }
  if (l < 13) L.root[l] = pre_root;
  float act = 0.0f;
  if (l < nd) {
    act = rclampf(pre_act * tp.action_scale, -tp.clip_actions, tp.clip_actions);
    A.actions[(size_t)e * nd + l] = act;
    L.dofb[l * DOF_STRIDE + 2] = act;   // parked for the chain lanes (the slot is otherwise unused on this mapping)
  }
  GROUP_SYNC();

  // Q2: base-frame velocities from the pre-physics root state (robot.py:222-229)
  if (l == 0) {
    const float gv[3] = {0.0f, 0.0f, -1.0f};
    float pg[3], blv[3], bav[3];
    quat_rotate_inverse(L.root + 3, L.root + 7, blv);
    quat_rotate_inverse(L.root + 3, L.root + 10, bav);
    quat_rotate_inverse(L.root + 3, gv, pg);
    float* o = A.base_vel + (size_t)e * 9;
#pragma unroll
    for (int k = 0; k < 3; k++) { o[k] = blv[k]; o[3 + k] = bav[k]; o[6 + k] = pg[k]; }
  }

  StepCtx C;
  C.m = m; C.sp = A.S.sp; C.terr.t = A.S.terr; C.terr.h = A.S.heights; C.scene = nullptr;
  const float mu = A.S.friction[e];
  const int nsub = tp.decimation + (tp.extra_substep ? 1 : 0);
  ChainState<CD> X;
  float acts[NLK], pg_[NLK], dg_[NLK], q0_[NLK], lim_[NLK];
#pragma unroll
  for (int k = 0; k < NLK; k++) {
    const int d = l < NCH ? l * NLK + k : 0;
    X.q[k] = L.dofb[d * DOF_STRIDE]; X.qd[k] = L.dofb[d * DOF_STRIDE + 1]; X.tau[k] = 0.0f;
    acts[k] = L.dofb[d * DOF_STRIDE + 2];
    pg_[k] = tp.p_gain[d]; dg_[k] = tp.d_gain[d]; q0_[k] = tp.default_dof_pos[d]; lim_[k] = m->effort[d];
  }
  ChainPoints<NR> LP;
  chain_points_load<G>(m, np, l, C.sp.contact_offset, LP);
  SlotBits mine[NLK];   // evaluation slots of this lane's bodies (chain lanes: their links; root lane: the root)
#pragma unroll
  for (int k = 0; k < NLK; k++) {
    const int b = l < NCH ? CD::body(l, k) : 0;
    const bool has = l < NCH || (l == NCH && k == 0);
    mine[k] = body_slot_mask(m, m->pt_start[b], has ? m->pt_start[b] + m->pt_count[b] : m->pt_start[b]);
  }
  PHASE_MARK(11);
  for (int it = 0; it < nsub; it++) {
    if (it < tp.decimation) {
      // A1Robot.step's explicit PD (a1_conditional.py:66-67); the extra refresh_state sub-step keeps the last torque (Q1)
#pragma unroll
      for (int k = 0; k < NLK; k++) {
        const float t = pg_[k] * (acts[k] + q0_[k] - X.q[k]) - dg_[k] * X.qd[k];
        X.tau[k] = rclampf(t, -lim_[k], lim_[k]);
      }
    }
    chain_substep<G, CD, TW>(C, L, l, X, LP, mine, (it == tp.decimation) ? A.push + (size_t)e * nb * 3 : nullptr, mu,
                             (it == nsub - 1) ? L.xch : nullptr);
  }
  PHASE_RESET();
  if (l < NCH) {
#pragma unroll
    for (int k = 0; k < NLK; k++) {
      float* D = L.dofb + (l * NLK + k) * DOF_STRIDE;
      D[0] = X.q[k]; D[1] = X.qd[k]; D[5] = X.tau[k];
    }
  }
  GROUP_SYNC();
  if (l < nd) A.torques[(size_t)e * nd + l] = L.dofb[l * DOF_STRIDE + 5];
  for (int i = l; i < 3 * nb; i += G) A.S.contact[(size_t)e * nb * 3 + i] = L.xch[i];
  // the contact-point region is idle from here on: it becomes scratch
  for (int i = l; i < nd * H; i += G) scr[SCR_HIST + i] = A.history[(size_t)e * nd * H + i];
  if (l < nd) scr[SCR_ACT + l] = act;
  PHASE_MARK(12);
  chain_body_states<G, CD>(m, L, l, X, scr + SCR_BODY);
  for (int i = l; i < 13 * nb; i += G) A.body_state[(size_t)e * nb * 13 + i] = scr[SCR_BODY + i];
  PHASE_RESET();
  a1_post_step<G>(A, m, tp, L, scr, e, l, stats_lds, stats_step);
}
