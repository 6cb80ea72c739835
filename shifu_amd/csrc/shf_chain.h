// shf_chain.h -- the fused A1 step with the tree's recursions on one lane per kinematic CHAIN.
//
// Tree shape served: a floating root with NCH serial chains of NLK revolute links, each chain ending in one welded
// reported body (the Unitree A1: trunk + 4 legs of hip / thigh / calf + foot).  The body-per-lane
// mapping of shf_device.h executes the tree level by level -- every level is one pass of the whole wavefront with a
// quarter of its lanes active and an LDS hand-off (write, fence, read: 70-110 clocks, tools/valu_microbench) on either
// side, ~20 group synchronisations per sub-step.  Here each kind of work runs where it is widest, and the recursions,
// which are serial whatever the mapping, run without hand-offs:
//   dof lanes   (lane j < ND)        joint j: integration, drive effort (t0, de), local rotation Rl = trot * Rot(axis, q)
//   chain lanes (lane c < NCH;       chain c: pose composition root -> tip, the ABA inward pass tip -> root and the
//    at 32 lanes: lanes 8 c)         outward pass, link after link with (S, c, U, 1/D, u) in registers
//   row lanes   (32 lanes per env:   the inward pass row-parallel: lane 8 c + i holds row i of chain c's articulated inertia
//    lanes 8 c + i, i < 6)           (6 fused operations per product / update instead of 36 / 21; chain_substep)
//   body lanes  (lane j <= ND)       moving body j (link j, or the root on lane ND): rigid inertia, external force,
//                                    fold of its active contacts; the root lane also solves the 6x6 and integrates the base
//   point lanes (all G lanes)        one contact sample point per lane and round
// Seven group synchronisations per sub-step.  G = 16 packs four envs into a wavefront (4096 envs = 1024 waves = one
// per SIMD); a single wave issues one VALU instruction per ~4.6 clocks whatever its active lanes (profiles/
// r03_valu_microbench.md), so what counts is the length of the instruction stream, not lane utilisation.
//
// ARITHMETIC: the same operations in the same order as oracle/shf_oracle.c (and as the body-mapped kernel) -- only
// which lane executes them changes, so every result stays bit-identical; tests/test_gpu_parity.py holds both mappings to
// the oracle.  Reference call sites replaced: gym.simulate (examples/a1_conditional/a1_conditional.py:69,
// shifu/gym/isaac_gym.py:140) inside ShifuVecEnv.step (shifu/gym/env.py:85-106).
#pragma once
#include "shf_task.h"
#include "shf_link.h"

template <int NCH_, int NLK_, int NP_, int NEV_>
struct ChainDims {
  static constexpr int NCH = NCH_, NLK = NLK_, NB = 1 + NCH_ * (NLK_ + 1), ND = NCH_ * NLK_, NPC = NP_;
  static constexpr int NEV = NEV_;  // evaluation slots (ShfModel.neval): the points plus the padding of the 32-slot blocks
  static constexpr int MAXPT = 9;   // contact points per moving body held as a packed slot list (7 bits each in 64)
  static_assert(NEV_ >= NP_ && NEV_ <= 128, "evaluation slots: two 64-bit ballot words");
  DEV static constexpr int body(int c, int k) { return 1 + c * (NLK_ + 1) + k; }   // k == NLK: the welded end body
  static bool matches(const ShfModel& m) {
    if (m.nb != NB || m.nd != ND || m.np != NPC || m.fixed_base || m.jtype[0] != SHF_JOINT_ROOT) return false;
    if (m.child_count[0] != NCH || m.nlevels != NLK) return false;
    if ((m.neval > 0 ? m.neval : m.np) != NEV) return false;
    int occupied = 0;
    for (int s = 0; s < NEV; s++) {   // pt_eval and pt_slot: each other's inverse over the occupied slots
      const int i = m.pt_eval[s];
      if (i == -1) continue;
      if (i < 0 || i >= NPC || m.pt_slot[i] != s) return false;
      occupied++;
    }
    if (occupied != NPC) return false;
    for (int b = 0; b < NB; b++) {
      const int n = m.pt_count[b], i0 = m.pt_start[b];
      if (n > MAXPT) return false;
      if (n == 0 || m.dyn[b] != b) continue;
      // the packed form: a moving body's points in consecutive slots, in point order, inside one block of 32
      const int s0 = m.pt_slot[i0];
      for (int j = 0; j < n; j++)
        if (m.pt_slot[i0 + j] != s0 + j) return false;
      if ((s0 >> 5) != ((s0 + n - 1) >> 5)) return false;
    }
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
typedef ChainDims<4, 3, 76, 88> A1Chain;   // the A1: 76 sample points in 88 slots (three blocks: 27 + 25 + 24)

// LDS of one env: pose records of all reported bodies, accelerations, one (IA, pA) exchange slot per link (body lane ->
// chain lane; the chain's first slot then carries its result to the root; the region is also the net-contact-force
// staging, nb x 3 floats), a record per joint, the dof block (epilogue layout), the root state, contact slots.
// Joint record: Rl[9] qd . . t0 de -- and, once the chain lane has composed the link (32 lanes per env), S[6] c[6] in the
// place of Rl and qd, for the row lanes of the inward pass and for the outward pass.
#define JREC_STRIDE 16
#define JREC_QD 9
#define JREC_T0 12
#define JREC_DE 13
#define JREC_S 0
#define JREC_C 6
// The chain kernels stage the model up to (not including) its link-contact records -- a single actor per env has no box
// actors to meet -- which leaves room for the self-collision slots in the 80 KB a workgroup may use at two per CU.
#define CHAIN_MODEL_BYTES ((int)offsetof(ShfModel, link_collide))
#define CHAIN_MODEL_WORDS ((CHAIN_MODEL_BYTES / 4 + 3) & ~3)
static_assert(CHAIN_MODEL_BYTES % 16 == 0, "stage_block copies 16-byte words");
template <class CD>
__host__ __device__ inline int chain_lds_words(int min_tail, bool self = false) {
  int tail = (CD::NPC + (self ? SHF_MAX_SELF_CONTACTS : 0)) * PT_STRIDE;
  if (tail < min_tail) tail = min_tail;
  const int w = CD::NB * POSE_STRIDE + ((CD::NB * 6 + 3) & ~3) + (CD::ND + 1) * XCH_STRIDE + CD::ND * JREC_STRIDE +
                ((CD::ND * DOF_STRIDE + 3) & ~3) + root_words(1) + tail;
  return (w + 3) & ~3;
}
struct ChainLds : EnvLds {
  float* jrec;
  float* xroot;   // the root's (IA, pA) for the element-wise sum with the chains' contributions
};
template <class CD>
DEV ChainLds chain_lds_carve(float* base) {
  static_assert(CD::ND * XCH_STRIDE >= 3 * CD::NB, "the exchange slots double as the contact-force staging");
  ChainLds L;
  L.pose = base;
  L.acc = L.pose + CD::NB * POSE_STRIDE;
  L.xch = L.acc + ((CD::NB * 6 + 3) & ~3);
  L.xroot = L.xch + CD::ND * XCH_STRIDE;
  L.jrec = L.xroot + XCH_STRIDE;
  L.dofb = L.jrec + CD::ND * JREC_STRIDE;
  L.root = L.dofb + ((CD::ND * DOF_STRIDE + 3) & ~3);
  L.pt = L.root + root_words(1);
  return L;
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
DEV void chain_dof_effort(const StepCtx& C, int d, float q, float qd, float tau_cmd, float* t0o, float* deo, float tq = 0.0f, float tv_in = 0.0f) {
  const ShfModel* m = C.m;
  const float dt = C.sp.dt;
  float t0 = 0.0f, de = m->armature[d];
  const int mode = m->drive_mode[d];
  if (mode == SHF_DOF_MODE_EFFORT) {
    t0 = tau_cmd;
  } else if (mode == SHF_DOF_MODE_POS || mode == SHF_DOF_MODE_VEL) {
    // the fused A1 step sets no drive targets (they read as zero, like the body-mapped kernel's null target pointers); the
    // hook path's simulate hands them in (tq; tv counts in VEL mode only, oracle substep())
    float kp = mode == SHF_DOF_MODE_POS ? m->kp[d] : 0.0f, kd = m->kd[d];
    const float tv = mode == SHF_DOF_MODE_VEL ? tv_in : 0.0f;
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

// Sample-point constants of a lane's evaluation slots (slot l + k*G evaluates point pt_eval[slot], ShfModel.pt_eval).
template <int NR>
struct ChainPoints {
  int idx[NR];         // point index, -1 for an empty slot
  int body[NR];        // reported body whose pose places the point
  float pos[NR][3], rad[NR], thr[NR];   // thr: clearance x nz_min above which the point cannot be within the contact offset
};
template <int G, int NR>
DEV void chain_points_load(const ShfModel* m, int nev, int l, float offset, ChainPoints<NR>& P) {
#pragma unroll
  for (int k = 0; k < NR; k++) {
    const int s = l + k * G;
    const int e = s < nev ? m->pt_eval[s] : -1;   // -1: padding of a 32-slot block
    const int i = e < 0 ? 0 : e;
    P.idx[k] = e;
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
  const float nz = rsqrt_spec(fmaf(gy, gy, fmaf(gx, gx, 1.0f)));
  n[0] = -gx * nz; n[1] = -gy * nz; n[2] = nz;
}

// Active evaluation slots of one env as two 64-bit words (slot s = bit s).
struct SlotBits {
  unsigned long long w[2];
  DEV bool test(int s) const { return ((s < 64 ? w[0] >> s : w[1] >> (s - 64)) & 1ull) != 0ull; }
  // the n < 32 slots from s0 on, which lie inside one block of 32 (the packed form of ShfModel.pt_eval): bit j = slot s0 + j
  DEV unsigned field(int s0, int n) const {
    const unsigned long long ww = s0 < 64 ? w[0] : w[1];
    return (unsigned)(ww >> (s0 & 63)) & ((1u << n) - 1u);
  }
};
// The evaluation slots of one body's points [i0, i0 + n), in point order, 7 bits each (constant per lane: read once per
// env step), and the same as a mask over SlotBits.
template <int MAXPT>
struct SlotList {
  unsigned long long list;
  int i0, n, s0;       // s0: slot of the first point (the packed form: point i0 + j sits in slot s0 + j)
  SlotBits mask;
  DEV int slot(int j) const { return (int)((list >> (7 * j)) & 127ull); }
};
template <int MAXPT>
DEV SlotList<MAXPT> slot_list_load(const ShfModel* m, int i0, int n) {
  static_assert(MAXPT * 7 <= 64, "packed slot list");
  SlotList<MAXPT> S;
  S.list = 0ull; S.i0 = i0; S.n = n; S.s0 = m->pt_slot[n > 0 ? i0 : 0]; S.mask.w[0] = 0ull; S.mask.w[1] = 0ull;
  int sl[MAXPT];
#pragma unroll
  for (int j = 0; j < MAXPT; j++) sl[j] = m->pt_slot[j < n ? i0 + j : 0];
#pragma unroll
  for (int j = 0; j < MAXPT; j++) {
    if (j < n) {
      const int s = sl[j];
      S.list |= (unsigned long long)s << (7 * j);
      if (s < 64) S.mask.w[0] |= 1ull << s; else S.mask.w[1] |= 1ull << (s - 64);
    }
  }
  return S;
}

// Per-lane state that lives across the sub-steps of one env step: lane j < ND is dof j.
struct DofLane {
  float q, qd, tau;   // joint position, velocity, commanded (explicit) effort
  float tq, tv;       // POS / VEL drive targets (hook path; zero in the fused step)
};

// Row lanes of the inward pass at 32 lanes per env: lane 8 c + i (i < 6) holds row i of chain c's 6x6 articulated inertia.
// off[j]: the packed-storage index of element (i, j).
struct RowLane {
  int c, i;
  bool on;
  int off[6];
};
template <class CD>
DEV RowLane row_lane_load(int l) {
  RowLane R;
  R.c = l >> 3; R.i = l & 7;
  R.on = R.i < 6 && R.c < CD::NCH;
  const int i = R.on ? R.i : 0;
#pragma unroll
  for (int j = 0; j < 6; j++) {
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    R.off[j] = lo * 6 - (lo * (lo - 1)) / 2 + (hi - lo);
  }
  return R;
}

// One gym.simulate() for one env.  Lane roles in the file header; contact_out (LDS, nb x 3) is written when non-null.
// At 32 lanes per env (ROWS) the chain lanes are lanes 0, 8, 16, 24 and the ABA inward pass runs row-parallel: the six lanes
// 8 c .. 8 c + 5 each hold one row of chain c's articulated inertia, so that IA S, the rank-1 update and IA c cost 6 fused
// operations per link instead of 36 + 21 + 36; U and pA travel between the rows through LDS.  Every element still sees the
// operations of chain_inward_link in its order (an element below the diagonal repeats its mirror image's: same operands).
template <int G, class CD, bool TW, bool SELF = false>
DEV void chain_substep(const StepCtx& C, const ChainLds& L, int l, DofLane& X, const ChainPoints<(CD::NEV + G - 1) / G>& P,
                       const SlotList<CD::MAXPT>& mine, const RowLane& RL, const float* fext, float mu_shape, float* contact_out) {
  static_assert(G > CD::ND && G <= 64, "a lane per dof plus one for the root");
  constexpr int NCH = CD::NCH, NLK = CD::NLK, NB = CD::NB, ND = CD::ND, NR = (CD::NEV + G - 1) / G;
  const ShfModel* m = C.m;
  const float dt = C.sp.dt;
  const float gon = (float)m->gravity_on;
  const float g[3] = {C.sp.gravity[0] * gon, C.sp.gravity[1] * gon, C.sp.gravity[2] * gon};
  // With 32 lanes per env each moving body has two lanes (lb and lb + 16) that share its 27 accumulators between them
  // (contact_accumulate_half): the rigid inertia is computed by both, each folds the contacts into the elements it owns.
  constexpr bool SPLIT = G >= 32;
  constexpr bool ROWS = G == 32;
  static_assert(!SPLIT || ND < 16, "second-half body lanes start at lane 16");
  static_assert(!ROWS || (NCH * 8 <= G && NB * 6 >= NCH * 12), "row lanes: eight lanes per chain; U / pA exchange in the acc region");
  const int lb = SPLIT ? (l & 15) : l;                       // body-lane index: link lb < ND, root lb == ND
  const int half = SPLIT ? ((l >> 4) & 1) : 0;
  const bool isdof = l < ND, isroot = l == ND;
  const bool ischain = ROWS ? ((l & 7) == 0 && (l >> 3) < NCH) : l < NCH;
  const int ci = ROWS ? (l >> 3) : l;                        // chain of a chain lane
  const bool isbody = SPLIT ? (lb <= ND && l < 32) : l <= ND;
  const bool islink = isbody && lb < ND;
  const int myb = lb < ND ? CD::body(lb / NLK, lb % NLK) : 0;   // moving body of this body lane (lb == ND: the root)
  PHASE_BEGIN();

  // ---- A. dof lanes: drive effort and the joint's local rotation -> joint record
  if (isdof) {
    float* rec = L.jrec + l * JREC_STRIDE;
    float Rl[9], t0, de;
    chain_dof_effort(C, l, X.q, X.qd, X.tau, &t0, &de);
    joint_local_rotation(m->trot[myb], m->axis[myb], X.q, Rl);
#pragma unroll
    for (int k = 0; k < 9; k++) rec[k] = Rl[k];
    rec[JREC_QD] = X.qd; rec[JREC_T0] = t0; rec[JREC_DE] = de;
  }
  GROUP_SYNC();
  PHASE_MARK(0);

  // ---- B. chain lanes: poses, velocities, motion subspaces root -> tip; the root lane publishes the root's pose
  ChainLink K[ROWS ? 1 : NLK];   // (ROWS: S and c go to the joint records instead)
  if (ischain || isroot) {
    float Rc[9], pc[3] = {0.0f, 0.0f, 0.0f}, vc[6];
    quat_to_mat(L.root + 3, Rc);
#pragma unroll
    for (int k = 0; k < 3; k++) { vc[k] = L.root[10 + k]; vc[3 + k] = L.root[7 + k]; }
    if (isroot && !(ROWS && ischain)) {
      pose_store(L.pose, Rc, pc, vc);
    } else {
      const int b0 = CD::body(ci, 0);
#pragma unroll
      for (int k = 0; k < NLK; k++) {
        const int b = b0 + k;
        float* rec = L.jrec + (ci * NLK + k) * JREC_STRIDE;
        float Rl[9];
#pragma unroll
        for (int j = 0; j < 9; j++) Rl[j] = rec[j];
        if constexpr (ROWS) {
          float Sx[6], cx[6];
          chain_compose_link(m->tpos[b], m->axis[b], Rl, rec[JREC_QD], Rc, pc, vc, Sx, cx);
#pragma unroll
          for (int j = 0; j < 6; j++) { rec[JREC_S + j] = Sx[j]; rec[JREC_C + j] = cx[j]; }
        } else {
          chain_compose_link(m->tpos[b], m->axis[b], Rl, rec[JREC_QD], Rc, pc, vc, K[k].S, K[k].c);
        }
        pose_store(L.pose + b * POSE_STRIDE, Rc, pc, vc);
      }
      const int b = b0 + NLK;   // the welded end body: reported pose, contact points and forces; inertia merged into the last link
      chain_kin_weld(m->tpos[b], m->trot[b], Rc, pc);
      pose_store(L.pose + b * POSE_STRIDE, Rc, pc, vc);
    }
  }
  GROUP_SYNC();
  PHASE_MARK(1);

  // ---- C. body lanes: rigid inertia and bias force of their moving body, external forces (body order within a moving body)
  float IA[21], pA[6];
  if (isbody) {
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
      if (islink && (lb % NLK) == NLK - 1) {   // the chain's last link also carries its welded end body
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
  }
  PHASE_MARK(2);

  // ---- contact sample points: one lane per evaluation slot and round.  Pass 1 places every round's point and reads its
  // terrain height (all rounds' loads in flight together); a round none of whose points can be within the contact offset
  // (clearance x nz_min above offset + radius: exact, ShfTerrain.nz_min) ends there.  Pass 2: unit normal, gap, response.
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
  // self-collision (ShfModel.self_collide; csrc/shf_boxes.h): capsule pairs, one lane each, the active ones in the slots
  // behind the sample points
  int nself = 0;
  if constexpr (SELF) nself = self_contacts_eval<G>(C, L, l, CD::NPC, mu_shape);
  GROUP_SYNC();
  PHASE_MARK(3);

  // ---- D. body lanes: fold their active contacts, ascending point order whatever slot evaluated them; links hand
  // (IA, pA) to their chain lane.  SPLIT: the body's slots are one bit field of the ballot words (ShfModel.pt_eval packs
  // them), so a lane visits exactly its active points -- the wave runs as many rounds as its busiest body has contacts.
  if (isbody) {
    if constexpr (SPLIT) {
      unsigned bits = act.field(mine.s0, mine.n);
      while (bits) {
        const int j = __builtin_ctz(bits);
        bits &= bits - 1u;
        const float* o = L.pt + (mine.i0 + j) * PT_STRIDE;
        if (half == 0) contact_accumulate_half<0>(o, dt, IA, pA);
        else contact_accumulate_half<1>(o, dt, IA, pA);
      }
    } else if (((act.w[0] & mine.mask.w[0]) | (act.w[1] & mine.mask.w[1])) != 0ull) {
      for (int j = 0; j < mine.n; j++) {
        if (!act.test(mine.slot(j))) continue;
        contact_accumulate_p(L.pt + (mine.i0 + j) * PT_STRIDE, dt, IA, pA);
      }
    }
    if constexpr (SELF) {
      // ... then the self-contacts of its body, slot order, +f on capsule a's body and -f on b's, each side scaled by
      // 1 + (own mass) / (other mass) (oracle: self_scales)
      for (int k = 0; k < nself; k++) {
        const float* o = L.pt + (CD::NPC + k) * PT_STRIDE;
        const int p = (int)o[PT_ON] - 1;
        const int da = m->dyn[m->cap_body[m->pair_a[p]]], db = m->dyn[m->cap_body[m->pair_b[p]]];
        if (da != myb && db != myb) continue;
        const float ma = body_mass(C, da), mb = body_mass(C, db);
        if constexpr (SPLIT) {
          if (half == 0) {
            if (da == myb) slot_accumulate_fb_half<0>(IA, pA, o, 1.0f, dt, 1.0f, 1.0f + ma / mb);
            if (db == myb) slot_accumulate_fb_half<0>(IA, pA, o, -1.0f, dt, 1.0f, 1.0f + mb / ma);
          } else {
            if (da == myb) slot_accumulate_fb_half<1>(IA, pA, o, 1.0f, dt, 1.0f, 1.0f + ma / mb);
            if (db == myb) slot_accumulate_fb_half<1>(IA, pA, o, -1.0f, dt, 1.0f, 1.0f + mb / ma);
          }
        } else {
          if (da == myb) slot_accumulate_fb(IA, pA, o, 1.0f, dt, 1.0f, 1.0f + ma / mb);
          if (db == myb) slot_accumulate_fb(IA, pA, o, -1.0f, dt, 1.0f, 1.0f + mb / ma);
        }
      }
    }
    // links hand (IA, pA) to their chain lane; the root's go to the same kind of slot for the element-wise sum below
    float* o = (islink ? L.xch + lb * XCH_STRIDE : L.xroot);
    if (!SPLIT) {
#pragma unroll
      for (int j = 0; j < 21; j++) o[j] = IA[j];
#pragma unroll
      for (int j = 0; j < 6; j++) o[21 + j] = pA[j];
    } else if (half == 0) {
#pragma unroll
      for (int j = 0; j < 11; j++) o[j] = IA[j];
    } else {
#pragma unroll
      for (int j = 11; j < 21; j++) o[j] = IA[j];
#pragma unroll
      for (int j = 0; j < 6; j++) o[21 + j] = pA[j];
    }
  }
  GROUP_SYNC();
  PHASE_MARK(4);

  // ---- E. inward pass tip -> root; the first link's result goes to the root through LDS
  float Ug[ROWS ? NLK : 1][6], invDk[ROWS ? NLK : 1], uk[ROWS ? NLK : 1];   // ROWS: what the outward pass needs, per link
  if constexpr (ROWS) {
    if (RL.on) {
      float Ic[6], pc = 0.0f;   // this row of the running child contribution
      float* sx = L.acc + RL.c * 12;   // U[6] pA[6] of the link in hand, row lanes -> every row lane (acc is idle until F)
#pragma unroll
      for (int k = NLK - 1; k >= 0; k--) {
        const float* o = L.xch + (RL.c * NLK + k) * XCH_STRIDE;
        const float* rec = L.jrec + (RL.c * NLK + k) * JREC_STRIDE;
        float row[6], S[6], cc[6], plg[6], Wg[6];
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
        for (int j = 0; j < 6; j++) { Ug[k][j] = sx[j]; plg[j] = sx[6 + j]; }
        GROUP_SYNC();   // every row has read before the next link's values are written
        float D = S[0] * Ug[k][0];
#pragma unroll
        for (int j = 1; j < 6; j++) D = fmaf(S[j], Ug[k][j], D);
        D += dex;
        float sp = S[0] * plg[0];
#pragma unroll
        for (int j = 1; j < 6; j++) sp = fmaf(S[j], plg[j], sp);
        const float invD = rcp_spec(D);
        invDk[k] = invD;
        uk[k] = tau0 - sp;
#pragma unroll
        for (int j = 0; j < 6; j++) Wg[j] = Ug[k][j] * invD;
        const float Wi = Ui * invD;
        // element (i, j): above the diagonal IA[i][j] -= U[i] W[j]; below it the mirror image's operands, IA[j][i] -= U[j] W[i]
#pragma unroll
        for (int j = 0; j < 6; j++) row[j] = j < RL.i ? fmaf(-Ug[k][j], Wi, row[j]) : fmaf(-Ui, Wg[j], row[j]);
        float acc = row[0] * cc[0];
#pragma unroll
        for (int j = 1; j < 6; j++) acc = fmaf(row[j], cc[j], acc);
        pc = fmaf(Wi, uk[k], pl + acc);
#pragma unroll
        for (int j = 0; j < 6; j++) Ic[j] = row[j];
      }
      float* o = L.xch + (RL.c * NLK) * XCH_STRIDE;   // packed upper triangle: row i writes its elements (i, j >= i)
#pragma unroll
      for (int j = 0; j < 6; j++)
        if (j >= RL.i) o[RL.off[j]] = Ic[j];
      o[21 + RL.i] = pc;
    }
  } else if (ischain) {
    float Ic[21], pc6[6];   // running child contribution
#pragma unroll
    for (int k = NLK - 1; k >= 0; k--) {
      const float* o = L.xch + (l * NLK + k) * XCH_STRIDE;
      const float* rec = L.jrec + (l * NLK + k) * JREC_STRIDE;
      float Il[21], pl[6], pa[6];
#pragma unroll
      for (int j = 0; j < 21; j++) Il[j] = o[j];
#pragma unroll
      for (int j = 0; j < 6; j++) pl[j] = o[21 + j];
      if (k < NLK - 1) {
#pragma unroll
        for (int j = 0; j < 21; j++) Il[j] += Ic[j];
#pragma unroll
        for (int j = 0; j < 6; j++) pl[j] += pc6[j];
      }
      chain_inward_link(K[k], Il, pl, rec[JREC_DE], rec[JREC_T0], pa);
#pragma unroll
      for (int j = 0; j < 21; j++) Ic[j] = Il[j];
#pragma unroll
      for (int j = 0; j < 6; j++) pc6[j] = pa[j];
    }
    float* o = L.xch + (l * NLK) * XCH_STRIDE;
#pragma unroll
    for (int j = 0; j < 21; j++) o[j] = Ic[j];
#pragma unroll
    for (int j = 0; j < 6; j++) o[21 + j] = pc6[j];
  }
  GROUP_SYNC();
  PHASE_MARK(6);

  // ---- F. root: its 27 accumulators plus the chains' in child_list order (= chain order, ChainDims::matches), one
  // accumulator per lane (the same additions in the same order as a serial sum); then the root lane: 6x6 solve,
  // integration of the base
  for (int j = l; j < 27; j += G) {
    float v = L.xroot[j];
#pragma unroll
    for (int c = 0; c < NCH; c++) v += L.xch[(c * NLK) * XCH_STRIDE + j];
    L.xroot[j] = v;
  }
  GROUP_SYNC();
  if (isroot) {
    float a[6];
#pragma unroll
    for (int j = 0; j < 21; j++) IA[j] = L.xroot[j];
#pragma unroll
    for (int j = 0; j < 6; j++) pA[j] = L.xroot[21 + j];
    ldlt_solve6(IA, pA, a);
#pragma unroll
    for (int j = 0; j < 6; j++) L.acc[j] = a[j];
    // semi-implicit Euler of the floating base (oracle substep(), last block)
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
    const float w2 = dot3(wn, wn), wmax = C.sp.max_ang_vel;
    if (w2 > wmax * wmax) {
      const float sc2 = wmax * rsqrt_spec(w2);
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
    const float inv = rsqrt_spec(fmaf(nw, nw, fmaf(nz, nz, fmaf(ny, ny, nx * nx))));
    Rt[3] = nx * inv; Rt[4] = ny * inv; Rt[5] = nz * inv; Rt[6] = nw * inv;
  }
  GROUP_SYNC();
  PHASE_MARK(7);

  // ---- G. chain lanes: outward pass; joint accelerations to the dof lanes
  if (ischain) {
    float ap[6];
#pragma unroll
    for (int j = 0; j < 6; j++) ap[j] = L.acc[j];
    const int b0 = CD::body(ci, 0);
#pragma unroll
    for (int k = 0; k < NLK; k++) {
      float Sk[6], ck[6], Uk[6], invD, uu;
      if constexpr (ROWS) {
        const float* rec = L.jrec + (ci * NLK + k) * JREC_STRIDE;
#pragma unroll
        for (int j = 0; j < 6; j++) { Sk[j] = rec[JREC_S + j]; ck[j] = rec[JREC_C + j]; Uk[j] = Ug[k][j]; }
        invD = invDk[k]; uu = uk[k];
      } else {
#pragma unroll
        for (int j = 0; j < 6; j++) { Sk[j] = K[k].S[j]; ck[j] = K[k].c[j]; Uk[j] = K[k].U[j]; }
        invD = K[k].invD; uu = K[k].u;
      }
#pragma unroll
      for (int j = 0; j < 6; j++) ap[j] = ap[j] + ck[j];
      float ua = Uk[0] * ap[0];
#pragma unroll
      for (int j = 1; j < 6; j++) ua = fmaf(Uk[j], ap[j], ua);
      const float qdd = (uu - ua) * invD;
#pragma unroll
      for (int j = 0; j < 6; j++) ap[j] = fmaf(Sk[j], qdd, ap[j]);
      if (contact_out) {
        float* o = L.acc + (b0 + k) * 6;
#pragma unroll
        for (int j = 0; j < 6; j++) o[j] = ap[j];
      }
      L.dofb[(ci * NLK + k) * DOF_STRIDE + 4] = qdd;
    }
  }
  GROUP_SYNC();
  PHASE_MARK(8);

  // ---- H. dof lanes: semi-implicit Euler of the joints
  if (isdof) {
    const float vl = m->vel_limit[l];
    const float qdn = rclampf(fmaf(dt, L.dofb[l * DOF_STRIDE + 4], X.qd), -vl, vl);
    X.qd = qdn;
    X.q = fmaf(dt, qdn, X.q);
  }

  // ---- net contact force per reported body (the sub-step whose forces the task reads)
  if (contact_out) {
#pragma unroll
    for (int k = 0; k < NR; k++) {
      if (P.idx[k] >= 0 && act.test(l + k * G)) contact_force_final(L.pt + P.idx[k] * PT_STRIDE, L.acc + m->dyn[P.body[k]] * 6, dt);
    }
    GROUP_SYNC();
    // a body lane owns the points of its moving body: those of the link itself and, on a chain's last link, those of
    // the welded end body -- two sums in point order (the oracle adds into contact_out[pt_body] in point order)
    if (isbody && half == 0) {
      const bool last = islink && (lb % NLK) == NLK - 1;
      float f[3] = {0.0f, 0.0f, 0.0f}, fw[3] = {0.0f, 0.0f, 0.0f};
      if constexpr (SPLIT) {
        unsigned bits = act.field(mine.s0, mine.n);
        while (bits) {
          const int i = mine.i0 + __builtin_ctz(bits);
          bits &= bits - 1u;
          const float* o = L.pt + i * PT_STRIDE;
          if (m->pt_body[i] == myb) { f[0] += o[PT_F]; f[1] += o[PT_F + 1]; f[2] += o[PT_F + 2]; }
          else { fw[0] += o[PT_F]; fw[1] += o[PT_F + 1]; fw[2] += o[PT_F + 2]; }
        }
      } else if (((act.w[0] & mine.mask.w[0]) | (act.w[1] & mine.mask.w[1])) != 0ull) {
        for (int j = 0; j < mine.n; j++) {
          if (!act.test(mine.slot(j))) continue;
          const int i = mine.i0 + j;
          const float* o = L.pt + i * PT_STRIDE;
          if (m->pt_body[i] == myb) { f[0] += o[PT_F]; f[1] += o[PT_F + 1]; f[2] += o[PT_F + 2]; }
          else { fw[0] += o[PT_F]; fw[1] += o[PT_F + 1]; fw[2] += o[PT_F + 2]; }
        }
      }
      contact_out[3 * myb] = f[0]; contact_out[3 * myb + 1] = f[1]; contact_out[3 * myb + 2] = f[2];
      if (last) { contact_out[3 * (myb + 1)] = fw[0]; contact_out[3 * (myb + 1) + 1] = fw[1]; contact_out[3 * (myb + 1) + 2] = fw[2]; }
    }
    GROUP_SYNC();
    if constexpr (SELF) {
      static_assert(!SELF || G >= CD::NB, "self_contact_forces: a lane per reported body");
      self_contact_forces(C, L, l, CD::NPC, nself, contact_out);
      GROUP_SYNC();
    }
  }
  PHASE_MARK(9);
}

// gym.refresh_rigid_body_state_tensor for one env: rows -> `stage` (LDS, nb x 13)
template <int G, class CD>
DEV void chain_body_states(const ShfModel* m, const ChainLds& L, int l, const DofLane& X, float* stage) {
  constexpr int NCH = CD::NCH, NLK = CD::NLK, ND = CD::ND, NB = CD::NB;
  if (l < ND) {
    const int b = CD::body(l / NLK, l % NLK);
    float* rec = L.jrec + l * JREC_STRIDE;
    float Rl[9];
    joint_local_rotation(m->trot[b], m->axis[b], X.q, Rl);
#pragma unroll
    for (int k = 0; k < 9; k++) rec[k] = Rl[k];
    rec[9] = X.qd;
  }
  GROUP_SYNC();
  if (l < NCH || l == ND) {
    float Rc[9], pc[3] = {0.0f, 0.0f, 0.0f}, vc[6], S[6], c[6];
    quat_to_mat(L.root + 3, Rc);
#pragma unroll
    for (int k = 0; k < 3; k++) { vc[k] = L.root[10 + k]; vc[3 + k] = L.root[7 + k]; }
    if (l == ND) {
      pose_store(L.pose, Rc, pc, vc);
    } else {
      const int b0 = CD::body(l, 0);
#pragma unroll
      for (int k = 0; k < NLK; k++) {
        const float* rec = L.jrec + (l * NLK + k) * JREC_STRIDE;
        float Rl[9];
#pragma unroll
        for (int j = 0; j < 9; j++) Rl[j] = rec[j];
        chain_compose_link(m->tpos[b0 + k], m->axis[b0 + k], Rl, rec[9], Rc, pc, vc, S, c);
        pose_store(L.pose + (b0 + k) * POSE_STRIDE, Rc, pc, vc);
      }
      chain_kin_weld(m->tpos[b0 + NLK], m->trot[b0 + NLK], Rc, pc);
      pose_store(L.pose + (b0 + NLK) * POSE_STRIDE, Rc, pc, vc);
    }
  }
  GROUP_SYNC();
  for (int b = l; b < NB; b += G) {
    const float* pb = L.pose + b * POSE_STRIDE;
    float Rw[9], pp[3], v[6], t[3], q[4];
#pragma unroll
    for (int k = 0; k < 9; k++) Rw[k] = pb[k];
#pragma unroll
    for (int k = 0; k < 3; k++) pp[k] = pb[9 + k];
#pragma unroll
    for (int k = 0; k < 6; k++) v[k] = pb[12 + k];
    float* o = stage + 13 * b;
#pragma unroll
    for (int k = 0; k < 3; k++) o[k] = L.root[k] + pp[k];
    mat_to_quat(Rw, q);
#pragma unroll
    for (int k = 0; k < 4; k++) o[3 + k] = q[k];
    cross3(v, pp, t);
#pragma unroll
    for (int k = 0; k < 3; k++) { o[7 + k] = v[3 + k] + t[k]; o[10 + k] = v[k]; }
  }
  GROUP_SYNC();
}

// ShifuVecEnv.step for the A1 task (env.py:85-106) on the chain mapping; the task glue after the physics is shared
// with the body-mapped kernel (a1_post_step, shf_task.h).
template <int G, class CD, bool TW, bool SELF, int KC, bool TGS>
DEV void chain_substep_hard(const StepCtx& C, const ChainLds& L, int l, DofLane& X, const ChainPoints<(CD::NEV + G - 1) / G>& P,
                            const RowLane& RL, const float* fext, float mu_shape, float* contact_out);   // shf_chain_hard.h
// HARD: the velocity-level contact solve (ShfSimParams.solver == SHF_SOLVER_PGS, csrc/shf_chain_hard.h)
// KC: constraints the solve holds per env (8: response matrix in full inside the contact-slot region; 16: its upper triangle, packed)
// TGS: ShfSimParams.solver = SHF_SOLVER_TGS, the sub-stepped sweeps (its own instantiations: the PGS kernels keep their code)
template <int G, class CD, bool TW, bool SELF = false, bool HARD = false, int KC = 8, bool TGS = false>
DEV void a1_chain_step_body(const A1Args& A) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NLK = CD::NLK, nb = CD::NB, nd = CD::ND, NR = (CD::NEV + G - 1) / G;
  PHASE_BEGIN();
  float* stats_lds = smem + CHAIN_MODEL_WORDS + TASK_WORDS;
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
  static_assert(G >= 13 && G > nd, "one root word and one action per lane");
  if (e < n) {
#pragma unroll
    for (int k = 0; k < NW; k++)
      if (l + k * G < 2 * nd) pre_dof[k] = A.S.dof[(size_t)e * nd * 2 + l + k * G];
    if (l < 13) pre_root = A.S.root[(size_t)e * 13 + l];
    if (l < nd) pre_act = raw_action(A, e, l, nd, stats_step);
  }
  stage_block<(int)sizeof(ShfA1TaskParams)>(A.tp, smem + CHAIN_MODEL_WORDS);
  stage_block<CHAIN_MODEL_BYTES>(A.S.model, smem);   // the model without its link-contact records (CHAIN_MODEL_BYTES)
  __syncthreads();
  const ShfModel* m = reinterpret_cast<const ShfModel*>(smem);
  const ShfA1TaskParams& tp = *reinterpret_cast<const ShfA1TaskParams*>(smem + CHAIN_MODEL_WORDS);
  if (e >= n) return;
  const int H = tp.num_history, P = tp.num_height_points;
  const int nobs = 12 + 2 * nd + nd * H + P;
  const int env_words = chain_lds_words<CD>(SCR_OBS + nobs, SELF);
  ChainLds L = chain_lds_carve<CD>(smem + CHAIN_MODEL_WORDS + TASK_WORDS + STATS_LDS_WORDS + es * env_words);
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
  if constexpr (HARD) C.dropped = env_dropped(A.S.dropped, A.S.sp, e);
  C.mscale = A.S.mscale ? A.S.mscale + (size_t)e * nb : nullptr;
  const float mu = A.S.friction[e];
  const int nsub = tp.decimation + (tp.extra_substep ? 1 : 0);
  const int dl = l < nd ? l : 0;
  DofLane X = {L.dofb[dl * DOF_STRIDE], L.dofb[dl * DOF_STRIDE + 1], 0.0f};
  const float pg_ = tp.p_gain[dl], dg_ = tp.d_gain[dl], q0_ = tp.default_dof_pos[dl], lim_ = m->effort[dl];
  ChainPoints<NR> LP;
  chain_points_load<G>(m, CD::NEV, l, HARD ? C.sp.contact_offset + C.sp.rest_offset : C.sp.contact_offset, LP);
  // evaluation slots of this body lane's points (lane j < nd: link j; lane nd: the root)
  const int lb = G >= 32 ? (l & 15) : l;
  const int mb = lb < nd ? CD::body(lb / NLK, lb % NLK) : 0;
  const SlotList<CD::MAXPT> mine = slot_list_load<CD::MAXPT>(m, m->pt_start[mb], (lb <= nd && l < 32) ? m->pt_count[mb] : 0);
  const RowLane RL = row_lane_load<CD>(l);
  PHASE_MARK(11);
  for (int it = 0; it < nsub; it++) {
    if (it < tp.decimation) {
      // A1Robot.step's explicit PD (a1_conditional.py:66-67); the extra refresh_state sub-step keeps the last torque (Q1)
      const float t = pg_ * (act + q0_ - X.q) - dg_ * X.qd;
      X.tau = rclampf(t, -lim_, lim_);
    }
    if constexpr (HARD)
      chain_substep_hard<G, CD, TW, SELF, KC, TGS>(C, L, l, X, LP, RL, (it == tp.decimation) ? A.push + (size_t)e * nb * 3 : nullptr, mu,
                                    (it == nsub - 1) ? L.xch : nullptr);
    else
      chain_substep<G, CD, TW, SELF>(C, L, l, X, LP, mine, RL, (it == tp.decimation) ? A.push + (size_t)e * nb * 3 : nullptr, mu,
                                     (it == nsub - 1) ? L.xch : nullptr);
  }
  PHASE_RESET();
  if (l < nd) {
    float* D = L.dofb + l * DOF_STRIDE;
    D[0] = X.q; D[1] = X.qd; D[5] = X.tau;
    A.torques[(size_t)e * nd + l] = X.tau;
    scr[SCR_ACT + l] = act;
  }
  for (int i = l; i < 3 * nb; i += G) A.S.contact[(size_t)e * nb * 3 + i] = L.xch[i];
  // the contact-point region is idle from here on: it becomes scratch
  for (int i = l; i < nd * H; i += G) scr[SCR_HIST + i] = A.history[(size_t)e * nd * H + i];
  PHASE_MARK(12);
  chain_body_states<G, CD>(m, L, l, X, scr + SCR_BODY);
  for (int i = l; i < 13 * nb; i += G) A.body_state[(size_t)e * nb * 13 + i] = scr[SCR_BODY + i];
  PHASE_RESET();
  a1_post_step<G>(A, m, tp, L, scr, e, l, stats_lds, stats_step);
}
