// shf_boxes.h -- single-body box actors (gym.create_box, reference shifu/units/object.py:19-39;
// the ABB push-box scene of examples/abb_pushbox_vision/task_config.py:13-46: fixed table, free
// cube, fixed goal pad) and their contacts.  Included by shf_device.h, same arithmetic contract.
//
// Lane roles added to a group:   lane nb + k          <-> box actor k
//                                lane i (+j*G)        <-> one (corner, target) or (sphere, box) slot
// Free boxes are rigid bodies under the same linearly-implicit contact law as the articulation:
// 8 corners against the terrain and every static box, the articulation's collision spheres against
// them.  Arm<->box coupling is staggered (each side sees the other's point moving with its
// start-of-step velocity); the articulation side is scaled by m_box / (m_box + dt*beta) so that
// both sides feel about the same force.
#pragma once

DEV bool box_is_dynamic(const ShfBoxDesc& b) { return !b.fixed && b.mass > 0.0f; }

// Scene shape: read from the LDS scene at run time (any mix of boxes; `in contact` flags travel through LDS), or fixed
// at compile time for a scene with exactly one free box, so that only its slots are evaluated -- one lane each, in one
// or two passes -- and the folds pick the active ones out of wave ballots (no flag reads, no idle iterations), like
// the articulation's own sample points.  Same operations on the same slots in the same order either way.
struct DynScene { static constexpr int NBX = 0, DYN = 0, NSPH = 0; };
template <int NBX_, int DYN_, int NSPH_>
struct FixedScene {
  static constexpr int NBX = NBX_, DYN = DYN_, NSPH = NSPH_;   // boxes per env, index of the free one, arm spheres
  static bool matches(int nboxes, const ShfBoxDesc* boxes, int nsph) {
    if (nboxes != NBX || nsph != NSPH) return false;
    for (int k = 0; k < nboxes; k++)
      if ((!boxes[k].fixed && boxes[k].mass > 0.0f) != (k == DYN)) return false;
    return true;
  }
};
typedef FixedScene<3, 1, 2> AbbScene;   // table (fixed), cube (free), goal pad (fixed); the rod capsule's two records (abb_task.py)
// per-lane constants of the fixed-scene path: which of the arm's spheres sit on this lane's body / moving body
struct BoxLane { unsigned sph_body = 0u, sph_dyn = 0u; };
DEV BoxLane box_lane_load(const ShfModel* m, int l) {
  BoxLane K;
  for (int si = 0; si < m->nsph; si++) {
    if (m->sph_body[si] == l) K.sph_body |= 1u << si;
    if (m->dyn[m->sph_body[si]] == l) K.sph_dyn |= 1u << si;
  }
  return K;
}
// ballots of one sub-step (fixed-scene path): the free box's corners against the terrain (bit c) and against the other
// boxes (bit c * (NBX - 1) + t, t counting the other boxes in ascending order), sphere bit si
struct BoxMasks { unsigned cplane = 0u; unsigned long long cbox = 0ull; unsigned cedge = 0u; unsigned spheres = 0u; int nlink = 0; };
// Next active corner slot in fold order -- corner ascending, within a corner the terrain (tg = 0) before the boxes
// (tg = 1 + box, ascending): the order of the run-time path's flag scan and of the oracle's loops.
// (ed: bit ks = the edge-edge slot against fixed box ks, which sits at corner ks, target 1 + KD)
template <int NO, int KD>
DEV bool corner_next(unsigned& pl, unsigned long long& bx, unsigned& ed, int* c, int* tg) {
  if (!pl && !bx && !ed) return false;
  const int cp = pl ? __builtin_ctz(pl) : 64;
  const int jb = bx ? __builtin_ctzll(bx) : 64 * NO;
  const int cb = jb / NO, tb = jb - cb * NO, tgb = 1 + tb + (tb >= KD ? 1 : 0);
  const int ce = ed ? __builtin_ctz(ed) : 64;
  // keys (corner, target): the smallest goes first
  const int kp = cp * 8, kb = cb * 8 + tgb, ke = ce * 8 + 1 + KD;
  if (kp <= kb && kp <= ke) { *c = cp; *tg = 0; pl &= pl - 1u; }
  else if (kb <= ke) { *c = cb; *tg = tgb; bx &= bx - 1ull; }
  else { *c = ce; *tg = 1 + KD; ed &= ed - 1u; }
  return true;
}

// slot layout in LDS (PT_STRIDE floats): r[3] n[3] f0[3] ct bn on (PT_* offsets, shf_device.h)
// ShfSimParams.solver == SHF_SOLVER_PGS (oracle: slot_eval with kc < 0): the slot only records the candidate constraint --
// location, normal, gap from rest_offset (`beta` carries it) in PT_F, friction in PT_F + 1; `offset` arrives as
// contact_offset + rest_offset.
DEV void slot_eval(float* o, float phi, const float* n, const float* r, const float* vs, const float* vp, float mu, float kc,
                   float beta, float veps, float vdep, float dt, float offset) {
  float on = 0.0f;
  if (kc < 0.0f) {
    if (phi < offset) {
      on = 1.0f;
      o[PT_CT] = 0.0f; o[PT_BN] = 0.0f;
#pragma unroll
      for (int k = 0; k < 3; k++) { o[PT_R + k] = r[k]; o[PT_N + k] = n[k]; }
      o[PT_F] = phi - beta; o[PT_F + 1] = mu; o[PT_F + 2] = 0.0f;
    }
    o[PT_ON] = on;
    return;
  }
  if (phi < offset) {
    const float vn = dot3(n, vp);
    const float pen = rminf(-kc * phi, beta * vdep);
    const float fn = fmaf(-beta, vn, pen);
    if ((phi < 0.0f || fmaf(dt, vn, phi) < 0.0f) && fn > 0.0f) {
      const float vt[3] = {fmaf(-vn, n[0], vp[0]), fmaf(-vn, n[1], vp[1]), fmaf(-vn, n[2], vp[2])};
      const float ct = friction_coefficient(n, vs, pen, mu, beta, veps);
      on = 1.0f;
      o[PT_CT] = ct;
      o[PT_BN] = beta;
#pragma unroll
      for (int k = 0; k < 3; k++) { o[PT_R + k] = r[k]; o[PT_N + k] = n[k]; o[PT_F + k] = fmaf(fn, n[k], -(ct * vt[k])); }
    }
  }
  o[PT_ON] = on;
}

DEV void slot_accumulate_fb(float* IA, float* pA, const float* o, float sign, float dt, float fscale, float bscale) {
  const float ss = sign * fscale, oct = o[PT_CT] * bscale, obn = o[PT_BN] * bscale;
  const float f0[3] = {ss * o[PT_F], ss * o[PT_F + 1], ss * o[PT_F + 2]};
  const float r[3] = {o[PT_R], o[PT_R + 1], o[PT_R + 2]}, n[3] = {o[PT_N], o[PT_N + 1], o[PT_N + 2]};
  float t[3], wn[6];
  cross3(r, f0, t);
#pragma unroll
  for (int k = 0; k < 3; k++) { pA[k] -= t[k]; pA[3 + k] -= f0[k]; }
  cross3(r, n, t);
#pragma unroll
  for (int k = 0; k < 3; k++) { wn[k] = t[k]; wn[3 + k] = n[k]; }
  const float a = dt * oct, bb = dt * (obn - oct);
  const float r2 = dot3(r, r);
#pragma unroll
  for (int i2 = 0; i2 < 3; i2++)
#pragma unroll
    for (int j2 = i2; j2 < 3; j2++) IA[SYM(i2, j2)] = fmaf(a, (i2 == j2 ? r2 : 0.0f) - r[i2] * r[j2], IA[SYM(i2, j2)]);
  IA[SYM(0, 4)] = fmaf(a, -r[2], IA[SYM(0, 4)]); IA[SYM(0, 5)] = fmaf(a, r[1], IA[SYM(0, 5)]);
  IA[SYM(1, 3)] = fmaf(a, r[2], IA[SYM(1, 3)]);  IA[SYM(1, 5)] = fmaf(a, -r[0], IA[SYM(1, 5)]);
  IA[SYM(2, 3)] = fmaf(a, -r[1], IA[SYM(2, 3)]); IA[SYM(2, 4)] = fmaf(a, r[0], IA[SYM(2, 4)]);
  IA[SYM(3, 3)] += a; IA[SYM(4, 4)] += a; IA[SYM(5, 5)] += a;
#pragma unroll
  for (int i2 = 0; i2 < 6; i2++) {
    const float bw = bb * wn[i2];
#pragma unroll
    for (int j2 = i2; j2 < 6; j2++) IA[SYM(i2, j2)] = fmaf(bw, wn[j2], IA[SYM(i2, j2)]);
  }
}

// The same for a lane that owns half of a body's accumulators (the chain-mapped A1 kernel at 32 lanes per env: HALF 0 owns
// IA[0..10], HALF 1 owns IA[11..20] and pA): every element sees the operations of slot_accumulate_fb, the other half's are
// dead code.
template <int HALF>
DEV void slot_accumulate_fb_half(float* IA, float* pA, const float* o, float sign, float dt, float fscale, float bscale) {
  float tI[21], tp[6];
#pragma unroll
  for (int k = 0; k < 21; k++) tI[k] = IA[k];
#pragma unroll
  for (int k = 0; k < 6; k++) tp[k] = pA[k];
  slot_accumulate_fb(tI, tp, o, sign, dt, fscale, bscale);
  if (HALF == 0) {
#pragma unroll
    for (int k = 0; k < 11; k++) IA[k] = tI[k];
  } else {
#pragma unroll
    for (int k = 11; k < 21; k++) IA[k] = tI[k];
#pragma unroll
    for (int k = 0; k < 6; k++) pA[k] = tp[k];
  }
}

DEV void slot_accumulate(float* IA, float* pA, const float* o, float sign, float dt, float scale) {
  slot_accumulate_fb(IA, pA, o, sign, dt, scale, scale);
}

DEV void slot_force_fb(const float* o, const float* ab, float sign, float dt, float fscale, float bscale, float* f) {
  const float ss = sign * fscale, oct = o[PT_CT] * bscale, obn = o[PT_BN] * bscale;
  const float r[3] = {o[PT_R], o[PT_R + 1], o[PT_R + 2]}, n[3] = {o[PT_N], o[PT_N + 1], o[PT_N + 2]}, al[3] = {ab[0], ab[1], ab[2]};
  float t[3], ap[3];
  cross3(al, r, t);
#pragma unroll
  for (int k = 0; k < 3; k++) ap[k] = ab[3 + k] + t[k];
  const float an = dot3(n, ap);
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const float Ba = fmaf(obn - oct, an * n[k], oct * ap[k]);
    f[k] += fmaf(-dt, Ba, ss * o[PT_F + k]);
  }
}

DEV void slot_force(const float* o, const float* ab, float sign, float dt, float scale, float* f) {
  slot_force_fb(o, ab, sign, dt, scale, scale, f);
}

// ---- consistent articulation <-> free-box pair law (oracle: pair_accumulate and its comment) ----
// Force on the articulation's point f = Feff - dt Keff aA_pt, with Keff = (I + dt K W)^-1 K and
// Feff = (I + dt K W)^-1 (f0 + dt K cfree): W = J IAbox^-1 J^T and cfree = J ba_free are the box's compliance and its
// supported free acceleration at the point, from its articulated inertia with its own (ground / table) contacts folded in.
// The box lane computes the law (pair_law) and later hands -f to its own solve; the articulation's lane folds it
// (pair_accumulate).  PR_F / PR_K: layout of the pair record.
#define PR_F 0
#define PR_K 3
DEV void pair_point_accel(const float* a6, const float* r, float* ap) {
  float t[3];
  cross3(a6, r, t);
#pragma unroll
  for (int k = 0; k < 3; k++) ap[k] = a6[3 + k] + t[k];
}
DEV void mat3_inv(const float* A, float* Ai) {
  const float c00 = fmaf(A[4], A[8], -(A[5] * A[7])), c01 = fmaf(A[5], A[6], -(A[3] * A[8])), c02 = fmaf(A[3], A[7], -(A[4] * A[6]));
  const float id = 1.0f / fmaf(A[0], c00, fmaf(A[1], c01, A[2] * c02));
  Ai[0] = c00 * id; Ai[1] = fmaf(A[2], A[7], -(A[1] * A[8])) * id; Ai[2] = fmaf(A[1], A[5], -(A[2] * A[4])) * id;
  Ai[3] = c01 * id; Ai[4] = fmaf(A[0], A[8], -(A[2] * A[6])) * id; Ai[5] = fmaf(A[2], A[3], -(A[0] * A[5])) * id;
  Ai[6] = c02 * id; Ai[7] = fmaf(A[1], A[6], -(A[0] * A[7])) * id; Ai[8] = fmaf(A[0], A[4], -(A[1] * A[3])) * id;
}
// box lane: IA / pA = the box with its own contacts folded, afree = its solve; o = the pair slot; pr = the pair record
// rsum, nshare: sum of the contact points and number of the pair slots active on this box -- each slot is eliminated as if
// the others pushed with the same force as itself (mass splitting with the true geometry; oracle boxes_pre)
DEV void pair_law(const Ldlt6& FIA, const float* afree, const float* o, float dt, float* pr, const float* rsum, float nshare) {
  const float r[3] = {o[PT_R], o[PT_R + 1], o[PT_R + 2]}, n[3] = {o[PT_N], o[PT_N + 1], o[PT_N + 2]};
  const float ct = o[PT_CT], bn = o[PT_BN];
  float W[9], cfree[3];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const float e[3] = {k == 0 ? 1.0f : 0.0f, k == 1 ? 1.0f : 0.0f, k == 2 ? 1.0f : 0.0f};
    float t[3], rhs[6], y[6], wp[3];
    cross3(rsum, e, t);
#pragma unroll
    for (int i = 0; i < 3; i++) { rhs[i] = -t[i]; rhs[3 + i] = -(e[i] * nshare); }
    ldlt_substitute6(FIA, rhs, y);
    pair_point_accel(y, r, wp);
#pragma unroll
    for (int i = 0; i < 3; i++) W[3 * i + k] = wp[i];
  }
  pair_point_accel(afree, r, cfree);
  float K[9], Mx[9], S[9], g0[3], Ke[9];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) K[3 * i + j] = fmaf(bn - ct, n[i] * n[j], i == j ? ct : 0.0f);
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++)
      Mx[3 * i + j] = fmaf(dt, fmaf(K[3 * i + 2], W[6 + j], fmaf(K[3 * i + 1], W[3 + j], K[3 * i] * W[j])), i == j ? 1.0f : 0.0f);
  mat3_inv(Mx, S);
#pragma unroll
  for (int i = 0; i < 3; i++)
    g0[i] = fmaf(dt, fmaf(K[3 * i + 2], cfree[2], fmaf(K[3 * i + 1], cfree[1], K[3 * i] * cfree[0])), o[PT_F + i]);
#pragma unroll
  for (int i = 0; i < 3; i++) {
    pr[PR_F + i] = fmaf(S[3 * i + 2], g0[2], fmaf(S[3 * i + 1], g0[1], S[3 * i] * g0[0]));
#pragma unroll
    for (int j = 0; j < 3; j++) Ke[3 * i + j] = fmaf(S[3 * i + 2], K[6 + j], fmaf(S[3 * i + 1], K[3 + j], S[3 * i] * K[j]));
  }
  // symmetrised upper triangle: 00 01 02 11 12 22
  pr[PR_K + 0] = Ke[0]; pr[PR_K + 1] = 0.5f * (Ke[1] + Ke[3]); pr[PR_K + 2] = 0.5f * (Ke[2] + Ke[6]);
  pr[PR_K + 3] = Ke[4]; pr[PR_K + 4] = 0.5f * (Ke[5] + Ke[7]); pr[PR_K + 5] = Ke[8];
}
DEV void pair_unpack(const float* pr, float* F, float* K) {
  F[0] = pr[PR_F]; F[1] = pr[PR_F + 1]; F[2] = pr[PR_F + 2];
  K[0] = pr[PR_K]; K[1] = pr[PR_K + 1]; K[2] = pr[PR_K + 2]; K[3] = pr[PR_K + 1]; K[4] = pr[PR_K + 3]; K[5] = pr[PR_K + 4];
  K[6] = pr[PR_K + 2]; K[7] = pr[PR_K + 4]; K[8] = pr[PR_K + 5];
}
// articulation lane: fold f = F - dt K a_pt at r into the packed (IA, pA): IA += dt J^T K J, pA -= J^T F, J = [-[r]x, I]
DEV void pair_accumulate(float* IA, float* pA, const float* r, const float* F, const float* K, float dt) {
  float t[3];
  cross3(r, F, t);
#pragma unroll
  for (int k = 0; k < 3; k++) { pA[k] -= t[k]; pA[3 + k] -= F[k]; }
  const float X[9] = {0.0f, r[2], -r[1], -r[2], 0.0f, r[0], r[1], -r[0], 0.0f};
  float KJ[3][6];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      KJ[i][j] = fmaf(K[3 * i + 2], X[6 + j], fmaf(K[3 * i + 1], X[3 + j], K[3 * i] * X[j]));
      KJ[i][3 + j] = K[3 * i + j];
    }
#pragma unroll
  for (int i = 0; i < 6; i++)
#pragma unroll
    for (int j = i; j < 6; j++) {
      float v;
      if (i < 3) v = fmaf(X[6 + i], KJ[2][j], fmaf(X[3 + i], KJ[1][j], X[i] * KJ[0][j]));
      else v = KJ[i - 3][j];
      IA[SYM(i, j)] = fmaf(dt, v, IA[SYM(i, j)]);
    }
}
// the pair force once the articulation's acceleration is known
DEV void pair_force(const float* pr, const float* a_art, const float* r, float dt, float* f) {
  float F[3], K[9], ap[3];
  pair_unpack(pr, F, K);
  pair_point_accel(a_art, r, ap);
#pragma unroll
  for (int i = 0; i < 3; i++) f[i] = fmaf(-dt, fmaf(K[3 * i + 2], ap[2], fmaf(K[3 * i + 1], ap[1], K[3 * i] * ap[0])), F[i]);
}

// ---- joint pair law: the two ends of one capsule in line contact with a free box (ShfModel.sph_part; oracle
// pair_law_joint and its comment).  (K^-1 + dt W) f = K^-1 f0 + dt c - dt a over both points: Keff = (K^-1 + dt W)^-1
// (symmetric 6x6, packed), Feff = Keff (K^-1 f0 + dt c); the record is Keff[21] Feff[6] flag (one exchange slot).
#define JR_K 0
#define JR_F 21
#define JR_ON 27
DEV void pair_law_joint(const Ldlt6& FIA, const float* afree, const float* o1, const float* o2, float dt, float* rec) {
  const float* o[2] = {o1, o2};
  float Kinv[2][9], Wc[6][6], g[6], negg[6], A[21], Feff[6];
#pragma unroll
  for (int j = 0; j < 2; j++) {
    const float ict = 1.0f / o[j][PT_CT], ibn = 1.0f / o[j][PT_BN];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int c = 0; c < 3; c++) Kinv[j][3 * r + c] = fmaf(ibn - ict, o[j][PT_N + r] * o[j][PT_N + c], r == c ? ict : 0.0f);
  }
#pragma unroll
  for (int j = 0; j < 2; j++)
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const float e[3] = {k == 0 ? 1.0f : 0.0f, k == 1 ? 1.0f : 0.0f, k == 2 ? 1.0f : 0.0f};
      const float rj[3] = {o[j][PT_R], o[j][PT_R + 1], o[j][PT_R + 2]};
      float t[3], rhs[6], y[6];
      cross3(rj, e, t);
#pragma unroll
      for (int i = 0; i < 3; i++) { rhs[i] = -t[i]; rhs[3 + i] = -e[i]; }
      ldlt_substitute6(FIA, rhs, y);
#pragma unroll
      for (int i = 0; i < 2; i++) {
        const float ri[3] = {o[i][PT_R], o[i][PT_R + 1], o[i][PT_R + 2]};
        pair_point_accel(y, ri, &Wc[3 * j + k][3 * i]);
      }
    }
#pragma unroll
  for (int p = 0; p < 6; p++)
#pragma unroll
    for (int q = p; q < 6; q++) {
      const float kinv = (p / 3 == q / 3) ? Kinv[p / 3][3 * (p % 3) + (q % 3)] : 0.0f;
      A[SYM(p, q)] = fmaf(dt, Wc[q][p], kinv);
    }
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const float ri[3] = {o[i][PT_R], o[i][PT_R + 1], o[i][PT_R + 2]};
    float cf[3];
    pair_point_accel(afree, ri, cf);
#pragma unroll
    for (int r = 0; r < 3; r++)
      g[3 * i + r] = fmaf(dt, cf[r], fmaf(Kinv[i][3 * r + 2], o[i][PT_F + 2], fmaf(Kinv[i][3 * r + 1], o[i][PT_F + 1], Kinv[i][3 * r] * o[i][PT_F])));
  }
#pragma unroll
  for (int p = 0; p < 6; p++) negg[p] = -g[p];
  Ldlt6 FA;
  ldlt_factor6(A, FA);
  ldlt_substitute6(FA, negg, Feff);
#pragma unroll
  for (int p = 0; p < 6; p++) rec[JR_F + p] = Feff[p];
#pragma unroll
  for (int q = 0; q < 6; q++) {
    float rhs[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f}, x[6];
    rhs[q] = -1.0f;
    ldlt_substitute6(FA, rhs, x);
#pragma unroll
    for (int p = 0; p <= q; p++) rec[JR_K + SYM(p, q)] = x[p];
  }
  rec[JR_ON] = 1.0f;
}
DEV void pair_joint_jacobian(const float* r1, const float* r2, float J[6][6]) {
  const float* rr[2] = {r1, r2};
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const float* r = rr[i];
    const float X[9] = {0.0f, r[2], -r[1], -r[2], 0.0f, r[0], r[1], -r[0], 0.0f};
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
      for (int c = 0; c < 3; c++) { J[3 * i + a][c] = X[3 * a + c]; J[3 * i + a][3 + c] = a == c ? 1.0f : 0.0f; }
  }
}
// the articulation's body folds the joint law: IA += dt J^T Keff J, pA -= J^T Feff
DEV void pair_accumulate_joint(float* IA, float* pA, const float* r1, const float* r2, const float* rec, float dt) {
  const float* rr[2] = {r1, r2};
  float Kf[21], Feff[6];
#pragma unroll
  for (int k = 0; k < 21; k++) Kf[k] = rec[JR_K + k];
#pragma unroll
  for (int k = 0; k < 6; k++) Feff[k] = rec[JR_F + k];
#pragma unroll
  for (int i = 0; i < 2; i++) {
    float t[3];
    cross3(rr[i], Feff + 3 * i, t);
#pragma unroll
    for (int k = 0; k < 3; k++) { pA[k] -= t[k]; pA[3 + k] -= Feff[3 * i + k]; }
  }
  float J[6][6], T[6][6];
  pair_joint_jacobian(r1, r2, J);
#pragma unroll
  for (int p = 0; p < 6; p++)
#pragma unroll
    for (int c = 0; c < 6; c++) {
      float acc = SYMG(Kf, p, 0) * J[0][c];
#pragma unroll
      for (int q = 1; q < 6; q++) acc = fmaf(SYMG(Kf, p, q), J[q][c], acc);
      T[p][c] = acc;
    }
#pragma unroll
  for (int a = 0; a < 6; a++)
#pragma unroll
    for (int c = a; c < 6; c++) {
      float acc = J[0][a] * T[0][c];
#pragma unroll
      for (int p = 1; p < 6; p++) acc = fmaf(J[p][a], T[p][c], acc);
      IA[SYM(a, c)] = fmaf(dt, acc, IA[SYM(a, c)]);
    }
}
// the two forces once the body's acceleration is known: f = Feff - dt Keff [a_pt1; a_pt2]
DEV void pair_force_joint(const float* rec, const float* a_body, const float* r1, const float* r2, float dt, float* f1, float* f2) {
  float ap[6], f[6];
  pair_point_accel(a_body, r1, ap);
  pair_point_accel(a_body, r2, ap + 3);
#pragma unroll
  for (int p = 0; p < 6; p++) {
    float acc = SYMG(rec + JR_K, p, 0) * ap[0];
#pragma unroll
    for (int q = 1; q < 6; q++) acc = fmaf(SYMG(rec + JR_K, p, q), ap[q], acc);
    f[p] = fmaf(-dt, acc, rec[JR_F + p]);
  }
#pragma unroll
  for (int k = 0; k < 3; k++) { f1[k] = f[k]; f2[k] = f[3 + k]; }
}
// Which capsule (first record) has exactly its two ends, and nothing else, as the active pair slots `sb` (bit si) of a
// box whose link slots are `nlink_on_box`: -1 if none (oracle boxes_pre: w->joint).  cts: PT_CT of sphere slot si.
template <class CT>
DEV int joint_pair_of(const ShfModel* m, unsigned sb, int nlink_on_box, CT ct_of) {
  if (nlink_on_box != 0 || __builtin_popcount(sb) != 2) return -1;
  const int si = __builtin_ctz(sb);
  if (si + 1 >= m->nsph || !((sb >> (si + 1)) & 1u)) return -1;
  if (m->sph_part[si] != 0 || m->sph_part[si + 1] != 1 || m->sph_body[si] != m->sph_body[si + 1]) return -1;
  if (!(ct_of(si) > 0.0f) || !(ct_of(si + 1) > 0.0f)) return -1;
  return si;
}

DEV bool point_in_box(const float* bR, const float* bpos, const float* h, const float* r, float* phi, float* n) {
  const float rel[3] = {r[0] - bpos[0], r[1] - bpos[1], r[2] - bpos[2]};
  float d[3], pen[3];
#pragma unroll
  for (int i = 0; i < 3; i++) d[i] = fmaf(bR[6 + i], rel[2], fmaf(bR[3 + i], rel[1], bR[i] * rel[0]));
#pragma unroll
  for (int i = 0; i < 3; i++) pen[i] = h[i] - fabsf(d[i]);
  if (!(pen[0] > 0.0f) || !(pen[1] > 0.0f) || !(pen[2] > 0.0f)) return false;
  int ax = 0;
  if (pen[1] < pen[ax]) ax = 1;
  if (pen[2] < pen[ax]) ax = 2;
  const float dax = ax == 0 ? d[0] : (ax == 1 ? d[1] : d[2]);
  const float pax = ax == 0 ? pen[0] : (ax == 1 ? pen[1] : pen[2]);
  const float sg = dax < 0.0f ? -1.0f : 1.0f;
  *phi = -pax;
  n[0] = sg * bR[ax]; n[1] = sg * bR[3 + ax]; n[2] = sg * bR[6 + ax];
  return true;
}

DEV void sphere_vs_box(const float* bR, const float* bpos, const float* h, const float* c, float rad, float* phi,
                       float* n, float* rc) {
  const float rel[3] = {c[0] - bpos[0], c[1] - bpos[1], c[2] - bpos[2]};
  float d[3], q[3], nl[3];
#pragma unroll
  for (int i = 0; i < 3; i++) d[i] = fmaf(bR[6 + i], rel[2], fmaf(bR[3 + i], rel[1], bR[i] * rel[0]));
  bool outside = false;
#pragma unroll
  for (int i = 0; i < 3; i++) { q[i] = rclampf(d[i], -h[i], h[i]); outside = outside || (q[i] != d[i]); }
  if (outside) {
    const float df[3] = {d[0] - q[0], d[1] - q[1], d[2] - q[2]};
    const float dist = sqrtf(dot3(df, df));
    *phi = dist - rad;
    const float inv = 1.0f / rmaxf(dist, 1e-12f);
#pragma unroll
    for (int i = 0; i < 3; i++) nl[i] = df[i] * inv;
  } else {
    const float pen[3] = {h[0] - fabsf(d[0]), h[1] - fabsf(d[1]), h[2] - fabsf(d[2])};
    int ax = 0;
    if (pen[1] < pen[ax]) ax = 1;
    if (pen[2] < pen[ax]) ax = 2;
    const float dax = ax == 0 ? d[0] : (ax == 1 ? d[1] : d[2]);
    const float pax = ax == 0 ? pen[0] : (ax == 1 ? pen[1] : pen[2]);
    const float hax = ax == 0 ? h[0] : (ax == 1 ? h[1] : h[2]);
    const float sg = dax < 0.0f ? -1.0f : 1.0f;
    *phi = -pax - rad;
    nl[0] = ax == 0 ? sg : 0.0f; nl[1] = ax == 1 ? sg : 0.0f; nl[2] = ax == 2 ? sg : 0.0f;
    if (ax == 0) q[0] = sg * hax; else if (ax == 1) q[1] = sg * hax; else q[2] = sg * hax;
  }
  mv3(bR, nl, n);
  mv3(bR, q, rc);
#pragma unroll
  for (int i = 0; i < 3; i++) rc[i] += bpos[i];
}


// Capsule against an oriented box: the parameter t in [0,1] of the point c0 + t s of the capsule's segment closest to the
// box (oracle: segment_box_param).  g(t) = sum_i d_i (p_i - clamp(p_i, -h_i, h_i)) -- half the derivative of the squared
// distance, non-decreasing and piecewise linear -- is sampled at the two ends and at the kinks inside (0,1) where a
// coordinate crosses a face plane; the root is interpolated between the last negative and the first positive sample, and
// a stretch of exact zeros gives its midpoint.  Sample order: ends, axes 0..2, face -h before +h.
// sample k of g (k = 0, 1: the ends; 2 + 2 i + f: the kink of axis i at face -h (f = 0) / +h (f = 1)); false: not in (0,1)
DEV bool seg_box_sample(int k, const float* a, const float* d, const float* h, float dd, float* t_out, float* g_out) {
  float t;
  bool use = true;
  if (k == 0) t = 0.0f;
  else if (k == 1) t = 1.0f;
  else {
    const int i = (k - 2) >> 1;
    const float ai = i == 0 ? a[0] : (i == 1 ? a[1] : a[2]), di = i == 0 ? d[0] : (i == 1 ? d[1] : d[2]);
    const float hi = i == 0 ? h[0] : (i == 1 ? h[1] : h[2]);
    t = (((k & 1) ? hi : -hi) - ai) * (1.0f / di);      // +-inf / nan when parallel to the faces: fails the (0,1) test
    use = t > 0.0f && t < 1.0f;
  }
  float g = 0.0f;
  if (use) {                                            // (one lane per capsule: an unused kink is really skipped)
    float ee = 0.0f;
#pragma unroll
    for (int i = 0; i < 3; i++) {
      const float p = fmaf(t, d[i], a[i]);
      const float e = p - rclampf(p, -h[i], h[i]);
      g = fmaf(d[i], e, g);
      ee = fmaf(e, e, ee);
    }
    // within 5e-4 rad of perpendicular to the distance vector the sample counts as flat (oracle: same line): a capsule
    // lying along a face touches at the middle of the overlap instead of hopping between its ends
    if (g * g <= 2.5e-7f * dd * ee) g = 0.0f;
  }
  *t_out = t; *g_out = g;
  return use;
}
struct SegBoxRoot {
  float tl = 0.0f, gl = 0.0f, th = 1.0f, gh = 0.0f, zl = 2.0f, zh = -1.0f;
  bool hasl = false, hash = false;
  DEV void push(float t, float g, bool use) {       // samples arrive in the order k = 0..7
    if (!use) return;
    if (g < 0.0f) { if (!hasl || t > tl) { tl = t; gl = g; hasl = true; } }
    else if (g > 0.0f) { if (!hash || t < th) { th = t; gh = g; hash = true; } }
    else { zl = rminf(zl, t); zh = rmaxf(zh, t); }
  }
  DEV float root() const {
    if (zh >= zl) return 0.5f * (zl + zh);
    if (!hasl) return 0.0f;
    if (!hash) return 1.0f;
    return fmaf(th - tl, gl / (gl - gh), tl);
  }
  // contact point `part` (ShfModel.sph_part; oracle segment_box_contact): a flat stretch of non-zero length is a line
  // contact held at both of its ends; otherwise part 0 is the closest point and part 1 does not exist
  DEV bool contact(int part, float* t) const {
    if (zh > zl) { *t = part == 0 ? zl : zh; return true; }
    if (part != 0) return false;
    *t = root();
    return true;
  }
};
DEV void seg_box_frame(const float* bR, const float* bpos, const float* c0, const float* s, float* a, float* d) {
  const float rel[3] = {c0[0] - bpos[0], c0[1] - bpos[1], c0[2] - bpos[2]};
#pragma unroll
  for (int i = 0; i < 3; i++) {
    a[i] = fmaf(bR[6 + i], rel[2], fmaf(bR[3 + i], rel[1], bR[i] * rel[0]));
    d[i] = fmaf(bR[6 + i], s[2], fmaf(bR[3 + i], s[1], bR[i] * s[0]));
  }
}
DEV bool segment_box_contact(const float* bR, const float* bpos, const float* h, const float* c0, const float* s, int part, float* tc) {
  float a[3], d[3];
  seg_box_frame(bR, bpos, c0, s, a, d);
  const float dd = dot3(d, d);
  SegBoxRoot Q;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    float t, g;
    const bool use = seg_box_sample(k, a, d, h, dd, &t, &g);
    Q.push(t, g, use);
  }
  return Q.contact(part, tc);
}
// world-frame centre of rounded shape si against box (Rk, bpos, hh): the sphere's centre, or the capsule's closest point.
// false: the shape's bounding sphere is further than the contact offset (+ 1 cm for the rounding of this test) from the
// box's bounding sphere -- the slot is off and nothing else is computed (oracle: rounded_far).
DEV bool rounded_centre(const ShfModel* m, int si, const float* Rb, const float* pbody, const float* Rk, const float* bpos,
                        const float* hh, float offset, float* c) {
  const float lp[3] = {m->sph_pos[si][0], m->sph_pos[si][1], m->sph_pos[si][2]};
  const float ls[3] = {m->sph_seg[si][0], m->sph_seg[si][1], m->sph_seg[si][2]};
  float sw[3];
  mv3(Rb, lp, c);
#pragma unroll
  for (int i = 0; i < 3; i++) c[i] += pbody[i];
  mv3(Rb, ls, sw);
  const float rel[3] = {fmaf(0.5f, sw[0], c[0]) - bpos[0], fmaf(0.5f, sw[1], c[1]) - bpos[1], fmaf(0.5f, sw[2], c[2]) - bpos[2]};
  const float reach = 0.5f * sqrtf(dot3(sw, sw)) + sqrtf(dot3(hh, hh)) + m->sph_radius[si] + offset + 0.01f;
  if (dot3(rel, rel) > reach * reach) return false;
  if (ls[0] != 0.0f || ls[1] != 0.0f || ls[2] != 0.0f) {
    float t;
    if (!segment_box_contact(Rk, bpos, hh, c, sw, m->sph_part[si], &t)) return false;   // the second end of no line contact
#pragma unroll
    for (int i = 0; i < 3; i++) c[i] = fmaf(t, sw[i], c[i]);
  }
  return true;
}

// ------------------------------------------------------------ self-collision --
// Capsule pairs of the articulation itself (ShfModel.self_collide; reference units.py:68, collision filter 0).
DEV void segment_closest(const float* p1, const float* q1, const float* p2, const float* q2, float* c1, float* c2) {
  const float eps = 1e-12f;
  const float d1[3] = {q1[0] - p1[0], q1[1] - p1[1], q1[2] - p1[2]}, d2[3] = {q2[0] - p2[0], q2[1] - p2[1], q2[2] - p2[2]};
  const float r[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]};
  const float a = dot3(d1, d1), e = dot3(d2, d2), f = dot3(d2, r);
  float s = 0.0f, t = 0.0f;
  if (a <= eps && e <= eps) {
    s = 0.0f; t = 0.0f;
  } else if (a <= eps) {
    t = rclampf(f / e, 0.0f, 1.0f);
  } else {
    const float c = dot3(d1, r);
    if (e <= eps) {
      s = rclampf(-c / a, 0.0f, 1.0f);
    } else {
      const float b = dot3(d1, d2);
      const float denom = fmaf(a, e, -(b * b));
      s = denom > eps ? rclampf(fmaf(b, f, -(c * e)) / denom, 0.0f, 1.0f) : 0.0f;
      t = fmaf(b, s, f) / e;
      if (t < 0.0f) { t = 0.0f; s = rclampf(-c / a, 0.0f, 1.0f); }
      else if (t > 1.0f) { t = 1.0f; s = rclampf((b - c) / a, 0.0f, 1.0f); }
    }
  }
#pragma unroll
  for (int k = 0; k < 3; k++) { c1[k] = fmaf(d1[k], s, p1[k]); c2[k] = fmaf(d2[k], t, p2[k]); }
}

// Edge-edge contact of two oriented boxes by the separating-axis test: the oracle's box_box_edge, operation for operation
// (A: the receiver, axes = columns of RA).  Run-time axis choices are selects, not indexed registers.
DEV float sel3(int i, float x0, float x1, float x2) { return i == 0 ? x0 : (i == 1 ? x1 : x2); }
DEV bool box_box_edge(const float* RA, const float* cA, const float* hA, const float* RB, const float* cB, const float* hB,
                      float offset, float* phi, float* n, float* r) {
  const float t[3] = {cA[0] - cB[0], cA[1] - cB[1], cA[2] - cB[2]};     // from B to A
  float Rm[3][3], Q[3][3], tA[3], tB[3];
#pragma unroll
  for (int i = 0; i < 3; i++) {
    tA[i] = fmaf(RA[6 + i], t[2], fmaf(RA[3 + i], t[1], RA[i] * t[0]));
    tB[i] = fmaf(RB[6 + i], t[2], fmaf(RB[3 + i], t[1], RB[i] * t[0]));
#pragma unroll
    for (int j = 0; j < 3; j++) {
      Rm[i][j] = fmaf(RA[6 + i], RB[6 + j], fmaf(RA[3 + i], RB[3 + j], RA[i] * RB[j]));
      Q[i][j] = fabsf(Rm[i][j]) + 1e-6f;
    }
  }
  float amax = 0.0f;     // an axis of A within 10 degrees of an axis of B: the cross products nearly repeat face normals
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) amax = rmaxf(amax, fabsf(Rm[i][j]));
  if (amax > 0.985f) return false;
  float sface = -1e30f;
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const float sa = fabsf(tA[i]) - (hA[i] + fmaf(hB[2], Q[i][2], fmaf(hB[1], Q[i][1], hB[0] * Q[i][0])));
    const float sb = fabsf(tB[i]) - (hB[i] + fmaf(hA[2], Q[2][i], fmaf(hA[1], Q[1][i], hA[0] * Q[0][i])));
    sface = rmaxf(sface, rmaxf(sa, sb));
  }
  if (!(sface < offset)) return false;
  float sedge = -1e30f, sg = 1.0f, il = 1.0f;
  int bi = -1, bj = -1;
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const int i1 = (i + 1) % 3, i2 = (i + 2) % 3;
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const int j1 = (j + 1) % 3, j2 = (j + 2) % 3;
      const float l2 = fmaf(-Rm[i][j], Rm[i][j], 1.0f);
      const float tl = fmaf(tA[i2], Rm[i1][j], -(tA[i1] * Rm[i2][j]));
      const float ra = fmaf(hA[i1], Q[i2][j], hA[i2] * Q[i1][j]), rb = fmaf(hB[j1], Q[i][j2], hB[j2] * Q[i][j1]);
      const float inv = rsqrt_spec(rmaxf(l2, 1e-4f));          // (the clamp only keeps the unused branch finite)
      const float se = (fabsf(tl) - (ra + rb)) * inv;
      if (l2 > 1e-4f && se > sedge) { sedge = se; bi = i; bj = j; sg = tl < 0.0f ? -1.0f : 1.0f; il = inv; }
    }
  }
  if (bi < 0 || !(sedge < offset)) return false;
  if (!(sedge > sface + fmaf(0.05f, fabsf(sface), 1e-5f))) return false;   // a face axis wins
  const int i1 = (bi + 1) % 3, i2 = (bi + 2) % 3, j1 = (bj + 1) % 3, j2 = (bj + 2) % 3;
  const float a[3] = {sel3(bi, RA[0], RA[1], RA[2]), sel3(bi, RA[3], RA[4], RA[5]), sel3(bi, RA[6], RA[7], RA[8])};
  const float b[3] = {sel3(bj, RB[0], RB[1], RB[2]), sel3(bj, RB[3], RB[4], RB[5]), sel3(bj, RB[6], RB[7], RB[8])};
  float nn[3];
  cross3(a, b, nn);
#pragma unroll
  for (int k = 0; k < 3; k++) nn[k] = sg * il * nn[k];
  float pA[3] = {cA[0], cA[1], cA[2]}, pB[3] = {cB[0], cB[1], cB[2]};
#pragma unroll
  for (int q = 0; q < 2; q++) {
    const int ka = q == 0 ? i1 : i2, kb = q == 0 ? j1 : j2;
    const float ax[3] = {sel3(ka, RA[0], RA[1], RA[2]), sel3(ka, RA[3], RA[4], RA[5]), sel3(ka, RA[6], RA[7], RA[8])};
    const float bx[3] = {sel3(kb, RB[0], RB[1], RB[2]), sel3(kb, RB[3], RB[4], RB[5]), sel3(kb, RB[6], RB[7], RB[8])};
    const float hka = sel3(ka, hA[0], hA[1], hA[2]), hkb = sel3(kb, hB[0], hB[1], hB[2]);
    const float sa = dot3(nn, ax) > 0.0f ? -hka : hka, sb = dot3(nn, bx) > 0.0f ? hkb : -hkb;
#pragma unroll
    for (int k = 0; k < 3; k++) { pA[k] = fmaf(sa, ax[k], pA[k]); pB[k] = fmaf(sb, bx[k], pB[k]); }
  }
  const float w[3] = {pA[0] - pB[0], pA[1] - pB[1], pA[2] - pB[2]};
  const float bb = sel3(bi, sel3(bj, Rm[0][0], Rm[0][1], Rm[0][2]), sel3(bj, Rm[1][0], Rm[1][1], Rm[1][2]), sel3(bj, Rm[2][0], Rm[2][1], Rm[2][2]));
  const float d = dot3(a, w), e = dot3(b, w);
  const float idn = il * il;
  const float alpha = fmaf(bb, e, -d) * idn, beta = fmaf(-bb, d, e) * idn;
  if (fabsf(alpha) > sel3(bi, hA[0], hA[1], hA[2]) || fabsf(beta) > sel3(bj, hB[0], hB[1], hB[2])) return false;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    r[k] = 0.5f * (fmaf(alpha, a[k], pA[k]) + fmaf(beta, b[k], pB[k]));
    n[k] = nn[k];
  }
  *phi = sedge;
  return true;
}

// One lane per capsule pair: pairs closer than the contact offset respond with the shared contact law; the active ones
// are compacted, in pair order, into at most SHF_MAX_SELF_CONTACTS slots starting at slot `slot0` (their `on` word
// holds pair index + 1), then folded into the two bodies (+f on a's, -f on b's).  Returns the number of active slots.
// (self_contacts_eval: evaluation and compaction only -- the chain-mapped A1 kernel folds them itself, shf_chain.h)
template <int G>
DEV int self_contacts_eval(const StepCtx& C, const EnvLds& L, int l, int slot0, float mu_shape) {
  const ShfModel* m = C.m;
  const bool hard = C.sp.solver != SHF_SOLVER_COMPLIANT;     // candidates only (slot_eval)
  const float dt = C.sp.dt, kc = hard ? -1.0f : C.sp.contact_k, veps = C.sp.friction_vel, vdep = C.sp.max_depen_vel;
  const float offset = hard ? C.sp.contact_offset + C.sp.rest_offset : C.sp.contact_offset;
  const float beta = hard ? C.sp.rest_offset : fmaf(kc, dt, C.sp.contact_d);
  const int lane0 = (int)(threadIdx.x & 63u) - l;
  const unsigned long long gmask = G >= 64 ? ~0ull : ((1ull << (G & 63)) - 1ull);
  const int npair = m->npair;
  int count = 0;
  for (int j = 0; j * G < npair; j++) {
    const int p = l + j * G;
    float slot[PT_STRIDE];
    slot[PT_ON] = 0.0f;
    // Broad phase (not in the oracle, and no need to be): the two capsules' bounding spheres -- centre the middle of the
    // segment, radius half its length + the capsule's -- further apart than the contact offset + 1 cm means gap > offset
    // wherever the closest points are, and the slot stays off.  A round none of whose pairs is near (the usual case: a
    // walking robot's links are apart) skips the closest-point search wave-wide.
    float A0[3], A1[3], B0[3], B1[3];
    bool near = false;
    if (p < npair) {
      const int ia = m->pair_a[p], ib = m->pair_b[p];
      const int ba = m->cap_body[ia], bb = m->cap_body[ib];
      const float* pa = L.pose + ba * POSE_STRIDE;
      const float* pb = L.pose + bb * POSE_STRIDE;
      float Ra[9], Rb[9];
#pragma unroll
      for (int k = 0; k < 9; k++) { Ra[k] = pa[k]; Rb[k] = pb[k]; }
      const float la[3] = {m->cap_a[ia][0], m->cap_a[ia][1], m->cap_a[ia][2]}, lb[3] = {m->cap_b[ia][0], m->cap_b[ia][1], m->cap_b[ia][2]};
      const float lc[3] = {m->cap_a[ib][0], m->cap_a[ib][1], m->cap_a[ib][2]}, ld[3] = {m->cap_b[ib][0], m->cap_b[ib][1], m->cap_b[ib][2]};
      mv3(Ra, la, A0); mv3(Ra, lb, A1); mv3(Rb, lc, B0); mv3(Rb, ld, B1);
#pragma unroll
      for (int k = 0; k < 3; k++) { A0[k] += pa[9 + k]; A1[k] += pa[9 + k]; B0[k] += pb[9 + k]; B1[k] += pb[9 + k]; }
      const float dm[3] = {(A0[0] + A1[0]) - (B0[0] + B1[0]), (A0[1] + A1[1]) - (B0[1] + B1[1]), (A0[2] + A1[2]) - (B0[2] + B1[2])};   // 2 x (centre a - centre b)
      const float da[3] = {A1[0] - A0[0], A1[1] - A0[1], A1[2] - A0[2]}, db[3] = {B1[0] - B0[0], B1[1] - B0[1], B1[2] - B0[2]};
      // 2 x reach <= |da| + |db| + 2 (ra + rb + offset + 0.01); compared squared, with (x + y)^2 <= 2 x^2 + 2 y^2 for the lengths
      const float rr = 2.0f * (m->cap_radius[ia] + m->cap_radius[ib] + offset + 0.01f);
      const float len2 = 2.0f * (dot3(da, da) + dot3(db, db));            // >= (|da| + |db|)^2
      const float bound = 2.0f * (len2 + rr * rr);                        // >= (|da| + |db| + rr)^2
      near = !(dot3(dm, dm) > bound * 1.0001f);
    }
    if (__ballot(near) == 0ull) continue;
    if (near) {
      const int ia = m->pair_a[p], ib = m->pair_b[p];
      const int ba = m->cap_body[ia], bb = m->cap_body[ib];
      float c1[3], c2[3];
      segment_closest(A0, A1, B0, B1, c1, c2);
      const float dv[3] = {c1[0] - c2[0], c1[1] - c2[1], c1[2] - c2[2]};
      const float d = sqrtf(dot3(dv, dv));
      const float ra = m->cap_radius[ia], rb = m->cap_radius[ib];
      const float gap = d - ra - rb;
      if (gap < offset) {
        float n[3] = {0.0f, 0.0f, 1.0f}, P[3];
        if (d > 1e-9f) { const float inv = 1.0f / d; n[0] = dv[0] * inv; n[1] = dv[1] * inv; n[2] = dv[2] * inv; }
        const float hp = fmaf(0.5f, gap, rb);
#pragma unroll
        for (int k = 0; k < 3; k++) P[k] = fmaf(n[k], hp, c2[k]);
        // spatial velocities of the two (moving) bodies: welded links share their carrier's
        const float* qa = L.pose + m->dyn[ba] * POSE_STRIDE;
        const float* qb = L.pose + m->dyn[bb] * POSE_STRIDE;
        const float va[3] = {qa[12], qa[13], qa[14]}, vb[3] = {qb[12], qb[13], qb[14]};
        float ta[3], tb[3], vs[3];
        cross3(va, P, ta); cross3(vb, P, tb);
#pragma unroll
        for (int k = 0; k < 3; k++) vs[k] = (qa[15 + k] + ta[k]) - (qb[15 + k] + tb[k]);
        slot_eval(slot, gap, n, P, vs, vs, mu_shape, kc, beta, veps, vdep, dt, offset);
      }
    }
    const bool on = slot[PT_ON] != 0.0f;
    const unsigned long long mask = (__ballot(on) >> lane0) & gmask;
    const int mine = count + __builtin_popcountll(mask & ((1ull << l) - 1ull));
    if (on && mine < SHF_MAX_SELF_CONTACTS) {
      float* o = L.pt + (slot0 + mine) * PT_STRIDE;
#pragma unroll
      for (int k = 0; k < PT_STRIDE - 1; k++) o[k] = slot[k];
      o[PT_ON] = (float)(p + 1);
    }
    count += __builtin_popcountll(mask);
  }
  if (count > SHF_MAX_SELF_CONTACTS) {
    if (l == 0 && C.dropped) *C.dropped += count - SHF_MAX_SELF_CONTACTS;     // dropped in pair order -- and counted (SHF_T_DROPPED)
    count = SHF_MAX_SELF_CONTACTS;
  }
  return count;
}
template <int G>
DEV int self_contacts(const StepCtx& C, const EnvLds& L, int l, bool isdyn, int slot0, BodyRegs& B, float mu_shape) {
  const ShfModel* m = C.m;
  const float dt = C.sp.dt;
  const int count = self_contacts_eval<G>(C, L, l, slot0, mu_shape);
  GROUP_SYNC();
  if (isdyn) {
    for (int k = 0; k < count; k++) {
      const float* o = L.pt + (slot0 + k) * PT_STRIDE;
      const int p = (int)o[PT_ON] - 1;
      const int da = m->dyn[m->cap_body[m->pair_a[p]]], db = m->dyn[m->cap_body[m->pair_b[p]]];
      // each side implicit in its own acceleration, scaled by 1 + (own mass) / (other mass): see the oracle's self_scales
      const float ma = body_mass(C, da), mb = body_mass(C, db);
      if (da == l) slot_accumulate_fb(B.IA, B.pA, o, 1.0f, dt, 1.0f, 1.0f + ma / mb);
      if (db == l) slot_accumulate_fb(B.IA, B.pA, o, -1.0f, dt, 1.0f, 1.0f + mb / ma);
    }
  }
  return count;
}

// Adds the end-of-step self-contact forces to the net contact force of reported body l (contact_out row l).
DEV void self_contact_forces(const StepCtx& C, const EnvLds& L, int l, int slot0, int count, float* contact_out) {
  const ShfModel* m = C.m;
  if (l >= m->nb || count == 0) return;
  float f[3] = {contact_out[3 * l], contact_out[3 * l + 1], contact_out[3 * l + 2]};
  const float* ab = L.acc + m->dyn[l] * 6;
  const float abr[6] = {ab[0], ab[1], ab[2], ab[3], ab[4], ab[5]};
  for (int k = 0; k < count; k++) {
    const float* o = L.pt + (slot0 + k) * PT_STRIDE;
    const int p = (int)o[PT_ON] - 1;
    const int ba = m->cap_body[m->pair_a[p]], bb = m->cap_body[m->pair_b[p]];
    const float ma = body_mass(C, m->dyn[ba]), mb = body_mass(C, m->dyn[bb]);
    if (ba == l) slot_force_fb(o, abr, 1.0f, C.sp.dt, 1.0f, 1.0f + ma / mb, f);
    if (bb == l) slot_force_fb(o, abr, -1.0f, C.sp.dt, 1.0f, 1.0f + mb / ma, f);
  }
  contact_out[3 * l] = f[0]; contact_out[3 * l + 1] = f[1]; contact_out[3 * l + 2] = f[2];
}

// Slot indexing inside the env's contact region, after the articulation's np sample points.  Only the FREE boxes own
// slots -- 8 corners x (terrain + every box) each, then one slot per (rounded shape, free box) and one pair record per such
// slot -- numbered by the free box's rank among the free boxes (a fixed box has no contacts of its own to fold).
struct SlotLay { int np, nbx, ndyn, nsph; unsigned ranks, dynmask; };   // ranks: 2 bits per box; dynmask: bit kd = box kd is free
__host__ __device__ inline int box_slot_count(int nbx, int ndyn, int nsph) { return ndyn * 8 * (1 + nbx) + 2 * nsph * ndyn; }
template <class SC>
DEV SlotLay slot_lay(const ShfModel* m, const ShfScene* S) {
  SlotLay Q;
  Q.np = m->np; Q.nsph = m->nsph;
  if constexpr (SC::NBX > 0) {
    Q.nbx = SC::NBX; Q.ndyn = 1; Q.ranks = 0u; Q.dynmask = 1u << SC::DYN;
  } else {
    Q.nbx = S ? S->nboxes : 0; Q.ndyn = 0; Q.ranks = 0u; Q.dynmask = 0u;
    for (int k = 0; k < Q.nbx; k++)
      if (box_is_dynamic(S->box[k])) { Q.ranks |= (unsigned)Q.ndyn << (2 * k); Q.dynmask |= 1u << k; Q.ndyn++; }
  }
  return Q;
}
DEV int lay_rank(const SlotLay& Q, int kd) { return (int)((Q.ranks >> (2 * kd)) & 3u); }
DEV int corner_slot(const SlotLay& Q, int kd, int c, int tg) { return Q.np + (lay_rank(Q, kd) * 8 + c) * (1 + Q.nbx) + tg; }
DEV int sphere_slot(const SlotLay& Q, int si, int kd) { return Q.np + Q.ndyn * 8 * (1 + Q.nbx) + si * Q.ndyn + lay_rank(Q, kd); }
// second record of a pair slot: the consistent law's (Feff[3], Keff upper triangle [6]) -- see pair_law
DEV int pair_slot(const SlotLay& Q, int si, int kd) { return sphere_slot(Q, si, kd) + Q.nsph * Q.ndyn; }
DEV int box_slots(const SlotLay& Q) { return box_slot_count(Q.nbx, Q.ndyn, Q.nsph); }
// joint pair law (pair_law_joint): which capsule, if any, has exactly its two ends as box kd's active pair slots; where the
// record lives (the box's exchange slot: boxes hand nothing to a parent); the force of sphere slot si either way
DEV int box_joint_pair(const ShfModel* m, const EnvLds& L, const SlotLay& Q, int kd, unsigned sb, int nlink_on_box) {
#ifdef SHF_EXP_NO_JOINT_LAW   /* timing experiment only (tools/mlp_probe.py-style build): the independent laws everywhere */
  return -1;
#endif
  return joint_pair_of(m, sb, nlink_on_box, [&](int si) { return L.pt[sphere_slot(Q, si, kd) * PT_STRIDE + PT_CT]; });
}
DEV float* joint_record(const ShfModel* m, const EnvLds& L, int kd) { return L.xch + (m->nb + kd) * XCH_STRIDE; }
DEV void sphere_pair_force(const ShfModel* m, const EnvLds& L, const SlotLay& Q, int si, int kd, int jp, float dt, float* f) {
  const float* ab = L.acc + m->dyn[m->sph_body[si]] * 6;
  const float abr[6] = {ab[0], ab[1], ab[2], ab[3], ab[4], ab[5]};
  if (jp < 0) {
    const float* o = L.pt + sphere_slot(Q, si, kd) * PT_STRIDE;
    const float r[3] = {o[PT_R], o[PT_R + 1], o[PT_R + 2]};
    pair_force(L.pt + pair_slot(Q, si, kd) * PT_STRIDE, abr, r, dt, f);
  } else {
    const float* o1 = L.pt + sphere_slot(Q, jp, kd) * PT_STRIDE;
    const float* o2 = L.pt + sphere_slot(Q, jp + 1, kd) * PT_STRIDE;
    const float r1[3] = {o1[PT_R], o1[PT_R + 1], o1[PT_R + 2]}, r2[3] = {o2[PT_R], o2[PT_R + 1], o2[PT_R + 2]};
    float f1[3], f2[3];
    pair_force_joint(joint_record(m, L, kd), abr, r1, r2, dt, f1, f2);
#pragma unroll
    for (int k = 0; k < 3; k++) f[k] = si == jp ? f1[k] : f2[k];
  }
}

// `in contact` flags of a lane's slots as bit masks, read in one batch: the folds below then visit only the active
// slots, in the same order as the plain nested loops (ascending bit index = loop order), instead of paying one
// LDS round trip per slot to find out that almost all of them are idle.
#define BOX_T (SHF_MAX_BOXES + 1)
DEV unsigned long long corner_flags(const ShfModel* m, const EnvLds& L, const SlotLay& Q, int kd) {   // bit c * BOX_T + tg
  unsigned long long bits = 0ull;
#pragma unroll
  for (int c = 0; c < 8; c++)
#pragma unroll
    for (int tg = 0; tg < BOX_T; tg++) {
      const bool ok = tg <= Q.nbx;
      const float f = L.pt[corner_slot(Q, kd, c, ok ? tg : 0) * PT_STRIDE + PT_ON];
      if (ok && f != 0.0f) bits |= 1ull << (c * BOX_T + tg);
    }
  return bits;
}
DEV unsigned box_sphere_flags(const ShfModel* m, const EnvLds& L, const SlotLay& Q, int kd) {   // bit si
  unsigned bits = 0u;
#pragma unroll
  for (int si = 0; si < SHF_MAX_SPHERES; si++) {
    const bool ok = si < m->nsph;
    const float f = L.pt[sphere_slot(Q, ok ? si : 0, kd) * PT_STRIDE + PT_ON];
    if (ok && f != 0.0f) bits |= 1u << si;
  }
  return bits;
}
// spheres of the articulation: bit si * SHF_MAX_BOXES + kd, restricted to spheres with owner[si] == l
DEV unsigned body_sphere_flags(const ShfModel* m, const EnvLds& L, const SlotLay& Q, int l, bool by_dyn) {
  unsigned bits = 0u;
#pragma unroll
  for (int si = 0; si < SHF_MAX_SPHERES; si++) {
    const bool mine = si < m->nsph && (by_dyn ? m->dyn[m->sph_body[si]] : m->sph_body[si]) == l;
#pragma unroll
    for (int kd = 0; kd < SHF_MAX_BOXES; kd++) {
      const bool ok = mine && ((Q.dynmask >> kd) & 1u) != 0u;   // only free boxes own sphere slots
      const float f = L.pt[sphere_slot(Q, ok ? si : 0, ok ? kd : 0) * PT_STRIDE + PT_ON];
      if (ok && f != 0.0f) bits |= 1u << (si * SHF_MAX_BOXES + kd);
    }
  }
  return bits;
}

#include "shf_hull.h"   // the convex narrow phase (hulls, face manifolds): uses segment_closest above

// ------------------------------------------------------------ link contacts --
// ShfModel.link_collide (include/shifu_amd.h; oracle boxes_pre "link contacts"): the articulation's collision shapes
// against the box actors of its env beyond the rounded-shape pair slots -- (A) sample points x boxes, (C) rounded shapes
// x fixed boxes, (B) box corners x articulation box volumes -- one lane per candidate and round, the active ones
// compacted in candidate order into at most SHF_MAX_LINK_CONTACTS slots from `slot0` on (their `on` word holds
// 1 + body * SHF_MAX_BOXES + box; the pair record of slot k sits at slot0 + SHF_MAX_LINK_CONTACTS + k).  A slot's
// normal points towards the articulation, which receives +f.  Returns the number of active slots.
DEV int link_code(int body, int box) { return 1 + body * SHF_MAX_BOXES + box; }
DEV int link_code_body(float on) { return ((int)on - 1) / SHF_MAX_BOXES; }
DEV int link_code_box(float on) { return ((int)on - 1) % SHF_MAX_BOXES; }
DEV int link_slots_on_box(const EnvLds& L, int link_slot0, int nlink, int kd) {
  int c = 0;
  for (int k = 0; k < nlink; k++) c += link_code_box(L.pt[(link_slot0 + k) * PT_STRIDE + PT_ON]) == kd ? 1 : 0;
  return c;
}
// Candidates pair by pair (reported body ascending, box actor ascending; oracle boxes_pre "link contacts"); within a pair
// four homogeneous passes -- (A) the body's sample points, (C) its rounded shapes when the box is fixed, (B) the box's
// corners in the body's volumes, (E) edge-edge crossings of its volumes with the box -- one lane per candidate and round.
// Stage 0, the broad phase: one lane per body, its bounding box (ShfModel.bbox) against every box actor, two oriented
// boxes tested along their six face normals; a pair separated by more than the contact offset (+ 1 cm) has no active
// candidate in any family -- every shape of the body lies inside its box, and a corner of the box actor inside one of the
// body's volumes lies inside it too -- and is skipped without touching the order of what is left.
struct LinkCtx {
  const StepCtx& C; const EnvLds& L;
  int l, lane0, slot0, count;
  unsigned long long gmask, below;
  float mu_shape; const float* g_art;
};
// append the lanes' active slots, in lane order
DEV void link_append(LinkCtx& X, const float* slot, int body, int box) {
  const bool on = slot[PT_ON] != 0.0f;
  const unsigned long long mask = (__ballot(on) >> X.lane0) & X.gmask;
  const int mine = X.count + __builtin_popcountll(mask & X.below);
  if (on && mine < SHF_MAX_LINK_CONTACTS) {
    float* o = X.L.pt + (X.slot0 + mine) * PT_STRIDE;
#pragma unroll
    for (int k = 0; k < PT_STRIDE - 1; k++) o[k] = slot[k];
    o[PT_ON] = (float)link_code(body, box);
  }
  X.count += __builtin_popcountll(mask);
}
// EXT: with the families (F) and (H) of the convex narrow phase compiled in (csrc/shf_hull.h) -- the run-time-shaped kernels of
// scenes with ShfScene.flags or hulls; the other instantiations are rounds 1-5's code
template <int G, bool EXT = false>
DEV int link_contacts(const StepCtx& C, const EnvLds& L, int l, int slot0, float mu_shape, const float* g_art) {
  const ShfModel* m = C.m;
  const SceneDev* S = C.scene;
  const int nb = m->nb, nbx = S->nboxes;
  const bool hard = C.sp.solver != SHF_SOLVER_COMPLIANT;     // candidates only (slot_eval); family B by signed distance
  const float dt = C.sp.dt, kc = hard ? -1.0f : C.sp.contact_k, veps = C.sp.friction_vel, vdep = C.sp.max_depen_vel;
  const float offset = hard ? C.sp.contact_offset + C.sp.rest_offset : C.sp.contact_offset;
  const float beta = hard ? C.sp.rest_offset : fmaf(kc, dt, C.sp.contact_d);
  const float gb[3] = {C.sp.gravity[0], C.sp.gravity[1], C.sp.gravity[2]};
  PHASE_BEGIN();
  LinkCtx X = {C, L, l, (int)(threadIdx.x & 63u) - l, slot0, 0, G >= 64 ? ~0ull : ((1ull << (G & 63)) - 1ull), (1ull << l) - 1ull,
               mu_shape, g_art};
  // (0) live: bit kd * 32 + b = body b may touch box kd.  (Run-time loops over the boxes here and below: unrolled four
  // times this function alone was 50 KB of code, and the step kernels already overflow the 64 KB instruction cache.)
  static_assert(SHF_MAX_BOXES * 32 <= 128 && SHF_MAX_BODIES <= 32, "live-pair masks");
  unsigned long long live[2] = {0ull, 0ull};
  for (int j = 0; j * G < nb; j++) {
    const int b = l + j * G;
    const bool has = b < nb && m->bbox[b][3] >= 0.0f;
    // the body's box (columns of Rb, centre ca, half extents ha) against box actor kd (columns of Rk, pk, hk):
    // separated along a face normal of either by more than the margin?
    const float* pb = L.pose + (has ? b : 0) * POSE_STRIDE;
    float Rb[9], ca[3];
#pragma unroll
    for (int i = 0; i < 9; i++) Rb[i] = pb[i];
    const float* bx = m->bbox[has ? b : 0];
    const float lc[3] = {bx[0], bx[1], bx[2]}, ha[3] = {bx[3], bx[4], bx[5]};
    mv3(Rb, lc, ca);
#pragma unroll
    for (int i = 0; i < 3; i++) ca[i] += pb[9 + i];
    const float margin = offset + 0.01f;
#pragma unroll 1
    for (int kd = 0; kd < nbx; kd++) {
      const float* pk = L.pose + (nb + kd) * POSE_STRIDE;
      const ShfBoxDesc& bd = S->box[kd];
      const float hk[3] = {0.5f * bd.dim[0], 0.5f * bd.dim[1], 0.5f * bd.dim[2]};
      const float t[3] = {pk[9] - ca[0], pk[10] - ca[1], pk[11] - ca[2]};
      float Rq[3][3];   // |a_i . b_j|
#pragma unroll
      for (int i = 0; i < 3; i++)
#pragma unroll
        for (int jj = 0; jj < 3; jj++) Rq[i][jj] = fabsf(fmaf(Rb[6 + i], pk[6 + jj], fmaf(Rb[3 + i], pk[3 + jj], Rb[i] * pk[jj])));
      bool apart = false;
#pragma unroll
      for (int i = 0; i < 3; i++) {
        const float ta = fabsf(fmaf(Rb[6 + i], t[2], fmaf(Rb[3 + i], t[1], Rb[i] * t[0])));
        const float ra = ha[i] + fmaf(hk[2], Rq[i][2], fmaf(hk[1], Rq[i][1], hk[0] * Rq[i][0]));
        apart = apart || ta > (ra + margin) * 1.0001f;
        const float tb = fabsf(fmaf(pk[6 + i], t[2], fmaf(pk[3 + i], t[1], pk[i] * t[0])));
        const float rb = hk[i] + fmaf(ha[2], Rq[2][i], fmaf(ha[1], Rq[1][i], ha[0] * Rq[0][i]));
        apart = apart || tb > (rb + margin) * 1.0001f;
      }
      const unsigned long long near = ((__ballot(has && !apart) >> X.lane0) & X.gmask) << (j * G + (kd & 1) * 32);
      if (kd < 2) live[0] |= near; else live[1] |= near;
    }
  }
  if (m->bounds_ok != SHF_BOUNDS_MAGIC) live[0] = live[1] = ~0ull;   // (cannot happen through shf_sim_bind) test every pair
  // the pairs some env group of this wavefront has to look at: the loops below are wave-uniform
  unsigned long long any[2];
#pragma unroll
  for (int w = 0; w < 2; w++) {
    unsigned lo = (unsigned)live[w], hi = (unsigned)(live[w] >> 32);
    if (G < 64) {
#pragma unroll
      for (int sh = G; sh < 64; sh <<= 1) { lo |= (unsigned)__shfl_xor((int)lo, sh, 64); hi |= (unsigned)__shfl_xor((int)hi, sh, 64); }
    }
    any[w] = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)lo) |
             ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)hi) << 32);
  }
#ifdef SHF_PHASE_CLOCK
  if (blockIdx.x == 0 && threadIdx.x == 0) {   // tools/phase_clock.py --link: live pairs of this wavefront / of its first env, calls
    g_phase_cycles[29] += __builtin_popcountll(any[0]) + __builtin_popcountll(any[1]);
    g_phase_cycles[30] += __builtin_popcountll(live[0]) + __builtin_popcountll(live[1]);
    g_phase_cycles[31] += 1;
  }
#endif
  PHASE_MARK(24);
#pragma unroll 1
  for (int b = 0; b < nb; b++) {
#pragma unroll 1
    for (int kd = 0; kd < nbx; kd++) {
      const int bit = (kd & 1) * 32 + b;
      if (!(((kd < 2 ? any[0] : any[1]) >> bit) & 1ull)) continue;
      const bool mine = ((kd < 2 ? live[0] : live[1]) >> bit) & 1ull;
      const ShfBoxDesc& bd = S->box[kd];
      const bool dynb = box_is_dynamic(bd);
      const float* pb = L.pose + b * POSE_STRIDE;
      const float* pk = L.pose + (nb + kd) * POSE_STRIDE;
      const float* qa = L.pose + m->dyn[b] * POSE_STRIDE;
      float Rb[9], Rk[9];
#pragma unroll
      for (int i = 0; i < 9; i++) { Rb[i] = pb[i]; Rk[i] = pk[i]; }
      const float hh[3] = {0.5f * bd.dim[0], 0.5f * bd.dim[1], 0.5f * bd.dim[2]}, bpos[3] = {pk[9], pk[10], pk[11]};
      const float va[3] = {qa[12], qa[13], qa[14]}, vla[3] = {qa[15], qa[16], qa[17]};
      const float vbx[3] = {pk[12], pk[13], pk[14]}, vlb[3] = {pk[15], pk[16], pk[17]};
      const float mu_pair = 0.5f * (mu_shape + bd.friction);
      const float hdiag = sqrtf(dot3(hh, hh));
      const int p0 = m->lc_range[b][0], pn = m->lc_range[b][1], a0 = m->lc_range[b][2], an = m->lc_range[b][3];
      const int count0 = X.count;      // (E) runs only for a pair without an active vertex / rounded-shape contact
      PHASE_MARK(25);
      // (A) the body's sample points against the box, each a sphere
      for (int j = 0; j * G < pn; j++) {
        const int q = l + j * G;
        float slot[PT_STRIDE];
        slot[PT_ON] = 0.0f;
        if (mine && q < pn) {
          const int i = m->lc_pt[p0 + q];
          const float lp[3] = {m->pt_pos[i][0], m->pt_pos[i][1], m->pt_pos[i][2]}, rad = m->pt_radius[i];
          float c[3];
          mv3(Rb, lp, c);
#pragma unroll
          for (int k = 0; k < 3; k++) c[k] += pb[9 + k];
          const float rel[3] = {c[0] - bpos[0], c[1] - bpos[1], c[2] - bpos[2]};
          const float reach = hdiag + rad + offset + 0.01f;       // oracle: rounded_far with a zero segment
          // squared distance from the box, a centimetre of slack: a point further away than its radius + the contact offset
          // has a gap > offset, slot_eval would switch it off -- the expensive part (square root, normal, velocities) is skipped
          float d2 = 0.0f;
#pragma unroll
          for (int k = 0; k < 3; k++) {
            const float d = fmaf(Rk[6 + k], rel[2], fmaf(Rk[3 + k], rel[1], Rk[k] * rel[0]));
            const float e = rmaxf(fabsf(d) - hh[k], 0.0f);
            d2 = fmaf(e, e, d2);
          }
          const float rs = rad + offset + 0.01f;
          if (!(dot3(rel, rel) > reach * reach) && !(d2 > rs * rs * 1.0001f)) {
            float phi, n[3], rc[3], ta[3], tb[3], vrel[3], vrs[3];
            sphere_vs_box(Rk, bpos, hh, c, rad, &phi, n, rc);
            if (phi < offset) {                    // (slot_eval's own first test: further away the slot is off)
              cross3(va, rc, ta);
              cross3(vbx, rc, tb);
#pragma unroll
              for (int k = 0; k < 3; k++) {
                const float pa = vla[k] + ta[k], pq = vlb[k] + tb[k];
                vrs[k] = pa - pq;
                vrel[k] = fmaf(dt, g_art[k], pa) - (dynb ? fmaf(dt, gb[k], pq) : pq);
              }
              slot_eval(slot, phi, n, rc, vrs, vrel, mu_pair, kc, beta, veps, vdep, dt, offset);
            }
          }
        }
        if (__ballot(slot[PT_ON] != 0.0f) != 0ull) link_append(X, slot, b, kd);
      }
      PHASE_MARK(26);
      // (C) its rounded shapes against a fixed box (a free box has the pair slots)
      if (!dynb && m->nsph > 0) {
        static_assert(SHF_MAX_SPHERES <= 16, "one round of rounded shapes");
        float slot[PT_STRIDE];
        slot[PT_ON] = 0.0f;
        if (mine && l < m->nsph && m->sph_body[l] == b) {
          const int si = l;
          const float lp[3] = {m->sph_pos[si][0], m->sph_pos[si][1], m->sph_pos[si][2]};
          const float ls[3] = {m->sph_seg[si][0], m->sph_seg[si][1], m->sph_seg[si][2]}, rad = m->sph_radius[si];
          float c[3], sw[3];
          mv3(Rb, lp, c);
#pragma unroll
          for (int k = 0; k < 3; k++) c[k] += pb[9 + k];
          mv3(Rb, ls, sw);
          const float rel[3] = {fmaf(0.5f, sw[0], c[0]) - bpos[0], fmaf(0.5f, sw[1], c[1]) - bpos[1], fmaf(0.5f, sw[2], c[2]) - bpos[2]};
          const float reach = 0.5f * sqrtf(dot3(sw, sw)) + hdiag + rad + offset + 0.01f;
          bool exists = !(dot3(rel, rel) > reach * reach);
          if (exists) {
            // not in the oracle, and no need to be: the shape's bounding sphere (centre: the middle of its segment) further
            // from the box than its radius + the contact offset + 1 cm means a gap > offset wherever the closest point is
            float d2 = 0.0f;
#pragma unroll
            for (int k = 0; k < 3; k++) {
              const float d = fmaf(Rk[6 + k], rel[2], fmaf(Rk[3 + k], rel[1], Rk[k] * rel[0]));
              const float e = rmaxf(fabsf(d) - hh[k], 0.0f);
              d2 = fmaf(e, e, d2);
            }
            const float rs = 0.5f * sqrtf(dot3(sw, sw)) + rad + offset + 0.01f;
            exists = !(d2 > rs * rs * 1.0001f);
          }
          if (exists && (ls[0] != 0.0f || ls[1] != 0.0f || ls[2] != 0.0f)) {
            float t = 0.0f;
            exists = segment_box_contact(Rk, bpos, hh, c, sw, m->sph_part[si], &t);   // part 1: only for a line contact
#pragma unroll
            for (int k = 0; k < 3; k++) c[k] = fmaf(t, sw[k], c[k]);
          }
          if (exists) {
            float phi, n[3], rc[3], ta[3], vrel[3], vrs[3];
            sphere_vs_box(Rk, bpos, hh, c, rad, &phi, n, rc);
            cross3(va, rc, ta);
#pragma unroll
            for (int k = 0; k < 3; k++) { vrs[k] = vla[k] + ta[k]; vrel[k] = fmaf(dt, g_art[k], vrs[k]); }   // the box does not move
            slot_eval(slot, phi, n, rc, vrs, vrel, mu_pair, kc, beta, veps, vdep, dt, offset);
          }
        }
        if (__ballot(slot[PT_ON] != 0.0f) != 0ull) link_append(X, slot, b, kd);
      }
      PHASE_MARK(27);
      // (B) the box's corners inside the body's volumes: candidate q = volume * 8 + corner; and behind them, in the same
      // rounds, (E) an edge of one of the body's volumes across an edge of the box actor (box_box_edge): candidate
      // q = 8 x volumes + volume -- only for a pair none of whose points / rounded shapes (A, C) is in contact
#ifdef SHF_EXP_NO_ELINK   /* timing experiment only (tools/experiment.py) */
      const bool edges = false;
#else
      const bool edges = mine && X.count == count0;
#endif
      // (when all corner candidates fit one round the edge tests ride in it, on the lanes of each volume's corner 0; otherwise
      // they take the lanes behind the corners)
      const bool oneround = an * 8 <= G;
      const int nbe = an * 8 + ((!oneround && __ballot(edges) != 0ull) ? an : 0);      // wave-uniform trip count
      for (int j = 0; j * G < nbe; j++) {
        const int q = l + j * G;
        float slot[PT_STRIDE], slotE[PT_STRIDE];
        slot[PT_ON] = 0.0f;
        slotE[PT_ON] = 0.0f;
        if (mine && q < an * 8) {
          const int jb = m->lc_abox[a0 + (q >> 3)], cn = q & 7;
          float lr[9], ar[9], ac[3], r[3], tb[3];
          const float lc[3] = {((cn & 4) ? 0.5f : -0.5f) * bd.dim[0], ((cn & 2) ? 0.5f : -0.5f) * bd.dim[1],
                               ((cn & 1) ? 0.5f : -0.5f) * bd.dim[2]};
          mv3(Rk, lc, r);
#pragma unroll
          for (int k = 0; k < 3; k++) r[k] += bpos[k];
          const float lp[3] = {m->abox_pos[jb][0], m->abox_pos[jb][1], m->abox_pos[jb][2]};
          const float ah[3] = {m->abox_half[jb][0], m->abox_half[jb][1], m->abox_half[jb][2]};
          mv3(Rb, lp, ac);
#pragma unroll
          for (int k = 0; k < 3; k++) ac[k] += pb[9 + k];
          const float rel[3] = {r[0] - ac[0], r[1] - ac[1], r[2] - ac[2]};
          float phi, n[3];
          bool inside = !(dot3(rel, rel) > fmaf(dot3(ah, ah), 1.01f, 1e-8f));   // outside the volume's bounding sphere: outside the volume
          if (hard) {     // ... or further from it than the contact offset (velocity-level solve: the corner's signed distance)
            const float reach = sqrtf(dot3(ah, ah)) + offset + 1e-4f;
            inside = !(dot3(rel, rel) > reach * reach * 1.01f);
          }
          if (inside) {
#pragma unroll
            for (int k = 0; k < 9; k++) lr[k] = m->abox_rot[jb][k];
            mm3(Rb, lr, ar);
            if (hard) {
              float rc[3];
              sphere_vs_box(ar, ac, ah, r, 0.0f, &phi, n, rc);
            } else {
              inside = point_in_box(ar, ac, ah, r, &phi, n);
            }
          }
          if (inside) {
            float ta[3], nn[3], vrel[3], vrs[3];
            cross3(vbx, r, tb);
            cross3(va, r, ta);
#pragma unroll
            for (int k = 0; k < 3; k++) {
              const float pa = vla[k] + ta[k], pq = vlb[k] + tb[k];
              nn[k] = -n[k];
              vrs[k] = pa - pq;
              vrel[k] = fmaf(dt, g_art[k], pa) - (dynb ? fmaf(dt, gb[k], pq) : pq);
            }
            slot_eval(slot, phi, nn, r, vrs, vrel, mu_pair, kc, beta, veps, vdep, dt, offset);
          }
        }
        if (edges && (oneround ? (q < an * 8 && (q & 7) == 0) : (q >= an * 8 && q < an * 9))) {
          const int jb = m->lc_abox[a0 + (oneround ? (q >> 3) : (q - an * 8))];
          float lr[9], ar[9], ac[3];
#pragma unroll
          for (int k = 0; k < 9; k++) lr[k] = m->abox_rot[jb][k];
          const float lp[3] = {m->abox_pos[jb][0], m->abox_pos[jb][1], m->abox_pos[jb][2]};
          const float ah[3] = {m->abox_half[jb][0], m->abox_half[jb][1], m->abox_half[jb][2]};
          mm3(Rb, lr, ar);
          mv3(Rb, lp, ac);
#pragma unroll
          for (int k = 0; k < 3; k++) ac[k] += pb[9 + k];
          float phi, n[3], r[3];
          if (box_box_edge(ar, ac, ah, Rk, bpos, hh, offset, &phi, n, r)) {
            float ta[3], tb[3], vrel[3], vrs[3];
            cross3(va, r, ta);
            cross3(vbx, r, tb);
#pragma unroll
            for (int k = 0; k < 3; k++) {
              const float pa = vla[k] + ta[k], pq = vlb[k] + tb[k];
              vrs[k] = pa - pq;
              vrel[k] = fmaf(dt, g_art[k], pa) - (dynb ? fmaf(dt, gb[k], pq) : pq);
            }
            slot_eval(slotE, phi, n, r, vrs, vrel, mu_pair, kc, beta, veps, vdep, dt, offset);
          }
        }
        if (__ballot(slot[PT_ON] != 0.0f) != 0ull) link_append(X, slot, b, kd);
        if (__ballot(slotE[PT_ON] != 0.0f) != 0ull) link_append(X, slotE, b, kd);
      }
      // (F) ShfScene.flags & SHF_SCENE_FACE_MANIFOLD: a pair without any contact of the families above gets the clipped face manifold
      // of each of the body's box volumes with the box actor; (H) the body's convex hulls (SHF_T_HULLS) against the box actor: the
      // convex narrow phase (csrc/shf_hull.h), the lanes of the env sharing its axes; up to four contacts each, on lanes 0..3
      if constexpr (EXT) {
        const bool flagF = (S->flags & SHF_SCENE_FACE_MANIFOLD) != 0;
        const int nh = (m->nhull > 0 && C.hulls != nullptr) ? C.hulls->nhull : 0;
        const bool wantF = flagF && mine && X.count == count0;
#pragma unroll 1
        for (int pass = 0; pass < 2; pass++) {
          const int count = pass == 0 ? ((flagF && __ballot(wantF) != 0ull) ? an : 0) : nh;
#pragma unroll 1
          for (int j = 0; j < count; j++) {
            if (pass == 1 && C.hulls->hull[j].body != b) continue;
            const bool active = pass == 0 ? wantF : mine;
            if (__ballot(active) == 0ull) continue;
            PolyDev PA, PB;
            if (pass == 0) {
              const int jb = m->lc_abox[a0 + j];
              float lr[9];
#pragma unroll
              for (int k = 0; k < 9; k++) lr[k] = m->abox_rot[jb][k];
              const float lp[3] = {m->abox_pos[jb][0], m->abox_pos[jb][1], m->abox_pos[jb][2]};
              mm3(Rb, lr, PA.R);
              mv3(Rb, lp, PA.p);
#pragma unroll
              for (int k = 0; k < 3; k++) { PA.p[k] += pb[9 + k]; PA.hx[k] = m->abox_half[jb][k]; }
              PA.h = nullptr;
            } else {
              PA.h = &C.hulls->hull[j];
#pragma unroll
              for (int k = 0; k < 9; k++) PA.R[k] = Rb[k];
#pragma unroll
              for (int k = 0; k < 3; k++) { PA.p[k] = pb[9 + k]; PA.hx[k] = 0.0f; }
            }
            PB.h = nullptr;
#pragma unroll
            for (int k = 0; k < 9; k++) PB.R[k] = Rk[k];
#pragma unroll
            for (int k = 0; k < 3; k++) { PB.p[k] = bpos[k]; PB.hx[k] = hh[k]; }
            float mr[4][3], mn[3] = {0.0f, 0.0f, 0.0f}, mphi[4];
            int nc = convex_manifold_dev<G>(PA, PB, offset, l, mr, mn, mphi);
            if (!active) nc = 0;
            float slot[PT_STRIDE];
            slot[PT_ON] = 0.0f;
            if (l < nc) {
              float rq[3] = {mr[0][0], mr[0][1], mr[0][2]}, ph = mphi[0];
#pragma unroll
              for (int q = 1; q < 4; q++)
                if (l == q) { rq[0] = mr[q][0]; rq[1] = mr[q][1]; rq[2] = mr[q][2]; ph = mphi[q]; }
              float ta[3], tb[3], vrel[3], vrs[3];
              cross3(va, rq, ta);
              cross3(vbx, rq, tb);
#pragma unroll
              for (int k = 0; k < 3; k++) {
                const float pa = vla[k] + ta[k], pq = vlb[k] + tb[k];
                vrs[k] = pa - pq;
                vrel[k] = fmaf(dt, g_art[k], pa) - (dynb ? fmaf(dt, gb[k], pq) : pq);
              }
              slot_eval(slot, ph, mn, rq, vrs, vrel, mu_pair, kc, beta, veps, vdep, dt, offset);
            }
            if (__ballot(slot[PT_ON] != 0.0f) != 0ull) link_append(X, slot, b, kd);
          }
        }
      }
      PHASE_MARK(28);
    }
  }
  int count = X.count;
  if (count > SHF_MAX_LINK_CONTACTS) {
    if (l == 0 && C.dropped) *C.dropped += count - SHF_MAX_LINK_CONTACTS;
    count = SHF_MAX_LINK_CONTACTS;
  }
  return count;
}

// Box lanes: pose / spatial velocity about O from the root-state rows, inertia of the free ones.
template <int G>
DEV void boxes_pose(const StepCtx& C, const EnvLds& L, int l, BodyRegs& B) {
  const ShfModel* m = C.m;
  const int nb = m->nb, nbx = C.scene->nboxes;
  const int k = l - nb;
  if (k >= 0 && k < nbx) {
    const float* row = L.root + 13 * (1 + k);
    const ShfBoxDesc& bd = C.scene->box[k];
    quat_to_mat(row + 3, B.Rw);
#pragma unroll
    for (int i = 0; i < 3; i++) B.p[i] = row[i] - L.root[i];
#pragma unroll
    for (int i = 0; i < 6; i++) B.v[i] = 0.0f;
    if (box_is_dynamic(bd)) {
      const float ang[3] = {row[10], row[11], row[12]};
      float t[3];
      cross3(ang, B.p, t);
#pragma unroll
      for (int i = 0; i < 3; i++) { B.v[i] = ang[i]; B.v[3 + i] = row[7 + i] - t[i]; }
      const float mass = bd.mass, dx = bd.dim[0], dy = bd.dim[1], dz = bd.dim[2];
      const float I6[6] = {mass * (dy * dy + dz * dz) / 12.0f, 0.0f, 0.0f, mass * (dx * dx + dz * dz) / 12.0f, 0.0f,
                           mass * (dx * dx + dy * dy) / 12.0f};
      const float com[3] = {0.0f, 0.0f, 0.0f};
      rigid_inertia(mass, com, I6, B);
    }
    float* o = L.pose + l * POSE_STRIDE;
#pragma unroll
    for (int i = 0; i < 9; i++) o[i] = B.Rw[i];
#pragma unroll
    for (int i = 0; i < 3; i++) o[9 + i] = B.p[i];
#pragma unroll
    for (int i = 0; i < 6; i++) o[12 + i] = B.v[i];
  }
  GROUP_SYNC();
}

// The fixed-scene form of boxes_contacts -- the free box's corner slots and NSPH sphere slots, one lane each -- in five
// pieces, so that the wave-specialised ABB step (shf_api.hip: k_abb_step_ws) can run the box's pieces and the
// articulation's on different waves; boxes_contacts_fixed below strings them together for one wave.
template <int G, class SC>
struct FixedSceneConsts {
  static constexpr int nbx = SC::NBX, T = 1 + SC::NBX, kd = SC::DYN, NO = SC::NBX - 1, NBS = 8 * (SC::NBX - 1);
  static_assert(8 * T <= 64 && SC::NSPH <= 16, "fixed scene too large for the ballot masks");
  static_assert(NO >= 0 && NBS <= 64, "corner ballots");
  DEV static int lane0(int l) { return (int)(threadIdx.x & 63u) - l; }
  DEV static unsigned long long gmask() { return G >= 64 ? ~0ull : ((1ull << (G & 63)) - 1ull); }
};
// Corner slots of the free box.  Round(s) 1: corner x other box, one code path for every lane.  Round 2: corner x
// terrain -- skipped (slots off) when the terrain is the plane z = 0 and the box's bounding sphere clears it by more
// than the contact offset: every corner's gap then exceeds the offset and slot_eval would switch the slot off.
// HARD (the velocity-level solve): candidate constraints only, with boxes_contacts<.., HARD>'s constants and a corner's signed
// distance to a fixed box
template <int G, class SC, bool HARD = false>
DEV void fixed_corner_slots(const StepCtx& C, const EnvLds& L, int l, BoxMasks& BM) {
  typedef FixedSceneConsts<G, SC> F;
  constexpr int nbx = F::nbx, kd = F::kd, NO = F::NO, NBS = F::NBS;
  const ShfModel* m = C.m;
  const SlotLay Q = slot_lay<SC>(m, C.scene);
  const SceneDev* S = C.scene;
  const int nb = m->nb;
  const float dt = C.sp.dt, kc = HARD ? -1.0f : C.sp.contact_k, veps = C.sp.friction_vel, vdep = C.sp.max_depen_vel;
  const float offset = HARD ? C.sp.contact_offset + C.sp.rest_offset : C.sp.contact_offset;
  const float beta = HARD ? C.sp.rest_offset : fmaf(kc, dt, C.sp.contact_d);
  const float gb[3] = {C.sp.gravity[0], C.sp.gravity[1], C.sp.gravity[2]};
  const int lane0 = F::lane0(l);
  const unsigned long long gmask = F::gmask();
  const ShfBoxDesc& bd = S->box[kd];
  const float* pk = L.pose + (nb + kd) * POSE_STRIDE;
  float Rk[9], pcen[3], vb[3], vlin[3];
#pragma unroll
  for (int i = 0; i < 9; i++) Rk[i] = pk[i];
#pragma unroll
  for (int i = 0; i < 3; i++) { pcen[i] = pk[9 + i]; vb[i] = pk[12 + i]; vlin[i] = pk[15 + i]; }
  auto corner_place = [&](int c, float* r, float* vs, float* vp) {
    const float lc[3] = {((c & 4) ? 0.5f : -0.5f) * bd.dim[0], ((c & 2) ? 0.5f : -0.5f) * bd.dim[1], ((c & 1) ? 0.5f : -0.5f) * bd.dim[2]};
    float t[3];
    mv3(Rk, lc, r);
#pragma unroll
    for (int i = 0; i < 3; i++) r[i] += pcen[i];
    cross3(vb, r, t);
#pragma unroll
    for (int i = 0; i < 3; i++) { vs[i] = vlin[i] + t[i]; vp[i] = fmaf(dt, gb[i], vs[i]); }
  };
  BM.cbox = 0ull;
#pragma unroll
  for (int j = 0; j < (NBS + G - 1) / G; j++) {
    const int idx = l + j * G;
    const bool valid = idx < NBS;
    const int c = valid ? idx / (NO > 0 ? NO : 1) : 0, t = valid ? idx % (NO > 0 ? NO : 1) : 0;
    const int ks = t + (t >= kd ? 1 : 0);
    bool on = false;
    if (valid) {
      float* o = L.pt + corner_slot(Q, kd, c, 1 + ks) * PT_STRIDE;
      o[PT_ON] = 0.0f;
      float r[3], vs[3], vp[3], n[3], phi;
      corner_place(c, r, vs, vp);
      const ShfBoxDesc& bs = S->box[ks];
      const float* ps = L.pose + (nb + ks) * POSE_STRIDE;
      float Rs[9], hh[3] = {0.5f * bs.dim[0], 0.5f * bs.dim[1], 0.5f * bs.dim[2]}, bpos[3] = {ps[9], ps[10], ps[11]};
#pragma unroll
      for (int i = 0; i < 9; i++) Rs[i] = ps[i];
      if constexpr (HARD) {
        float rc[3];
        sphere_vs_box(Rs, bpos, hh, r, 0.0f, &phi, n, rc);
        slot_eval(o, phi, n, r, vs, vp, 0.5f * (bd.friction + bs.friction), kc, beta, veps, vdep, dt, offset);
      } else {
        if (point_in_box(Rs, bpos, hh, r, &phi, n))
          slot_eval(o, phi, n, r, vs, vp, 0.5f * (bd.friction + bs.friction), kc, beta, veps, vdep, dt, offset);
      }
      on = o[PT_ON] != 0.0f;
    }
    BM.cbox |= ((__ballot(on) >> lane0) & gmask) << (j * G);
  }
  {
    static_assert(G >= 8 + nbx, "one lane per corner, one per edge-edge slot");
    const bool valid = l < 8;
    bool on = false;
    // lanes 8 .. 8 + NBX - 1: the free box's edge-edge contact with fixed box ks = l - 8 (box_box_edge), kept in the slot
    // corner ks would have against its own box
    bool eon = false;
    if (l >= 8 && l < 8 + nbx && l - 8 != kd) {
      const int ks = l - 8;
      float* o = L.pt + corner_slot(Q, kd, ks, 1 + kd) * PT_STRIDE;
      o[PT_ON] = 0.0f;
      // a corner of the free box already in contact with box ks (a vertex-face situation): the edge test is not run
      unsigned long long vmask = 0ull;
#pragma unroll
      for (int c = 0; c < 8; c++) vmask |= 1ull << (c * (NO > 0 ? NO : 1) + (ks - (ks > kd ? 1 : 0)));
      const ShfBoxDesc& bs = S->box[ks];
#ifdef SHF_EXP_NO_EBOX   /* timing experiment only (tools/experiment.py) */
      if (false) {
#else
      if ((BM.cbox & vmask) == 0ull) {
#endif
      const float* ps = L.pose + (nb + ks) * POSE_STRIDE;
      float Rs[9];
#pragma unroll
      for (int i = 0; i < 9; i++) Rs[i] = ps[i];
      const float hd[3] = {0.5f * bd.dim[0], 0.5f * bd.dim[1], 0.5f * bd.dim[2]}, hs[3] = {0.5f * bs.dim[0], 0.5f * bs.dim[1], 0.5f * bs.dim[2]};
      const float cs[3] = {ps[9], ps[10], ps[11]};
      float phi, n[3], re[3], t[3], vs[3], vp[3];
      if (box_box_edge(Rk, pcen, hd, Rs, cs, hs, offset, &phi, n, re)) {
        cross3(vb, re, t);
#pragma unroll
        for (int i = 0; i < 3; i++) { vs[i] = vlin[i] + t[i]; vp[i] = fmaf(dt, gb[i], vs[i]); }
        slot_eval(o, phi, n, re, vs, vp, 0.5f * (bd.friction + bs.friction), kc, beta, veps, vdep, dt, offset);
        eon = o[PT_ON] != 0.0f;
      }
      }
    }
    BM.cedge = (unsigned)(((__ballot(eon) >> lane0) & gmask) >> 8);
    if (valid) {
      float* o = L.pt + corner_slot(Q, kd, l, 0) * PT_STRIDE;
      if (l >= nbx || l == kd) L.pt[corner_slot(Q, kd, l, 1 + kd) * PT_STRIDE + PT_ON] = 0.0f;   // no box l to cross edges with
      const float reach = 0.5f * sqrtf(fmaf(bd.dim[2], bd.dim[2], fmaf(bd.dim[1], bd.dim[1], bd.dim[0] * bd.dim[0])));
      // 1 % and a millimetre over the exact bound (the corner's own rounding cannot bridge it)
      const bool clear = C.terr.t.rows == 0 && (L.root[2] + pcen[2]) - reach * 1.01f - 1e-3f > offset;
      if (clear) {
        o[PT_ON] = 0.0f;
      } else {
        float r[3], vs[3], vp[3], n[3], h;
        corner_place(l, r, vs, vp);
        terrain_query(C.terr, L.root[0] + r[0], L.root[1] + r[1], &h, n);
        const float phi = (L.root[2] + r[2] - h) * n[2];
        slot_eval(o, phi, n, r, vs, vp, 0.5f * (bd.friction + C.terr.t.friction), kc, beta, veps, vdep, dt, offset);
        on = o[PT_ON] != 0.0f;
      }
    }
    BM.cplane = (unsigned)((__ballot(on) >> lane0) & gmask);
  }
}
// Rounded shapes of the articulation (sphere / capsule) against the free box: lane si < NSPH
template <int G, class SC, bool HARD = false>
DEV void fixed_sphere_slots(const StepCtx& C, const EnvLds& L, int l, float mu_shape, const float* g_art, BoxMasks& BM) {
  typedef FixedSceneConsts<G, SC> F;
  constexpr int nbx = F::nbx, kd = F::kd;
  const ShfModel* m = C.m;
  const SlotLay Q = slot_lay<SC>(m, C.scene);
  const int nb = m->nb;
  const float dt = C.sp.dt, kc = HARD ? -1.0f : C.sp.contact_k, veps = C.sp.friction_vel, vdep = C.sp.max_depen_vel;
  const float offset = HARD ? C.sp.contact_offset + C.sp.rest_offset : C.sp.contact_offset;
  const float beta = HARD ? C.sp.rest_offset : fmaf(kc, dt, C.sp.contact_d);
  const float gb[3] = {C.sp.gravity[0], C.sp.gravity[1], C.sp.gravity[2]};
  const ShfBoxDesc& bd = C.scene->box[kd];
  const float* pk = L.pose + (nb + kd) * POSE_STRIDE;
  const bool valid = l < SC::NSPH;
  const int si = valid ? l : 0;
  float* o = L.pt + sphere_slot(Q, si, kd) * PT_STRIDE;
  bool on = false;
  if (valid) {
    const int b = m->sph_body[si];
    const float* pb = L.pose + b * POSE_STRIDE;
    float Rb[9], Rk[9], c[3];
#pragma unroll
    for (int i = 0; i < 9; i++) { Rb[i] = pb[i]; Rk[i] = pk[i]; }
    const float hh[3] = {0.5f * bd.dim[0], 0.5f * bd.dim[1], 0.5f * bd.dim[2]}, bpos[3] = {pk[9], pk[10], pk[11]};
    if (rounded_centre(m, si, Rb, pb + 9, Rk, bpos, hh, offset, c)) {
      float phi, n[3], rc[3], ta[3], tb[3], vrel[3], vrs[3];
      sphere_vs_box(Rk, bpos, hh, c, m->sph_radius[si], &phi, n, rc);
      const float va[3] = {pb[12], pb[13], pb[14]}, vbx[3] = {pk[12], pk[13], pk[14]};
      cross3(va, rc, ta);
      cross3(vbx, rc, tb);
#pragma unroll
      for (int i = 0; i < 3; i++) {
        const float pa = pb[15 + i] + ta[i], pq = pk[15 + i] + tb[i];
        vrs[i] = pa - pq;
        vrel[i] = fmaf(dt, g_art[i], pa) - fmaf(dt, gb[i], pq);
      }
      slot_eval(o, phi, n, rc, vrs, vrel, 0.5f * (mu_shape + bd.friction), kc, beta, veps, vdep, dt, offset);
      on = o[PT_ON] != 0.0f;
    } else {
      o[PT_ON] = 0.0f;
    }
  }
  BM.spheres = (unsigned)((__ballot(on) >> F::lane0(l)) & F::gmask());
}
// fold, the box lane: its own contacts (corners ascending, terrain before boxes) ...
template <int G, class SC>
DEV void fixed_box_fold(const StepCtx& C, const EnvLds& L, int l, BodyRegs& B, const BoxMasks& BM) {
  typedef FixedSceneConsts<G, SC> F;
  const ShfModel* m = C.m;
  const SlotLay Q = slot_lay<SC>(m, C.scene);
  if (l == m->nb + F::kd) {
    unsigned pl = BM.cplane, ed = BM.cedge;
    unsigned long long bx = BM.cbox;
    int c, tg;
    while (corner_next<F::NO, F::kd>(pl, bx, ed, &c, &tg))
      slot_accumulate(B.IA, B.pA, L.pt + corner_slot(Q, F::kd, c, tg) * PT_STRIDE, 1.0f, C.sp.dt, 1.0f);
  }
}
// ... then the consistent law of every active pair slot (pair_law) into the pair records, by the lane `mine` that holds
// (or has read) the box's folded (IA, pA)
template <int G, class SC>
DEV void fixed_pair_laws(const StepCtx& C, const EnvLds& L, bool mine, const float* IA, const float* pA, unsigned spheres,
                         unsigned lb = 0u, int link_slot0 = 0) {
  typedef FixedSceneConsts<G, SC> F;
  const ShfModel* m = C.m;
  const SlotLay Q = slot_lay<SC>(m, C.scene);
  if (mine) {
    unsigned sb = spheres;
    const int jp = box_joint_pair(m, L, Q, F::kd, sb, __builtin_popcount(lb));
    if (jp >= 0) {                                // the two ends of one capsule: eliminated together
      float afree[6];
      Ldlt6 FIA;
      ldlt_factor6(IA, FIA);
      ldlt_substitute6(FIA, pA, afree);
      pair_law_joint(FIA, afree, L.pt + sphere_slot(Q, jp, F::kd) * PT_STRIDE, L.pt + sphere_slot(Q, jp + 1, F::kd) * PT_STRIDE,
                     C.sp.dt, joint_record(m, L, F::kd));
    } else if (sb || lb) {
      float afree[6];
      Ldlt6 FIA;
      ldlt_factor6(IA, FIA);
      ldlt_substitute6(FIA, pA, afree);
      const float nshare = (float)(__builtin_popcount(sb) + __builtin_popcount(lb));
      float rsum[3] = {0.0f, 0.0f, 0.0f};
      for (unsigned bb = sb; bb; bb &= bb - 1u) {
        const float* o = L.pt + sphere_slot(Q, __builtin_ctz(bb), F::kd) * PT_STRIDE;
#pragma unroll
        for (int i = 0; i < 3; i++) rsum[i] += o[PT_R + i];
      }
      for (unsigned bb = lb; bb; bb &= bb - 1u) {
        const float* o = L.pt + (link_slot0 + __builtin_ctz(bb)) * PT_STRIDE;
#pragma unroll
        for (int i = 0; i < 3; i++) rsum[i] += o[PT_R + i];
      }
      while (sb) {
        const int si = __builtin_ctz(sb);
        sb &= sb - 1u;
        pair_law(FIA, afree, L.pt + sphere_slot(Q, si, F::kd) * PT_STRIDE, C.sp.dt,
                 L.pt + pair_slot(Q, si, F::kd) * PT_STRIDE, rsum, nshare);
      }
      while (lb) {                                // the link contacts on the free box, slot order
        const int k = __builtin_ctz(lb);
        lb &= lb - 1u;
        pair_law(FIA, afree, L.pt + (link_slot0 + k) * PT_STRIDE, C.sp.dt, L.pt + (link_slot0 + SHF_MAX_LINK_CONTACTS + k) * PT_STRIDE, rsum, nshare);
      }
    }
  }
}
// after the hand-off the articulation's lanes fold their pair records, shapes ascending
template <int G, class SC>
DEV void fixed_arm_fold(const StepCtx& C, const EnvLds& L, int l, BodyRegs& B, const BoxLane& BL, unsigned spheres,
                        int nlink = 0, unsigned lb = 0u, int link_slot0 = 0) {
  typedef FixedSceneConsts<G, SC> F;
  const ShfModel* m = C.m;
  const SlotLay Q = slot_lay<SC>(m, C.scene);
  if (l < m->nb && m->dyn[l] == l) {
    unsigned bits = spheres & BL.sph_dyn;
    const int jp = bits ? box_joint_pair(m, L, Q, F::kd, spheres, __builtin_popcount(lb)) : -1;
    while (bits) {
      const int si = __builtin_ctz(bits);
      bits &= bits - 1u;
      const float* o = L.pt + sphere_slot(Q, si, F::kd) * PT_STRIDE;
      const float r[3] = {o[PT_R], o[PT_R + 1], o[PT_R + 2]};
      if (jp >= 0) {
        if (si == jp) {                           // both ends at once; the second end's bit is skipped
          const float* o2 = L.pt + sphere_slot(Q, jp + 1, F::kd) * PT_STRIDE;
          const float r2[3] = {o2[PT_R], o2[PT_R + 1], o2[PT_R + 2]};
          pair_accumulate_joint(B.IA, B.pA, r, r2, joint_record(m, L, F::kd), C.sp.dt);
        }
        continue;
      }
      float Fp[3], K[9];
      pair_unpack(L.pt + pair_slot(Q, si, F::kd) * PT_STRIDE, Fp, K);
      pair_accumulate(B.IA, B.pA, r, Fp, K, C.sp.dt);
    }
    // ... then its link contacts, slot order: the pair law against the free box, the plain contact law against a fixed one
    for (int k = 0; k < nlink; k++) {
      const float* o = L.pt + (link_slot0 + k) * PT_STRIDE;
      if (m->dyn[link_code_body(o[PT_ON])] != l) continue;
      if ((lb >> k) & 1u) {
        const float r[3] = {o[PT_R], o[PT_R + 1], o[PT_R + 2]};
        float Fp[3], K[9];
        pair_unpack(L.pt + (link_slot0 + SHF_MAX_LINK_CONTACTS + k) * PT_STRIDE, Fp, K);
        pair_accumulate(B.IA, B.pA, r, Fp, K, C.sp.dt);
      } else {
        slot_accumulate(B.IA, B.pA, o, 1.0f, C.sp.dt, 1.0f);
      }
    }
  }
}
// bit k: link slot k sits on box kd
DEV unsigned link_box_bits(const EnvLds& L, int link_slot0, int nlink, int kd) {
  unsigned lb = 0u;
  for (int k = 0; k < nlink; k++)
    if (link_code_box(L.pt[(link_slot0 + k) * PT_STRIDE + PT_ON]) == kd) lb |= 1u << k;
  return lb;
}
template <int G, class SC, bool LINK = false>
DEV void boxes_contacts_fixed(const StepCtx& C, const EnvLds& L, int l, BodyRegs& B, float mu_shape, const float* g_art,
                              const BoxLane& BL, BoxMasks& BM, int link_slot0 = 0) {
  PHASE_BEGIN();
  fixed_corner_slots<G, SC>(C, L, l, BM);
  PHASE_MARK(17);
  fixed_sphere_slots<G, SC>(C, L, l, mu_shape, g_art, BM);
  PHASE_MARK(18);
  if constexpr (LINK) BM.nlink = link_contacts<G>(C, L, l, link_slot0, mu_shape, g_art);
  GROUP_SYNC();
  PHASE_MARK(19);
  unsigned lb = 0u;
  if constexpr (LINK) lb = link_box_bits(L, link_slot0, BM.nlink, SC::DYN);
  fixed_box_fold<G, SC>(C, L, l, B, BM);
  PHASE_MARK(20);
  fixed_pair_laws<G, SC>(C, L, l == C.m->nb + SC::DYN, B.IA, B.pA, BM.spheres, lb, link_slot0);
  PHASE_MARK(21);
  GROUP_SYNC();
  PHASE_MARK(22);
  fixed_arm_fold<G, SC>(C, L, l, B, BL, BM.spheres, BM.nlink, lb, link_slot0);
  PHASE_MARK(23);
}

// Evaluate every box contact slot (one lane each), then fold them into the owning bodies.
// HARD (ShfSimParams.solver == SHF_SOLVER_PGS): the slots only record candidate constraints (slot_eval), a corner's gap to a
// fixed box is its signed distance (a constraint needs the gap of a corner still outside), and nothing is folded.
template <int G, class SC, bool LINK = false, bool HARD = false, bool EXT = false>
DEV void boxes_contacts(const StepCtx& C, const EnvLds& L, int l, BodyRegs& B, float mu_shape, const float* g_art,
                        const BoxLane& BL, BoxMasks& BM, int link_slot0 = 0) {
  static_assert(!EXT || SC::NBX == 0, "the convex narrow phase is compiled into the run-time-shaped scene path only");
  if constexpr (SC::NBX > 0) { boxes_contacts_fixed<G, SC, LINK>(C, L, l, B, mu_shape, g_art, BL, BM, link_slot0); return; }
  const ShfModel* m = C.m;
  const SceneDev* S = C.scene;
  const SlotLay Q = slot_lay<SC>(m, S);
  const int nb = m->nb, nbx = S->nboxes, T = 1 + nbx;
  const float dt = C.sp.dt, kc = HARD ? -1.0f : C.sp.contact_k, veps = C.sp.friction_vel, vdep = C.sp.max_depen_vel;
  const float offset = HARD ? C.sp.contact_offset + C.sp.rest_offset : C.sp.contact_offset;
  const float beta = HARD ? C.sp.rest_offset : fmaf(kc, dt, C.sp.contact_d);
  const float gb[3] = {C.sp.gravity[0], C.sp.gravity[1], C.sp.gravity[2]};
  PHASE_BEGIN();
  // corner slots
  for (int idx = l; idx < nbx * 8 * T; idx += G) {
    const int kd = idx / (8 * T), c = (idx / T) % 8, tg = idx % T;
    const ShfBoxDesc& bd = S->box[kd];
    if (!box_is_dynamic(bd)) continue;           // a fixed box owns no slots (SlotLay)
    float* o = L.pt + corner_slot(Q, kd, c, tg) * PT_STRIDE;
    o[PT_ON] = 0.0f;
    const float* pk = L.pose + (nb + kd) * POSE_STRIDE;
    float Rk[9], lc[3] = {((c & 4) ? 0.5f : -0.5f) * bd.dim[0], ((c & 2) ? 0.5f : -0.5f) * bd.dim[1],
                          ((c & 1) ? 0.5f : -0.5f) * bd.dim[2]};
#pragma unroll
    for (int i = 0; i < 9; i++) Rk[i] = pk[i];
    float r[3], t[3], vs[3], vp[3], n[3], h, phi, vb[3] = {pk[12], pk[13], pk[14]};
    mv3(Rk, lc, r);
#pragma unroll
    for (int i = 0; i < 3; i++) r[i] += pk[9 + i];
    cross3(vb, r, t);
#pragma unroll
    for (int i = 0; i < 3; i++) { vs[i] = pk[15 + i] + t[i]; vp[i] = fmaf(dt, gb[i], vs[i]); }
    if (tg == 0) {
      terrain_query(C.terr, L.root[0] + r[0], L.root[1] + r[1], &h, n);
      phi = (L.root[2] + r[2] - h) * n[2];
      slot_eval(o, phi, n, r, vs, vp, 0.5f * (bd.friction + C.terr.t.friction), kc, beta, veps, vdep, dt, offset);
    } else if (tg - 1 == kd) {
      continue;       // the box against itself: this slot carries an edge-edge contact, evaluated below
    } else {
      const int ks = tg - 1;
      const ShfBoxDesc& bs = S->box[ks];
      if (box_is_dynamic(bs)) continue;
      const float* ps = L.pose + (nb + ks) * POSE_STRIDE;
      float Rs[9], hh[3] = {0.5f * bs.dim[0], 0.5f * bs.dim[1], 0.5f * bs.dim[2]}, bpos[3] = {ps[9], ps[10], ps[11]};
#pragma unroll
      for (int i = 0; i < 9; i++) Rs[i] = ps[i];
      if constexpr (HARD) {
        float rc[3];
        sphere_vs_box(Rs, bpos, hh, r, 0.0f, &phi, n, rc);
      } else {
        if (!point_in_box(Rs, bpos, hh, r, &phi, n)) continue;
      }
      slot_eval(o, phi, n, r, vs, vp, 0.5f * (bd.friction + bs.friction), kc, beta, veps, vdep, dt, offset);
    }
  }
  // edge-edge contacts of the free boxes with the fixed ones (oracle boxes_pre; box_box_edge): one per (free kd, fixed ks),
  // kept in the slot corner ks of kd would have against its own box -- folded, reported and sized like any corner slot --
  // and only when no corner of kd is in contact with ks (then it is a vertex-face situation)
  GROUP_SYNC();
  for (int idx = l; idx < nbx * nbx; idx += G) {
    const int kd = idx / nbx, ks = idx % nbx;
    const ShfBoxDesc& bd = S->box[kd];
    const ShfBoxDesc& bs = S->box[ks];
    if (ks == kd || ks >= 8 || !box_is_dynamic(bd) || box_is_dynamic(bs)) continue;
    bool vertex = false;
#pragma unroll
    for (int c = 0; c < 8; c++) vertex = vertex || L.pt[corner_slot(Q, kd, c, 1 + ks) * PT_STRIDE + PT_ON] != 0.0f;
    if (vertex) continue;
    float* o = L.pt + corner_slot(Q, kd, ks, 1 + kd) * PT_STRIDE;
    const float* pk = L.pose + (nb + kd) * POSE_STRIDE;
    const float* ps = L.pose + (nb + ks) * POSE_STRIDE;
    float Rk[9], Rs[9];
#pragma unroll
    for (int i = 0; i < 9; i++) { Rk[i] = pk[i]; Rs[i] = ps[i]; }
    const float hd[3] = {0.5f * bd.dim[0], 0.5f * bd.dim[1], 0.5f * bd.dim[2]}, hs[3] = {0.5f * bs.dim[0], 0.5f * bs.dim[1], 0.5f * bs.dim[2]};
    const float cd[3] = {pk[9], pk[10], pk[11]}, cs[3] = {ps[9], ps[10], ps[11]}, vb[3] = {pk[12], pk[13], pk[14]};
    float phi, n[3], re[3], t[3], vs[3], vp[3];
    if (!box_box_edge(Rk, cd, hd, Rs, cs, hs, offset, &phi, n, re)) continue;
    cross3(vb, re, t);
#pragma unroll
    for (int i = 0; i < 3; i++) { vs[i] = pk[15 + i] + t[i]; vp[i] = fmaf(dt, gb[i], vs[i]); }
    slot_eval(o, phi, n, re, vs, vp, 0.5f * (bd.friction + bs.friction), kc, beta, veps, vdep, dt, offset);
  }
  // ShfScene.flags & SHF_SCENE_FACE_MANIFOLD (oracle boxes_pre): a fixed box that a free box touches with neither a corner nor an
  // edge crossing gets the clipped face manifold (csrc/shf_hull.h), its <= 4 points in the slots corners 0..3 would have against it
  if constexpr (EXT) if (S->flags & SHF_SCENE_FACE_MANIFOLD) {
    GROUP_SYNC();
#pragma unroll 1
    for (int kd = 0; kd < nbx; kd++) {
#pragma unroll 1
      for (int ks = 0; ks < nbx && ks < 8; ks++) {
        const ShfBoxDesc& bd = S->box[kd];
        const ShfBoxDesc& bs = S->box[ks];
        if (ks == kd || !box_is_dynamic(bd) || box_is_dynamic(bs)) continue;
        bool any = L.pt[corner_slot(Q, kd, ks, 1 + kd) * PT_STRIDE + PT_ON] != 0.0f;
#pragma unroll
        for (int c = 0; c < 8; c++) any = any || L.pt[corner_slot(Q, kd, c, 1 + ks) * PT_STRIDE + PT_ON] != 0.0f;
        if (__ballot(!any) == 0ull) continue;
        const float* pk = L.pose + (nb + kd) * POSE_STRIDE;
        const float* ps = L.pose + (nb + ks) * POSE_STRIDE;
        PolyDev PA, PB;
        PA.h = nullptr; PB.h = nullptr;
#pragma unroll
        for (int i = 0; i < 9; i++) { PA.R[i] = pk[i]; PB.R[i] = ps[i]; }
#pragma unroll
        for (int i = 0; i < 3; i++) { PA.p[i] = pk[9 + i]; PB.p[i] = ps[9 + i]; PA.hx[i] = 0.5f * bd.dim[i]; PB.hx[i] = 0.5f * bs.dim[i]; }
        float mr[4][3], mn[3] = {0.0f, 0.0f, 0.0f}, mphi[4];
        int nc = convex_manifold_dev<G>(PA, PB, offset, l, mr, mn, mphi);
        if (any) nc = 0;
        if (l < nc) {
          float rq[3] = {mr[0][0], mr[0][1], mr[0][2]}, ph = mphi[0];
#pragma unroll
          for (int q = 1; q < 4; q++)
            if (l == q) { rq[0] = mr[q][0]; rq[1] = mr[q][1]; rq[2] = mr[q][2]; ph = mphi[q]; }
          const float vb[3] = {pk[12], pk[13], pk[14]};
          float t[3], vs[3], vp[3];
          cross3(vb, rq, t);
#pragma unroll
          for (int i = 0; i < 3; i++) { vs[i] = pk[15 + i] + t[i]; vp[i] = fmaf(dt, gb[i], vs[i]); }
          slot_eval(L.pt + corner_slot(Q, kd, l, 1 + ks) * PT_STRIDE, ph, mn, rq, vs, vp, 0.5f * (bd.friction + bs.friction), kc, beta, veps, vdep, dt, offset);
        }
        GROUP_SYNC();
      }
    }
  }
  PHASE_MARK(17);
  // sphere slots
  for (int idx = l; idx < m->nsph * nbx; idx += G) {
    const int si = idx / nbx, kd = idx % nbx;
    const ShfBoxDesc& bd = S->box[kd];
    if (!box_is_dynamic(bd)) continue;
    float* o = L.pt + sphere_slot(Q, si, kd) * PT_STRIDE;
    o[PT_ON] = 0.0f;
    const int b = m->sph_body[si];
    const float* pb = L.pose + b * POSE_STRIDE;
    const float* pk = L.pose + (nb + kd) * POSE_STRIDE;
    float Rb[9], Rk[9], c[3];
#pragma unroll
    for (int i = 0; i < 9; i++) { Rb[i] = pb[i]; Rk[i] = pk[i]; }
    const float hh[3] = {0.5f * bd.dim[0], 0.5f * bd.dim[1], 0.5f * bd.dim[2]}, bpos[3] = {pk[9], pk[10], pk[11]};
    if (!rounded_centre(m, si, Rb, pb + 9, Rk, bpos, hh, offset, c)) continue;
    float phi, n[3], rc[3], ta[3], tb[3], vrel[3], vrs[3];
    sphere_vs_box(Rk, bpos, hh, c, m->sph_radius[si], &phi, n, rc);
    const float va[3] = {pb[12], pb[13], pb[14]}, vbx[3] = {pk[12], pk[13], pk[14]};
    cross3(va, rc, ta);
    cross3(vbx, rc, tb);
#pragma unroll
    for (int i = 0; i < 3; i++) {
      const float pa = pb[15 + i] + ta[i], pq = pk[15 + i] + tb[i];
      vrs[i] = pa - pq;
      vrel[i] = fmaf(dt, g_art[i], pa) - fmaf(dt, gb[i], pq);
    }
    slot_eval(o, phi, n, rc, vrs, vrel, 0.5f * (mu_shape + bd.friction), kc, beta, veps, vdep, dt, offset);
  }
  int nlink = 0;
  PHASE_MARK(18);
  if constexpr (LINK) nlink = link_contacts<G, EXT>(C, L, l, link_slot0, mu_shape, g_art);
  BM.nlink = nlink;
  GROUP_SYNC();
  PHASE_MARK(20);
  if constexpr (HARD) return;       // substep_hard_finish gathers the slots as constraints
  // fold (as the fixed-scene path: box lanes first, with the pair laws; then the articulation's lanes)
  const int kd = l - nb;
  if (kd >= 0 && kd < nbx && box_is_dynamic(S->box[kd])) {
    unsigned long long cb = corner_flags(m, L, Q, kd);
    unsigned sb = box_sphere_flags(m, L, Q, kd);
    while (cb) {
      const int j = __builtin_ctzll(cb);
      cb &= cb - 1ull;
      slot_accumulate(B.IA, B.pA, L.pt + corner_slot(Q, kd, j / BOX_T, j % BOX_T) * PT_STRIDE, 1.0f, dt, 1.0f);
    }
    unsigned lb = 0u;     // this box's link contacts (bit k = slot k)
    for (int k = 0; k < nlink; k++)
      if (link_code_box(L.pt[(link_slot0 + k) * PT_STRIDE + PT_ON]) == kd) lb |= 1u << k;
    const int jp = box_joint_pair(m, L, Q, kd, sb, __builtin_popcount(lb));
    if (jp >= 0) {                                // the two ends of one capsule: eliminated together
      float afree[6];
      Ldlt6 FIA;
      ldlt_factor6(B.IA, FIA);
      ldlt_substitute6(FIA, B.pA, afree);
      pair_law_joint(FIA, afree, L.pt + sphere_slot(Q, jp, kd) * PT_STRIDE, L.pt + sphere_slot(Q, jp + 1, kd) * PT_STRIDE, dt,
                     joint_record(m, L, kd));
    } else if (sb || lb) {
      float afree[6];
      Ldlt6 FIA;
      ldlt_factor6(B.IA, FIA);
      ldlt_substitute6(FIA, B.pA, afree);
      const float nshare = (float)(__builtin_popcount(sb) + __builtin_popcount(lb));
      float rsum[3] = {0.0f, 0.0f, 0.0f};
      for (unsigned bb = sb; bb; bb &= bb - 1u) {
        const float* o = L.pt + sphere_slot(Q, __builtin_ctz(bb), kd) * PT_STRIDE;
#pragma unroll
        for (int i = 0; i < 3; i++) rsum[i] += o[PT_R + i];
      }
      for (unsigned bb = lb; bb; bb &= bb - 1u) {
        const float* o = L.pt + (link_slot0 + __builtin_ctz(bb)) * PT_STRIDE;
#pragma unroll
        for (int i = 0; i < 3; i++) rsum[i] += o[PT_R + i];
      }
      while (sb) {
        const int si = __builtin_ctz(sb);
        sb &= sb - 1u;
        pair_law(FIA, afree, L.pt + sphere_slot(Q, si, kd) * PT_STRIDE, dt, L.pt + pair_slot(Q, si, kd) * PT_STRIDE, rsum, nshare);
      }
      while (lb) {
        const int k = __builtin_ctz(lb);
        lb &= lb - 1u;
        pair_law(FIA, afree, L.pt + (link_slot0 + k) * PT_STRIDE, dt, L.pt + (link_slot0 + SHF_MAX_LINK_CONTACTS + k) * PT_STRIDE, rsum, nshare);
      }
    }
  }
  GROUP_SYNC();
  if (l < nb && m->dyn[l] == l) {
    unsigned bits = body_sphere_flags(m, L, Q, l, true);
    while (bits) {
      const int j = __builtin_ctz(bits), si = j / SHF_MAX_BOXES, k2 = j % SHF_MAX_BOXES;
      bits &= bits - 1u;
      const float* o = L.pt + sphere_slot(Q, si, k2) * PT_STRIDE;
      const float r[3] = {o[PT_R], o[PT_R + 1], o[PT_R + 2]};
      const int jp = box_joint_pair(m, L, Q, k2, box_sphere_flags(m, L, Q, k2), link_slots_on_box(L, link_slot0, nlink, k2));
      if (jp >= 0) {
        if (si == jp) {
          const float* o2 = L.pt + sphere_slot(Q, jp + 1, k2) * PT_STRIDE;
          const float r2[3] = {o2[PT_R], o2[PT_R + 1], o2[PT_R + 2]};
          pair_accumulate_joint(B.IA, B.pA, r, r2, joint_record(m, L, k2), dt);
        }
        continue;
      }
      float F[3], K[9];
      pair_unpack(L.pt + pair_slot(Q, si, k2) * PT_STRIDE, F, K);
      pair_accumulate(B.IA, B.pA, r, F, K, dt);
    }
    // ... then its link contacts, slot order: pair law against a free box, the plain contact law against a fixed one
    for (int k = 0; k < nlink; k++) {
      const float* o = L.pt + (link_slot0 + k) * PT_STRIDE;
      if (m->dyn[link_code_body(o[PT_ON])] != l) continue;
      if (box_is_dynamic(S->box[link_code_box(o[PT_ON])])) {
        const float r[3] = {o[PT_R], o[PT_R + 1], o[PT_R + 2]};
        float F[3], K[9];
        pair_unpack(L.pt + (link_slot0 + SHF_MAX_LINK_CONTACTS + k) * PT_STRIDE, F, K);
        pair_accumulate(B.IA, B.pA, r, F, K, dt);
      } else {
        slot_accumulate(B.IA, B.pA, o, 1.0f, dt, 1.0f);
      }
    }
  }
  PHASE_MARK(19);
}

// Solve the free boxes, report contact forces (boxes and the articulation's sphere contacts),
// integrate the boxes.  contact_out has nb + nboxes rows.
// force of link slot k on the articulation once its acceleration is known (oracle boxes_post: flink)
DEV void link_force(const StepCtx& C, const EnvLds& L, int link_slot0, int k, float* f) {
  const ShfModel* m = C.m;
  const float* o = L.pt + (link_slot0 + k) * PT_STRIDE;
  const float* ab = L.acc + m->dyn[link_code_body(o[PT_ON])] * 6;
  const float abr[6] = {ab[0], ab[1], ab[2], ab[3], ab[4], ab[5]};
  if (box_is_dynamic(C.scene->box[link_code_box(o[PT_ON])])) {
    const float r[3] = {o[PT_R], o[PT_R + 1], o[PT_R + 2]};
    pair_force(L.pt + (link_slot0 + SHF_MAX_LINK_CONTACTS + k) * PT_STRIDE, abr, r, C.sp.dt, f);
  } else {
    f[0] = f[1] = f[2] = 0.0f;
    slot_force(o, abr, 1.0f, C.sp.dt, 1.0f, f);
  }
}

// PART: 1 = the boxes' lanes (pair forces, solve, their contact rows, integration), 2 = the articulation's contact rows,
// 3 = both (one wave does everything).
template <int G, class SC, int PART = 3>
DEV void boxes_finish(const StepCtx& C, const EnvLds& L, int l, BodyRegs& B, float* contact_out, const BoxLane& BL,
                      const BoxMasks& BM, int link_slot0 = 0) {
  const ShfModel* m = C.m;
  const SceneDev* S = C.scene;
  const SlotLay Q = slot_lay<SC>(m, S);
  const int nb = m->nb, nbx = S->nboxes, T = 1 + nbx;
  const float dt = C.sp.dt;
  const int kd = l - nb;
  const bool isbox = kd >= 0 && kd < nbx;
  const bool dynbox = isbox && box_is_dynamic(S->box[kd]);
  float a[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
  if ((PART & 1) && dynbox) {
    // the articulation's solved acceleration fixes every pair force; the box receives exactly its opposite, then solves
    unsigned sb;
    if constexpr (SC::NBX > 0) sb = BM.spheres; else sb = box_sphere_flags(m, L, Q, kd);
    const int jp = box_joint_pair(m, L, Q, kd, sb, link_slots_on_box(L, link_slot0, BM.nlink, kd));
    while (sb) {
      const int si = __builtin_ctz(sb);
      sb &= sb - 1u;
      const float* o = L.pt + sphere_slot(Q, si, kd) * PT_STRIDE;
      const float r[3] = {o[PT_R], o[PT_R + 1], o[PT_R + 2]};
      float f[3], t[3];
      sphere_pair_force(m, L, Q, si, kd, jp, dt, f);
      cross3(r, f, t);
#pragma unroll
      for (int k = 0; k < 3; k++) { B.pA[k] += t[k]; B.pA[3 + k] += f[k]; }
    }
    for (int k = 0; k < BM.nlink; k++) {
      const float* o = L.pt + (link_slot0 + k) * PT_STRIDE;
      if (link_code_box(o[PT_ON]) != kd) continue;
      const float r[3] = {o[PT_R], o[PT_R + 1], o[PT_R + 2]};
      float f[3], t[3];
      link_force(C, L, link_slot0, k, f);
      cross3(r, f, t);
#pragma unroll
      for (int i = 0; i < 3; i++) { B.pA[i] += t[i]; B.pA[3 + i] += f[i]; }
    }
    ldlt_solve6(B.IA, B.pA, a);
  }
  if (contact_out) {
    if ((PART & 2) && l < nb) {
      float f[3] = {contact_out[3 * l], contact_out[3 * l + 1], contact_out[3 * l + 2]};
      const float* ab = L.acc + m->dyn[l] * 6;
      const float abr[6] = {ab[0], ab[1], ab[2], ab[3], ab[4], ab[5]};
      unsigned bits;
      if constexpr (SC::NBX > 0) {       // ballots of this sub-step: bit si -> the run-time path's bit si * MAX + DYN
        bits = 0u;
        unsigned sb = BM.spheres & BL.sph_body;
        while (sb) { const int si = __builtin_ctz(sb); sb &= sb - 1u; bits |= 1u << (si * SHF_MAX_BOXES + SC::DYN); }
      } else {
        bits = body_sphere_flags(m, L, Q, l, false);
      }
      while (bits) {
        const int j = __builtin_ctz(bits), si = j / SHF_MAX_BOXES, k2 = j % SHF_MAX_BOXES;
        bits &= bits - 1u;
        unsigned sall;
        if constexpr (SC::NBX > 0) sall = BM.spheres; else sall = box_sphere_flags(m, L, Q, k2);
        const int jp = box_joint_pair(m, L, Q, k2, sall, link_slots_on_box(L, link_slot0, BM.nlink, k2));
        float fp[3];
        sphere_pair_force(m, L, Q, si, k2, jp, dt, fp);
#pragma unroll
        for (int k = 0; k < 3; k++) f[k] += fp[k];
      }
      for (int k = 0; k < BM.nlink; k++) {
        if (link_code_body(L.pt[(link_slot0 + k) * PT_STRIDE + PT_ON]) != l) continue;
        float fp[3];
        link_force(C, L, link_slot0, k, fp);
#pragma unroll
        for (int i = 0; i < 3; i++) f[i] += fp[i];
      }
      contact_out[3 * l] = f[0]; contact_out[3 * l + 1] = f[1]; contact_out[3 * l + 2] = f[2];
    }
    if ((PART & 1) && isbox) {
      float f[3] = {0.0f, 0.0f, 0.0f};
      if (dynbox) {
        unsigned sb;
        if constexpr (SC::NBX > 0) {
          unsigned pl = BM.cplane, ed = BM.cedge;
          unsigned long long bx = BM.cbox;
          int c, tg;
          sb = BM.spheres;
          while (corner_next<SC::NBX - 1, SC::DYN>(pl, bx, ed, &c, &tg))
            slot_force(L.pt + corner_slot(Q, kd, c, tg) * PT_STRIDE, a, 1.0f, dt, 1.0f, f);
        } else {
          unsigned long long cb = corner_flags(m, L, Q, kd);
          sb = box_sphere_flags(m, L, Q, kd);
          while (cb) {
            const int j = __builtin_ctzll(cb);
            cb &= cb - 1ull;
            slot_force(L.pt + corner_slot(Q, kd, j / BOX_T, j % BOX_T) * PT_STRIDE, a, 1.0f, dt, 1.0f, f);
          }
        }
        const int jp_rows = box_joint_pair(m, L, Q, kd, sb, link_slots_on_box(L, link_slot0, BM.nlink, kd));
        while (sb) {
          const int si = __builtin_ctz(sb);
          sb &= sb - 1u;
          float fp[3];
          sphere_pair_force(m, L, Q, si, kd, jp_rows, dt, fp);
#pragma unroll
          for (int k = 0; k < 3; k++) f[k] -= fp[k];
        }
        for (int k = 0; k < BM.nlink; k++) {
          if (link_code_box(L.pt[(link_slot0 + k) * PT_STRIDE + PT_ON]) != kd) continue;
          float fp[3];
          link_force(C, L, link_slot0, k, fp);
#pragma unroll
          for (int i = 0; i < 3; i++) f[i] -= fp[i];
        }
      }
      contact_out[3 * l] = f[0]; contact_out[3 * l + 1] = f[1]; contact_out[3 * l + 2] = f[2];
    }
  }
  if ((PART & 1) && dynbox) {
    float* row = L.root + 13 * (1 + kd);
    const float gb[3] = {C.sp.gravity[0], C.sp.gravity[1], C.sp.gravity[2]};
    const float ang[3] = {row[10], row[11], row[12]}, lin[3] = {row[7], row[8], row[9]}, al[3] = {a[0], a[1], a[2]};
    float wxv[3], axp[3];
    cross3(ang, lin, wxv);
    cross3(al, B.p, axp);
    const float damp = 1.0f / fmaf(dt, C.sp.angular_damping, 1.0f);
    float wn[3], vn[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
      wn[k] = fmaf(dt, a[k], ang[k]) * damp;
      vn[k] = fmaf(dt, a[3 + k] + gb[k] + axp[k] + wxv[k], lin[k]);
    }
    const float w2 = dot3(wn, wn), wmax = C.sp.max_ang_vel;
    if (w2 > wmax * wmax) {
      const float sc2 = wmax * rsqrt_spec(w2);
#pragma unroll
      for (int k = 0; k < 3; k++) wn[k] *= sc2;
    }
#pragma unroll
    for (int k = 0; k < 3; k++) { row[10 + k] = wn[k]; row[7 + k] = vn[k]; row[k] = fmaf(dt, vn[k], row[k]); }
    const float hx = 0.5f * dt * wn[0], hy = 0.5f * dt * wn[1], hz = 0.5f * dt * wn[2];
    const float x = row[3], y = row[4], z = row[5], ww = row[6];
    const float nx = x + fmaf(hx, ww, fmaf(hy, z, -(hz * y)));
    const float ny = y + fmaf(hy, ww, fmaf(hz, x, -(hx * z)));
    const float nz = z + fmaf(hz, ww, fmaf(hx, y, -(hy * x)));
    const float nw = ww - fmaf(hx, x, fmaf(hy, y, hz * z));
    const float inv = rsqrt_spec(fmaf(nw, nw, fmaf(nz, nz, fmaf(ny, ny, nx * nx))));
    row[3] = nx * inv; row[4] = ny * inv; row[5] = nz * inv; row[6] = nw * inv;
  }
}
