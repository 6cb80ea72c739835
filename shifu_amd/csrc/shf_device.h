// shf_device.h -- gfx950 device code: one lane group (G lanes, G = 64 by default:
// one wavefront) simulates one environment.
//
// Lane roles inside a group:   lane b < nb        <-> reported rigid body b
//                              lane d < nd        <-> degree of freedom d
//                              lane i (+k*G) < np <-> contact sample point i
// The flattened articulation (ShfModel) and the per-env working set live in LDS;
// parent->child (kinematics, accelerations) and child->parent (articulated
// inertia) hand-offs go through LDS slots owned by the producing lane, cross-lane
// scalars through wave shuffles.  HBM is touched only at the start and the end of
// an env step, in the Isaac-Gym tensor layouts, which are contiguous per env and
// therefore coalesced across the lanes of a group.
//
// ARITHMETIC CONTRACT: every floating-point operation below is written out
// (explicit fmaf, no contraction: the file is compiled with -ffp-contract=off
// and without fast-math) in the same order as the float build of the CPU oracle
// (oracle/shf_oracle.c, test-only), so results are reproducible bit for bit.
// This file does not include, link or call the oracle.
//
// Reference call sites replaced: gym.simulate (shifu/units/robot.py:69,
// examples/a1_conditional/a1_conditional.py:69, shifu/gym/isaac_gym.py:140).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/shifu_amd.h"

#define DEV __device__ __forceinline__

// LDS hand-offs stay inside one wavefront: LDS requests of a wave are served in
// order, so only the compiler has to be kept from reordering.
#define GROUP_SYNC()                                         \
  do {                                                       \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   \
    __builtin_amdgcn_wave_barrier();                         \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");   \
  } while (0)

// Per-phase cycle accounting for tools/phase_clock.py (debug builds with -DSHF_PHASE_CLOCK only):
// thread 0 of block 0 adds the s_memtime delta since the previous mark to g_phase_cycles[k].
#ifdef SHF_PHASE_CLOCK
__device__ unsigned long long g_phase_cycles[48];   // 0-23 sub-step phases, 24-31 link stages / counters, 32-37 the generic solve, 38-46 constraint-count histogram
#define PHASE_BEGIN() unsigned long long _pc = clock64()
#define PHASE_MARK(k)                                                                    \
  do {                                                                                   \
    const unsigned long long _now = clock64();                                           \
    if (blockIdx.x == 0 && threadIdx.x == 0) g_phase_cycles[k] += _now - _pc;            \
    _pc = clock64();                                                                     \
  } while (0)
#define PHASE_RESET() _pc = clock64()
// the same for one other thread of block 0 (the first lane of the box wave in k_abb_step_ws): its own running clock
#define PHASE_BEGIN_T() unsigned long long _pct = clock64()
#define PHASE_MARK_T(k, tid)                                                             \
  do {                                                                                   \
    const unsigned long long _now = clock64();                                           \
    if (blockIdx.x == 0 && (int)threadIdx.x == (tid)) g_phase_cycles[k] += _now - _pct;  \
    _pct = clock64();                                                                    \
  } while (0)
#else
#define PHASE_BEGIN_T() do {} while (0)
#define PHASE_MARK_T(k, tid) do {} while (0)
#define PHASE_BEGIN() do {} while (0)
#define PHASE_MARK(k) do {} while (0)
#define PHASE_RESET() do {} while (0)
#endif

// 1/x (x > 0) and 1/sqrt(x) (x > 0) by Newton's iteration from an integer seed: the fixed operation sequences of the
// oracle's rcp_spec / rsqrt_spec (7 and 12 dependent operations; IEEE `1.0f / x` and `sqrtf` are 11 and ~20 on gfx950).
// DOMAIN: normal positive floats, 1e-30 <= x <= 1e30 (relative error <= 1.0 / 2.4 units of 2^-24 there,
// tests/test_oracle_physics.py).  Outside it: zero and denormals give nan / inf (they propagate, as IEEE's inf would), a
// negative argument of rcp_spec its negative reciprocal, rsqrt_spec of a negative number -inf (pinned by the same test).
// Every call site feeds a quantity that is positive
// by construction: the ABA's D = S^T IA S + armature + dt (damping + ...) and the LDL^T pivots of an articulated inertia
// (positive definite: rigid inertias with mass > 0, model.py refuses massless moving bodies; contact terms only add
// positive semi-definite parts), 1 + |grad h|^2, |q|^2 of a unit-ish quaternion, max(|v_t|^2, v_eps^2), the regularised
// diagonal blocks of the contact-space response (SPD + 1e-6 trace), squared lengths guarded by explicit thresholds
// (box_box_edge: l2 > 1e-4).  A non-positive pivot would mean a broken model, not a reachable state.
DEV float rcp_spec(float x) {
  float y = __uint_as_float(0x7EF311C7u - __float_as_uint(x));
#pragma unroll
  for (int k = 0; k < 3; k++) { const float e = fmaf(-x, y, 1.0f); y = fmaf(y, e, y); }
  return y;
}
DEV float rsqrt_spec(float x) {
  float y = __uint_as_float(0x5F375A86u - (__float_as_uint(x) >> 1));
  const float hx = 0.5f * x;
#pragma unroll
  for (int k = 0; k < 3; k++) { const float t = y * y; const float w = fmaf(-hx, t, 1.5f); y = y * w; }
  return y;
}

// ------------------------------------------------------------------ math --
DEV float rminf(float a, float b) { return a < b ? a : b; }
DEV float rmaxf(float a, float b) { return a > b ? a : b; }
DEV float rclampf(float x, float lo, float hi) { return rminf(rmaxf(x, lo), hi); }
DEV float dot3(const float* a, const float* b) { return fmaf(a[2], b[2], fmaf(a[1], b[1], a[0] * b[0])); }
DEV void cross3(const float* a, const float* b, float* o) {
  float x = fmaf(a[1], b[2], -(a[2] * b[1]));
  float y = fmaf(a[2], b[0], -(a[0] * b[2]));
  float z = fmaf(a[0], b[1], -(a[1] * b[0]));
  o[0] = x; o[1] = y; o[2] = z;
}
DEV void mv3(const float* M, const float* v, float* o) {
  float x = dot3(M, v), y = dot3(M + 3, v), z = dot3(M + 6, v);
  o[0] = x; o[1] = y; o[2] = z;
}
DEV void mm3(const float* A, const float* B, float* o) {
  float t[9];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) t[3 * i + j] = fmaf(A[3 * i + 2], B[6 + j], fmaf(A[3 * i + 1], B[3 + j], A[3 * i] * B[j]));
#pragma unroll
  for (int k = 0; k < 9; k++) o[k] = t[k];
}
DEV void sincos_spec(float x, float* s, float* c) {
  float k = rintf(x * 0.636619772367581343f);
  float r = fmaf(k, -1.5703125f, x);
  r = fmaf(k, -4.83751296997070312e-4f, r);
  r = fmaf(k, -7.54978995489188216e-8f, r);
  float r2 = r * r;
  float ps = fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f);
  ps = fmaf(r2, ps, -1.6666654611e-1f);
  float sn = fmaf(r * r2, ps, r);
  float pc = fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f);
  pc = fmaf(r2, pc, 4.166664568298827e-2f);
  float cs = fmaf(r2 * r2, pc, fmaf(r2, -0.5f, 1.0f));
  int q = ((int)k) & 3;
  float so = (q & 1) ? cs : sn, co = (q & 1) ? sn : cs;
  if (q == 1 || q == 2) co = -co;
  if (q >= 2) so = -so;
  *s = so; *c = co;
}
DEV float exp_spec(float x) {
  if (x < -87.0f) return 0.0f;
  float k = rintf(x * 1.44269504088896341f);
  float r = fmaf(k, -0.693359375f, x);
  r = fmaf(k, 2.12194440e-4f, r);
  float p = fmaf(r, 1.9875691500e-4f, 1.3981999507e-3f);
  p = fmaf(r, p, 8.3334519073e-3f);
  p = fmaf(r, p, 4.1665795894e-2f);
  p = fmaf(r, p, 1.6666665459e-1f);
  p = fmaf(r, p, 5.0000001201e-1f);
  float e = fmaf(r * r, p, r) + 1.0f;
  const uint32_t bits = __float_as_uint(e) + ((uint32_t)(int32_t)k << 23);   // unsigned: k may be negative (same bits)
  return __uint_as_float(bits);
}
DEV void quat_to_mat(const float* q, float* M) {
  float x = q[0], y = q[1], z = q[2], w = q[3];
  float xx = x * x, yy = y * y, zz = z * z, xy = x * y, xz = x * z, yz = y * z, wx = w * x, wy = w * y, wz = w * z;
  M[0] = fmaf(-2.0f, yy + zz, 1.0f); M[1] = 2.0f * (xy - wz);          M[2] = 2.0f * (xz + wy);
  M[3] = 2.0f * (xy + wz);          M[4] = fmaf(-2.0f, xx + zz, 1.0f); M[5] = 2.0f * (yz - wx);
  M[6] = 2.0f * (xz - wy);          M[7] = 2.0f * (yz + wx);          M[8] = fmaf(-2.0f, xx + yy, 1.0f);
}
DEV void mat_to_quat(const float* M, float* q) {
  float tr = M[0] + M[4] + M[8];
  if (tr > 0.0f) {
    float s = sqrtf(tr + 1.0f) * 2.0f;
    q[3] = 0.25f * s; q[0] = (M[7] - M[5]) / s; q[1] = (M[2] - M[6]) / s; q[2] = (M[3] - M[1]) / s;
  } else if (M[0] > M[4] && M[0] > M[8]) {
    float s = sqrtf(1.0f + M[0] - M[4] - M[8]) * 2.0f;
    q[3] = (M[7] - M[5]) / s; q[0] = 0.25f * s; q[1] = (M[1] + M[3]) / s; q[2] = (M[2] + M[6]) / s;
  } else if (M[4] > M[8]) {
    float s = sqrtf(1.0f + M[4] - M[0] - M[8]) * 2.0f;
    q[3] = (M[2] - M[6]) / s; q[0] = (M[1] + M[3]) / s; q[1] = 0.25f * s; q[2] = (M[5] + M[7]) / s;
  } else {
    float s = sqrtf(1.0f + M[8] - M[0] - M[4]) * 2.0f;
    q[3] = (M[3] - M[1]) / s; q[0] = (M[2] + M[6]) / s; q[1] = (M[5] + M[7]) / s; q[2] = 0.25f * s;
  }
}
DEV void crm(const float* a, const float* m, float* o) {
  float t0[3], t1[3], t2[3];
  cross3(a, m, t0);
  cross3(a, m + 3, t1);
  cross3(a + 3, m, t2);
  o[0] = t0[0]; o[1] = t0[1]; o[2] = t0[2];
  o[3] = t1[0] + t2[0]; o[4] = t1[1] + t2[1]; o[5] = t1[2] + t2[2];
}

// packed upper triangle of a symmetric 6x6: (i,j), i<=j
#define SYM(i, j) ((i) * 6 - ((i) * ((i) - 1)) / 2 + ((j) - (i)))
#define SYMG(A, i, j) ((i) <= (j) ? (A)[SYM(i, j)] : (A)[SYM(j, i)])

// --------------------------------------------------------------- terrain --
struct TerrainDev {
  ShfTerrain t;
  const int16_t* h;
};

// One triangle of the warped mesh against the query point (cell units; Z in metres): keeps the highest surface that
// covers the point, edges inclusive.  Only coverage and height are evaluated per triangle; the winner's corners are kept
// and its normal is computed once after the search (warped_normal) -- the same operations on the same values as
// computing it for every covering triangle (what the oracle does), a third of the instructions.
DEV void warped_triangle(float px, float py, const float* P0, const float* P1, const float* P2, float* best, float* BP,
                         bool* found) {
  const float ax = P0[0] - px, ay = P0[1] - py, bx = P1[0] - px, by = P1[1] - py, cx = P2[0] - px, cy = P2[1] - py;
  float e0 = fmaf(bx, cy, -(cx * by)), e1 = fmaf(cx, ay, -(ax * cy)), e2 = fmaf(ax, by, -(bx * ay));
  float area = e0 + e1 + e2;
  if (area < 0.0f) { e0 = -e0; e1 = -e1; e2 = -e2; area = -area; }
  const float z = fmaf(e2, P2[2], fmaf(e1, P1[2], e0 * P0[2])) / area;
  // a riser (no extent in the horizontal plane), a miss, or not above the best so far: no change
  const bool take = area > 1e-6f && !(e0 < 0.0f || e1 < 0.0f || e2 < 0.0f) && !(*found && !(z > *best));
  if (take) {
    *best = z; *found = true;
#pragma unroll
    for (int k = 0; k < 3; k++) { BP[k] = P0[k]; BP[3 + k] = P1[k]; BP[6 + k] = P2[k]; }
  }
}
DEV void warped_normal(float hs, const float* BP, float* bn) {
  const float* P0 = BP; const float* P1 = BP + 3; const float* P2 = BP + 6;
  const float ux = (P1[0] - P0[0]) * hs, uy = (P1[1] - P0[1]) * hs, uz = P1[2] - P0[2];
  const float vx = (P2[0] - P0[0]) * hs, vy = (P2[1] - P0[1]) * hs, vz = P2[2] - P0[2];
  float nx = fmaf(uy, vz, -(uz * vy)), ny = fmaf(uz, vx, -(ux * vz)), nz = fmaf(ux, vy, -(uy * vx));
  if (nz < 0.0f) { nx = -nx; ny = -ny; nz = -nz; }
  const float inv = rsqrt_spec(fmaf(nz, nz, fmaf(ny, ny, nx * nx)));
  bn[0] = nx * inv; bn[1] = ny * inv; bn[2] = nz * inv;
}

DEV void terrain_query_heightfield(const TerrainDev& T, float x, float y, float* h, float* n);

// ShfTerrain.warped: the trimesh convert_heightfield_to_trimesh makes of the samples; vertex shifts in
// the bytes after the samples, with the per-cell search range (one cell where no neighbour reaches in, up to 3 x 3).
DEV void terrain_query_warped(const TerrainDev& T, float x, float y, float* h, float* n) {
  const int rows = T.t.rows, cols = T.t.cols;
  const uint8_t* W = reinterpret_cast<const uint8_t*>(T.h + (size_t)rows * cols);
  const float hs = T.t.hscale, inv = 1.0f / hs, vs = T.t.vscale;
  const float fx = (x + T.t.border) * inv, fy = (y + T.t.border) * inv;
  const int i0 = (int)rclampf(floorf(fx), 0.0f, (float)(rows - 2)), j0 = (int)rclampf(floorf(fy), 0.0f, (float)(cols - 2));
  // search range: the cell, plus the neighbouring rows / columns its hint bits ask for (include/shifu_amd.h)
  const int wc = W[(size_t)i0 * cols + j0];
  const int ilo = ((wc >> 4) & 1) && i0 > 0 ? i0 - 1 : i0, ihi = ((wc >> 5) & 1) && i0 < rows - 2 ? i0 + 1 : i0;
  const int jlo = ((wc >> 6) & 1) && j0 > 0 ? j0 - 1 : j0, jhi = ((wc >> 7) & 1) && j0 < cols - 2 ? j0 + 1 : j0;
  float best = 0.0f, BP[9];
  bool found = false;
  for (int i = ilo; i <= ihi; i++)
    for (int j = jlo; j <= jhi; j++) {
      float P[4][3];
#pragma unroll
      for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) {
          const size_t idx = (size_t)(i + a) * cols + (j + b);
          const int wv = W[idx];
          P[2 * a + b][0] = (float)(i + a + (wv & 3) - 1);
          P[2 * a + b][1] = (float)(j + b + ((wv >> 2) & 3) - 1);
          P[2 * a + b][2] = (float)T.h[idx] * vs;
        }
      warped_triangle(fx, fy, P[0], P[3], P[1], &best, BP, &found);
      warped_triangle(fx, fy, P[0], P[2], P[3], &best, BP, &found);
    }
  if (!found) { terrain_query_heightfield(T, x, y, h, n); return; }   // outside the mesh / numerical gap
  *h = best;
  warped_normal(hs, BP, n);
}

// TW ("trimesh-capable") instantiations look at ShfTerrain.warped; the others are height-field only and the host
// does not launch them on a warped terrain (the extra path costs the default A1 step 3.5 % just by being compiled in).
template <bool TW = false>
DEV void terrain_query(const TerrainDev& T, float x, float y, float* h, float* n) {
  if (T.t.rows == 0) { *h = 0.0f; n[0] = 0.0f; n[1] = 0.0f; n[2] = 1.0f; return; }
  if (TW && T.t.warped) { terrain_query_warped(T, x, y, h, n); return; }
  terrain_query_heightfield(T, x, y, h, n);
}
DEV void terrain_query_heightfield(const TerrainDev& T, float x, float y, float* h, float* n) {
  float inv = 1.0f / T.t.hscale;
  float fx = (x + T.t.border) * inv, fy = (y + T.t.border) * inv;
  float fi = rclampf(floorf(fx), 0.0f, (float)(T.t.rows - 2)), fj = rclampf(floorf(fy), 0.0f, (float)(T.t.cols - 2));
  int i = (int)fi, j = (int)fj;
  float u = rclampf(fx - fi, 0.0f, 1.0f), v = rclampf(fy - fj, 0.0f, 1.0f);
  float vs = T.t.vscale;
  const int16_t* row0 = T.h + (size_t)i * T.t.cols + j;
  const int16_t* row1 = row0 + T.t.cols;
  float h00 = (float)row0[0] * vs, h10 = (float)row1[0] * vs, h01 = (float)row0[1] * vs, h11 = (float)row1[1] * vs;
  // lower triangle (u+v <= 1): gx = h10-h00, gy = h01-h00, h = fma(v, gy, fma(u, gx, h00));
  // upper: gx = h11-h01, gy = h11-h10, h = fma(1-v, -gy, fma(1-u, -gx, h11)).  Written with selects so that
  // the four loads are issued together instead of one per branch (same operations either way).
  const bool lo = u + v <= 1.0f;
  float gx = (lo ? h10 : h11) - (lo ? h00 : h01);
  float gy = (lo ? h01 : h11) - (lo ? h00 : h10);
  const float hh = fmaf(lo ? v : 1.0f - v, lo ? gy : -gy, fmaf(lo ? u : 1.0f - u, lo ? gx : -gx, lo ? h00 : h11));
  gx *= inv; gy *= inv;
  float nz = rsqrt_spec(fmaf(gy, gy, fmaf(gx, gx, 1.0f)));
  *h = hh; n[0] = -gx * nz; n[1] = -gy * nz; n[2] = nz;
}

// ------------------------------------------------------------- LDS views --
// per-env LDS working set (float offsets); sizes follow the model actually
// loaded so that four 256-thread blocks fit one CU (160 KiB LDS).
#define POSE_STRIDE 20 /* Rw[9] p[3] v[6], padded to 16 bytes                */
#define XCH_STRIDE 28  /* Ia[21] pa[6], padded to 16 bytes                    */
#define PT_STRIDE 12   /* r[3] n[3] f[3] ct bn on: 11 contiguous floats from a 16-byte boundary, flag last */
#define PT_R 0
#define PT_N 3
#define PT_F 6
#define PT_CT 9
#define PT_BN 10
#define PT_ON 11
#define DOF_STRIDE 6   /* q qd tau0 dex qdd tau_cmd                         */

struct EnvLds {
  float* pose;
  float* acc;
  float* xch;
  float* pt;
  float* dofb;
  float* root;
};

// The contact-point region comes last: kernels that need post-physics scratch
// reuse it and may ask for `min_tail` words there.
// `nb` counts pose/acc/xch entries (reported bodies + box actors), `np` contact slots
// (articulation sample points + box corner/sphere slots), `nactors` root-state rows.
__host__ __device__ inline int root_words(int nactors) { return (13 * nactors + 3 + 3) & ~3; }
__host__ __device__ inline int env_lds_words(int nb, int nd, int np, int min_tail = 0, int nactors = 1) {
  int tail = np * PT_STRIDE;
  if (tail < min_tail) tail = min_tail;
  int w = nb * POSE_STRIDE + ((nb * 6 + 3) & ~3) + nb * XCH_STRIDE + ((nd * DOF_STRIDE + 3) & ~3) + root_words(nactors) + tail;
  return (w + 3) & ~3;
}
DEV EnvLds env_lds_carve(float* base, int nb, int nd, int np, int nactors = 1) {
  EnvLds L;
  L.pose = base;
  L.acc = L.pose + nb * POSE_STRIDE;
  L.xch = L.acc + ((nb * 6 + 3) & ~3);
  L.dofb = L.xch + nb * XCH_STRIDE;
  L.root = L.dofb + ((nd * DOF_STRIDE + 3) & ~3);
  L.pt = L.root + root_words(nactors);
  return L;
}

// Model dimensions: read from the LDS copy of the model at run time (any URDF), or fixed at
// compile time for a known robot so that loop bounds and LDS offsets fold to immediates
// (the A1 instantiation is ~6 % faster; the host picks it only when the model matches).
struct DynDims {
  DEV static int nb(const ShfModel* m) { return m->nb; }
  DEV static int nd(const ShfModel* m) { return m->nd; }
  DEV static int np(const ShfModel* m) { return m->np; }
  DEV static int nlevels(const ShfModel* m) { return m->nlevels; }
  DEV static int nklevels(const ShfModel* m) { return m->nklevels; }
  static constexpr int NPC = 0;   // contact-point count unknown at compile time
  static constexpr int NKC = 0;   // kinematic depth unknown at compile time
};
template <int NB, int ND, int NP, int NL, int NK>
struct FixedDims {
  static constexpr int NPC = NP;
  static constexpr int NKC = NK;
  DEV static int nb(const ShfModel*) { return NB; }
  DEV static int nd(const ShfModel*) { return ND; }
  DEV static int np(const ShfModel*) { return NP; }
  DEV static int nlevels(const ShfModel*) { return NL; }
  DEV static int nklevels(const ShfModel*) { return NK; }
  static bool matches(const ShfModel& m) {
    return m.nb == NB && m.nd == ND && m.np == NP && m.nlevels == NL && m.nklevels == NK;
  }
};
typedef FixedDims<17, 12, 76, 3, 4> A1Dims;  // Unitree A1 as compiled from a1.urdf
typedef FixedDims<7, 6, 3, 6, 6> AbbDims;    // ABB IRB1200 + rod as compiled from abb_rod.urdf: a 6-deep chain (appendix A.2)
typedef FixedDims<7, 6, 59, 6, 6> AbbLinkDims;   // the same arm with link contacts: its seven box volumes' vertices are sample points too

// per-lane persistent body registers between phases of one sub-step
struct BodyRegs {
  float Rw[9], p[3], S[6], v[6], c[6];
  float IA[21], pA[6], U[6], invD, u;
};

// Per-lane model constants.  Lane l is body l, dof l (and its share of the contact points) for the whole
// launch, so everything the sub-step would re-read from the LDS model copy -- and the dependent
// index -> data LDS round trips that come with it -- is read once into registers.
// HOIST = false keeps the float constants in LDS behind the same member names (register-limited
// instantiations: one wavefront per env has to fit 128 VGPRs to keep 4096 envs resident).
#define LANE_CHILDREN 4
#define LANE_ANCESTORS 6
template <int N> struct RegVec {
  float v[N];
  DEV operator const float*() const { return v; }
  DEV float operator[](int k) const { return v[k]; }
  DEV void load(const float* p) {
#pragma unroll
    for (int k = 0; k < N; k++) v[k] = p[k];
  }
};
template <int N> struct LdsVec {
  const float* p;
  DEV operator const float*() const { return p; }
  DEV float operator[](int k) const { return p[k]; }
  DEV void load(const float* q) { p = q; }
};
struct RegF { float v; DEV operator float() const { return v; } DEV void load(const float* p) { v = *p; } };
struct LdsF { const float* p; DEV operator float() const { return *p; } DEV void load(const float* q) { p = q; } };
template <bool C, class A, class B> struct pick { typedef A type; };
template <class A, class B> struct pick<false, A, B> { typedef B type; };
template <bool HOIST>
struct LaneModelT {
  typedef typename pick<HOIST, RegF, LdsF>::type F;
  typedef typename pick<HOIST, RegVec<3>, LdsVec<3>>::type V3;
  typedef typename pick<HOIST, RegVec<6>, LdsVec<6>>::type V6;
  typedef typename pick<HOIST, RegVec<9>, LdsVec<9>>::type V9;
  bool isbody, isdyn, moving;
  // small tree indices share one register: jt+1 (3 bits) | klev+1 (5) | level+1 (5) | dof+1 (6) | dyn[parent] (5) | parent (5)
  unsigned tree;
  DEV int jt() const { return (int)(tree & 7u) - 1; }
  DEV int klev() const { return (int)((tree >> 3) & 31u) - 1; }
  DEV int level() const { return (int)((tree >> 8) & 31u) - 1; }
  DEV int dofi() const { return (int)((tree >> 13) & 63u) - 1; }
  DEV int dynpar() const { return (int)((tree >> 19) & 31u); }
  DEV int par() const { return (int)((tree >> 24) & 31u); }
  int nchild, child0, child[LANE_CHILDREN], pt0, npt;
  unsigned ancs;             // kinematic chain below the root in 5-bit fields, field klev-1 = this body (fixed-depth models only)
  DEV int anc(int k) const { return (int)((ancs >> (5 * k)) & 31u); }
  V3 tp, ax, com;
  V9 tr;
  V6 I6;
  F mass;
  int mode;                                                      // lane l as dof l
  F armature, kp, kd, effort, damping, lower, upper, vel_limit;
};
typedef LaneModelT<true> LaneModel;
template <class DM, class LM>
DEV void lane_model_load(const ShfModel* m, int l, LM& M) {
  const int nb = DM::nb(m), nd = DM::nd(m);
  M.isbody = l < nb;
  const int b = M.isbody ? l : 0;
  const int jt = M.isbody ? m->jtype[b] : -1;
  M.isdyn = M.isbody && m->dyn[b] == b;
  M.moving = M.isdyn && jt != SHF_JOINT_ROOT;
  const int par = jt > SHF_JOINT_ROOT ? m->parent[b] : 0;
  const int klev = (M.isbody && jt != SHF_JOINT_ROOT) ? m->klevel[b] : -1;
  const int level = M.isbody ? m->level[b] : -1;
  M.tree = (unsigned)(jt + 1) | (unsigned)(klev + 1) << 3 | (unsigned)(level + 1) << 8 | (unsigned)(m->dof[b] + 1) << 13 |
           (unsigned)m->dyn[par] << 19 | (unsigned)par << 24;
  M.child0 = m->child_start[b];
  M.nchild = M.isdyn ? m->child_count[b] : 0;
#pragma unroll
  for (int k = 0; k < LANE_CHILDREN; k++) M.child[k] = k < M.nchild ? m->child_list[M.child0 + k] : 0;
  M.pt0 = m->pt_start[b];
  M.npt = M.isdyn ? m->pt_count[b] : 0;
  if constexpr (DM::NKC > 0) {
    static_assert(DM::NKC <= LANE_ANCESTORS, "raise LANE_ANCESTORS");
    int a = b;
    M.ancs = 0u;
#pragma unroll
    for (int k = LANE_ANCESTORS - 1; k >= 0; k--) {
      if (k < klev) { M.ancs |= (unsigned)a << (5 * k); a = m->parent[a]; }
    }
  }
  M.tp.load(m->tpos[b]); M.ax.load(m->axis[b]); M.com.load(m->com[b]);
  M.tr.load(m->trot[b]);
  M.I6.load(m->inertia[b]);
  M.mass.load(&m->mass[b]);
  const int d = l < nd ? l : 0;
  M.mode = m->drive_mode[d];
  M.armature.load(&m->armature[d]); M.kp.load(&m->kp[d]); M.kd.load(&m->kd[d]); M.effort.load(&m->effort[d]);
  M.damping.load(&m->damping[d]); M.lower.load(&m->lower[d]); M.upper.load(&m->upper[d]); M.vel_limit.load(&m->vel_limit[d]);
}

// Forward kinematics: pose, motion subspace, velocity, bias acceleration of every reported
// body.  Phase 1 (all lanes at once): the joint's local rotation Rl = trot * Rot(axis, q).
// Phase 2 (level by level through LDS): R = Rp * Rl, p = pp + Rp * tpos, a_w = R * axis.
template <int G, class DM = DynDims, class LM = LaneModel>
DEV void kinematics(const ShfModel* m, const EnvLds& L, int l, const LM& M, BodyRegs& B) {
  const bool isbody = M.isbody;
  const int jt = M.jt();
  PHASE_BEGIN();
  float Rl[9], qv = 0.0f, qdv = 0.0f;
  const float* tp = M.tp;
  const float* ax = M.ax;
  const float* tr = M.tr;
  const int par = M.par(), klev = M.klev();
  if (isbody && jt != SHF_JOINT_ROOT) {
    if (jt == SHF_JOINT_WELD) {
#pragma unroll
      for (int k = 0; k < 9; k++) Rl[k] = tr[k];
    } else {
      const int d = M.dofi();
      qv = L.dofb[d * DOF_STRIDE];
      qdv = L.dofb[d * DOF_STRIDE + 1];
      if (jt == SHF_JOINT_REVOLUTE) {
        float sn, cs;
        sincos_spec(qv, &sn, &cs);
        const float oc = 1.0f - cs;
        const float Rq[9] = {fmaf(oc, ax[0] * ax[0], cs),           fmaf(oc, ax[0] * ax[1], -(sn * ax[2])), fmaf(oc, ax[0] * ax[2], sn * ax[1]),
                             fmaf(oc, ax[1] * ax[0], sn * ax[2]),    fmaf(oc, ax[1] * ax[1], cs),            fmaf(oc, ax[1] * ax[2], -(sn * ax[0])),
                             fmaf(oc, ax[2] * ax[0], -(sn * ax[1])), fmaf(oc, ax[2] * ax[1], sn * ax[0]),    fmaf(oc, ax[2] * ax[2], cs)};
        mm3(tr, Rq, Rl);
      } else {
#pragma unroll
        for (int k = 0; k < 9; k++) Rl[k] = tr[k];
      }
    }
  }
  if (l == 0) {
    quat_to_mat(L.root + 3, B.Rw);
    B.p[0] = B.p[1] = B.p[2] = 0.0f;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      B.v[k] = m->fixed_base ? 0.0f : L.root[10 + k];
      B.v[3 + k] = m->fixed_base ? 0.0f : L.root[7 + k];
    }
#pragma unroll
    for (int k = 0; k < 6; k++) { B.S[k] = 0.0f; B.c[k] = 0.0f; }
    float* o = L.pose;
#pragma unroll
    for (int k = 0; k < 9; k++) o[k] = B.Rw[k];
#pragma unroll
    for (int k = 0; k < 3; k++) o[9 + k] = B.p[k];
#pragma unroll
    for (int k = 0; k < 6; k++) o[12 + k] = B.v[k];
  }
  PHASE_MARK(0);
  if constexpr (DM::NKC > 0) {
    // Known depth: instead of one LDS hand-off per kinematic level, every lane publishes its joint's local
    // rotation and state once, then walks its own chain from the root down, repeating the ancestors'
    // compositions (same operands, same order => the same poses the level loop produces).
    if (isbody && jt != SHF_JOINT_ROOT) {
      float* o = L.pose + l * POSE_STRIDE;
#pragma unroll
      for (int k = 0; k < 9; k++) o[k] = Rl[k];
      o[9] = qv; o[10] = qdv;
    }
    GROUP_SYNC();
    if (klev >= 1) {
      float Rp[9], pp[3], vp[6], Sn[6], cn[6];
#pragma unroll
      for (int k = 0; k < 9; k++) Rp[k] = L.pose[k];
#pragma unroll
      for (int k = 0; k < 3; k++) pp[k] = L.pose[9 + k];
#pragma unroll
      for (int k = 0; k < 6; k++) { vp[k] = L.pose[12 + k]; Sn[k] = 0.0f; cn[k] = 0.0f; }
#pragma unroll
      for (int d = 0; d < DM::NKC; d++) {
        if (d < klev) {
          const int a = M.anc(d);
          const float* rec = L.pose + a * POSE_STRIDE;
          const int ja = m->jtype[a];
          float Ra[9], ta[3], aa[3], t[3], Rn[9], pn[3];
#pragma unroll
          for (int k = 0; k < 9; k++) Ra[k] = rec[k];
          const float qa = rec[9], qda = rec[10];
#pragma unroll
          for (int k = 0; k < 3; k++) { ta[k] = m->tpos[a][k]; aa[k] = m->axis[a][k]; }
          mv3(Rp, ta, t);
#pragma unroll
          for (int k = 0; k < 3; k++) pn[k] = pp[k] + t[k];
          mm3(Rp, Ra, Rn);
          if (ja == SHF_JOINT_WELD) {
#pragma unroll
            for (int k = 0; k < 6; k++) { Sn[k] = 0.0f; cn[k] = 0.0f; }
          } else {
            float aw[3], t2[3];
            mv3(Rn, aa, aw);
            if (ja == SHF_JOINT_REVOLUTE) {
              cross3(pn, aw, t2);
#pragma unroll
              for (int k = 0; k < 3; k++) { Sn[k] = aw[k]; Sn[3 + k] = t2[k]; }
            } else {
#pragma unroll
              for (int k = 0; k < 3; k++) { pn[k] = fmaf(aw[k], qa, pn[k]); Sn[k] = 0.0f; Sn[3 + k] = aw[k]; }
            }
            float vJ[6], cc[6];
#pragma unroll
            for (int k = 0; k < 6; k++) vJ[k] = Sn[k] * qda;
            crm(vp, vJ, cc);
#pragma unroll
            for (int k = 0; k < 6; k++) { cn[k] = cc[k]; vp[k] = vp[k] + vJ[k]; }
          }
#pragma unroll
          for (int k = 0; k < 9; k++) Rp[k] = Rn[k];
#pragma unroll
          for (int k = 0; k < 3; k++) pp[k] = pn[k];
        }
      }
#pragma unroll
      for (int k = 0; k < 9; k++) B.Rw[k] = Rp[k];
#pragma unroll
      for (int k = 0; k < 3; k++) B.p[k] = pp[k];
#pragma unroll
      for (int k = 0; k < 6; k++) { B.v[k] = vp[k]; B.S[k] = Sn[k]; B.c[k] = cn[k]; }
    }
    GROUP_SYNC();   // every lane has read the records it needs before the slots turn into poses
    if (klev >= 1) {
      float* o = L.pose + l * POSE_STRIDE;
#pragma unroll
      for (int k = 0; k < 9; k++) o[k] = B.Rw[k];
#pragma unroll
      for (int k = 0; k < 3; k++) o[9 + k] = B.p[k];
#pragma unroll
      for (int k = 0; k < 6; k++) o[12 + k] = B.v[k];
    }
    GROUP_SYNC();
    PHASE_MARK(1);
    return;
  }
  const int nk = DM::nklevels(m);
  for (int lev = 1; lev <= nk; lev++) {
    GROUP_SYNC();
    if (klev == lev) {
      const float* pp = L.pose + par * POSE_STRIDE;
      float Rp[9], vp[6], t[3];
#pragma unroll
      for (int k = 0; k < 9; k++) Rp[k] = pp[k];
#pragma unroll
      for (int k = 0; k < 6; k++) vp[k] = pp[12 + k];
      mv3(Rp, tp, t);
#pragma unroll
      for (int k = 0; k < 3; k++) B.p[k] = pp[9 + k] + t[k];
      mm3(Rp, Rl, B.Rw);
      if (jt == SHF_JOINT_WELD) {
#pragma unroll
        for (int k = 0; k < 6; k++) { B.v[k] = vp[k]; B.S[k] = 0.0f; B.c[k] = 0.0f; }
      } else {
        float aw[3];
        mv3(B.Rw, ax, aw);
        if (jt == SHF_JOINT_REVOLUTE) {
          cross3(B.p, aw, t);
#pragma unroll
          for (int k = 0; k < 3; k++) { B.S[k] = aw[k]; B.S[3 + k] = t[k]; }
        } else {
#pragma unroll
          for (int k = 0; k < 3; k++) { B.p[k] = fmaf(aw[k], qv, B.p[k]); B.S[k] = 0.0f; B.S[3 + k] = aw[k]; }
        }
        float vJ[6];
#pragma unroll
        for (int k = 0; k < 6; k++) vJ[k] = B.S[k] * qdv;
        crm(vp, vJ, B.c);
#pragma unroll
        for (int k = 0; k < 6; k++) B.v[k] = vp[k] + vJ[k];
      }
      float* o = L.pose + l * POSE_STRIDE;
#pragma unroll
      for (int k = 0; k < 9; k++) o[k] = B.Rw[k];
#pragma unroll
      for (int k = 0; k < 3; k++) o[9 + k] = B.p[k];
#pragma unroll
      for (int k = 0; k < 6; k++) o[12 + k] = B.v[k];
    }
  }
  GROUP_SYNC();
  PHASE_MARK(1);
}

// spatial inertia about O in world axes (packed) and velocity-product bias force
DEV void rigid_inertia_p(float mass, const float* com, const float* I6, const float* Rw, const float* p, const float* v,
                         float* IA, float* pA) {
  float cw[3], Ic[9], T[9];
  mv3(Rw, com, cw);
#pragma unroll
  for (int k = 0; k < 3; k++) cw[k] += p[k];
  Ic[0] = I6[0]; Ic[1] = I6[1]; Ic[2] = I6[2];
  Ic[3] = I6[1]; Ic[4] = I6[3]; Ic[5] = I6[4];
  Ic[6] = I6[2]; Ic[7] = I6[4]; Ic[8] = I6[5];
  mm3(Rw, Ic, T);
  const float c2 = dot3(cw, cw);
#pragma unroll
  for (int k = 0; k < 21; k++) IA[k] = 0.0f;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = i; j < 3; j++) {
      float iw = dot3(T + 3 * i, Rw + 3 * j);
      float par = (i == j ? c2 : 0.0f) - cw[i] * cw[j];
      IA[SYM(i, j)] = fmaf(mass, par, iw);
    }
  float h[3] = {mass * cw[0], mass * cw[1], mass * cw[2]};
  IA[SYM(0, 4)] = -h[2]; IA[SYM(0, 5)] = h[1];
  IA[SYM(1, 3)] = h[2];  IA[SYM(1, 5)] = -h[0];
  IA[SYM(2, 3)] = -h[1]; IA[SYM(2, 4)] = h[0];
  IA[SYM(3, 3)] = mass; IA[SYM(4, 4)] = mass; IA[SYM(5, 5)] = mass;
  float J[9] = {IA[SYM(0, 0)], IA[SYM(0, 1)], IA[SYM(0, 2)], IA[SYM(0, 1)], IA[SYM(1, 1)],
                IA[SYM(1, 2)], IA[SYM(0, 2)], IA[SYM(1, 2)], IA[SYM(2, 2)]};
  float n[3], f[3], t[3], t2[3];
  mv3(J, v, n);
  cross3(h, v + 3, t);
#pragma unroll
  for (int k = 0; k < 3; k++) n[k] += t[k];
  cross3(v, h, t);
#pragma unroll
  for (int k = 0; k < 3; k++) f[k] = fmaf(mass, v[3 + k], t[k]);
  cross3(v, n, t);
  cross3(v + 3, f, t2);
#pragma unroll
  for (int k = 0; k < 3; k++) pA[k] = t[k] + t2[k];
  cross3(v, f, t);
#pragma unroll
  for (int k = 0; k < 3; k++) pA[3 + k] = t[k];
}
DEV void rigid_inertia(float mass, const float* com, const float* I6, BodyRegs& B) {
  rigid_inertia_p(mass, com, I6, B.Rw, B.p, B.v, B.IA, B.pA);
}
// mscale: this env's row of SHF_T_BODY_MASS_SCALE or null -- the body's mass and inertia tensor times its factor (read here,
// once per sub-step and only when bound, rather than carried in a register by every launch)
template <class LM>
DEV void body_inertia(const LM& M, BodyRegs& B, const float* mscale = nullptr, int b = 0) {
  float mass = M.mass, I6[6];
#pragma unroll
  for (int k = 0; k < 6; k++) I6[k] = M.I6[k];
  if (mscale) {
    const float s = mscale[b];
    mass *= s;
#pragma unroll
    for (int k = 0; k < 6; k++) I6[k] *= s;
  }
  rigid_inertia(mass, M.com, I6, B);
}

typedef ShfScene SceneDev;  // box actors of the scene (gym.create_box), staged in LDS next to the model
struct StepCtx {
  const ShfModel* m;   // LDS copy
  ShfSimParams sp;
  TerrainDev terr;
  const SceneDev* scene;  // LDS copy, may be null when the scene has no boxes
  int32_t* dropped = nullptr;   // this env's word of SHF_T_DROPPED (contacts beyond the per-env limits), may be null
  const float* mscale = nullptr;  // this env's row of SHF_T_BODY_MASS_SCALE (factor on each body's mass and inertia), may be null
  const ShfHullSet* hulls = nullptr;   // the articulation's convex hulls (global memory), may be null
};
// SHF_T_CONTACT_HIST bound: SimArgs.dropped points at it instead of SHF_T_DROPPED -- rows of SHF_CONTACT_HIST_BINS + 1 words, the
// histogram and, in the last word, the env's drop counter -- and this bit is set in the launch's copy of sp.max_contacts (the
// kernels read the cap through hard_kmax()).  No kernel argument of its own: the fused A1 step is at its scalar-register limit.
#define SHF_HIST_FLAG 0x100
// this env's drop counter from the launch's `dropped` argument: a word of SHF_T_DROPPED, or the last word of the env's row of
// SHF_T_CONTACT_HIST when that is bound (SHF_HIST_FLAG in the launch's max_contacts)
DEV int32_t* env_dropped(int32_t* base, const ShfSimParams& sp, int e) {
  if (!base) return nullptr;
  return (sp.max_contacts & SHF_HIST_FLAG) ? base + (size_t)e * (SHF_CONTACT_HIST_BINS + 1) + SHF_CONTACT_HIST_BINS : base + e;
}
// one sub-step offered `total` candidate constraints (before the cap): this env's histogram row, when bound
DEV void contact_hist_count(const StepCtx& C, int l, int total) {
  if (l == 0 && C.dropped && (C.sp.max_contacts & SHF_HIST_FLAG))
    C.dropped[(total < SHF_CONTACT_HIST_BINS - 1 ? total : SHF_CONTACT_HIST_BINS - 1) - SHF_CONTACT_HIST_BINS] += 1;
}
DEV int hard_kmax_of(const ShfSimParams& sp, int cap) {
  const int mc = sp.max_contacts & 0xff;
  return mc > 0 ? (mc < cap ? mc : cap) : cap;
}
// mass of body b in this env (oracle: body_mass)
DEV float body_mass(const StepCtx& C, int b) { return C.mscale ? C.m->mass[b] * C.mscale[b] : C.m->mass[b]; }

// solve IA x = -pA for a symmetric positive definite 6x6 in packed storage (LDL^T).  Factorisation and substitution are
// separate so that several right-hand sides share one factorisation (the pair laws: up to 13 solves with two matrices);
// ldlt_solve6 = both, the same operations in the same order as the oracle's ldlt_solve6.
struct Ldlt6 { float Lm[6][6], Dg[6], iD[6]; };
DEV void ldlt_factor6(const float* IA, Ldlt6& F) {
#pragma unroll
  for (int j = 0; j < 6; j++) {
    float d = IA[SYM(j, j)];
#pragma unroll
    for (int k = 0; k < j; k++) d = fmaf(-(F.Lm[j][k] * F.Lm[j][k]), F.Dg[k], d);
    F.Dg[j] = d;
    const float id = rcp_spec(d);
    F.iD[j] = id;
#pragma unroll
    for (int i = j + 1; i < 6; i++) {
      float v = IA[SYM(j, i)];
#pragma unroll
      for (int k = 0; k < j; k++) v = fmaf(-(F.Lm[i][k] * F.Lm[j][k]), F.Dg[k], v);
      F.Lm[i][j] = v * id;
    }
  }
}
DEV void ldlt_substitute6(const Ldlt6& F, const float* pA, float* x) {
  float y[6];
#pragma unroll
  for (int i = 0; i < 6; i++) {
    float v = -pA[i];
#pragma unroll
    for (int k = 0; k < i; k++) v = fmaf(-F.Lm[i][k], y[k], v);
    y[i] = v;
  }
#pragma unroll
  for (int i = 0; i < 6; i++) y[i] = y[i] * F.iD[i];
#pragma unroll
  for (int i = 5; i >= 0; i--) {
    float v = y[i];
#pragma unroll
    for (int k = i + 1; k < 6; k++) v = fmaf(-F.Lm[k][i], x[k], v);
    x[i] = v;
  }
}
DEV void ldlt_solve6(const float* IA, const float* pA, float* x) {
  Ldlt6 F;
  ldlt_factor6(IA, F);
  ldlt_substitute6(F, pA, x);
}

// Tangential coefficient c_t of the regularised Coulomb law f_t = -c_t v_t(end of step), c_t = mu f_n / max(|v_t|, v_eps),
// with the bound taken from the START-of-step relative velocity `vs` of the point (normal force k pen - beta v_n and
// tangential speed as they are now).  The force itself is linearised about the free-fall prediction vp = vs + dt g;
// taking the bound from vp too would add beta dt g ~ 17 N to the normal force of every touching point.
DEV float friction_coefficient(const float* n, const float* vs, float pen, float mu, float beta, float veps) {
  const float vns = dot3(n, vs);
  const float fns = rmaxf(fmaf(-beta, vns, pen), 0.0f);
  const float vts[3] = {fmaf(-vns, n[0], vs[0]), fmaf(-vns, n[1], vs[1]), fmaf(-vns, n[2], vs[2])};
  return mu * fns * rsqrt_spec(rmaxf(dot3(vts, vts), veps * veps));   // = mu fns / max(|v_t|, v_eps)
}

#include "shf_boxes.h"
// tools/experiment.py builds with -DSHF_EXP_SHUFFLE_HANDOFF: the inward pass's child -> parent hand-off through
// cross-lane moves (measured slower, profiles/r02_experiments.md); the product build takes the LDS slots
#ifdef SHF_EXP_SHUFFLE_HANDOFF
#include "experiments/shuffle_handoff.h"
#define SHF_HANDOFF_SHUFFLE true
#else
#define SHF_HANDOFF_SHUFFLE false
template <int G, class LM> DEV void exp_shuffle_handoff(const LM&, int, bool, BodyRegs&) {}
#endif

// Sample-point constants of the lane's contact rounds (point l + k*G in round k), for models whose point
// count is a compile-time constant.
template <int NR>
struct LanePoints {
  unsigned bodies;   // reported body of round k's point in 6-bit fields (NR <= 5)
  float pos[NR][3], rad[NR];
  DEV int body(int k) const { return (int)((bodies >> (6 * k)) & 63u); }
};
template <int G, int NR>
DEV void lane_points_load(const ShfModel* m, int np, int l, LanePoints<NR>& P) {
  static_assert(NR <= 5, "LanePoints packs at most five rounds");
  P.bodies = 0u;
#pragma unroll
  for (int k = 0; k < NR; k++) {
    const int i = l + k * G < np ? l + k * G : 0;
    P.bodies |= (unsigned)m->pt_body[i] << (6 * k);
    P.rad[k] = m->pt_radius[i];
#pragma unroll
    for (int j = 0; j < 3; j++) P.pos[k][j] = m->pt_pos[i][j];
  }
}

struct ContactConsts {
  float dt, g[3], kc, beta, mu, veps, max_depen, offset;
};
// Response of one sample point within the contact offset (phi < offset): r is moved to the contact point; the point
// responds if it is below the surface now or would be at the end of the step (speculative contact); writes the slot,
// returns `on`.
DEV float contact_point_response(const ContactConsts& K, const float* pb, float* r, const float* n, float rad, float phi, float* o) {
  float on = 0.0f;
#pragma unroll
  for (int k = 0; k < 3; k++) r[k] = fmaf(-rad, n[k], r[k]);
  float vb[6], vs[3], vp[3], t[3];
#pragma unroll
  for (int k = 0; k < 6; k++) vb[k] = pb[12 + k];
  cross3(vb, r, t);
#pragma unroll
  for (int k = 0; k < 3; k++) { vs[k] = vb[3 + k] + t[k]; vp[k] = fmaf(K.dt, K.g[k], vs[k]); }
  const float vn = dot3(n, vp);
  const float pen = rminf(-K.kc * phi, K.beta * K.max_depen);
  const float fn = fmaf(-K.beta, vn, pen);
  if ((phi < 0.0f || fmaf(K.dt, vn, phi) < 0.0f) && fn > 0.0f) {
    float vt[3] = {fmaf(-vn, n[0], vp[0]), fmaf(-vn, n[1], vp[1]), fmaf(-vn, n[2], vp[2])};
    const float ct = friction_coefficient(n, vs, pen, K.mu, K.beta, K.veps);
    on = 1.0f;
    o[PT_CT] = ct;
    o[PT_BN] = K.beta;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      o[PT_R + k] = r[k];
      o[PT_N + k] = n[k];
      o[PT_F + k] = fmaf(fn, n[k], -(ct * vt[k]));
    }
  }
  return on;
}

// Fold one active contact slot into its body's bias force and articulated inertia:
// pA -= [r x f0; f0],  IA += dt * (c_t * PointMass(r) + (beta - c_t) * w w^T),  w = [r x n; n].
DEV void contact_accumulate_p(const float* o, float dt, float* IA, float* pA) {
  const float r[3] = {o[PT_R], o[PT_R + 1], o[PT_R + 2]}, n[3] = {o[PT_N], o[PT_N + 1], o[PT_N + 2]}, f0[3] = {o[PT_F], o[PT_F + 1], o[PT_F + 2]};
  float t[3], wn[6];
  cross3(r, f0, t);
#pragma unroll
  for (int k = 0; k < 3; k++) { pA[k] -= t[k]; pA[3 + k] -= f0[k]; }
  cross3(r, n, t);
#pragma unroll
  for (int k = 0; k < 3; k++) { wn[k] = t[k]; wn[3 + k] = n[k]; }
  const float a = dt * o[PT_CT], bb = dt * (o[PT_BN] - o[PT_CT]);
  const float r2 = dot3(r, r);
#pragma unroll
  for (int i2 = 0; i2 < 3; i2++)
#pragma unroll
    for (int j2 = i2; j2 < 3; j2++)
      IA[SYM(i2, j2)] = fmaf(a, (i2 == j2 ? r2 : 0.0f) - r[i2] * r[j2], IA[SYM(i2, j2)]);
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
// The same fold with the 27 accumulators (IA[21], pA[6]) shared between two lanes: half 0 owns IA[0..10], half 1 owns
// IA[11..20] and pA.  Each accumulator sees exactly the operations contact_accumulate_p applies to it, in the same
// order; an element a lane does not own is left untouched (and its arithmetic is not generated).
template <int HALF>
DEV void contact_accumulate_half(const float* o, float dt, float* IA, float* pA) {
  auto own = [](int idx) { return HALF == 0 ? idx <= 10 : idx >= 11; };
  const float r[3] = {o[PT_R], o[PT_R + 1], o[PT_R + 2]}, n[3] = {o[PT_N], o[PT_N + 1], o[PT_N + 2]}, f0[3] = {o[PT_F], o[PT_F + 1], o[PT_F + 2]};
  float t[3], wn[6];
  if (HALF == 1) {
    cross3(r, f0, t);
#pragma unroll
    for (int k = 0; k < 3; k++) { pA[k] -= t[k]; pA[3 + k] -= f0[k]; }
  }
  cross3(r, n, t);
#pragma unroll
  for (int k = 0; k < 3; k++) { wn[k] = t[k]; wn[3 + k] = n[k]; }
  const float a = dt * o[PT_CT], bb = dt * (o[PT_BN] - o[PT_CT]);
  const float r2 = dot3(r, r);
#pragma unroll
  for (int i2 = 0; i2 < 3; i2++)
#pragma unroll
    for (int j2 = i2; j2 < 3; j2++)
      if (own(SYM(i2, j2))) IA[SYM(i2, j2)] = fmaf(a, (i2 == j2 ? r2 : 0.0f) - r[i2] * r[j2], IA[SYM(i2, j2)]);
  if (own(SYM(0, 4))) IA[SYM(0, 4)] = fmaf(a, -r[2], IA[SYM(0, 4)]);
  if (own(SYM(0, 5))) IA[SYM(0, 5)] = fmaf(a, r[1], IA[SYM(0, 5)]);
  if (own(SYM(1, 3))) IA[SYM(1, 3)] = fmaf(a, r[2], IA[SYM(1, 3)]);
  if (own(SYM(1, 5))) IA[SYM(1, 5)] = fmaf(a, -r[0], IA[SYM(1, 5)]);
  if (own(SYM(2, 3))) IA[SYM(2, 3)] = fmaf(a, -r[1], IA[SYM(2, 3)]);
  if (own(SYM(2, 4))) IA[SYM(2, 4)] = fmaf(a, r[0], IA[SYM(2, 4)]);
  if (own(SYM(3, 3))) IA[SYM(3, 3)] += a;
  if (own(SYM(4, 4))) IA[SYM(4, 4)] += a;
  if (own(SYM(5, 5))) IA[SYM(5, 5)] += a;
#pragma unroll
  for (int i2 = 0; i2 < 6; i2++) {
    const float bw = bb * wn[i2];
#pragma unroll
    for (int j2 = i2; j2 < 6; j2++)
      if (own(SYM(i2, j2))) IA[SYM(i2, j2)] = fmaf(bw, wn[j2], IA[SYM(i2, j2)]);
  }
}
DEV void contact_accumulate(const float* o, float dt, BodyRegs& B) { contact_accumulate_p(o, dt, B.IA, B.pA); }

// End-of-step force of an active slot: f = f0 - dt * B * a_point with the body's solved acceleration `ab`.
DEV void contact_force_final(float* o, const float* ab, float dt) {
  const float r[3] = {o[PT_R], o[PT_R + 1], o[PT_R + 2]}, n[3] = {o[PT_N], o[PT_N + 1], o[PT_N + 2]};
  float al[3] = {ab[0], ab[1], ab[2]}, t[3], ap[3];
  cross3(al, r, t);
#pragma unroll
  for (int k = 0; k < 3; k++) ap[k] = ab[3 + k] + t[k];
  const float an = dot3(n, ap);
  const float ct = o[PT_CT], bn = o[PT_BN];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const float Ba = fmaf(bn - ct, an * n[k], ct * ap[k]);
    o[PT_F + k] = fmaf(-dt, Ba, o[PT_F + k]);
  }
}

#include "shf_hard.h"

// One gym.simulate() for one env, executed by the G lanes of its group.
//   HARD: the velocity-level contact solve (ShfSimParams.solver == SHF_SOLVER_PGS; shf_hard.h): the contact passes only record
//   candidate constraints, the articulated-body solve runs free, substep_hard_finish solves and integrates.
//   dofb[d]: q, qd in;  tau_cmd (explicit effort), pos/vel targets via pt_tgt/vt_tgt (LDS, may be null)
//   fext: world force per reported body (global memory, this env) or nullptr; fpos: its world point of application or nullptr (CoM)
//   contact_out: LDS/global float[nb*3] written by body lanes (may be nullptr)
#define LANE_ROUNDS(G, DM) ((DM::NPC + (G) - 1) / (G) > 0 ? (DM::NPC + (G) - 1) / (G) : 1)
template <int G, bool BOX, class DM, class LM, class SC, bool SELF, bool LINK, bool RECORDS = true, int ARMNL = 0>
DEV void substep_hard_finish(const StepCtx& C, const EnvLds& L, int l, const LM& M, BodyRegs& B, const float* g, float* a, int nself,
                             int self_slot0, int link_slot0, int nlink, float* contact_out);
template <int G, bool BOX = false, class DM = DynDims, bool TW = false, class LM = LaneModel, class SC = DynScene,
          bool SELF = false, bool LINK = false, bool HARD = false, bool EXT = false>
DEV void substep(const StepCtx& C, const EnvLds& L, int l, const LM& M, const LanePoints<LANE_ROUNDS(G, DM)>& P,
                 const float* pos_tgt, const float* vel_tgt, const float* fext, float mu_shape, float* contact_out,
                 const BoxLane& BL = BoxLane(), const float* fpos = nullptr) {
  static_assert(!HARD || (DM::NPC == 0 && SC::NBX == 0 && G == 32), "the generic velocity-level solve: run-time shapes, 32 lanes per env");
  const ShfModel* m = C.m;
  const int nb = DM::nb(m), nd = DM::nd(m), np = DM::np(m);
  const float dt = C.sp.dt;
  const float gon = (float)m->gravity_on;
  const float g[3] = {C.sp.gravity[0] * gon, C.sp.gravity[1] * gon, C.sp.gravity[2] * gon};
  const bool isbody = M.isbody, isdyn = M.isdyn, moving = M.moving;
  const int mylevel = M.level();

  BodyRegs B;
  kinematics<G, DM, LM>(m, L, l, M, B);
  PHASE_BEGIN();
  if (isdyn) body_inertia(M, B, C.mscale, l);
  if (BOX) boxes_pose<G>(C, L, l, B);

  // external forces on reported bodies -- at the CoM, or at the world point fpos[3 b ..] when given
  // (gym.apply_rigid_body_force_at_pos_tensors(force, pos | None), robot.py:231-236) -- folded in ascending body order
  if (fext) {
    if (isbody) {
      float F[3] = {fext[3 * l], fext[3 * l + 1], fext[3 * l + 2]};
      float cw[3], t[3];
      if (fpos) {
#pragma unroll
        for (int k = 0; k < 3; k++) cw[k] = fpos[3 * l + k] - L.root[k];   // its arm about O, the root's position
      } else {
        mv3(B.Rw, M.com, cw);
#pragma unroll
        for (int k = 0; k < 3; k++) cw[k] += B.p[k];
      }
      cross3(cw, F, t);
      float* o = L.xch + l * XCH_STRIDE;
      o[0] = t[0]; o[1] = t[1]; o[2] = t[2]; o[3] = F[0]; o[4] = F[1]; o[5] = F[2];
      o[6] = (F[0] == 0.0f && F[1] == 0.0f && F[2] == 0.0f) ? 0.0f : 1.0f;
    }
    GROUP_SYNC();
    if (isdyn) {
      for (int b = l; b < nb; b++) {
        if (m->dyn[b] != l) continue;
        const float* o = L.xch + b * XCH_STRIDE;
        if (o[6] == 0.0f) continue;
#pragma unroll
        for (int k = 0; k < 6; k++) B.pA[k] -= o[k];
      }
    }
    GROUP_SYNC();
  }

  PHASE_MARK(2);
  // contacts: one lane per sample point, results parked in LDS
  const float kc = C.sp.contact_k, dc = C.sp.contact_d, veps = C.sp.friction_vel;
  const float beta = fmaf(kc, dt, dc);
  const float mu = 0.5f * (mu_shape + C.terr.t.friction);
  const ContactConsts K = {dt, {g[0], g[1], g[2]}, kc, beta, mu, veps, C.sp.max_depen_vel, C.sp.contact_offset};
  unsigned long long active[LANE_ROUNDS(G, DM)];
  if constexpr (DM::NPC > 0 && G < 64) {
    // known point count: pass 1 places every round's point and queries the terrain without branches, so the
    // pose reads and height loads of all rounds are in flight together; pass 2 is the (divergent) response
    constexpr int NR = (DM::NPC + G - 1) / G;
    float r[NR][3], n[NR][3], phi[NR];
#pragma unroll
    for (int k = 0; k < NR; k++) {
      const float* pb = L.pose + P.body(k) * POSE_STRIDE;
      float Rb[9], h;
#pragma unroll
      for (int j = 0; j < 9; j++) Rb[j] = pb[j];
      mv3(Rb, P.pos[k], r[k]);
#pragma unroll
      for (int j = 0; j < 3; j++) r[k][j] += pb[9 + j];
      terrain_query<TW>(C.terr, L.root[0] + r[k][0], L.root[1] + r[k][1], &h, n[k]);
      phi[k] = fmaf(L.root[2] + r[k][2] - h, n[k][2], -P.rad[k]);
    }
    const int lane0 = (int)(threadIdx.x & 63u) - l;   // first lane of this env's group within the wavefront
#pragma unroll
    for (int k = 0; k < NR; k++) {
      const int i = l + k * G;
      float on = 0.0f;
      if (i < DM::NPC) {
        float* o = L.pt + i * PT_STRIDE;
        if (phi[k] < K.offset) on = contact_point_response(K, L.pose + P.body(k) * POSE_STRIDE, r[k], n[k], P.rad[k], phi[k], o);
        o[PT_ON] = on;
      }
      // bit j = point k*G + j of this env is in contact
      active[k] = (__ballot(on != 0.0f) >> lane0) & (G >= 64 ? ~0ull : ((1ull << (G & 63)) - 1ull));
    }
  } else {
    for (int i = l; i < np; i += G) {
      const int b = m->pt_body[i];
      const float* pb = L.pose + b * POSE_STRIDE;
      float Rb[9], lp[3] = {m->pt_pos[i][0], m->pt_pos[i][1], m->pt_pos[i][2]}, r[3], n[3], h;
#pragma unroll
      for (int k = 0; k < 9; k++) Rb[k] = pb[k];
      mv3(Rb, lp, r);
#pragma unroll
      for (int k = 0; k < 3; k++) r[k] += pb[9 + k];
      terrain_query<TW>(C.terr, L.root[0] + r[0], L.root[1] + r[1], &h, n);
      const float rad = m->pt_radius[i];
      const float phi = fmaf(L.root[2] + r[2] - h, n[2], -rad);
      float* o = L.pt + i * PT_STRIDE;
      float on = 0.0f;
      if constexpr (HARD) {
        // candidate constraint: gap from rest_offset inside the contact offset (oracle: substep, hard point loop)
        if (phi < C.sp.contact_offset + C.sp.rest_offset) {
          on = 1.0f;
#pragma unroll
          for (int k = 0; k < 3; k++) { o[PT_R + k] = fmaf(-rad, n[k], r[k]); o[PT_N + k] = n[k]; }
          o[PT_F] = phi - C.sp.rest_offset; o[PT_F + 1] = mu; o[PT_F + 2] = 0.0f; o[PT_CT] = 0.0f; o[PT_BN] = 0.0f;
        }
      } else {
        if (phi < K.offset) on = contact_point_response(K, pb, r, n, rad, phi, o);
      }
      o[PT_ON] = on;
    }
  }
  GROUP_SYNC();
  PHASE_MARK(3);
  if constexpr (DM::NPC > 0 && G < 64) {
    // the body lane picks its active points (ascending) out of the ballots: no flag reads, no idle iterations
    constexpr int NR = (DM::NPC + G - 1) / G;
    if (isdyn) {
      const int i0 = M.pt0, i1 = i0 + M.npt;
#pragma unroll
      for (int k = 0; k < NR; k++) {
        const int a0 = (i0 > k * G ? i0 : k * G) - k * G, a1 = (i1 < (k + 1) * G ? i1 : (k + 1) * G) - k * G;
        unsigned long long bits = a1 > a0 ? (active[k] >> a0) & (a1 - a0 >= 64 ? ~0ull : ((1ull << (a1 - a0)) - 1ull)) : 0ull;
        while (bits) {
          const int j = __builtin_ctzll(bits);
          bits &= bits - 1ull;
          contact_accumulate(L.pt + (k * G + a0 + j) * PT_STRIDE, dt, B);
        }
      }
    }
  } else if constexpr (!HARD) {
    if (isdyn) {
      const int i0 = M.pt0, i1 = i0 + M.npt;
      for (int i = i0; i < i1; i++) {
        const float* o = L.pt + i * PT_STRIDE;
        if (o[PT_ON] == 0.0f) continue;
        contact_accumulate(o, dt, B);
      }
    }
  }

  // self-collision slots sit behind the articulation's sample points and the box slots
  int nself = 0;
  const int self_slot0 = np + (BOX ? box_slots(slot_lay<SC>(m, C.scene)) : 0);
  if constexpr (SELF) nself = HARD ? self_contacts_eval<G>(C, L, l, self_slot0, mu_shape) : self_contacts<G>(C, L, l, isdyn, self_slot0, B, mu_shape);
  BoxMasks BM;
  const int link_slot0 = self_slot0 + (SELF ? SHF_MAX_SELF_CONTACTS : 0);   // 2 x SHF_MAX_LINK_CONTACTS slots when LINK
  if (BOX) boxes_contacts<G, SC, LINK && BOX, HARD, EXT && LINK && BOX>(C, L, l, B, mu_shape, g, BL, BM, link_slot0);
  PHASE_MARK(4);

  // joint-space efforts: one lane per dof
  if (l < nd) {
    float* D = L.dofb + l * DOF_STRIDE;
    const float q = D[0], qd = D[1];
    float t0 = 0.0f, de = M.armature;
    const int mode = M.mode;
    if (mode == SHF_DOF_MODE_EFFORT) {
      t0 = D[5];
    } else if (mode == SHF_DOF_MODE_POS || mode == SHF_DOF_MODE_VEL) {
      float kp = mode == SHF_DOF_MODE_POS ? M.kp : 0.0f, kd = M.kd;
      const float tq = pos_tgt ? pos_tgt[l] : 0.0f, tv = (mode == SHF_DOF_MODE_VEL && vel_tgt) ? vel_tgt[l] : 0.0f;
      const float est = fmaf(kp, tq - q, kd * (tv - qd));
      const float lim = M.effort;
      // drive saturation: scale both gains so the torque starts at the effort limit and stays implicit
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
  GROUP_SYNC();
  PHASE_MARK(5);

  // inward pass
  const int nl = DM::nlevels(m);
  for (int lev = nl; lev >= 1; lev--) {
    if (moving && mylevel == lev) {
      const int d = M.dofi();
#pragma unroll
      for (int i = 0; i < 6; i++) {
        float acc = SYMG(B.IA, i, 0) * B.S[0];
#pragma unroll
        for (int j = 1; j < 6; j++) acc = fmaf(SYMG(B.IA, i, j), B.S[j], acc);
        B.U[i] = acc;
      }
      float D = B.S[0] * B.U[0];
#pragma unroll
      for (int j = 1; j < 6; j++) D = fmaf(B.S[j], B.U[j], D);
      D += L.dofb[d * DOF_STRIDE + 3];
      float sp = B.S[0] * B.pA[0];
#pragma unroll
      for (int j = 1; j < 6; j++) sp = fmaf(B.S[j], B.pA[j], sp);
      const float invD = rcp_spec(D);
      B.invD = invD;
      B.u = L.dofb[d * DOF_STRIDE + 2] - sp;
      float W[6];
#pragma unroll
      for (int i = 0; i < 6; i++) W[i] = B.U[i] * invD;
#pragma unroll
      for (int i = 0; i < 6; i++)
#pragma unroll
        for (int j = i; j < 6; j++) B.IA[SYM(i, j)] = fmaf(-B.U[i], W[j], B.IA[SYM(i, j)]);
      float pa[6];
#pragma unroll
      for (int i = 0; i < 6; i++) {
        float acc = SYMG(B.IA, i, 0) * B.c[0];
#pragma unroll
        for (int j = 1; j < 6; j++) acc = fmaf(SYMG(B.IA, i, j), B.c[j], acc);
        pa[i] = fmaf(W[i], B.u, B.pA[i] + acc);
      }
#pragma unroll
      for (int k = 0; k < 6; k++) B.pA[k] = pa[k];
    }
    if constexpr (SHF_HANDOFF_SHUFFLE) {
      // experiment (profiles/r02_experiments.md): child -> parent through DPP / ds_bpermute instead of LDS slots
      exp_shuffle_handoff<G>(M, l, isdyn && mylevel == lev - 1, B);
    } else {
      if (moving && mylevel == lev) {
        float* o = L.xch + l * XCH_STRIDE;
#pragma unroll
        for (int k = 0; k < 21; k++) o[k] = B.IA[k];
#pragma unroll
        for (int k = 0; k < 6; k++) o[21 + k] = B.pA[k];
      }
      GROUP_SYNC();
      if (isdyn && mylevel == lev - 1) {
        // children in child_list order: the first LANE_CHILDREN from registers, any further ones through LDS
#pragma unroll
        for (int kk = 0; kk < LANE_CHILDREN; kk++) {
          if (kk < M.nchild) {
            const float* o = L.xch + M.child[kk] * XCH_STRIDE;
#pragma unroll
            for (int k = 0; k < 21; k++) B.IA[k] += o[k];
#pragma unroll
            for (int k = 0; k < 6; k++) B.pA[k] += o[21 + k];
          }
        }
        for (int kk = LANE_CHILDREN; kk < M.nchild; kk++) {
          const float* o = L.xch + m->child_list[M.child0 + kk] * XCH_STRIDE;
#pragma unroll
          for (int k = 0; k < 21; k++) B.IA[k] += o[k];
#pragma unroll
          for (int k = 0; k < 6; k++) B.pA[k] += o[21 + k];
        }
      }
    }
    // no hand-off here: every exchange slot is written once per sub-step, by its owner, before the sync above
  }

  PHASE_MARK(6);
  // root acceleration (primed)
  float a[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
  if (l == 0) {
    if (m->fixed_base) {
      a[3] = -g[0]; a[4] = -g[1]; a[5] = -g[2];
    } else {
      ldlt_solve6(B.IA, B.pA, a);
    }
#pragma unroll
    for (int k = 0; k < 6; k++) L.acc[k] = a[k];
  }
  PHASE_MARK(7);
  // outward pass
  for (int lev = 1; lev <= nl; lev++) {
    GROUP_SYNC();
    if (moving && mylevel == lev) {
      const float* pa = L.acc + M.dynpar() * 6;
      float ap[6];
#pragma unroll
      for (int i = 0; i < 6; i++) ap[i] = pa[i] + B.c[i];
      float ua = B.U[0] * ap[0];
#pragma unroll
      for (int j = 1; j < 6; j++) ua = fmaf(B.U[j], ap[j], ua);
      const float qdd = (B.u - ua) * B.invD;
      L.dofb[M.dofi() * DOF_STRIDE + 4] = qdd;
#pragma unroll
      for (int i = 0; i < 6; i++) { a[i] = fmaf(B.S[i], qdd, ap[i]); L.acc[l * 6 + i] = a[i]; }
    }
  }
  GROUP_SYNC();
  PHASE_MARK(8);

  if constexpr (HARD) {
    substep_hard_finish<G, BOX, DM, LM, SC, SELF, LINK>(C, L, l, M, B, g, a, nself, self_slot0, link_slot0, BM.nlink, contact_out);
    return;
  }
  // net contact force per reported body
  if (contact_out) {
    if constexpr (DM::NPC > 0 && G < 64) {
      // active points come from the ballots taken in the contact pass
      constexpr int NR = (DM::NPC + G - 1) / G;
#pragma unroll
      for (int k = 0; k < NR; k++) {
        if ((active[k] >> l) & 1ull) contact_force_final(L.pt + (l + k * G) * PT_STRIDE, L.acc + m->dyn[P.body(k)] * 6, dt);
      }
      GROUP_SYNC();
      if (isbody) {
        float f[3] = {0.0f, 0.0f, 0.0f};
        const int dl = m->dyn[l];
        const int i0 = m->pt_start[dl], i1 = i0 + m->pt_count[dl];
#pragma unroll
        for (int k = 0; k < NR; k++) {
          const int a0 = (i0 > k * G ? i0 : k * G) - k * G, a1 = (i1 < (k + 1) * G ? i1 : (k + 1) * G) - k * G;
          unsigned long long bits = a1 > a0 ? (active[k] >> a0) & (a1 - a0 >= 64 ? ~0ull : ((1ull << (a1 - a0)) - 1ull)) : 0ull;
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
    } else {
      for (int i = l; i < np; i += G) {
        float* o = L.pt + i * PT_STRIDE;
        if (o[PT_ON] == 0.0f) continue;
        contact_force_final(o, L.acc + m->dyn[m->pt_body[i]] * 6, dt);
      }
      GROUP_SYNC();
      if (isbody) {
        float f[3] = {0.0f, 0.0f, 0.0f};
        const int dl = m->dyn[l];
        const int i0 = m->pt_start[dl], i1 = i0 + m->pt_count[dl];
        for (int i = i0; i < i1; i++) {
          const float* o = L.pt + i * PT_STRIDE;
          if (o[PT_ON] == 0.0f || m->pt_body[i] != l) continue;
          f[0] += o[PT_F]; f[1] += o[PT_F + 1]; f[2] += o[PT_F + 2];
        }
        contact_out[3 * l] = f[0]; contact_out[3 * l + 1] = f[1]; contact_out[3 * l + 2] = f[2];
      }
    }
  }

  if constexpr (SELF) {
    if (contact_out) {
      GROUP_SYNC();
      self_contact_forces(C, L, l, self_slot0, nself, contact_out);
    }
  }
  if (BOX) {
    GROUP_SYNC();
    boxes_finish<G, SC>(C, L, l, B, contact_out, BL, BM, link_slot0);
  }

  PHASE_MARK(9);
  // semi-implicit Euler
  if (l < nd) {
    float* D = L.dofb + l * DOF_STRIDE;
    const float vl = M.vel_limit;
    const float qd = rclampf(fmaf(dt, D[4], D[1]), -vl, vl);
    D[1] = qd;
    D[0] = fmaf(dt, qd, D[0]);
  }
  if (l == 0 && !m->fixed_base) {
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
  PHASE_MARK(10);
}

// rigid_body_state rows of this env -> `stage` (LDS, nb*13 floats); caller copies out coalesced
template <int G, class DM = DynDims, class LM = LaneModel>
DEV void body_states(const ShfModel* m, const EnvLds& L, int l, const LM& M, float* stage) {
  BodyRegs B;
  kinematics<G, DM, LM>(m, L, l, M, B);
  if (l < DM::nb(m)) {
    float* o = stage + 13 * l;
    float t[3], q[4];
#pragma unroll
    for (int k = 0; k < 3; k++) o[k] = L.root[k] + B.p[k];
    mat_to_quat(B.Rw, q);
#pragma unroll
    for (int k = 0; k < 4; k++) o[3 + k] = q[k];
    cross3(B.v, B.p, t);
#pragma unroll
    for (int k = 0; k < 3; k++) { o[7 + k] = B.v[3 + k] + t[k]; o[10 + k] = B.v[k]; }
  }
  GROUP_SYNC();
}
