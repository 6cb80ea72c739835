// shf_hull.h -- the convex narrow phase: mesh colliders as convex hulls.  The reference's links collide through
// <mesh> colliders (asset/urdf/abb_rod_description/urdf/abb_rod_isaac.urdf:38-113) that PhysX cooks into convex hulls [EXT]; every
// shape of an env meets every other (shifu/units/units.py:68, filter 0).  The published algorithm: the separating-axis test over the
// face normals of both polytopes and the edge pairs that span a face of their Minkowski difference (D. Gregorius, "The separating
// axis test between convex polyhedra", GDC 2013), then ONE contact where two edges cross, or the clipped face manifold
// (Sutherland-Hodgman; E. Catto, "Contact manifolds", GDC 2007) reduced to at most four points.
//
// ARITHMETIC: the operations of oracle/shf_oracle.c (poly_from_box / poly_from_hull, poly_face_query, poly_edge_query,
// convex_manifold) in the same order on the same values.  What differs is who executes them: the G lanes of an env share the
// axes -- a lane per face, a lane per edge of A -- and agree on the best by shuffles (largest separation, lowest index on a tie:
// the order-independent form of the oracle's ascending loops); the manifold itself is computed by every lane alike.
#pragma once

static __device__ const unsigned char SHF_BOX_LOOP[6][4] = {{4, 6, 7, 5}, {0, 1, 3, 2}, {2, 3, 7, 6}, {0, 4, 5, 1}, {1, 5, 7, 3}, {0, 2, 6, 4}};
static __device__ const unsigned char SHF_BOX_EDGE[12][4] = {{0, 4, 3, 5}, {1, 5, 3, 4}, {2, 6, 2, 5}, {3, 7, 2, 4}, {0, 2, 1, 5}, {1, 3, 1, 4},
                                                             {4, 6, 0, 5}, {5, 7, 0, 4}, {0, 1, 1, 3}, {2, 3, 1, 2}, {4, 5, 0, 3}, {6, 7, 0, 2}};

// A convex polytope about O in world axes: hull `h` (global memory) on the pose (R, p), or -- h == nullptr -- the box with axes =
// columns of R, centre p, half extents hx.  Vertices, planes and topology are produced on demand (nothing is stored per env).
struct PolyDev { const ShfHull* h; float R[9], p[3], hx[3]; };
__device__ inline int poly_nv(const PolyDev& P) { return P.h ? P.h->nv : 8; }
__device__ inline int poly_nf(const PolyDev& P) { return P.h ? P.h->nf : 6; }
__device__ inline int poly_ne(const PolyDev& P) { return P.h ? P.h->ne : 12; }
__device__ inline void poly_vert(const PolyDev& P, int i, float* v) {
  float lv[3];
  if (P.h) { lv[0] = P.h->vert[i][0]; lv[1] = P.h->vert[i][1]; lv[2] = P.h->vert[i][2]; }
  else { lv[0] = (i & 4) ? P.hx[0] : -P.hx[0]; lv[1] = (i & 2) ? P.hx[1] : -P.hx[1]; lv[2] = (i & 1) ? P.hx[2] : -P.hx[2]; }
  mv3(P.R, lv, v);
#pragma unroll
  for (int k = 0; k < 3; k++) v[k] += P.p[k];
}
__device__ inline void poly_plane(const PolyDev& P, int f, float* pl) {
  if (P.h) {
    const float ln[3] = {P.h->plane[f][0], P.h->plane[f][1], P.h->plane[f][2]};
    mv3(P.R, ln, pl);
    pl[3] = P.h->plane[f][3] + dot3(pl, P.p);
  } else {
    const int ax = f >> 1;
    const float sg = (f & 1) ? -1.0f : 1.0f;
    pl[0] = sg * P.R[ax]; pl[1] = sg * P.R[3 + ax]; pl[2] = sg * P.R[6 + ax];
    pl[3] = dot3(pl, P.p) + P.hx[ax];
  }
}
__device__ inline void poly_edge(const PolyDev& P, int e, int* rec) {
#pragma unroll
  for (int k = 0; k < 4; k++) rec[k] = P.h ? (int)P.h->edge[e][k] : (int)SHF_BOX_EDGE[e][k];
}
__device__ inline int poly_face_count(const PolyDev& P, int f) { return P.h ? (int)P.h->face_count[f] : 4; }
__device__ inline int poly_loop(const PolyDev& P, int f, int k) { return P.h ? (int)P.h->face_loop[P.h->face_start[f] + k] : (int)SHF_BOX_LOOP[f][k]; }
__device__ inline void poly_centroid(const PolyDev& P, float* c) {
  if (P.h) {
    const float lc[3] = {P.h->centroid[0], P.h->centroid[1], P.h->centroid[2]};
    mv3(P.R, lc, c);
#pragma unroll
    for (int k = 0; k < 3; k++) c[k] += P.p[k];
  } else {
#pragma unroll
    for (int k = 0; k < 3; k++) c[k] = P.p[k];
  }
}

// the group's best (largest value, lowest index on a tie) to every lane of the group
template <int G>
__device__ inline void hull_group_best(float& s, int& idx) {
#pragma unroll
  for (int sh = 1; sh < G; sh <<= 1) {
    const float os = __shfl_xor(s, sh, 64);
    const int oi = __shfl_xor(idx, sh, 64);
    if (os > s || (os == s && oi < idx)) { s = os; idx = oi; }
  }
}
// face query (oracle: poly_face_query): a lane per face of P, every vertex of Q against it
template <int G>
__device__ inline float hull_face_query(const PolyDev& P, const PolyDev& Q, int l, int* face) {
  float best = -1e30f;
  int bf = 0x7fffffff;
  const int nf = poly_nf(P), nv = poly_nv(Q);
  for (int f = l; f < nf; f += G) {
    float pl[4], smin = 1e30f;
    poly_plane(P, f, pl);
    for (int i = 0; i < nv; i++) {
      float v[3];
      poly_vert(Q, i, v);
      smin = rminf(smin, dot3(pl, v));
    }
    const float s = smin - pl[3];
    if (s > best) { best = s; bf = f; }
  }
  hull_group_best<G>(best, bf);
  *face = bf == 0x7fffffff ? 0 : bf;
  return best;
}
// one edge pair (oracle: the body of poly_edge_query's loops): true when the pair spans a face of the Minkowski difference and the
// edges are not parallel; then `n` = the unit axis, away from A, and *s the separation along it
__device__ inline bool hull_edge_pair(const float* pa, const float* da, const float* a, const float* b, const float* bxa, const float* ca,
                                      const float* pb, const float* qb, const float* cp, const float* dp, float* n, float* s) {
  const float c[3] = {-cp[0], -cp[1], -cp[2]}, d[3] = {-dp[0], -dp[1], -dp[2]};
  float dxc[3];
  cross3(d, c, dxc);
  const float cba = dot3(c, bxa), dba = dot3(d, bxa), adc = dot3(a, dxc), bdc = dot3(b, dxc);
  if (!(cba * dba < 0.0f && adc * bdc < 0.0f && cba * bdc > 0.0f)) return false;
  const float db[3] = {qb[0] - pb[0], qb[1] - pb[1], qb[2] - pb[2]};
  cross3(da, db, n);
  const float l2 = dot3(n, n);
  if (!(l2 > 1e-6f * (dot3(da, da) * dot3(db, db)))) return false;
  const float il = rsqrt_spec(l2);
#pragma unroll
  for (int k = 0; k < 3; k++) n[k] *= il;
  const float out[3] = {pa[0] - ca[0], pa[1] - ca[1], pa[2] - ca[2]};
  if (dot3(n, out) < 0.0f) { n[0] = -n[0]; n[1] = -n[1]; n[2] = -n[2]; }
  const float w[3] = {pb[0] - pa[0], pb[1] - pa[1], pb[2] - pa[2]};
  *s = dot3(n, w);
  return true;
}
struct HullEdgeA { float pa[3], qa[3], da[3], a[4], b[4], bxa[3]; };
__device__ inline void hull_edge_a(const PolyDev& A, int i, HullEdgeA& E) {
  int rec[4];
  poly_edge(A, i, rec);
  poly_vert(A, rec[0], E.pa);
  poly_vert(A, rec[1], E.qa);
  poly_plane(A, rec[2], E.a);
  poly_plane(A, rec[3], E.b);
#pragma unroll
  for (int k = 0; k < 3; k++) E.da[k] = E.qa[k] - E.pa[k];
  cross3(E.b, E.a, E.bxa);
}
__device__ inline bool hull_edge_ab(const PolyDev& B, const HullEdgeA& E, const float* ca, int j, float* n, float* s) {
  int rec[4];
  float pb[3], qb[3], cp[4], dp[4];
  poly_edge(B, j, rec);
  poly_vert(B, rec[0], pb);
  poly_vert(B, rec[1], qb);
  poly_plane(B, rec[2], cp);
  poly_plane(B, rec[3], dp);
  return hull_edge_pair(E.pa, E.da, E.a, E.b, E.bxa, ca, pb, qb, cp, dp, n, s);
}
// edge query (oracle: poly_edge_query): a lane per edge of A, every edge of B against it; pair number i * neB + j
template <int G>
__device__ inline float hull_edge_query(const PolyDev& A, const PolyDev& B, int l, int* ea, int* eb, float* axis) {
  float best = -1e30f;
  int bq = 0x7fffffff;
  const int nea = poly_ne(A), neb = poly_ne(B);
  float ca[3];
  poly_centroid(A, ca);
  for (int i = l; i < nea; i += G) {
    HullEdgeA E;
    hull_edge_a(A, i, E);
    for (int j = 0; j < neb; j++) {
      float n[3], s;
      if (!hull_edge_ab(B, E, ca, j, n, &s)) continue;
      if (s > best) { best = s; bq = i * neb + j; }
    }
  }
  hull_group_best<G>(best, bq);
  *ea = -1; *eb = -1;
  axis[0] = 0.0f; axis[1] = 0.0f; axis[2] = 0.0f;
  if (bq != 0x7fffffff) {
    *ea = bq / neb; *eb = bq - (bq / neb) * neb;
    HullEdgeA E;
    float s;
    hull_edge_a(A, *ea, E);
    hull_edge_ab(B, E, ca, *eb, axis, &s);        // (the winning lane's own arithmetic again: the same values)
  }
  return best;
}

#define HULL_CLIP (2 * SHF_HULL_MAX_FACE_VERTS)
// Contacts of polytope A (receives +f: the normal points from B towards A) with polytope B: up to four points r with their gaps phi
// (< offset) and the common normal n; returns their number -- the same in every lane of the group (oracle: convex_manifold).
// Not inlined: its arrays (the clipped loop) live in scratch, and the step kernels keep their own register allocation.
template <int G>
__device__ __noinline__ int convex_manifold_dev(const PolyDev& A, const PolyDev& B, float offset, int l, float (*r)[3], float* n, float* phi) {
  int fa, fb, ea, eb;
  float axis[3];
  // (a face that separates the pair by the contact offset or more ends the test: the oracle computes all three queries and returns
  // nothing in that case too -- most live pairs of the broad phase end here, before the edge pairs, the expensive query)
  const float sb = hull_face_query<G>(B, A, l, &fb);
  if (!(sb < offset)) return 0;
  const float sa = hull_face_query<G>(A, B, l, &fa);
  if (!(sa < offset)) return 0;
  const float se = hull_edge_query<G>(A, B, l, &ea, &eb, axis);
  const float sface = rmaxf(sa, sb);
  if (!(sface < offset) || !(se < offset)) return 0;
  for (int pass = 0; pass < 3; pass++) {
    if (ea >= 0 && ((pass == 2 && se >= sface - fmaf(0.05f, fabsf(sface), 5e-4f)) || (pass == 0 && se > sface + fmaf(0.05f, fabsf(sface), 5e-4f)))) {
      int ra[4], rb[4];
      float p1[3], q1[3], p2[3], q2[3], c1[3], c2[3];
      poly_edge(A, ea, ra);
      poly_edge(B, eb, rb);
      poly_vert(A, ra[0], p1); poly_vert(A, ra[1], q1); poly_vert(B, rb[0], p2); poly_vert(B, rb[1], q2);
      segment_closest(p1, q1, p2, q2, c1, c2);
      // (the edges really cross there: closest points inside both segments, `se` apart along the axis -- oracle: the same test)
      const float ex[3] = {(c2[0] - c1[0]) - se * axis[0], (c2[1] - c1[1]) - se * axis[1], (c2[2] - c1[2]) - se * axis[2]};
      if (dot3(ex, ex) <= 1e-8f) {
#pragma unroll
        for (int k = 0; k < 3; k++) { r[0][k] = 0.5f * (c1[k] + c2[k]); n[k] = -axis[k]; }
        phi[0] = se;
        return 1;
      }
    }
    if (pass == 2) return 0;
    const bool refa = (sa > sb + fmaf(0.05f, fabsf(sb), 5e-4f)) != (pass == 1);
    if (pass == 1) {
      const float so = refa ? sa : sb, sp = refa ? sb : sa;
      if (!(so >= sp - fmaf(0.05f, fabsf(sp), 5e-4f))) continue;
    }
    const PolyDev& Pr = refa ? A : B;
    const PolyDev& Pi = refa ? B : A;
    const int fr = refa ? fa : fb;
    float pr[4];
    poly_plane(Pr, fr, pr);
    int fi = 0;
    float dmin = 1e30f;
    const int nfi = poly_nf(Pi);
    for (int f = 0; f < nfi; f++) {
      float pl[4];
      poly_plane(Pi, f, pl);
      const float d = dot3(pl, pr);
      if (d < dmin) { dmin = d; fi = f; }
    }
    float poly[2][HULL_CLIP][3];
    int cur = 0, m = poly_face_count(Pi, fi);
    for (int i = 0; i < m; i++) poly_vert(Pi, poly_loop(Pi, fi, i), poly[0][i]);
    const int mr = poly_face_count(Pr, fr);
    for (int e = 0; e < mr && m > 0; e++) {
      float v0[3], v1[3], side[3];
      poly_vert(Pr, poly_loop(Pr, fr, e), v0);
      poly_vert(Pr, poly_loop(Pr, fr, e + 1 == mr ? 0 : e + 1), v1);
      const float ed[3] = {v1[0] - v0[0], v1[1] - v0[1], v1[2] - v0[2]};
      cross3(ed, pr, side);
      int mo = 0;
      for (int i = 0; i < m; i++) {
        const float* pc = poly[cur][i];
        const float* pn = poly[cur][i + 1 == m ? 0 : i + 1];
        const float wc[3] = {pc[0] - v0[0], pc[1] - v0[1], pc[2] - v0[2]}, wn[3] = {pn[0] - v0[0], pn[1] - v0[1], pn[2] - v0[2]};
        const float dc = dot3(side, wc), dn = dot3(side, wn);
        const bool inc = dc <= 0.0f, inn = dn <= 0.0f;
        if (inc && mo < HULL_CLIP) {
#pragma unroll
          for (int k = 0; k < 3; k++) poly[1 - cur][mo][k] = pc[k];
          mo++;
        }
        if (inc != inn && mo < HULL_CLIP) {
          const float t = dc / (dc - dn);
#pragma unroll
          for (int k = 0; k < 3; k++) poly[1 - cur][mo][k] = fmaf(t, pn[k] - pc[k], pc[k]);
          mo++;
        }
      }
      cur = 1 - cur; m = mo;
    }
    float cp[HULL_CLIP][3], cd[HULL_CLIP];
    int nc = 0;
    for (int i = 0; i < m; i++) {
      const float d = dot3(pr, poly[cur][i]) - pr[3];
      if (!(d < offset)) continue;
#pragma unroll
      for (int k = 0; k < 3; k++) cp[nc][k] = poly[cur][i][k];
      cd[nc] = d; nc++;
    }
    if (nc == 0) continue;
    int keep[4] = {0, 1, 2, 3}, nk = nc;
    if (nc > 4) {
      int i0 = 0, i1 = -1, i2 = -1, i3 = -1;
      for (int i = 1; i < nc; i++) if (cd[i] < cd[i0]) i0 = i;
      float best = -1.0f;
      for (int i = 0; i < nc; i++) {
        if (i == i0) continue;
        const float w[3] = {cp[i][0] - cp[i0][0], cp[i][1] - cp[i0][1], cp[i][2] - cp[i0][2]};
        const float d2 = dot3(w, w);
        if (d2 > best) { best = d2; i1 = i; }
      }
      const float e1[3] = {cp[i1][0] - cp[i0][0], cp[i1][1] - cp[i0][1], cp[i1][2] - cp[i0][2]};
      float area[HULL_CLIP];
      for (int i = 0; i < nc; i++) {
        const float w[3] = {cp[i][0] - cp[i0][0], cp[i][1] - cp[i0][1], cp[i][2] - cp[i0][2]};
        float t[3];
        cross3(e1, w, t);
        area[i] = dot3(pr, t);
      }
      best = -1.0f;
      for (int i = 0; i < nc; i++) {
        if (i == i0 || i == i1) continue;
        if (fabsf(area[i]) > best) { best = fabsf(area[i]); i2 = i; }
      }
      best = 0.0f;
      for (int i = 0; i < nc; i++) {
        if (i == i0 || i == i1 || i == i2) continue;
        const float a = area[i2] > 0.0f ? -area[i] : area[i];
        if (a > best) { best = a; i3 = i; }
      }
      nk = 0;
      for (int i = 0; i < nc; i++) if (i == i0 || i == i1 || i == i2 || i == i3) keep[nk++] = i;
    }
    for (int q = 0; q < nk; q++) {
      const int i = keep[q];
#pragma unroll
      for (int k = 0; k < 3; k++) r[q][k] = fmaf(-0.5f * cd[i], pr[k], cp[i][k]);
      phi[q] = cd[i];
    }
#pragma unroll
    for (int k = 0; k < 3; k++) n[k] = refa ? -pr[k] : pr[k];
    return nk;
  }
  return 0;
}
