// shf_hard.h -- the velocity-level contact solve (ShfSimParams.solver == SHF_SOLVER_PGS) for the body-per-lane kernels: any
// articulation, with box actors, self-collision and link contacts.  What gymapi.SimParams.physx configures in the reference
// (shifu/configs/env_config.py:50-58) and gym.simulate runs under (shifu/units/robot.py:69, shifu/gym/isaac_gym.py:140).
// The chain-mapped A1 kernels have their own lane mapping of the same solve (shf_chain_hard.h); the pieces both share --
// the constraint record, the contact frame, the sweep's arithmetic -- are defined here.
//
// ARITHMETIC: the operations of oracle/shf_oracle.c (substep with hard = 1, hard_solve, hc_apply) in the same order on the
// same values; tests/test_gpu_parity.py holds the kernels to the oracle bit for bit.
//
// Per env (32 lanes; two envs per wavefront), after the FREE articulated-body solve of substep():
//   records   body lane b -> S, U, 1/D and the velocity rate Dl of its body in its exchange slot; the root's / the free boxes'
//             LDL^T factors in theirs
//   gather    every active candidate slot of the sub-step (sample points, self-contacts, box corners / edges, rounded
//             shapes, link contacts: slot_eval recorded location, normal, gap, friction) in the oracle's candidate order;
//             the K <= 8 with the smallest gap become constraint records
//   columns   W = J M^-1 J^T, a lane per (contact, axis of its frame): impulse up the tree, root solve, down to every
//             constrained body; a free box answers with its own inverse inertia
//   sweeps    projected Gauss-Seidel, a lane per contact (hard_sweeps)
//   apply     the impulses (after the position iterations / after the velocity iterations) through the tree, level by level
#pragma once

#define HCK 8           /* constraints one env's solve holds (ShfSimParams.max_contacts <= HCK) */
#define HC_STRIDE 24    /* r[3] n[3] phi mu body rep p[3] bodyb repb pv0 t1[3] t2[3] pv1 pv2 */
#define HC_R 0
#define HC_N 3
#define HC_PHI 6
#define HC_MU 7
#define HC_BODY 8       /* solver body that receives +p: a moving body of the articulation, nb + k for free box k */
#define HC_REP 9        /* reported body the force is logged on */
#define HC_P 10         /* impulse after the position iterations, world axes */
#define HC_BODYB 13     /* solver body that receives -p, or -1 (terrain, fixed box) */
#define HC_REPB 14
#define HC_T1 16        /* tangents of the contact frame */
#define HC_T2 19
#define HC_PV0 15       /* impulse after the velocity iterations, world axes (three spare words) */
#define HC_PV1 22
#define HC_PV2 23
#define HG_LEV 8        /* deepest tree the generic solve walks (ShfModel.nlevels) */
// LDS of the generic solve inside the env's contact-slot region (PT_STRIDE-word slots; `nbase` = the scene's own slots: sample
// points, box slots, self slots and, with link contacts, 2 x SHF_MAX_LINK_CONTACTS link slots):
//   W (HCK x HCK blocks of 9 words = 48 slots)  in the place of slots [0, 48): every slot is dead once the constraints are gathered
//   constraint records (HCK x HC_STRIDE words = 16 slots)  from slot hc0 >= 48 on: the link slots' pair-record half (the compliant
//     pair laws' second records, unused here) when there are link contacts, else behind the scene's slots
//   the gather's candidate list: words PT_CT, PT_BN of slot k < hc0 (written as zeros by the evaluation, read by nobody here)
#define HARD_W_SLOTS ((HCK * HCK * 9) / PT_STRIDE)
#define HARD_HC_SLOTS ((HCK * HC_STRIDE) / PT_STRIDE)
static_assert((HCK * HCK * 9) % PT_STRIDE == 0 && (HCK * HC_STRIDE) % PT_STRIDE == 0, "the solve's LDS in whole contact slots");
static_assert(HARD_HC_SLOTS == SHF_MAX_LINK_CONTACTS, "the constraint records fill the link slots' pair-record half");
__host__ __device__ inline int hard_hc_slot0(int nbase, bool link) {
  const int want = link ? nbase - SHF_MAX_LINK_CONTACTS : nbase;
  return want < HARD_W_SLOTS ? HARD_W_SLOTS : want;
}
__host__ __device__ inline int hard_total_slots(int nbase, bool link) {   // contact slots of an env under the solve
  const int t = hard_hc_slot0(nbase, link) + HARD_HC_SLOTS;
  return t > nbase ? t : nbase;
}
// body record in the body's exchange slot (XCH_STRIDE words)
#define HB_S 0
#define HB_U 6
#define HB_INVD 12
#define HB_DL 14        /* velocity rate; after the owners' setup: impulse-pass words of set 1 */
#define HB_PC 20        /* impulse-pass words of set 0 */
#define HB_FDL 21       /* root / box record: LDL^T factors in words 0..20, then the velocity rate */

DEV float hard_readlane(float x, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), lane)); }
// point velocity of the spatial velocity v6 (about O) at r
DEV void hard_point(const float* v6, const float* r, float* o) {
  float t[3];
  cross3(v6, r, t);
#pragma unroll
  for (int k = 0; k < 3; k++) o[k] = v6[3 + k] + t[k];
}
// the tangents of a contact frame: t1 = (x or y) cross n normalised -- x unless n leans on it; t2 = n cross t1 (oracle: hard_solve)
DEV void hard_frame(const float* n, float* t1, float* t2) {
  float t[3];
  if (fabsf(n[0]) < 0.7f) { t[0] = 0.0f; t[1] = -n[2]; t[2] = n[1]; }
  else { t[0] = n[2]; t[1] = 0.0f; t[2] = -n[0]; }
  const float il = rsqrt_spec(dot3(t, t));
  float a[3];
#pragma unroll
  for (int k = 0; k < 3; k++) a[k] = t[k] * il;
  float b[3];
  cross3(n, a, b);
#pragma unroll
  for (int k = 0; k < 3; k++) { t1[k] = a[k]; t2[k] = b[k]; }
}
// LDL^T factors of a 6x6 in LDS: L below the diagonal row by row (15), then 1/D (6)
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

// Per-lane state of the solve: the owner lane of contact c; velocity u and impulse p in the contact frame (n, t1, t2)
struct HardOwner {
  float mu, u[3], p[3], tgt, tgt_v, w10, w20, iwnn, Ti[3], rt;
  float gap, rfloor;      // SHF_SOLVER_TGS: the contact's gap as the sub-iterations advance it; the restitution floor of its target (or -1e30)
};
// SHF_SOLVER_TGS (physx.solver_type = 1, shifu/configs/env_config.py:50; oracle: hard_solve "tgs"): the position iterations as
// sub-iterations of h = dt / npos -- targets from the gaps as they stand (horizon h), gaps advanced by the normal velocities each
// sweep leaves, the poses moved by the mean of the impulses after the sweeps; velocity iterations against what is left (horizon dt).
struct HardTgs { bool on; float hN, ihN, inv_n, idt, erp, vdep; };
DEV HardTgs hard_tgs(const ShfSimParams& sp, int npos) {
  HardTgs T;
  T.on = sp.solver == SHF_SOLVER_TGS && npos > 0;
  const float dt = sp.dt, idt = 1.0f / dt;
  T.hN = T.on ? dt / (float)npos : dt;
  T.ihN = T.on ? (float)npos * idt : idt;
  T.inv_n = T.on ? 1.0f / (float)npos : 1.0f;
  T.idt = idt;
  T.erp = sp.erp > 0.0f ? sp.erp : 0.2f;
  T.vdep = sp.max_depen_vel;
  return T;
}
DEV float hard_tgs_target_pos(const HardTgs& T, const HardOwner& O) {
  return rmaxf(O.gap >= 0.0f ? -(O.gap * T.ihN) : rminf(T.erp * -(O.gap) * T.ihN, T.vdep), O.rfloor);
}
DEV float hard_tgs_target_vel(const HardTgs& T, const HardOwner& O) { return rmaxf(O.gap >= 0.0f ? -(O.gap * T.idt) : 0.0f, O.rfloor); }
// owner lane: targets from the gap, the regularised diagonal block of W (written back) and its inverses (oracle: hard_solve,
// "free velocities, targets, block inverses").  vs: start-of-step relative velocity, vf: under v_free (world axes).
DEV void hard_owner_setup(const ShfSimParams& sp, const float* h, const float* vs, const float* vf, float* Wd, bool own, float idt, HardOwner& O) {
  const float n[3] = {h[HC_N], h[HC_N + 1], h[HC_N + 2]};
#pragma unroll
  for (int k = 0; k < 3; k++) O.p[k] = 0.0f;
  O.mu = h[HC_MU];
  O.u[0] = dot3(n, vf); O.u[1] = dot3(h + HC_T1, vf); O.u[2] = dot3(h + HC_T2, vf);
  const float phi = h[HC_PHI];
  const float erp = sp.erp > 0.0f ? sp.erp : 0.2f;
  float tg = phi >= 0.0f ? -(phi * idt) : rminf(erp * -(phi) * idt, sp.max_depen_vel);
  float tv = phi >= 0.0f ? tg : 0.0f;
  const float vn0 = dot3(n, vs);
  if (sp.restitution > 0.0f && vn0 < -sp.bounce_threshold) { tg = rmaxf(tg, -(sp.restitution * vn0)); tv = rmaxf(tv, -(sp.restitution * vn0)); }
  O.tgt = tg; O.tgt_v = tv;
  O.gap = phi;
  O.rfloor = (sp.restitution > 0.0f && vn0 < -sp.bounce_threshold) ? -(sp.restitution * vn0) : -1e30f;
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
// Projected Gauss-Seidel at 32 lanes per env (two envs per wavefront): lane c < K owns contact c.  Position iterations,
// then velocity iterations; after each phase the owner writes its impulse in world axes into the record (HC_P / HC_PV).
// W: blocks (i, j) at W + (j * HCK + i) * 9.  oracle: hard_solve, "sweeps".
DEV void hard_sweeps(HardOwner& O, float* hc, const float* W, int l, int K, int npos, int nvel, const ShfSimParams& sp) {
  const HardTgs TG = hard_tgs(sp, npos);
  float psum[3] = {0.0f, 0.0f, 0.0f};
  const int lane0 = (int)(threadIdx.x & 63u) - l;
  const bool own = l < K;
  int Kw = 0;      // the larger constraint count of the wavefront's envs (wave-uniform)
#pragma unroll
  for (int c = 0; c < HCK; c++)
    if (__ballot(c < K) != 0ull) Kw = c + 1;
  const bool hi = lane0 != 0;
  const float* Wcol = W + (own ? l : 0) * 9;     // block (l, c) sits at Wcol + c * HCK * 9
#pragma unroll 1
  for (int phase = 0; phase < 2; phase++) {
    const int sweeps = phase == 0 ? npos : nvel;
    float tg = phase == 0 ? O.tgt : O.tgt_v;
    if (TG.on && phase == 1) tg = hard_tgs_target_vel(TG, O);
#pragma unroll 1
    for (int it = 0; it < sweeps; it++) {
      if (TG.on && phase == 0) tg = hard_tgs_target_pos(TG, O);
#pragma unroll 1
      for (int c = 0; c < Kw; c++) {
        // an open, unloaded contact whose normal velocity keeps it open asks for nothing (oracle: the same test): when that is
        // so for contact c of every env of the wavefront the visit is skipped
        const bool act = !(O.p[0] == 0.0f && O.p[1] == 0.0f && O.p[2] == 0.0f && !(O.u[0] < tg));
        if (__ballot(l == c && c < K && act) == 0ull) continue;
        // this lane's block of column c: in flight while the update is computed
        float Wb[9];
#pragma unroll
        for (int k = 0; k < 9; k++) Wb[k] = Wcol[c * HCK * 9 + k];
        // every owner lane computes its own update; lane c's is the one that counts
        const float pn0 = O.p[0];
        const float pn = rmaxf(fmaf(-(O.u[0] - tg), O.iwnn, pn0), 0.0f);
        const float dn = pn - pn0;
        const float ut1 = fmaf(dn, O.w10, O.u[1]), ut2 = fmaf(dn, O.w20, O.u[2]);
        float ps1 = O.p[1] - fmaf(O.Ti[1], ut2, O.Ti[0] * ut1), ps2 = O.p[2] - fmaf(O.Ti[2], ut2, O.Ti[1] * ut1);
        const float lim = O.mu * pn, lim2 = lim * lim;
        const bool commit = l == c && c < K && act;
        if (commit && fmaf(ps2, ps2, ps1 * ps1) > lim2) {      // (only the committing lanes: the others' updates are discarded)
          ps1 = fmaf(-O.rt, ut1, O.p[1]); ps2 = fmaf(-O.rt, ut2, O.p[2]);
          const float nt2 = fmaf(ps2, ps2, ps1 * ps1);
          const float sc1 = nt2 > rmaxf(lim2, 1e-30f) ? lim * rsqrt_spec(nt2) : 1.0f;   // (1e-30: a subnormal |p_t|^2 over a zero cone would make 0 * inf)
          ps1 *= sc1; ps2 *= sc1;
        }
        float dp0 = commit ? dn : 0.0f, dp1 = commit ? ps1 - O.p[1] : 0.0f, dp2 = commit ? ps2 - O.p[2] : 0.0f;
        if (commit) { O.p[0] = pn; O.p[1] = ps1; O.p[2] = ps2; }
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
      }
      if (TG.on && phase == 0) {
        O.gap = fmaf(TG.hN, O.u[0], O.gap);
#pragma unroll
        for (int r = 0; r < 3; r++) psum[r] += O.p[r];
      }
    }
    if (phase == 1 && nvel == 0) break;
    if (own) {
      float* h = hc + l * HC_STRIDE;
      float pw[3];
      const bool mean = TG.on && phase == 0;
      const float q0 = mean ? psum[0] * TG.inv_n : O.p[0], q1 = mean ? psum[1] * TG.inv_n : O.p[1], q2 = mean ? psum[2] * TG.inv_n : O.p[2];
#pragma unroll
      for (int r = 0; r < 3; r++) pw[r] = fmaf(q2, h[HC_T2 + r], fmaf(q1, h[HC_T1 + r], q0 * h[HC_N + r]));
      if (phase == 0) { h[HC_P] = pw[0]; h[HC_P + 1] = pw[1]; h[HC_P + 2] = pw[2]; }
      // (without velocity iterations the second set is not used, but the impulse passes carry it along: defined values, not stale LDS)
      if (phase == 1 || nvel == 0) { h[HC_PV0] = pw[0]; h[HC_PV1] = pw[1]; h[HC_PV2] = pw[2]; }
    }
  }
}

// ============================================================ the generic (body-per-lane) form ============================
// Response to the impulse e at r on solver body `src`: an articulation body (walk up its ancestors, root solve) or a free
// box (its inverse inertia).  oracle: hc_impulse / hc_box_impulse.
struct HgResp { int src, slev, spath[HG_LEV + 1]; float ub[HG_LEV + 1], dv0[6]; };
DEV void hg_impulse(const ShfModel* m, const EnvLds& L, int nb, int src, const float* r, const float* e, HgResp& q) {
  float p6[6], t[3];
  cross3(r, e, t);
#pragma unroll
  for (int k = 0; k < 3; k++) { p6[k] = -t[k]; p6[3 + k] = -e[k]; }
  q.src = src; q.slev = 0;
  if (src >= nb) { root_factors_apply(L.xch + src * XCH_STRIDE, p6, q.dv0); return; }
  q.slev = m->level[src];
  int b = src;
#pragma unroll
  for (int lev = HG_LEV; lev >= 1; lev--) {
    q.ub[lev] = 0.0f; q.spath[lev] = -1;
    if (lev <= q.slev) {
      const float* rec = L.xch + b * XCH_STRIDE;
      float sp = rec[HB_S] * p6[0];
#pragma unroll
      for (int j = 1; j < 6; j++) sp = fmaf(rec[HB_S + j], p6[j], sp);
      q.ub[lev] = -sp;
      q.spath[lev] = b;
      const float tt = q.ub[lev] * rec[HB_INVD];
#pragma unroll
      for (int j = 0; j < 6; j++) p6[j] = fmaf(rec[HB_U + j], tt, p6[j]);
      b = m->dyn[m->parent[b]];
    }
  }
  if (m->fixed_base) {
#pragma unroll
    for (int k = 0; k < 6; k++) q.dv0[k] = 0.0f;
  } else {
    root_factors_apply(L.xch, p6, q.dv0);
  }
}
// velocity change of the point r of solver body `tgt` under that response (oracle: hc_cross)
DEV void hg_velocity(const ShfModel* m, const EnvLds& L, int nb, const HgResp& q, int tgt, const float* r, float* vel) {
  vel[0] = 0.0f; vel[1] = 0.0f; vel[2] = 0.0f;
  if (tgt < 0 || q.src < 0) return;
  if (q.src >= nb) { if (tgt == q.src) hard_point(q.dv0, r, vel); return; }
  if (tgt >= nb) return;
  const int lt = m->level[tgt];
  int path[HG_LEV + 1];
  int b = tgt;
#pragma unroll
  for (int lev = HG_LEV; lev >= 1; lev--) {
    path[lev] = -1;
    if (lev <= lt) { path[lev] = b; b = m->dyn[m->parent[b]]; }
  }
  float dv[6];
#pragma unroll
  for (int k = 0; k < 6; k++) dv[k] = q.dv0[k];
#pragma unroll
  for (int lev = 1; lev <= HG_LEV; lev++) {
    if (lev <= lt) {
      const float* rec = L.xch + path[lev] * XCH_STRIDE;
      const float ubk = (lev <= q.slev && q.spath[lev] == path[lev]) ? q.ub[lev] : 0.0f;
      float ua = rec[HB_U] * dv[0];
#pragma unroll
      for (int j = 1; j < 6; j++) ua = fmaf(rec[HB_U + j], dv[j], ua);
      const float dq = (ubk - ua) * rec[HB_INVD];
#pragma unroll
      for (int j = 0; j < 6; j++) dv[j] = fmaf(rec[HB_S + j], dq, dv[j]);
    }
  }
  hard_point(dv, r, vel);
}

// Candidate slots of one sub-step in the oracle's candidate order: evaluation slots of the sample points, self-contacts,
// the free boxes' corner / edge slots, rounded shapes x boxes, link contacts.  Element `idx` of that sequence -> its slot
// (nullptr: no such slot) and the constraint's bodies.
struct HgSeq {
  int nev, nself, nbx, T, nsph, nlink, self_slot0, link_slot0, P1, P2, P3, P4;
};
template <class SC>
DEV const float* hg_slot_raw(const ShfModel* m, const SceneDev* S, const EnvLds& L, const SlotLay& Q, const HgSeq& H, int idx,
                             int* ba, int* bb, int* ra, int* rb);
// ... with the root of a fixed-base articulation as what it is to the solve: immovable, like terrain or a fixed box (-1); a
// contact between two immovable things is no constraint (oracle: hc_offer)
template <class SC>
DEV const float* hg_slot(const ShfModel* m, const SceneDev* S, const EnvLds& L, const SlotLay& Q, const HgSeq& H, int idx,
                         int* ba, int* bb, int* ra, int* rb) {
  const float* o = hg_slot_raw<SC>(m, S, L, Q, H, idx, ba, bb, ra, rb);
  if (o == nullptr) return nullptr;
  if (m->fixed_base && *ba == 0) *ba = -1;
  if (m->fixed_base && *bb == 0) *bb = -1;
  return (*ba < 0 && *bb < 0) ? nullptr : o;
}
// (SC: a compile-time scene makes the box count a constant -- the decoding's divisions become shifts and multiplications)
template <class SC>
DEV const float* hg_slot_raw(const ShfModel* m, const SceneDev* S, const EnvLds& L, const SlotLay& Q, const HgSeq& H, int idx,
                             int* ba, int* bb, int* ra, int* rb) {
  const int nb = m->nb;
  const int HT = SC::NBX > 0 ? 1 + SC::NBX : H.T, Hnbx = SC::NBX > 0 ? SC::NBX : H.nbx;
  *bb = -1; *rb = -1;
  if (idx < H.nev) {
    const int i = m->neval > 0 ? m->pt_eval[idx] : idx;
    if (i < 0) return nullptr;
    *ra = m->pt_body[i]; *ba = m->dyn[*ra];
    return L.pt + i * PT_STRIDE;
  }
  if (idx < H.P1) {
    const float* o = L.pt + (H.self_slot0 + (idx - H.nev)) * PT_STRIDE;
    const int pr = (int)o[PT_ON] - 1;
    if (pr < 0) return nullptr;
    *ra = m->cap_body[m->pair_a[pr]]; *rb = m->cap_body[m->pair_b[pr]];
    *ba = m->dyn[*ra]; *bb = m->dyn[*rb];
    return o;
  }
  if (idx < H.P2) {
    const int j = idx - H.P1, kd = j / (8 * HT), c = (j / HT) % 8, tg = j % HT;
    if (!box_is_dynamic(S->box[kd])) return nullptr;
    *ba = nb + kd; *ra = nb + kd;
    return L.pt + corner_slot(Q, kd, c, tg) * PT_STRIDE;
  }
  if (idx < H.P3) {
    const int j = idx - H.P2, si = j / Hnbx, kd = j % Hnbx;
    if (!box_is_dynamic(S->box[kd])) return nullptr;
    *ra = m->sph_body[si]; *ba = m->dyn[*ra];
    *bb = nb + kd; *rb = nb + kd;
    return L.pt + sphere_slot(Q, si, kd) * PT_STRIDE;
  }
  const float* o = L.pt + (H.link_slot0 + (idx - H.P3)) * PT_STRIDE;
  if (o[PT_ON] == 0.0f) return nullptr;
  const int body = link_code_body(o[PT_ON]), box = link_code_box(o[PT_ON]);
  *ra = body; *ba = m->dyn[body];
  if (box_is_dynamic(S->box[box])) { *bb = nb + box; *rb = nb + box; }
  return o;
}

// gather: the K <= kmax candidates with the smallest gap (ties: candidate order), in candidate order, as constraint records.
// Returns K.  oracle: hc_offer / hc_finish.
template <int G, class SC>
DEV int hg_gather(const StepCtx& C, const EnvLds& L, const SlotLay& Q, const HgSeq& H, int l, float* hc, int kmax, int LISTMAX) {
  const ShfModel* m = C.m;
  const SceneDev* S = C.scene;
  const int lane0 = (int)(threadIdx.x & 63u) - l;
  const unsigned long long gmask = G >= 64 ? ~0ull : ((1ull << (G & 63)) - 1ull);
  // pass 1: how many candidates -- and their (gap, candidate number) pairs, in candidate order, for the ranking below: entry k in
  // the words PT_CT, PT_BN of slot k (k < LISTMAX = the first slot of the constraint records)
  float* scratch = L.pt + PT_CT;
  constexpr int LS = PT_STRIDE;
  int total = 0;
  for (int base = 0; base < H.P4; base += G) {
    const int idx = base + l;
    int ba, bb, ra, rb;
    const float* o = idx < H.P4 ? hg_slot<SC>(m, S, L, Q, H, idx, &ba, &bb, &ra, &rb) : nullptr;
    const bool on = o != nullptr && o[PT_ON] != 0.0f;
    const unsigned long long mask = (__ballot(on) >> lane0) & gmask;
    if (on) {
      const int k = total + __builtin_popcountll(mask & ((1ull << l) - 1ull));
      if (k < LISTMAX) { scratch[LS * k] = o[PT_F]; scratch[LS * k + 1] = __int_as_float(idx); }
    }
    total += __builtin_popcountll(mask);
  }
  contact_hist_count(C, l, total);
  const bool overflow = total > kmax;
  if (overflow && l == 0 && C.dropped) *C.dropped += total - kmax;
  GROUP_SYNC();
  auto record = [&](int k, const float* o, int ba, int bb, int ra, int rb) {
    float* h = hc + k * HC_STRIDE;
#pragma unroll
    for (int j = 0; j < 3; j++) { h[HC_R + j] = o[PT_R + j]; h[HC_N + j] = o[PT_N + j]; }
    h[HC_PHI] = o[PT_F]; h[HC_MU] = o[PT_F + 1];
    h[HC_BODY] = __int_as_float(ba); h[HC_REP] = __int_as_float(ra);
    h[HC_BODYB] = __int_as_float(bb); h[HC_REPB] = __int_as_float(rb);
    const float n[3] = {o[PT_N], o[PT_N + 1], o[PT_N + 2]};
    hard_frame(n, h + HC_T1, h + HC_T2);
  };
  // pass 2: the selected ones (all of them, or those with fewer than kmax candidates ahead in (gap, order)), numbered in order
  int count = 0;
  if (__ballot(total > LISTMAX) == 0ull) {
    // ... walking the list: a lane per candidate (one round for up to G of them, instead of one per G slots)
    for (int base = 0; __ballot(base < total) != 0ull; base += G) {
      const int k0 = base + l;
      const bool cand = k0 < total;
      const float ph = cand ? scratch[LS * k0] : 0.0f;
      const int idx = cand ? __float_as_int(scratch[LS * k0 + 1]) : 0;
      bool sel = cand;
      if (overflow && cand) {
        int rank = 0;
        for (int j = 0; j < total; j++) {
          const float pj = scratch[LS * j];
          const int ij = __float_as_int(scratch[LS * j + 1]);
          rank += (pj < ph || (pj == ph && ij < idx)) ? 1 : 0;
        }
        sel = rank < kmax;
      }
      const unsigned long long mask = (__ballot(sel) >> lane0) & gmask;
      if (sel) {
        int ba = -1, bb = -1, ra = -1, rb = -1;
        const float* o = hg_slot<SC>(m, S, L, Q, H, idx, &ba, &bb, &ra, &rb);
        record(count + __builtin_popcountll(mask & ((1ull << l) - 1ull)), o, ba, bb, ra, rb);
      }
      count += __builtin_popcountll(mask);
    }
    GROUP_SYNC();       // (the slots' place becomes the response matrix)
    return count;
  }
  // (more candidates than the list holds in some env of the wavefront: slot by slot, ranking by a scan over every slot)
  for (int base = 0; base < H.P4; base += G) {
    const int idx = base + l;
    int ba = -1, bb = -1, ra = -1, rb = -1;
    const float* o = idx < H.P4 ? hg_slot<SC>(m, S, L, Q, H, idx, &ba, &bb, &ra, &rb) : nullptr;
    bool sel = o != nullptr && o[PT_ON] != 0.0f;
    if (__ballot(overflow && sel) != 0ull) {
      int rank = 0;
      const float ph = sel ? o[PT_F] : 0.0f;
      if (overflow && sel) {
        for (int j = 0; j < H.P4; j++) {
          int a0, a1, a2, a3;
          const float* oj = hg_slot<SC>(m, S, L, Q, H, j, &a0, &a1, &a2, &a3);
          if (oj == nullptr || oj[PT_ON] == 0.0f) continue;
          const float pj = oj[PT_F];
          rank += (pj < ph || (pj == ph && j < idx)) ? 1 : 0;
        }
        sel = rank < kmax;
      }
    }
    const unsigned long long mask = (__ballot(sel) >> lane0) & gmask;
    if (sel) record(count + __builtin_popcountll(mask & ((1ull << l) - 1ull)), o, ba, bb, ra, rb);
    count += __builtin_popcountll(mask);
  }
  return count;
}

// columns of W (oracle: hard_solve, "columns"): lane = 3 j + axis
template <int G>
DEV void hg_columns(const ShfModel* m, const EnvLds& L, int nb, int l, int K, const float* hc, float* W) {
  static_assert(G >= 3 * HCK, "a lane per column");
  const int j = (l * 11) >> 5, ax = l - 3 * j;
  const bool col = l < 3 * K;
  const float* hj = hc + (col ? j : 0) * HC_STRIDE;
  const float rj[3] = {hj[HC_R], hj[HC_R + 1], hj[HC_R + 2]};
  const int bsa = col ? __float_as_int(hj[HC_BODY]) : -1, bsb = col ? __float_as_int(hj[HC_BODYB]) : -1;
  const float* ej = hj + (ax == 0 ? HC_N : (ax == 1 ? HC_T1 : HC_T2));
  const float e[3] = {ej[0], ej[1], ej[2]};
  HgResp qa, qb;
  qa.src = -1; qb.src = -1;
  if (bsa >= 0) hg_impulse(m, L, nb, bsa, rj, e, qa);
  if (bsb >= 0) hg_impulse(m, L, nb, bsb, rj, e, qb);
  // one block of every symmetric pair (oracle: hard_solve, "columns"): block (i, j) from this column for i = j, j - 1, .. j - K / 2
  // (modulo K; for even K the pair at distance K / 2 belongs to the columns j >= K / 2): K / 2 + 1 targets per lane at most
  for (int d = 0; 2 * d <= HCK; d++) {
    if (__ballot(2 * d <= K) == 0ull) break;
    const bool todo = col && 2 * d <= K && d < K && !(d > 0 && 2 * d == K && 2 * j < K);
    int i = j - d;
    if (i < 0) i += K;
    if (!todo) continue;
    const float* hi = hc + i * HC_STRIDE;
    const float ri[3] = {hi[HC_R], hi[HC_R + 1], hi[HC_R + 2]};
    const int bta = __float_as_int(hi[HC_BODY]), btb = __float_as_int(hi[HC_BODYB]);
    float aa[3], ab[3], ba[3], bb[3], vw[3];
    hg_velocity(m, L, nb, qa, bta, ri, aa);
    hg_velocity(m, L, nb, qa, btb, ri, ab);
    hg_velocity(m, L, nb, qb, bta, ri, ba);
    hg_velocity(m, L, nb, qb, btb, ri, bb);
#pragma unroll
    for (int r = 0; r < 3; r++) vw[r] = (aa[r] - ab[r]) - (ba[r] - bb[r]);
    float* Wb = W + (j * HCK + i) * 9 + ax;
    const float w0 = dot3(hi + HC_N, vw), w1 = dot3(hi + HC_T1, vw), w2 = dot3(hi + HC_T2, vw);
    Wb[0] = w0; Wb[3] = w1; Wb[6] = w2;
    if (i != j) { float* Wt = W + (i * HCK + j) * 9 + 3 * ax; Wt[0] = w0; Wt[1] = w1; Wt[2] = w2; }     // row ax of block (j, i)
  }
}

// start-of-step and free velocity of the point r of solver body `body` (oracle: hc_body_point)
DEV void hg_body_point(const EnvLds& L, int nb, int body, const float* r, float dt, float* vs, float* vf) {
  vs[0] = vs[1] = vs[2] = 0.0f; vf[0] = vf[1] = vf[2] = 0.0f;
  if (body < 0) return;
  const float* pv = L.pose + body * POSE_STRIDE + 12;
  const float* D = L.xch + body * XCH_STRIDE + ((body == 0 || body >= nb) ? HB_FDL : HB_DL);
  float v[6], v6[6];
#pragma unroll
  for (int k = 0; k < 6; k++) { v[k] = pv[k]; v6[k] = fmaf(dt, D[k], v[k]); }
  hard_point(v, r, vs);
  hard_point(v6, r, vf);
}

// where the accelerations the impulses add travel down the tree: set 0 in the acceleration array, set 1 in words of the
// exchange slots that are dead by then (the body records after the columns; the root's / boxes' velocity-rate words)
DEV float* hg_ac(const EnvLds& L, int q, int b) {
  return q == 0 ? L.acc + b * 6 : (b == 0 ? L.xch + HB_FDL : L.xch + b * XCH_STRIDE + HB_S);
}

// records (oracle: the solver bodies' response data): body lane b -> S, U, 1/D in its exchange slot; the root's / the free boxes'
// LDL^T factors in theirs; then the velocity rates Dl = S qdd + Dl(parent) down the tree.  abox: a free box's free acceleration
// (its lane; the integration needs it again).  A group sync has to follow.  Its own function since round 6: k_abb_step_ws_hard
// runs it on the arm wave (16 lanes per env) while the box wave is still busy with the link candidates.
// PARTS: 1 = the articulation's lanes (records, root, velocity rates), 2 = the free boxes' lanes, 3 = both.
template <int G, bool BOX, class DM, class LM, int PARTS = 3>
DEV void hard_records(const StepCtx& C, const EnvLds& L, int l, const LM& M, const BodyRegs& B, const float* g, const float* a, float* abox) {
  const ShfModel* m = C.m;
  const SceneDev* S = C.scene;
  const int nb = DM::nb(m), nbx = BOX ? S->nboxes : 0;
  const bool moving = M.moving;
  const int mylevel = M.level(), nl = DM::nlevels(m);
  const int kd = l - nb;
  const bool dynbox = BOX && kd >= 0 && kd < nbx && box_is_dynamic(S->box[kd]);
  const float gb[3] = {C.sp.gravity[0], C.sp.gravity[1], C.sp.gravity[2]};
  if ((PARTS & 1) && moving) {
    float* rec = L.xch + l * XCH_STRIDE;
#pragma unroll
    for (int k = 0; k < 6; k++) { rec[HB_S + k] = B.S[k]; rec[HB_U + k] = B.U[k]; }
    rec[HB_INVD] = B.invD;
  }
  if ((PARTS & 1) && l == 0) {
    float* rec = L.xch;
    if (m->fixed_base) {
#pragma unroll
      for (int k = 0; k < 6; k++) rec[HB_FDL + k] = 0.0f;
    } else {
      Ldlt6 F;
      ldlt_factor6(B.IA, F);
      root_factors_store(F, rec);
      const float ang[3] = {L.root[10], L.root[11], L.root[12]}, lin[3] = {L.root[7], L.root[8], L.root[9]};
      float wxv[3];
      cross3(ang, lin, wxv);
#pragma unroll
      for (int k = 0; k < 3; k++) { rec[HB_FDL + k] = a[k]; rec[HB_FDL + 3 + k] = (a[3 + k] + g[k]) + wxv[k]; }
    }
  }
  if ((PARTS & 2) && dynbox) {
    float* rec = L.xch + l * XCH_STRIDE;
    Ldlt6 F;
    ldlt_factor6(B.IA, F);
    ldlt_substitute6(F, B.pA, abox);
    root_factors_store(F, rec);
    const float* row = L.root + 13 * (1 + kd);
    const float ang[3] = {row[10], row[11], row[12]}, lin[3] = {row[7], row[8], row[9]};
    float wxv[3];
    cross3(ang, lin, wxv);
#pragma unroll
    for (int k = 0; k < 3; k++) { rec[HB_FDL + k] = abox[k]; rec[HB_FDL + 3 + k] = (abox[3 + k] + gb[k]) + wxv[k]; }
  }
  if constexpr ((PARTS & 1) != 0) {
    // velocity rates down the tree: Dl = S qdd + Dl(parent)
    for (int lev = 1; lev <= nl; lev++) {
      GROUP_SYNC();
      if (moving && mylevel == lev) {
        const int pd = M.dynpar();
        const float* dp = L.xch + pd * XCH_STRIDE + (pd == 0 ? HB_FDL : HB_DL);
        const float qdd = L.dofb[M.dofi() * DOF_STRIDE + 4];
        float* rec = L.xch + l * XCH_STRIDE;
#pragma unroll
        for (int k = 0; k < 6; k++) rec[HB_DL + k] = fmaf(B.S[k], qdd, dp[k]);
      }
    }
  }
}

// The velocity-level solve of the body-per-lane sub-step, after its FREE articulated-body solve: records, gather, columns,
// owners, sweeps, the impulse passes, integration (poses with the accelerations after the position iterations, velocities
// after the velocity iterations), net contact forces.  `a`: lane 0 holds the root's free acceleration.
template <int G, bool BOX, class DM, class LM, class SC, bool SELF, bool LINK, bool RECORDS, int ARMNL>
DEV void substep_hard_finish(const StepCtx& C, const EnvLds& L, int l, const LM& M, BodyRegs& B, const float* g, float* a, int nself,
                             int self_slot0, int link_slot0, int nlink, float* contact_out) {
  static_assert(G == 32, "two envs per wavefront (hard_sweeps)");
  const ShfModel* m = C.m;
  const SceneDev* S = C.scene;
  const int nb = DM::nb(m), nd = DM::nd(m), nbx = BOX ? S->nboxes : 0, nbt = nb + nbx;
  const float dt = C.sp.dt, idt = 1.0f / dt;
  const bool isdyn = M.isdyn, moving = M.moving;
  const int mylevel = M.level(), nl = DM::nlevels(m);
  const int kd = l - nb;
  const bool dynbox = BOX && kd >= 0 && kd < nbx && box_is_dynamic(S->box[kd]);
  const float gb[3] = {C.sp.gravity[0], C.sp.gravity[1], C.sp.gravity[2]};
  const int hc0 = hard_hc_slot0(link_slot0 + (LINK ? 2 * SHF_MAX_LINK_CONTACTS : 0), LINK);
  float* hc = L.pt + hc0 * PT_STRIDE;
  float* W = L.pt;           // (in the place of the first 48 slots, from the columns on: see hard_hc_slot0)
  const int npos = C.sp.pos_iters > 0 ? C.sp.pos_iters : 0, nvel = C.sp.vel_iters > 0 ? C.sp.vel_iters : 0;
  const int kmax = hard_kmax_of(C.sp, HCK);

  PHASE_BEGIN();
  // ---- records
  float abox[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};      // a free box's free acceleration (its lane)
  if constexpr (RECORDS) {
    hard_records<G, BOX, DM, LM>(C, L, l, M, B, g, a, abox);
  } else if (dynbox) {
    // (written beside the records by whoever ran hard_records: the box actors' rows of the acceleration array are free)
#pragma unroll
    for (int k = 0; k < 6; k++) abox[k] = L.acc[l * 6 + k];
  }
  GROUP_SYNC();
  PHASE_MARK(32);

  // ---- gather
  const SlotLay Q = BOX ? slot_lay<SC>(m, S) : SlotLay();
  HgSeq H;
  H.nev = m->neval > 0 ? m->neval : DM::np(m);
  H.nself = SELF ? nself : 0; H.nbx = nbx; H.T = 1 + nbx; H.nsph = BOX ? m->nsph : 0; H.nlink = LINK ? nlink : 0;
  H.self_slot0 = self_slot0; H.link_slot0 = link_slot0;
  H.P1 = H.nev + H.nself; H.P2 = H.P1 + nbx * 8 * H.T; H.P3 = H.P2 + H.nsph * nbx; H.P4 = H.P3 + H.nlink;
  const int K = hg_gather<G, SC>(C, L, Q, H, l, hc, kmax, hc0);
  GROUP_SYNC();
  PHASE_MARK(33);

  float ac0[2][6] = {{0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f}};   // root lane / box lanes: what the contacts add
  if (__ballot(K > 0) != 0ull) {
    // ---- columns
    hg_columns<G>(m, L, nb, l, K, hc, W);
    GROUP_SYNC();
    PHASE_MARK(34);
    // ---- owners
    HardOwner O;
    const bool own = l < K;
    {
      const float* h = hc + (own ? l : 0) * HC_STRIDE;
      const float r[3] = {h[HC_R], h[HC_R + 1], h[HC_R + 2]};
      const int ba = own ? __float_as_int(h[HC_BODY]) : -1, bb = own ? __float_as_int(h[HC_BODYB]) : -1;
      float vsa[3], vfa[3], vsb[3], vfb[3], vs[3], vf[3];
      hg_body_point(L, nb, ba, r, dt, vsa, vfa);
      hg_body_point(L, nb, bb, r, dt, vsb, vfb);
#pragma unroll
      for (int k = 0; k < 3; k++) { vf[k] = vfa[k] - vfb[k]; vs[k] = vsa[k] - vsb[k]; }
      const int oi = own ? l : 0;
      hard_owner_setup(C.sp, h, vs, vf, W + (oi * HCK + oi) * 9, own, idt, O);
    }
    GROUP_SYNC();
    // ---- sweeps
    PHASE_MARK(35);
    hard_sweeps(O, hc, W, l, K, npos, nvel, C.sp);
    GROUP_SYNC();
    PHASE_MARK(36);
    // ---- the impulses through the tree, both sets (oracle: hc_apply)
    const int NQ = nvel > 0 ? 2 : 1;
    float pc[2][6] = {{0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f}};
    if (isdyn || dynbox) {
      // which constraints act on this body: the ids of all of them in flight at once (not a load per iteration behind its branch)
      unsigned mine_a = 0u, mine_b = 0u;
#pragma unroll
      for (int c = 0; c < HCK; c++) {
        const float* h = hc + c * HC_STRIDE;
        const int ia = __float_as_int(h[HC_BODY]), ib = __float_as_int(h[HC_BODYB]);
        if (c < K && ia == l) mine_a |= 1u << c;
        if (c < K && ib == l) mine_b |= 1u << c;
      }
      for (unsigned bits = mine_a | mine_b; bits; bits &= bits - 1u) {
        const int c = __builtin_ctz(bits);
        const float* h = hc + c * HC_STRIDE;
        const bool ona = (mine_a >> c) & 1u;
        const float r[3] = {h[HC_R], h[HC_R + 1], h[HC_R + 2]};
#pragma unroll
        for (int q = 0; q < 2; q++) {
          const float f[3] = {h[q == 0 ? HC_P : HC_PV0] * idt, h[q == 0 ? HC_P + 1 : HC_PV1] * idt, h[q == 0 ? HC_P + 2 : HC_PV2] * idt};
          float t[3];
          cross3(r, f, t);
          if (ona) {
#pragma unroll
            for (int k = 0; k < 3; k++) { pc[q][k] -= t[k]; pc[q][3 + k] -= f[k]; }
          } else {
#pragma unroll
            for (int k = 0; k < 3; k++) { pc[q][k] += t[k]; pc[q][3 + k] += f[k]; }
          }
        }
      }
    }
    if constexpr (ARMNL > 0) {
      // A fixed base with ONE serial chain of ARMNL links (ArmChain<ARMNL>::matches: body b's parent is b - 1, its dof b - 1, its
      // level b): every level holds one body, so the level-by-level passes below are 2 x ARMNL LDS hand-offs with a group
      // synchronisation each for nothing -- the body lanes publish the impulses they gathered, lane 0 runs both recursions link
      // after link in registers (the same operations on the same values: own + child's, S . pc, U tt; then U . a, S qc).
      if (moving) {
        float* rec = L.xch + l * XCH_STRIDE;
#pragma unroll
        for (int j = 0; j < 6; j++) { rec[HB_PC + j] = pc[0][j]; rec[HB_DL + j] = pc[1][j]; }
      }
      GROUP_SYNC();
      if (l == 0) {
        float cp[2][6], ucs[ARMNL][2];
#pragma unroll
        for (int b = ARMNL; b >= 1; b--) {
          const float* rec = L.xch + b * XCH_STRIDE;
          float Sb[6], Ub[6];
#pragma unroll
          for (int j = 0; j < 6; j++) { Sb[j] = rec[HB_S + j]; Ub[j] = rec[HB_U + j]; }
          const float iD = rec[HB_INVD];
#pragma unroll
          for (int q = 0; q < 2; q++) {
            float pb[6];
#pragma unroll
            for (int j = 0; j < 6; j++) {
              const float own = rec[(q == 0 ? HB_PC : HB_DL) + j];
              pb[j] = b == ARMNL ? own : own + cp[q][j];         // (the tip has no child to add)
            }
            float sp = Sb[0] * pb[0];
#pragma unroll
            for (int j = 1; j < 6; j++) sp = fmaf(Sb[j], pb[j], sp);
            ucs[b - 1][q] = -sp;
            const float tt = ucs[b - 1][q] * iD;
#pragma unroll
            for (int j = 0; j < 6; j++) cp[q][j] = fmaf(Ub[j], tt, pb[j]);
          }
        }
        float ac[2][6] = {{0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f}};       // the base does not move
#pragma unroll
        for (int b = 1; b <= ARMNL; b++) {
          const float* rec = L.xch + b * XCH_STRIDE;
          float Sb[6], Ub[6];
#pragma unroll
          for (int j = 0; j < 6; j++) { Sb[j] = rec[HB_S + j]; Ub[j] = rec[HB_U + j]; }
          const float iD = rec[HB_INVD];
#pragma unroll
          for (int q = 0; q < 2; q++) {
            float ua = Ub[0] * ac[q][0];
#pragma unroll
            for (int j = 1; j < 6; j++) ua = fmaf(Ub[j], ac[q][j], ua);
            const float qc = (ucs[b - 1][q] - ua) * iD;
#pragma unroll
            for (int j = 0; j < 6; j++) ac[q][j] = fmaf(Sb[j], qc, ac[q][j]);
            L.dofb[(b - 1) * DOF_STRIDE + 2 + q] = qc;
          }
        }
      }
      if (dynbox) {
#pragma unroll
        for (int q = 0; q < 2; q++) root_factors_apply(L.xch + l * XCH_STRIDE, pc[q], ac0[q]);
      }
    } else {
      float uc[2] = {0.0f, 0.0f};
      for (int lev = nl; lev >= 1; lev--) {
        if (moving && mylevel == lev) {
          float* rec = L.xch + l * XCH_STRIDE;
#pragma unroll
          for (int q = 0; q < 2; q++) {
            float sp = B.S[0] * pc[q][0];
#pragma unroll
            for (int j = 1; j < 6; j++) sp = fmaf(B.S[j], pc[q][j], sp);
            uc[q] = -sp;
            const float tt = uc[q] * B.invD;
#pragma unroll
            for (int j = 0; j < 6; j++) { pc[q][j] = fmaf(B.U[j], tt, pc[q][j]); rec[(q == 0 ? HB_PC : HB_DL) + j] = pc[q][j]; }
          }
        }
        GROUP_SYNC();
        if (isdyn && mylevel == lev - 1) {
          for (int kk = 0; kk < M.nchild; kk++) {
            const int cb = kk < LANE_CHILDREN ? M.child[kk < LANE_CHILDREN ? kk : 0] : m->child_list[M.child0 + kk];
            const float* o = L.xch + cb * XCH_STRIDE;
#pragma unroll
            for (int q = 0; q < 2; q++)
#pragma unroll
              for (int j = 0; j < 6; j++) pc[q][j] += o[(q == 0 ? HB_PC : HB_DL) + j];
          }
        }
      }
      if (l == 0) {
#pragma unroll
        for (int q = 0; q < 2; q++) {
          if (m->fixed_base) {
#pragma unroll
            for (int k = 0; k < 6; k++) ac0[q][k] = 0.0f;
          } else {
            root_factors_apply(L.xch, pc[q], ac0[q]);
          }
        }
#pragma unroll
        for (int q = 0; q < 2; q++) {
          float* o = hg_ac(L, q, 0);
#pragma unroll
          for (int k = 0; k < 6; k++) o[k] = ac0[q][k];
        }
      }
      if (dynbox) {
#pragma unroll
        for (int q = 0; q < 2; q++) root_factors_apply(L.xch + l * XCH_STRIDE, pc[q], ac0[q]);
      }
      for (int lev = 1; lev <= nl; lev++) {
        GROUP_SYNC();
        if (moving && mylevel == lev) {
          const int pd = M.dynpar();
#pragma unroll
          for (int q = 0; q < 2; q++) {
            const float* pa = hg_ac(L, q, pd);
            float acp[6];
#pragma unroll
            for (int j = 0; j < 6; j++) acp[j] = pa[j];
            float ua = B.U[0] * acp[0];
#pragma unroll
            for (int j = 1; j < 6; j++) ua = fmaf(B.U[j], acp[j], ua);
            const float qc = (uc[q] - ua) * B.invD;
            float* o = hg_ac(L, q, l);
#pragma unroll
            for (int j = 0; j < 6; j++) o[j] = fmaf(B.S[j], qc, acp[j]);
            L.dofb[M.dofi() * DOF_STRIDE + 2 + q] = qc;
          }
        }
      }
    }
    GROUP_SYNC();
    (void)NQ;
    PHASE_MARK(37);
  }

  // ---- integration (oracle substep(), "semi-implicit Euler" with hard = 1)
  const int vq = nvel > 0 ? 1 : 0;      // the set the velocities take
  if (l < nd) {
    float* D = L.dofb + l * DOF_STRIDE;
    const float vl = M.vel_limit;
    const float qf = D[4];
    const float qdv = K > 0 ? qf + D[2 + vq] : qf, qdp = K > 0 ? qf + D[2] : qf;
    const float qd = rclampf(fmaf(dt, qdv, D[1]), -vl, vl);
    const float qp = rclampf(fmaf(dt, qdp, D[1]), -vl, vl);
    D[0] = fmaf(dt, qp, D[0]);
    D[1] = qd;
  }
  if ((l == 0 && !m->fixed_base) || dynbox) {
    // a free body: the root (its origin is O: p = 0) or a box at B.p
    float* Rt = l == 0 ? L.root : L.root + 13 * (1 + kd);
    // (selected by value: a select between pointers to local arrays puts them on the stack)
    const float af[6] = {l == 0 ? a[0] : abox[0], l == 0 ? a[1] : abox[1], l == 0 ? a[2] : abox[2],
                         l == 0 ? a[3] : abox[3], l == 0 ? a[4] : abox[4], l == 0 ? a[5] : abox[5]};
    const float gg[3] = {l == 0 ? g[0] : gb[0], l == 0 ? g[1] : gb[1], l == 0 ? g[2] : gb[2]};
    float av[6], apz[6];
#pragma unroll
    for (int j = 0; j < 6; j++) {
      av[j] = K > 0 ? af[j] + ac0[vq][j] : af[j];
      apz[j] = K > 0 ? af[j] + ac0[0][j] : af[j];
    }
    const float ang[3] = {Rt[10], Rt[11], Rt[12]}, lin[3] = {Rt[7], Rt[8], Rt[9]};
    const float pp[3] = {l == 0 ? 0.0f : B.p[0], l == 0 ? 0.0f : B.p[1], l == 0 ? 0.0f : B.p[2]};
    float wxv[3], axp[3], axq[3];
    cross3(ang, lin, wxv);
    cross3(av, pp, axp);
    cross3(apz, pp, axq);
    const float damp = 1.0f / fmaf(dt, C.sp.angular_damping, 1.0f);
    float wn[3], vn[3], wp[3], vp[3];
    const float wmax = C.sp.max_ang_vel;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      wn[k] = fmaf(dt, av[k], ang[k]) * damp;
      wp[k] = fmaf(dt, apz[k], ang[k]) * damp;
      if (l == 0) {
        vn[k] = fmaf(dt, av[3 + k] + gg[k] + wxv[k], lin[k]);
        vp[k] = fmaf(dt, apz[3 + k] + gg[k] + wxv[k], lin[k]);
      } else {
        vn[k] = fmaf(dt, av[3 + k] + gg[k] + axp[k] + wxv[k], lin[k]);
        vp[k] = fmaf(dt, apz[3 + k] + gg[k] + axq[k] + wxv[k], lin[k]);
      }
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
  // ---- net contact force per reported body and box actor: the final impulses / dt, constraint order (oracle: hc_forces)
  if (contact_out) {
    GROUP_SYNC();
    float f[3] = {0.0f, 0.0f, 0.0f};
    if (l < nbt) {
      for (int c = 0; c < K; c++) {
        const float* h = hc + c * HC_STRIDE;
        const int rep = __float_as_int(h[HC_REP]), repb = __float_as_int(h[HC_REPB]);
        if (rep != l && repb != l) continue;
        const float pf[3] = {(nvel > 0 ? h[HC_PV0] : h[HC_P]) * idt, (nvel > 0 ? h[HC_PV1] : h[HC_P + 1]) * idt, (nvel > 0 ? h[HC_PV2] : h[HC_P + 2]) * idt};
        if (rep == l) { f[0] += pf[0]; f[1] += pf[1]; f[2] += pf[2]; }
        else { f[0] -= pf[0]; f[1] -= pf[1]; f[2] -= pf[2]; }
      }
    }
    GROUP_SYNC();
    if (l < nbt) { contact_out[3 * l] = f[0]; contact_out[3 * l + 1] = f[1]; contact_out[3 * l + 2] = f[2]; }
  }
  GROUP_SYNC();
}
