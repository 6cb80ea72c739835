// shf_link.h -- one revolute link of a serial chain, executed by the lane that owns the whole chain: pose composition
// below the parent and the ABA inward step, with the link's (S, c, U, 1/D, u) in the caller's hands.  Shared by the
// chain-mapped A1 step (shf_chain.h) and the serial-arm sub-step (shf_arm.h).  Same operations in the same order as the
// body-per-lane kinematics() / substep() of shf_device.h and as oracle/shf_oracle.c.
#pragma once
#include "shf_device.h"

// One link's share of the ABA, kept in the chain lane's registers across the sub-step.
struct ChainLink {
  float S[6], c[6], U[6], invD, u;
};

DEV void pose_store(float* o, const float* Rw, const float* p, const float* v) {
#pragma unroll
  for (int k = 0; k < 9; k++) o[k] = Rw[k];
#pragma unroll
  for (int k = 0; k < 3; k++) o[9 + k] = p[k];
#pragma unroll
  for (int k = 0; k < 6; k++) o[12 + k] = v[k];
}

// The joint's local rotation Rl = trot * Rot(axis, q): depends on its own angle only (dof lanes, all joints at once).
DEV void joint_local_rotation(const float* tr, const float* ax, float qv, float* Rl) {
  float sn, cs;
  sincos_spec(qv, &sn, &cs);
  const float oc = 1.0f - cs;
  const float Rq[9] = {fmaf(oc, ax[0] * ax[0], cs),           fmaf(oc, ax[0] * ax[1], -(sn * ax[2])), fmaf(oc, ax[0] * ax[2], sn * ax[1]),
                       fmaf(oc, ax[1] * ax[0], sn * ax[2]),    fmaf(oc, ax[1] * ax[1], cs),            fmaf(oc, ax[1] * ax[2], -(sn * ax[0])),
                       fmaf(oc, ax[2] * ax[0], -(sn * ax[1])), fmaf(oc, ax[2] * ax[1], sn * ax[0]),    fmaf(oc, ax[2] * ax[2], cs)};
  mm3(tr, Rq, Rl);
}
// Composition of one revolute link below (Rp, pp, vp) -- the parent's pose and velocity, replaced by the link's own on
// return -- with its motion subspace and bias acceleration (oracle kinematics(), same operations).
DEV void chain_compose_link(const float* tp, const float* ax, const float* Rl, float qdv, float* Rp, float* pp,
                            float* vp, float* S, float* c) {
  float t[3], Rn[9], pn[3], aw[3], t2[3], vJ[6], cc[6];
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

// Inward step of one link: U = IA S, D, u, IA -= U U^T / D, pa = pA + IA c + U u / D  (pa returned, IA updated).
DEV void chain_inward_link(ChainLink& K, float* IA, const float* pA, float dex, float tau0, float* pa) {
#pragma unroll
  for (int i = 0; i < 6; i++) {
    float acc = SYMG(IA, i, 0) * K.S[0];
#pragma unroll
    for (int j = 1; j < 6; j++) acc = fmaf(SYMG(IA, i, j), K.S[j], acc);
    K.U[i] = acc;
  }
  float D = K.S[0] * K.U[0];
#pragma unroll
  for (int j = 1; j < 6; j++) D = fmaf(K.S[j], K.U[j], D);
  D += dex;
  float sp = K.S[0] * pA[0];
#pragma unroll
  for (int j = 1; j < 6; j++) sp = fmaf(K.S[j], pA[j], sp);
  const float invD = rcp_spec(D);
  K.invD = invD;
  K.u = tau0 - sp;
  float W[6];
#pragma unroll
  for (int i = 0; i < 6; i++) W[i] = K.U[i] * invD;
#pragma unroll
  for (int i = 0; i < 6; i++)
#pragma unroll
    for (int j = i; j < 6; j++) IA[SYM(i, j)] = fmaf(-K.U[i], W[j], IA[SYM(i, j)]);
#pragma unroll
  for (int i = 0; i < 6; i++) {
    float acc = SYMG(IA, i, 0) * K.c[0];
#pragma unroll
    for (int j = 1; j < 6; j++) acc = fmaf(SYMG(IA, i, j), K.c[j], acc);
    pa[i] = fmaf(W[i], K.u, pA[i] + acc);
  }
}

