// shf_task.h -- device-side structures and helpers shared by the kernels of shf_api.hip and the chain-mapped A1 step
// (shf_a1_chain.hip): kernel argument blocks, the cooperative parameter staging, the in-kernel episode statistics,
// the counter-based random draws of a reset.  Device code only; no host state.
#pragma once
#include "shf_device.h"

// ---------------------------------------------------------------- kernels --
struct SimArgs {
  ShfSimParams sp;
  ShfTerrain terr;
  const int16_t* heights;
  const ShfModel* model;  // device copy
  const ShfScene* scene;  // device copy (box actors)
  int nboxes;
  int n;
  float* dof;            // (n*nd,2)
  float* root;           // (n*A,13)
  int actors;            // root rows per env
  const float* effort;
  const float* pos_tgt;
  const float* vel_tgt;
  const float* body_force;  // nullptr unless armed
  const float* body_force_pos;  // world points of application, nullptr = at the centres of mass
  const float* friction;
  const float* mscale;   // (n,nb) per-env factors on the bodies' mass and inertia (SHF_T_BODY_MASS_SCALE), nullptr = 1
  float* contact;  // (n*B,3)
  int32_t* dropped;  // (n) contacts dropped at the per-env limits, accumulated (SHF_T_DROPPED), may be null
  const ShfHullSet* hulls;   // device copy of the articulation's convex hulls (SHF_T_HULLS), nullptr = none
};
// (SHF_T_CONTACT_HIST bound: `dropped` points at its rows -- csrc/shf_device.h: SHF_HIST_FLAG, env_dropped)

// Cooperative global -> LDS copy of a fixed-size parameter block by the 256 threads of a block: all loads
// are issued before the first store (one memory round trip instead of one per loop iteration).
// Both pointers are 16-byte aligned (device allocations; LDS offsets are multiples of 4 words).
template <int NBYTES, int THREADS = 256>
DEV void stage_block(const void* gsrc, float* ldst) {
  constexpr int NV = NBYTES / 16, NT = (NBYTES % 16) / 4, IV = (NV + THREADS - 1) / THREADS;
  const uint4* src = reinterpret_cast<const uint4*>(gsrc);
  uint4* dst = reinterpret_cast<uint4*>(ldst);
  uint4 v[IV > 0 ? IV : 1];
  uint32_t tail = 0;
#pragma unroll
  for (int k = 0; k < IV; k++) {
    const int i = (int)threadIdx.x + k * THREADS;
    v[k] = src[i < NV ? i : 0];
  }
  if (NT > 0 && (int)threadIdx.x < NT) tail = reinterpret_cast<const uint32_t*>(gsrc)[NV * 4 + threadIdx.x];
#pragma unroll
  for (int k = 0; k < IV; k++) {
    const int i = (int)threadIdx.x + k * THREADS;
    if (i < NV) dst[i] = v[k];
  }
  if (NT > 0 && (int)threadIdx.x < NT) reinterpret_cast<uint32_t*>(ldst)[NV * 4 + threadIdx.x] = tail;
}
// the flattened articulation
template <int THREADS = 256>
DEV const ShfModel* stage_model(const ShfModel* gm, float* smem) {
  stage_block<(int)sizeof(ShfModel), THREADS>(gm, smem);
  __syncthreads();
  return reinterpret_cast<const ShfModel*>(smem);
}
#define MODEL_WORDS ((int)((sizeof(ShfModel) / 4 + 3) & ~3))
#define TASK_WORDS ((int)((sizeof(ShfA1TaskParams) / 4 + 3) & ~3))

#define SCENE_WORDS ((int)((sizeof(ShfScene) / 4 + 3) & ~3))

template <int THREADS = 256>
DEV const ShfScene* stage_scene(const ShfScene* gs, float* dst_words) {
  stage_block<(int)sizeof(ShfScene), THREADS>(gs, dst_words);
  return reinterpret_cast<const ShfScene*>(dst_words);
}

// ------------------------------------------------- episode statistics --
// log_info (env.py:149-158) folded into the step kernel: no second launch, no host-side slot argument (the vec-step
// can be replayed from a hipGraph).  Every env contributes integers -- finished-episode reward sums in 2^-20 fixed
// point, counts and terrain levels as they are -- so the reduction is exact and order-independent: env groups add
// into LDS, the last group of a block adds the block totals to the slot's global accumulators and takes a ticket, the
// block holding the last ticket turns the totals into the ring row, clears the slot and advances the step counter.
// Layout of the accumulator tensor: (ring + 1) rows of STATS_COLS u64; row r < ring = [keys..., col 8 = tickets],
// row `ring` col 0 = vec-steps completed.
#define STATS_COLS 10
#define STATS_LDS_WORDS 24     /* 9 u64 block accumulators + the group counter, 16-byte padded */
#define STATS_FIX 1048576.0f   /* 2^20 */
struct StatsArgs {
  unsigned long long* acc;
  float* out;          // (ring, out_cols)
  int ring, out_cols;
};
DEV long long stats_fix(float x) { return (long long)rintf(x * STATS_FIX); }
DEV void stats_block_init(float* lds_words) {
  if (threadIdx.x < STATS_LDS_WORDS) reinterpret_cast<uint32_t*>(lds_words)[threadIdx.x] = 0u;   // before stage_model's barrier
}
// stats_contribute: called by lane 0 of every env group of the block once its NK values are final; the last group of the
// block adds the block totals to the slot's global accumulators and returns the slot's row (else nullptr).
// stats_ticket / stats_finish: called by that same lane later in the kernel with the row (the additions have had time to
// be acknowledged): takes the block's ticket; the block that drew the last one finalises.  FIN(acc[NK], out) -> ring row.
// Ordering without agent-scope fences (on a multi-XCD part an agent-scope release writes the XCD's whole dirty L2 back
// -- here the ~13 MB of outputs the launch has just produced -- once per block): every global operation below is an
// agent-scope atomic, which is performed at the device's point of coherence; a block waits for its additions to be
// acknowledged (s_waitcnt through a workgroup-scope fence) before it takes its ticket, so whoever draws the last
// ticket reads complete totals -- with atomic exchanges, which also clear the slot.
// stats_step_load: the vec-step counter, read once at the top of the kernel (it only changes when the last block of a
// launch finishes) so that its round trip is not on post_step's critical path; wave-uniform, kept in scalar registers.
DEV unsigned long long stats_step_load(const StatsArgs& S) {
  const unsigned long long v = __hip_atomic_load(S.acc + (size_t)S.ring * STATS_COLS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return ((unsigned long long)hi << 32) | lo;
}
template <int NK>
DEV unsigned long long* stats_contribute(const StatsArgs& S, float* lds_words, const long long* v, int envs_in_block,
                                         unsigned long long step) {
  unsigned long long* blk = reinterpret_cast<unsigned long long*>(lds_words);
#pragma unroll
  for (int k = 0; k < NK; k++)
    if (v[k] != 0) __hip_atomic_fetch_add(&blk[k], (unsigned long long)v[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  const unsigned done = __hip_atomic_fetch_add(reinterpret_cast<unsigned*>(&blk[9]), 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
  if ((int)done != envs_in_block - 1) return nullptr;
  // last group of this block
  unsigned long long* row = S.acc + (size_t)(step % (unsigned long long)S.ring) * STATS_COLS;
#pragma unroll
  for (int k = 0; k < NK; k++) {
    const unsigned long long t = __hip_atomic_load(&blk[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (t != 0ull) __hip_atomic_fetch_add(&row[k], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  return row;
}
// stats_ticket: after the block's additions are acknowledged; the returned count is only looked at by stats_finish, at
// the very end of the kernel, so the atomic's round trip overlaps the output stores in between.
DEV unsigned long long stats_ticket(unsigned long long* row) {
  if (!row) return 0ull;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // s_waitcnt: this block's additions are acknowledged
  return __hip_atomic_fetch_add(&row[8], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int NK, class FIN>
DEV void stats_finish(const StatsArgs& S, unsigned long long* row, unsigned long long ticket, unsigned long long step, FIN finalize) {
  if (!row) return;
  if (ticket != (unsigned long long)gridDim.x - 1ull) return;
  // last block of the launch
  long long tot[NK];
#pragma unroll
  for (int k = 0; k < NK; k++)
    tot[k] = (long long)__hip_atomic_exchange(&row[k], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(&row[8], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  float* o = S.out + (size_t)(step % (unsigned long long)S.ring) * S.out_cols;
  finalize(tot, o);
  float* latest = S.out + (size_t)S.ring * S.out_cols;     // row `ring`: always the step that ran last
  for (int k = 0; k < S.out_cols; k++) latest[k] = o[k];
  unsigned long long* ctl = S.acc + (size_t)S.ring * STATS_COLS;
  __hip_atomic_store(ctl, step + 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ------------------------------------------------------- fused A1 step --
struct A1Args {
  SimArgs S;
  const ShfA1TaskParams* tp;  // device copy
  int64_t env_off;
  const float* raw_actions;
  float *actions, *obs, *rew;
  uint8_t *reset, *timeout;
  int64_t* ep_len;
  float *command, *history, *rew_sums, *torques, *base_vel, *heights_out;
  const float* hpoints;
  float *push, *origins;
  int64_t* levels;
  const int64_t* types;
  const float* torigins;
  int32_t* reset_count;
  float* done_sums;
  float* body_state;
  StatsArgs stats;
};

DEV void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t* out) {
#pragma unroll
  for (int r = 0; r < 10; r++) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1,
                   n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
DEV float u01(uint32_t x) { return (float)(x >> 8) * 5.9604644775390625e-08f; }
DEV float urange(uint32_t x, float lo, float hi) { return (hi - lo) * u01(x) + lo; }
// run_policy('random') (shifu/runner/policy_runner.py:38-41) without a per-step RNG launch: the raw action of dof d of
// global env gid at vec-step `step` is U(-1, 1) from the counter-based generator that also drives the resets
// (Philox4x32-10 keyed by the task seed; counter words gid, step, 0x40000000 + d / 4 -- the reset draws use 0, 1, 2 there),
// so a sharded run draws what the unsharded one does.  Oracle: shf_oracle_random_actions.
DEV float random_action(int64_t seed, int64_t gid, unsigned long long step, int d) {
  uint32_t r[4];
  philox4x32((uint32_t)gid, (uint32_t)step, 0x40000000u + (uint32_t)(d >> 2), (uint32_t)(gid >> 32) ^ ((uint32_t)(step >> 32) << 16),
             (uint32_t)seed, (uint32_t)(seed >> 32), r);
  return urange(r[d & 3], -1.0f, 1.0f);
}
// the policy's raw action: the caller's tensor, or (null) the generator above
template <class ARGS>
DEV float raw_action(const ARGS& A, int e, int d, int nd, unsigned long long step) {
  if (A.raw_actions) return A.raw_actions[(size_t)e * nd + d];
  return random_action(A.tp->seed, A.env_off + e, step, d);
}
DEV void quat_rotate_inverse(const float* q, const float* v, float* o) {
  const float w = q[3];
  const float s = 2.0f * (w * w) - 1.0f;
  const float cx = q[1] * v[2] - q[2] * v[1], cy = q[2] * v[0] - q[0] * v[2], cz = q[0] * v[1] - q[1] * v[0];
  const float d = q[0] * v[0] + q[1] * v[1] + q[2] * v[2];
  o[0] = v[0] * s - cx * w * 2.0f + q[0] * d * 2.0f;
  o[1] = v[1] * s - cy * w * 2.0f + q[1] * d * 2.0f;
  o[2] = v[2] * s - cz * w * 2.0f + q[2] * d * 2.0f;
}

// Everything random or curriculum-dependent about one env's reset
// (a1_conditional.py:204-221 curriculum, :43-50 spawn, :82-87 push, :194-200 command).
struct ResetOut {
  float root[13], cmd[3], push[3], org[3];
  int64_t level;
};
DEV void a1_reset_draw(const ShfA1TaskParams& tp, int64_t gid, uint32_t cnt, const float* root_pos, const float* cmd_old,
                       const float* org_old, int64_t level, int64_t type, const float* torigins, ResetOut& R) {
  uint32_t r0[4], r1[4], r2[4];
  const uint32_t k0 = (uint32_t)tp.seed, k1 = (uint32_t)(tp.seed >> 32);
  philox4x32((uint32_t)gid, cnt, 0u, (uint32_t)(gid >> 32), k0, k1, r0);
  philox4x32((uint32_t)gid, cnt, 1u, (uint32_t)(gid >> 32), k0, k1, r1);
  philox4x32((uint32_t)gid, cnt, 2u, (uint32_t)(gid >> 32), k0, k1, r2);
  float og[3] = {org_old[0], org_old[1], org_old[2]};
  if (tp.curriculum) {
    const float dx = root_pos[0] - og[0], dy = root_pos[1] - og[1];
    const float dist = sqrtf(dx * dx + dy * dy);
    const int up = dist > tp.env_length / 2.0f;
    const float cn = sqrtf(cmd_old[0] * cmd_old[0] + cmd_old[1] * cmd_old[1]);
    const int down = (dist < cn * tp.max_episode_length_s * 0.5f) && !up;
    level += up - down;
    if (level >= tp.max_terrain_level) level = (int64_t)(r2[3] % (uint32_t)tp.max_terrain_level);
    else if (level < 0) level = 0;
    const float* to = torigins + ((size_t)level * tp.num_terrain_cols + (size_t)type) * 3;
    og[0] = to[0]; og[1] = to[1]; og[2] = to[2];
  }
  R.level = level;
#pragma unroll
  for (int k = 0; k < 3; k++) R.org[k] = og[k];
  R.root[0] = tp.default_pos[0] + og[0] + urange(r0[0], -tp.spawn_xy, tp.spawn_xy);
  R.root[1] = tp.default_pos[1] + og[1] + urange(r0[1], -tp.spawn_xy, tp.spawn_xy);
  R.root[2] = tp.default_pos[2] + og[2];
#pragma unroll
  for (int k = 0; k < 4; k++) R.root[3 + k] = tp.default_quat[k];
#pragma unroll
  for (int k = 0; k < 6; k++) R.root[7 + k] = 0.0f;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    R.push[k] = urange(r1[k], -tp.max_push_force, tp.max_push_force);
    R.cmd[k] = urange(r2[k], -1.0f, 1.0f);
  }
}

// scratch carved out of the (then idle) contact-point region of the env's LDS
#define SCR_BODY 0   /* nb*13 body_state staging (<= 224)      */
#define SCR_MH 224   /* measured heights (<= 192)              */
#define SCR_HIST 416 /* action history (nd*H <= 96)            */
#define SCR_ACT 512  /* clipped actions (<= 32)                */
#define SCR_OBS 544  /* observation staging                    */

// Everything of ShifuVecEnv.step after the physics and the rigid-body-state refresh (isaac_gym.py:412-433 get_heights,
// env.py:93-106 post_step, a1_conditional.py:131-144 compute_observations, train.py:12-14 HistoryRecorder.add), shared by
// the body-mapped and the chain-mapped A1 kernels.  On entry: L.dofb / L.root hold the post-physics state, L.xch the net
// contact forces (nb x 3), scr[SCR_HIST] the action history, scr[SCR_ACT] the clipped actions; the G lanes of the
// env's group call it together.
template <int G>
DEV void a1_post_step(const A1Args& A, const ShfModel* m, const ShfA1TaskParams& tp, const EnvLds& L, float* scr, int e,
                      int l, float* stats_lds, unsigned long long stats_step) {
  PHASE_BEGIN();
  const int n = A.S.n, epb = 256 / G;
  const int nb = m->nb, nd = m->nd, H = tp.num_history, P = tp.num_height_points;
  const int nobs = 12 + 2 * nd + nd * H + P;
  float* dof = A.S.dof + (size_t)e * nd * 2;
  float* root = A.S.root + (size_t)e * 13;
  const float gv[3] = {0.0f, 0.0f, -1.0f};
  // post_step's inputs (lane 0): requested here so that their round trips overlap the height scan
  float ps_bv[6] = {0, 0, 0, 0, 0, 0}, ps_cmd[3] = {0, 0, 0}, ps_sums[6] = {0, 0, 0, 0, 0, 0};
  int64_t ps_ep = 0, ps_level = 0, ps_type = 0;
  int32_t ps_rc = 0;
  if (l == 0) {
    // pre-physics base-frame velocities: written to base_vel at the top of the kernel by this lane and read back
    // here rather than held in six registers across the sub-steps
    const float* bv = A.base_vel + (size_t)e * 9;
#pragma unroll
    for (int k = 0; k < 6; k++) ps_bv[k] = bv[k];
#pragma unroll
    for (int k = 0; k < 3; k++) ps_cmd[k] = A.command[(size_t)e * 3 + k];
#pragma unroll
    for (int k = 0; k < 6; k++) ps_sums[k] = A.rew_sums[(size_t)k * n + e];
    ps_ep = A.ep_len[e]; ps_level = A.levels[e]; ps_type = A.types[e]; ps_rc = A.reset_count[e];
  }

  // get_heights (isaac_gym.py:412-433)
  {
    float qz = L.root[5], qw = L.root[6];
    const float nrm = rmaxf(sqrtf(qz * qz + qw * qw), 1e-9f);
    qz = qz / nrm; qw = qw / nrm;
    // HC sample points per lane and trip: their point loads, then their height loads, go out together
    constexpr int HC = 3;
    const float rx = L.root[0], ry = L.root[1];
    // `points / horizontal_scale` as torch evaluates it on a GPU tensor with a Python-float divisor: x * (1 / s)
    const float inv_hs = 1.0f / A.S.terr.hscale;
    for (int base = l; base < P; base += HC * G) {
      float bx[HC], by[HC], hh[HC];
#pragma unroll
      for (int k = 0; k < HC; k++) {
        const int i = base + k * G < P ? base + k * G : 0;
        bx[k] = A.hpoints[2 * i]; by[k] = A.hpoints[2 * i + 1];
      }
      int16_t h1[HC], h2[HC], h3[HC];
      if (A.S.terr.rows > 0) {
#pragma unroll
        for (int k = 0; k < HC; k++) {
          const float tx = (-qz * by[k]) * 2.0f, ty = (qz * bx[k]) * 2.0f;
          float px = bx[k] + qw * tx + (-qz * ty) + rx;
          float py = by[k] + qw * ty + (qz * tx) + ry;
          px += A.S.terr.border; py += A.S.terr.border;
          int ix = (int)truncf(px * inv_hs), iy = (int)truncf(py * inv_hs);
          ix = ix < 0 ? 0 : ix; ix = ix > A.S.terr.rows - 2 ? A.S.terr.rows - 2 : ix;
          iy = iy < 0 ? 0 : iy; iy = iy > A.S.terr.cols - 2 ? A.S.terr.cols - 2 : iy;
          const int16_t* p0 = A.S.heights + (size_t)ix * A.S.terr.cols + iy;
          h1[k] = p0[0]; h2[k] = p0[A.S.terr.cols]; h3[k] = p0[1];
        }
#pragma unroll
        for (int k = 0; k < HC; k++) {
          int16_t hm = h1[k] < h2[k] ? h1[k] : h2[k];
          hm = hm < h3[k] ? hm : h3[k];
          hh[k] = (float)hm * A.S.terr.vscale;
        }
      } else {
#pragma unroll
        for (int k = 0; k < HC; k++) hh[k] = 0.0f;
      }
#pragma unroll
      for (int k = 0; k < HC; k++) {
        const int i = base + k * G;
        if (i < P) {
          scr[SCR_MH + i] = hh[k];
          A.heights_out[(size_t)e * P + i] = hh[k];
        }
      }
    }
  }
  GROUP_SYNC();
  PHASE_MARK(14);

  // post_step (env.py:93-106): one lane runs the scalar bookkeeping
  unsigned long long* stats_row = nullptr;
  if (l == 0) {
    const float blv[3] = {ps_bv[0], ps_bv[1], ps_bv[2]}, bav[3] = {ps_bv[3], ps_bv[4], ps_bv[5]};
    const float* cf = L.xch;
    int64_t ep = ps_ep + 1;
    const float* fb = cf + 3 * tp.base_body;
    const int contact_term = sqrtf(fb[0] * fb[0] + fb[1] * fb[1] + fb[2] * fb[2]) > 1.0f;
    const int timeout = (float)ep > tp.max_episode_length;
    const int reset = timeout | contact_term;
    A.timeout[e] = (uint8_t)timeout;
    A.reset[e] = (uint8_t)reset;
    float cmd[3] = {ps_cmd[0], ps_cmd[1], ps_cmd[2]};
    const float* hist = scr + SCR_HIST;
    float rterm[6];
    {
      const float e0 = cmd[0] - blv[0], e1 = cmd[1] - blv[1];
      rterm[0] = 1.0f * exp_spec(-(e0 * e0 + e1 * e1) / 0.25f);
      const float e2 = cmd[2] - bav[2];
      rterm[1] = 0.5f * exp_spec(-(e2 * e2) / 0.25f);
      rterm[2] = -2.0f * (blv[2] * blv[2]) + -0.005f * (bav[0] * bav[0] + bav[1] * bav[1]);
      float first = 0.0f, second = 0.0f;
      for (int d = 0; d < nd; d++) {
        const float a0 = hist[d * H + 0], a1 = hist[d * H + 1], a2 = hist[d * H + 2];
        first += (a1 - a0) * (a1 - a0);
        const float t = a2 - 2.0f * a1 + a0;
        second += t * t;
      }
      rterm[3] = -0.005f * (first + second);
      float cnt = 0.0f;
      for (int k = 0; k < tp.num_leg_bodies; k++) {
        const float* f = cf + 3 * tp.leg_bodies[k];
        if (sqrtf(f[0] * f[0] + f[1] * f[1] + f[2] * f[2]) > 0.1f) cnt += 1.0f;
      }
      rterm[4] = -1.0f * cnt;
      float t2 = 0.0f;
      for (int d = 0; d < nd; d++) { const float t = L.dofb[d * DOF_STRIDE + 5]; t2 += t * t; }
      rterm[5] = -2e-5f * t2;
    }
    float rew = 0.0f;
    float sums[6];
#pragma unroll
    for (int k = 0; k < 6; k++) {
      sums[k] = ps_sums[k] + rterm[k];
      rew += rterm[k];
    }
    A.rew[e] = rew;
    float done[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int64_t level = ps_level;
    if (reset) {
      ResetOut R;
      a1_reset_draw(tp, A.env_off + e, (uint32_t)ps_rc, L.root, cmd, A.origins + (size_t)e * 3, level,
                    ps_type, A.torigins, R);
      level = R.level;
      A.levels[e] = level;
#pragma unroll
      for (int k = 0; k < 3; k++) A.origins[(size_t)e * 3 + k] = R.org[k];
#pragma unroll
      for (int k = 0; k < 6; k++) { done[k] = sums[k]; sums[k] = 0.0f; }
      done[7] = 1.0f;
      for (int d = 0; d < nd; d++) { L.dofb[d * DOF_STRIDE] = tp.default_dof_pos[d]; L.dofb[d * DOF_STRIDE + 1] = 0.0f; }
#pragma unroll
      for (int k = 0; k < 13; k++) L.root[k] = R.root[k];
#pragma unroll
      for (int k = 0; k < 3; k++) A.push[((size_t)e * nb + tp.base_body) * 3 + k] = R.push[k];
      ep = 0;
      for (int k = 0; k < nd * H; k++) scr[SCR_HIST + k] = 0.0f;
#pragma unroll
      for (int k = 0; k < 3; k++) { cmd[k] = R.cmd[k]; A.command[(size_t)e * 3 + k] = cmd[k]; }
      A.reset_count[e] = ps_rc + 1;
    }
    done[6] = (float)level;
#pragma unroll
    for (int k = 0; k < 6; k++) A.rew_sums[(size_t)k * n + e] = sums[k];
#pragma unroll
    for (int k = 0; k < 8; k++) A.done_sums[(size_t)k * n + e] = done[k];
    A.ep_len[e] = ep;
    {
      // extras["episode"] of this vec-step (env.py:149-158)
      long long sv[8];
#pragma unroll
      for (int k = 0; k < 6; k++) sv[k] = stats_fix(done[k]);
      sv[6] = (long long)level; sv[7] = reset ? 1ll : 0ll;
      const int first = (int)blockIdx.x * epb, eib = n - first < epb ? n - first : epb;
      stats_row = stats_contribute<8>(A.stats, stats_lds, sv, eib, stats_step);
    }
    const float co = tp.clip_obs;
    float* o = scr + SCR_OBS;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      o[k] = rclampf(cmd[k], -co, co);
      o[3 + k] = rclampf(blv[k], -co, co);
      o[6 + k] = rclampf(bav[k], -co, co);
      o[9 + k] = rclampf(gv[k], -co, co);
    }
  }
  GROUP_SYNC();
  PHASE_MARK(15);
  {
    // compute_observations (a1_conditional.py:131-144) staged in LDS, then one coalesced store
    const float co = tp.clip_obs;
    float* o = scr + SCR_OBS;
    if (l < nd) {
      o[12 + l] = rclampf(L.dofb[l * DOF_STRIDE] - tp.default_dof_pos[l], -co, co);
      o[12 + nd + l] = rclampf(L.dofb[l * DOF_STRIDE + 1], -co, co);
    }
    for (int i = l; i < nd * H; i += G) {
      const int h = i / nd, d = i % nd;
      o[12 + 2 * nd + i] = rclampf(scr[SCR_HIST + d * H + h], -co, co);
    }
    const float bz = L.root[2];
    for (int i = l; i < P; i += G)
      o[12 + 2 * nd + nd * H + i] = rclampf(rclampf(bz - 0.5f - scr[SCR_MH + i], -1.0f, 1.0f), -co, co);
  }
  GROUP_SYNC();
  // the block's statistics additions were issued in post_step; by now they have been acknowledged
  unsigned long long stats_tk = 0ull;
  if (l == 0) stats_tk = stats_ticket(stats_row);
  for (int i = l; i < nobs; i += G) A.obs[(size_t)e * nobs + i] = scr[SCR_OBS + i];
  // HistoryRecorder.add (train.py:12-14), after the observation was taken (Q12)
  for (int i = l; i < nd * H; i += G) {
    const int d = i / H, h = i % H;
    A.history[(size_t)e * nd * H + i] = h == 0 ? scr[SCR_ACT + d] : scr[SCR_HIST + d * H + h - 1];
  }
  for (int i = l; i < 2 * nd; i += G) dof[i] = L.dofb[(i >> 1) * DOF_STRIDE + (i & 1)];
  if (l < 13) root[l] = L.root[l];
  if (l == 0) {
    const float Ts = tp.max_episode_length_s;
    stats_finish<8>(A.stats, stats_row, stats_tk, stats_step, [n, Ts](const long long* t, float* o) {
      const float c = (float)t[7];
#pragma unroll
      for (int k = 0; k < 6; k++) {
        const float sum = (float)t[k] * (1.0f / STATS_FIX);
        o[k] = sum;
        o[8 + k] = c > 0.0f ? sum / c / Ts : 0.0f;
      }
      o[6] = (float)t[6]; o[7] = c;
      o[14] = o[6] / (float)n; o[15] = (float)n;
    });
  }
  PHASE_MARK(16);
}
