// shf_kernels.h -- the env-step kernels (templates) of the body-per-lane and wave-specialised mappings.  Included by shf_api.hip
// (the C ABI: launches) and by the shf_k_*.hip translation units, each of which instantiates one family explicitly so that
// the families compile side by side (shifu_amd/build.py); shf_api.hip sees them as extern templates (shf_kernel_list.h).
#pragma once
#include <hip/hip_runtime.h>
#include "shf_device.h"
#include "shf_task.h"
#include "shf_arm.h"

// gym.simulate: one sub-step for every env
// HARD: the velocity-level contact solve (ShfSimParams.solver == SHF_SOLVER_PGS; csrc/shf_hard.h): 32 lanes per env, and
// hard_total_slots() contact slots per env (the solve's records and response matrix live inside the slot region, csrc/shf_hard.h)
// (WT threads per workgroup: 256, or 512 for k_sim_step_pgs_wide below)
// EXT: the convex narrow phase compiled in (hulls, face manifolds: csrc/shf_hull.h) -- scenes with ShfScene.flags or hulls
template <int G, bool BOX, bool SELF, bool LINK, bool HARD, int WT, bool EXT = false>
DEV void sim_step_body(const SimArgs& A) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const ShfScene* scene = BOX ? stage_scene<WT>(A.scene, smem + MODEL_WORDS) : nullptr;
  const ShfModel* m = stage_model<WT>(A.model, smem);
  const int epb = WT / G, es = threadIdx.x / G, l = threadIdx.x % G;
  const int e = blockIdx.x * epb + es;
  if (e >= A.n) return;
  const int nbx = BOX ? A.nboxes : 0, actors = 1 + nbx;
  const int nb = m->nb, nd = m->nd, nbt = nb + nbx;
  const int nslots = m->np + (BOX ? box_slots(slot_lay<DynScene>(m, scene)) : 0) + (SELF ? SHF_MAX_SELF_CONTACTS : 0) + (LINK ? 2 * SHF_MAX_LINK_CONTACTS : 0);
  EnvLds L = env_lds_carve(smem + MODEL_WORDS + (BOX ? SCENE_WORDS : 0) + es * env_lds_words(nbt, nd, HARD ? hard_total_slots(nslots, LINK) : nslots, 0, actors),
                           nbt, nd, nslots, actors);
  float* dof = A.dof + (size_t)e * nd * 2;
  float* root = A.root + (size_t)e * actors * 13;
  for (int i = l; i < 2 * nd; i += G) L.dofb[(i >> 1) * DOF_STRIDE + (i & 1)] = dof[i];
  for (int i = l; i < 13 * actors; i += G) L.root[i] = root[i];
  if (l < nd) L.dofb[l * DOF_STRIDE + 5] = A.effort ? A.effort[(size_t)e * nd + l] : 0.0f;
  GROUP_SYNC();
  StepCtx C;
  C.m = m; C.sp = A.sp; C.terr.t = A.terr; C.terr.h = A.heights; C.scene = scene;
  C.dropped = env_dropped(A.dropped, A.sp, e);
  C.mscale = A.mscale ? A.mscale + (size_t)e * nb : nullptr;
  C.hulls = A.hulls;
  const float mu = A.friction ? A.friction[e] : 1.0f;
  LaneModel M;
  lane_model_load<DynDims>(m, l, M);
  LanePoints<1> P;   // unused: point count only known at run time
  substep<G, BOX, DynDims, !BOX, LaneModel, DynScene, SELF, LINK, HARD, EXT>(C, L, l, M, P, A.pos_tgt ? A.pos_tgt + (size_t)e * nd : nullptr,
                  A.vel_tgt ? A.vel_tgt + (size_t)e * nd : nullptr, A.body_force ? A.body_force + (size_t)e * nbt * 3 : nullptr, mu, L.xch,
                  BoxLane(), (A.body_force && A.body_force_pos) ? A.body_force_pos + (size_t)e * nbt * 3 : nullptr);
  GROUP_SYNC();
  for (int i = l; i < 2 * nd; i += G) dof[i] = L.dofb[(i >> 1) * DOF_STRIDE + (i & 1)];
  for (int i = l; i < 13 * actors; i += G) root[i] = L.root[i];
  for (int i = l; i < 3 * nbt; i += G) A.contact[(size_t)e * nbt * 3 + i] = L.xch[i];
}
template <int G, bool BOX, bool SELF, bool LINK = false, bool HARD = false, bool EXT = false>
__global__ __launch_bounds__(256) void k_sim_step(SimArgs A) {
  sim_step_body<G, BOX, SELF, LINK, HARD, 256, EXT>(A);
}
// gym.simulate under the velocity-level solve for a scene with box actors and link contacts, sixteen envs per workgroup of 512
// threads (as k_abb_step_pgs_wide: where their LDS fits one CU and eight envs would leave the CU to one workgroup)
template <bool SELF>
__global__ __launch_bounds__(512) void k_sim_step_pgs_wide(SimArgs A) {
  sim_step_body<32, true, SELF, true, true, 512>(A);
}

// gym.refresh_rigid_body_state_tensor (+ refresh_jacobian_tensors for fixed bases) for one env:
// body rows from forward kinematics, box rows copied from their root rows; jacobian
// (nb-1, 6, nd), rows [linear; angular] of each link origin (robot.py:125-128).
// `body_state` / `jacobian` point at this env's slices (either may be null).  L.xch is scratch.
template <int G, class DM = DynDims>
DEV void refresh_body_jac(const ShfModel* m, const EnvLds& L, int l, int actors, float* body_state, float* jacobian,
                          float* keep_xy = nullptr, int keep_body = 0) {
  const int nb = DM::nb(m), nd = DM::nd(m);
  BodyRegs B;
  LaneModel M;
  lane_model_load<DM>(m, l, M);
  kinematics<G, DM>(m, L, l, M, B);
  if (body_state) {
    if (l < nb) {
      float* o = L.xch + 13 * l;
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
    if (keep_xy && l == 0) { keep_xy[0] = L.xch[13 * keep_body]; keep_xy[1] = L.xch[13 * keep_body + 1]; }
    for (int i = l; i < 13 * nb; i += G) body_state[i] = L.xch[i];
    for (int i = l; i < 13 * (actors - 1); i += G) body_state[nb * 13 + i] = L.root[13 + i];
    GROUP_SYNC();
  }
  if (jacobian && m->fixed_base) {
    if (l < nb) {
      float* o = L.xch + 6 * l;
#pragma unroll
      for (int k = 0; k < 6; k++) o[k] = B.S[k];
    }
    GROUP_SYNC();
    if (l >= 1 && l < nb) {
      float* J = jacobian + (size_t)(l - 1) * 6 * nd;
      for (int k = 0; k < 6 * nd; k++) J[k] = 0.0f;
      for (int b = l; b > 0; b = m->parent[b]) {
        const int d = m->dof[b];
        if (d < 0) continue;
        const float* S = L.xch + 6 * b;
        const float ax[3] = {S[0], S[1], S[2]};
        float t[3];
        cross3(ax, B.p, t);
#pragma unroll
        for (int k = 0; k < 3; k++) { J[k * nd + d] = t[k] + S[3 + k]; J[(3 + k) * nd + d] = S[k]; }
      }
    }
    GROUP_SYNC();
  }
}

template <int G>
__global__ __launch_bounds__(256) void k_body_state(const ShfModel* gm, int n, const float* dof, const float* root,
                                                    int actors, float* body_state, float* jacobian) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const ShfModel* m = stage_model(gm, smem);
  const int epb = 256 / G, es = threadIdx.x / G, l = threadIdx.x % G;
  const int e = blockIdx.x * epb + es;
  if (e >= n) return;
  const int nb = m->nb, nd = m->nd, np = m->np, nbt = nb + actors - 1;
  EnvLds L = env_lds_carve(smem + MODEL_WORDS + es * env_lds_words(nb, nd, np, 0, actors), nb, nd, np, actors);
  for (int i = l; i < 2 * nd; i += G) L.dofb[(i >> 1) * DOF_STRIDE + (i & 1)] = dof[(size_t)e * nd * 2 + i];
  for (int i = l; i < 13 * actors; i += G) L.root[i] = root[(size_t)e * actors * 13 + i];
  GROUP_SYNC();
  refresh_body_jac<G>(m, L, l, actors, body_state ? body_state + (size_t)e * nbt * 13 : nullptr,
                      jacobian ? jacobian + (size_t)e * (nb - 1) * 6 * nd : nullptr);
}

#ifdef SHF_DEFINE_SMALL_KERNELS
__global__ void k_commit_rows(const float* src, float* dst, const int32_t* idx, int n, int row_words, int idx_div,
                              int rows_per_idx_words, int num_rows) {
  // one block per index entry: copies `rows_per_idx_words` floats of row (idx/idx_div).  Indices come from user
  // tensors (set_*_tensor_indexed): an entry outside [0, num_rows * idx_div) is skipped rather than written through.
  const int i = blockIdx.x;
  if (i >= n) return;
  const int32_t a = idx[i];
  if (a < 0 || a / idx_div >= num_rows) return;
  const size_t r = (size_t)(a / idx_div) * row_words;
  for (int k = threadIdx.x; k < rows_per_idx_words; k += blockDim.x) dst[r + k] = src[r + k];
}
// the three indexed commits of one reset (shf_sim_commit_reset): blocks [0, n_root) copy root rows, the next n_dof the dof-state
// rows, the last n_dof the position-target rows -- k_commit_rows three times over
__global__ void k_commit_reset(const float* root_src, float* root_dst, const int32_t* root_idx, int n_root, int root_rows,
                               const float* dof_src, float* dof_dst, const float* tgt_src, float* tgt_dst, const int32_t* dof_idx,
                               int n_dof, int nd, int actors, int num_envs) {
  int i = blockIdx.x;
  if (i < n_root) {
    const int32_t a = root_idx[i];
    if (a < 0 || a >= root_rows) return;
    const size_t r = (size_t)a * 13;
    for (int k = threadIdx.x; k < 13; k += blockDim.x) root_dst[r + k] = root_src[r + k];
    return;
  }
  i -= n_root;
  const bool state = i < n_dof;
  if (!state) i -= n_dof;
  if (i >= n_dof) return;
  const int32_t a = dof_idx[i];
  if (a < 0 || a / actors >= num_envs) return;
  const int words = state ? nd * 2 : nd;
  const float* src = state ? dof_src : tgt_src;
  float* dst = state ? dof_dst : tgt_dst;
  if (!src) return;
  const size_t r = (size_t)(a / actors) * words;
  for (int k = threadIdx.x; k < words; k += blockDim.x) dst[r + k] = src[r + k];
}
#else
__global__ void k_commit_rows(const float* src, float* dst, const int32_t* idx, int n, int row_words, int idx_div,
                              int rows_per_idx_words, int num_rows);
__global__ void k_commit_reset(const float* root_src, float* root_dst, const int32_t* root_idx, int n_root, int root_rows,
                               const float* dof_src, float* dof_dst, const float* tgt_src, float* tgt_dst, const int32_t* dof_idx,
                               int n_dof, int nd, int actors, int num_envs);
#endif

#ifdef SHF_DEFINE_SMALL_KERNELS
__global__ void k_reset_all(const ShfModel* gm, int n, int actors, const float* default_root, const float* default_dof,
                            const float* origins, float* dof_a, float* dof_b, float* root_a, float* root_b) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  const int nd = gm->nd;
  for (int d = 0; d < nd; d++) {
    const size_t o = ((size_t)e * nd + d) * 2;
    dof_a[o] = default_dof[d]; dof_a[o + 1] = 0.0f;
    dof_b[o] = default_dof[d]; dof_b[o + 1] = 0.0f;
  }
  for (int a = 0; a < actors; a++) {
    const size_t o = ((size_t)e * actors + a) * 13;
    for (int k = 0; k < 13; k++) {
      float v = default_root[a * 13 + k];
      if (k < 3 && origins) v += origins[(size_t)e * 3 + k];
      root_a[o + k] = v; root_b[o + k] = v;
    }
  }
}
#else
__global__ void k_reset_all(const ShfModel* gm, int n, int actors, const float* default_root, const float* default_dof,
                            const float* origins, float* dof_a, float* dof_b, float* root_a, float* root_b);
#endif



// ShifuVecEnv.reset_idx(arange(N)) (env.py:108-130) for the A1 task: one thread per env
#ifdef SHF_DEFINE_SMALL_KERNELS
__global__ void k_a1_reset_all(A1Args A) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = A.S.n;
  if (e >= n) return;
  const ShfA1TaskParams& tp = *A.tp;
  const ShfModel* m = A.S.model;
  const int nd = m->nd, nb = m->nb, H = tp.num_history;
  float* root = A.S.root + (size_t)e * 13;
  float cmd[3] = {A.command[(size_t)e * 3], A.command[(size_t)e * 3 + 1], A.command[(size_t)e * 3 + 2]};
  ResetOut R;
  a1_reset_draw(tp, A.env_off + e, (uint32_t)A.reset_count[e], root, cmd, A.origins + (size_t)e * 3, A.levels[e],
                A.types[e], A.torigins, R);
  A.levels[e] = R.level;
  for (int k = 0; k < 3; k++) A.origins[(size_t)e * 3 + k] = R.org[k];
  for (int k = 0; k < 6; k++) { A.done_sums[(size_t)k * n + e] = A.rew_sums[(size_t)k * n + e]; A.rew_sums[(size_t)k * n + e] = 0.0f; }
  A.done_sums[(size_t)6 * n + e] = (float)R.level;
  A.done_sums[(size_t)7 * n + e] = 1.0f;
  for (int d = 0; d < nd; d++) {
    A.S.dof[((size_t)e * nd + d) * 2] = tp.default_dof_pos[d];
    A.S.dof[((size_t)e * nd + d) * 2 + 1] = 0.0f;
  }
  for (int k = 0; k < 13; k++) root[k] = R.root[k];
  for (int k = 0; k < 3; k++) A.push[((size_t)e * nb + tp.base_body) * 3 + k] = R.push[k];
  A.ep_len[e] = 0;
  A.reset[e] = 1;
  for (int k = 0; k < nd * H; k++) A.history[(size_t)e * nd * H + k] = 0.0f;
  for (int k = 0; k < 3; k++) A.command[(size_t)e * 3 + k] = R.cmd[k];
  A.reset_count[e] += 1;
}
#else
__global__ void k_a1_reset_all(A1Args A);
#endif


// Occupancy: at one wavefront per env (G = 64) 4096 envs are 4096 waves = 4 per SIMD, so
// the kernel is held to 128 VGPRs (measured 0.147 ms vs 0.165 ms at 165 VGPRs / 3 waves,
// profiles/r01_*); at G = 32 the grid is 2 waves per SIMD and the unconstrained
// allocation is faster (0.104 ms vs 0.112 ms).
template <int G, class DM, bool TW, bool SELF = false>
DEV void a1_step_body(const A1Args& A) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  PHASE_BEGIN();
  float* stats_lds = smem + MODEL_WORDS + TASK_WORDS;
  stats_block_init(stats_lds);
  const unsigned long long stats_step = stats_step_load(A.stats);
  const int epb = 256 / G, es = threadIdx.x / G, l = threadIdx.x % G;
  const int e = blockIdx.x * epb + es;
  const int n = A.S.n;
  // compile-time dimensions: the env's state and action loads go out before the model is staged, so their round trip
  // overlaps the staging copy instead of following its barrier
  constexpr bool EARLY = DM::NPC > 0;
  float pre_dof = 0.0f, pre_root = 0.0f, pre_act = 0.0f;
  if constexpr (EARLY) {
    static_assert(!EARLY || G >= 24, "one state word per lane");
    const int ndc = DM::nd(nullptr);
    if (e < n) {
      if (l < 2 * ndc) pre_dof = A.S.dof[(size_t)e * ndc * 2 + l];
      if (l < 13) pre_root = A.S.root[(size_t)e * 13 + l];
      if (l < ndc) pre_act = raw_action(A, e, l, ndc, stats_step);
    }
  }
  stage_block<(int)sizeof(ShfA1TaskParams)>(A.tp, smem + MODEL_WORDS);
  const ShfModel* m = stage_model(A.S.model, smem);
  const ShfA1TaskParams& tp = *reinterpret_cast<const ShfA1TaskParams*>(smem + MODEL_WORDS);
  if (e >= n) return;
  const int nb = DM::nb(m), nd = DM::nd(m), np = DM::np(m), H = tp.num_history, P = tp.num_height_points;
  const int nobs = 12 + 2 * nd + nd * H + P;
  // post-physics scratch reuses the contact-point region (last in the carve)
  const int nslots = np + (SELF ? SHF_MAX_SELF_CONTACTS : 0);
  const int env_words = env_lds_words(nb, nd, nslots, SCR_OBS + nobs);
  EnvLds L = env_lds_carve(smem + MODEL_WORDS + TASK_WORDS + STATS_LDS_WORDS + es * env_words, nb, nd, nslots);
  float* scr = L.pt;

  float* dof = A.S.dof + (size_t)e * nd * 2;
  float* root = A.S.root + (size_t)e * 13;
  float act = 0.0f;
  if constexpr (EARLY) {
    if (l < 2 * nd) L.dofb[(l >> 1) * DOF_STRIDE + (l & 1)] = pre_dof;
    if (l < 13) L.root[l] = pre_root;
    if (l < nd) act = rclampf(pre_act * tp.action_scale, -tp.clip_actions, tp.clip_actions);
  } else {
    for (int i = l; i < 2 * nd; i += G) L.dofb[(i >> 1) * DOF_STRIDE + (i & 1)] = dof[i];
    if (l < 13) L.root[l] = root[l];
    if (l < nd) act = rclampf(raw_action(A, e, l, nd, stats_step) * tp.action_scale, -tp.clip_actions, tp.clip_actions);
  }
  if (l < nd) A.actions[(size_t)e * nd + l] = act;
  GROUP_SYNC();

  // Q2: base-frame velocities from the pre-physics root state (robot.py:222-229)
  const float gv[3] = {0.0f, 0.0f, -1.0f};
  if (l == 0) {
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
  C.dropped = env_dropped(A.S.dropped, A.S.sp, e);
  C.mscale = A.S.mscale ? A.S.mscale + (size_t)e * nb : nullptr;
  C.hulls = A.S.hulls;
  const float mu = A.S.friction[e];
  const int nsub = tp.decimation + (tp.extra_substep ? 1 : 0);
  // (with self-collision the per-lane model constants stay in LDS: the pair tests need the registers)
  LaneModelT<(G < 64) && !SELF> M;
  lane_model_load<DM>(m, l, M);
  LanePoints<LANE_ROUNDS(G, DM)> LP;
  lane_points_load<G>(m, np, l, LP);
  PHASE_MARK(11);
  for (int it = 0; it < nsub; it++) {
    if (it < tp.decimation && l < nd) {
      float* D = L.dofb + l * DOF_STRIDE;
      const float t = tp.p_gain[l] * (act + tp.default_dof_pos[l] - D[0]) - tp.d_gain[l] * D[1];
      D[5] = rclampf(t, -m->effort[l], m->effort[l]);
    }
    GROUP_SYNC();
    substep<G, false, DM, TW, LaneModelT<(G < 64) && !SELF>, DynScene, SELF>(C, L, l, M, LP, nullptr, nullptr,
               (it == tp.decimation) ? A.push + (size_t)e * nb * 3 : nullptr, mu, (it == nsub - 1) ? L.xch : nullptr);
  }
  GROUP_SYNC();
  PHASE_RESET();
  if (l < nd) A.torques[(size_t)e * nd + l] = L.dofb[l * DOF_STRIDE + 5];
  for (int i = l; i < 3 * nb; i += G) A.S.contact[(size_t)e * nb * 3 + i] = L.xch[i];
  // the contact-point region is idle from here on: it becomes scratch
  for (int i = l; i < nd * H; i += G) scr[SCR_HIST + i] = A.history[(size_t)e * nd * H + i];
  if (l < nd) scr[SCR_ACT + l] = act;
  PHASE_MARK(12);
  body_states<G, DM>(m, L, l, M, scr + SCR_BODY);
  for (int i = l; i < 13 * nb; i += G) A.body_state[(size_t)e * nb * 13 + i] = scr[SCR_BODY + i];
  PHASE_RESET();

  a1_post_step<G>(A, m, tp, L, scr, e, l, stats_lds, stats_step);
}
template <int G, class DM>
__global__ __launch_bounds__(256, (G == 64 ? 4 : 1)) void k_a1_step(A1Args A) { a1_step_body<G, DM, (G < 64)>(A); }
// The default instantiation (A1, two envs per wavefront) has to stay within 256 VGPRs: two blocks per CU = two
// waves per SIMD keep all 4096 envs resident.  Launch bounds of (256, 2) would say the same but also switch the
// scheduler to its occupancy-preserving mode (measured +13 %), so the register cap is given directly.
#ifdef SHF_DEFINE_A1_KERNELS
__global__ __launch_bounds__(256, 1) __attribute__((amdgpu_num_vgpr(256))) void k_a1_step_a1_g32(A1Args A) {
  a1_step_body<32, A1Dims, false>(A);
}
#else
__global__ __launch_bounds__(256, 1) __attribute__((amdgpu_num_vgpr(256))) void k_a1_step_a1_g32(A1Args A);
#endif
// The same with self-collision (ShfModel.self_collide: capsule pairs of the articulation), its own instantiations so
// that the default one keeps its register budget.
template <int G, class DM>
__global__ __launch_bounds__(256, (G == 64 ? 4 : 1)) void k_a1_step_self(A1Args A) { a1_step_body<G, DM, (G < 64), true>(A); }
// A1 with self-collision at two envs per wavefront, held to 256 VGPRs for the same reason as k_a1_step_a1_g32
#ifdef SHF_DEFINE_A1_KERNELS
__global__ __launch_bounds__(256, 1) __attribute__((amdgpu_num_vgpr(256))) void k_a1_step_self_a1_g32(A1Args A) {
  a1_step_body<32, A1Dims, false, true>(A);
}
#else
__global__ __launch_bounds__(256, 1) __attribute__((amdgpu_num_vgpr(256))) void k_a1_step_self_a1_g32(A1Args A);
#endif

// ------------------------------------------------------ fused ABB step --
struct AbbArgs {
  SimArgs S;
  const ShfAbbTaskParams* tp;  // device copy
  int64_t env_off;
  const float* raw_actions;
  float *actions, *obs, *rew;
  uint8_t *reset, *timeout, *success;
  int64_t* ep_len;
  float *rew_sums, *dof_targets;
  int32_t* reset_count;
  float* done_sums;
  float *body_state, *jacobian;
  StatsArgs stats;
};
#define ABB_WORDS ((int)((sizeof(ShfAbbTaskParams) / 4 + 3) & ~3))

// shifu/utils/torch_utils.py:12-33 (xyzw)
DEV void quat_mul_ref(const float* a, const float* b, float* o) {
  const float x1 = a[0], y1 = a[1], z1 = a[2], w1 = a[3], x2 = b[0], y2 = b[1], z2 = b[2], w2 = b[3];
  const float ww = (z1 + x1) * (x2 + y2);
  const float yy = (w1 - y1) * (w2 + z2);
  const float zz = (w1 + y1) * (w2 - z2);
  const float xx = ww + yy + zz;
  const float qq = 0.5f * (xx + (z1 - x1) * (x2 - y2));
  o[3] = qq - ww + (z1 - y1) * (y2 - z2);
  o[0] = qq - xx + (x1 + w1) * (x2 + w2);
  o[1] = qq - yy + (w1 - x1) * (y2 + z2);
  o[2] = qq - zz + (z1 + y1) * (w2 - x2);
}

// per-env reset of arm, table, cube, goal into the LDS copies (a_prior_stage.py:24-58, robot.py:74-86)
DEV void abb_reset_env(const ShfAbbTaskParams& tp, int nd, int nbx, int64_t gid, uint32_t cnt, float* dofb, float* rootl) {
  uint32_t r0[4], r1[4];
  const uint32_t k0 = (uint32_t)tp.seed, k1 = (uint32_t)(tp.seed >> 32);
  philox4x32((uint32_t)gid, cnt, 0u, (uint32_t)(gid >> 32), k0, k1, r0);
  philox4x32((uint32_t)gid, cnt, 1u, (uint32_t)(gid >> 32), k0, k1, r1);
  for (int d = 0; d < nd; d++) { dofb[d * DOF_STRIDE] = tp.default_dof_pos[d]; dofb[d * DOF_STRIDE + 1] = 0.0f; }
  for (int k = 0; k < 7; k++) rootl[k] = tp.actor_default[0][k];
  for (int k = 7; k < 13; k++) rootl[k] = 0.0f;
  for (int b = 0; b < nbx; b++) {
    float* rb = rootl + 13 * (1 + b);
    for (int k = 7; k < 13; k++) rb[k] = 0.0f;
    if (b + 1 == tp.cube_actor || b + 1 == tp.goal_actor) {
      const bool cube = b + 1 == tp.cube_actor;
      uint32_t rr[4];                    // (values, not a pointer to one of the two arrays: that put both on the stack)
#pragma unroll
      for (int k = 0; k < 4; k++) rr[k] = cube ? r0[k] : r1[k];
      const float* lo = cube ? tp.cube_lo : tp.goal_lo;
      const float* hi = cube ? tp.cube_hi : tp.goal_hi;
      for (int k = 0; k < 3; k++) rb[k] = urange(rr[k], lo[k], hi[k]);
      const float yaw = urange(rr[3], -3.14159265358979323846f, 3.14159265358979323846f);
      float sn, cs;
      sincos_spec(yaw * 0.5f, &sn, &cs);
      rb[3] = 0.0f; rb[4] = 0.0f; rb[5] = sn; rb[6] = cs;
    } else {
      for (int k = 0; k < 7; k++) rb[k] = tp.actor_default[b + 1][k];
    }
  }
}

// AbbRobot.step (shifu/units/robot.py:103-160) on one lane: EE-delta -> clip -> damped least squares on the Jacobian tensor
// (possibly stale pose) -> POS targets of this env step (tgtl: LDS; also written to the dof_targets tensor)
// q, qstride: the joint positions (LDS dof block, DOF_STRIDE; or this env's rows of the dof_state tensor, 2)
DEV void abb_ik_targets(const AbbArgs& A, const ShfAbbTaskParams& tp, const float* q, int qstride, int e, int nd, const float* bstate,
                        const float* jac, float* tgtl, unsigned long long step) {
    float act[3], dpose[6], eq[4], cc[4], qr[4];
    const float* ee = bstate + 13 * tp.ee_body;
    float eep[3] = {ee[0], ee[1], ee[2]};
#pragma unroll
    for (int k = 0; k < 4; k++) eq[k] = ee[3 + k];
#pragma unroll
    for (int k = 0; k < 3; k++) {
      act[k] = rclampf(raw_action(A, e, k, 3, step), -tp.clip_actions, tp.clip_actions);
      A.actions[(size_t)e * 3 + k] = act[k];
      const float tar = rclampf(eep[k] + act[k] * tp.ee_velocity * tp.env_dt, tp.min_ee_pos[k], tp.max_ee_pos[k]);
      dpose[k] = tar - eep[k];
    }
    cc[0] = -eq[0]; cc[1] = -eq[1]; cc[2] = -eq[2]; cc[3] = eq[3];
    const float tq[4] = {tp.target_quat[0], tp.target_quat[1], tp.target_quat[2], tp.target_quat[3]};
    quat_mul_ref(tq, cc, qr);
    const float sg = qr[3] > 0.0f ? 1.0f : (qr[3] < 0.0f ? -1.0f : 0.0f);
#pragma unroll
    for (int k = 0; k < 3; k++) dpose[3 + k] = qr[k] * sg;
    const float* J = jac + (size_t)(tp.ee_body - 1) * 6 * nd;
    float Am[21], neg[6], x[6];
    const float lam2 = tp.ik_damping * tp.ik_damping;
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
      for (int j = i; j < 6; j++) {
        float acc = J[i * nd] * J[j * nd];
        for (int d = 1; d < nd; d++) acc = fmaf(J[i * nd + d], J[j * nd + d], acc);
        Am[SYM(i, j)] = (i == j) ? acc + lam2 : acc;
      }
#pragma unroll
    for (int k = 0; k < 6; k++) neg[k] = -dpose[k];
    ldlt_solve6(Am, neg, x);
    for (int d = 0; d < nd; d++) {
      float u = J[d] * x[0];
#pragma unroll
      for (int k = 1; k < 6; k++) u = fmaf(J[k * nd + d], x[k], u);
      const float t = q[d * qstride] + u;
      tgtl[d] = t;
      A.dof_targets[(size_t)e * nd + d] = t;
    }
  }

// Everything of the AbbPushBox env step after the sub-steps: contact-force copy-out, state refresh (body states, Jacobian),
// then abb_post_step.
template <int G, class DM>
DEV void abb_post_step(const AbbArgs& A, const ShfAbbTaskParams& tp, const ShfModel* m, const EnvLds& L, float* rootl, int l, int e,
                       int epb, int nbx, float* tgtl, float* stats_lds, unsigned long long stats_step);
template <int G, class DM>
DEV void abb_after_physics(const AbbArgs& A, const ShfAbbTaskParams& tp, const ShfModel* m, const EnvLds& L, int l, int e, int epb,
                           int nbx, float* tgtl, float* stats_lds, unsigned long long stats_step) {
  const int actors = 1 + nbx;
  const int nb = DM::nb(m), nd = DM::nd(m), nbt = nb + nbx;
  float* bstate = A.body_state + (size_t)e * nbt * 13;
  float* jac = A.jacobian + (size_t)e * (nb - 1) * 6 * nd;
  PHASE_BEGIN();
  GROUP_SYNC();
  for (int i = l; i < 3 * nbt; i += G) A.S.contact[(size_t)e * nbt * 3 + i] = L.xch[i];
  GROUP_SYNC();
  refresh_body_jac<G, DM>(m, L, l, actors, bstate, jac, tgtl + nd, tp.ee_body);
  PHASE_MARK(13);
  abb_post_step<G, DM>(A, tp, m, L, L.root, l, e, epb, nbx, tgtl, stats_lds, stats_step);
}
// post_step on one lane (env.py:93-106, a_prior_stage.py:97-135), statistics, state stores.  tgtl[nd], tgtl[nd + 1]: this
// step's end-effector x, y.  rootl: the root-state rows in LDS that a reset rewrites and the stores read (L.root, or a
// private copy when another wave still reads L.root).
template <int G, class DM>
DEV void abb_post_step(const AbbArgs& A, const ShfAbbTaskParams& tp, const ShfModel* m, const EnvLds& L, float* rootl, int l, int e,
                       int epb, int nbx, float* tgtl, float* stats_lds, unsigned long long stats_step) {
  const int n = A.S.n, actors = 1 + nbx;
  const int nd = DM::nd(m);
  float* dof = A.S.dof + (size_t)e * nd * 2;
  float* root = A.S.root + (size_t)e * actors * 13;
  PHASE_BEGIN();
  // post_step on one lane (env.py:93-106, a_prior_stage.py:97-135)
  unsigned long long* stats_row = nullptr;
  if (l == 0) {
    int64_t ep = A.ep_len[e] + 1;
    const float* cube = rootl + 13 * tp.cube_actor;
    const float* goal = rootl + 13 * tp.goal_actor;
    // this step's ee position, parked in LDS by refresh_body_jac
    const float eex = tgtl[nd], eey = tgtl[nd + 1];
    const int timeout = (float)ep > tp.max_episode_length;
    const float gx = goal[0] - cube[0], gy = goal[1] - cube[1];
    const float dgoal = sqrtf(gx * gx + gy * gy);
    const int success = dgoal < 0.02f;
    const int outbound = cube[0] < tp.min_ee_pos[0] || cube[1] < tp.min_ee_pos[1] || cube[0] > tp.max_ee_pos[0] ||
                         cube[1] > tp.max_ee_pos[1] || eex < tp.min_ee_pos[0] || eey < tp.min_ee_pos[1] ||
                         eex > tp.max_ee_pos[0] || eey > tp.max_ee_pos[1];
    const int reset = timeout | outbound | success;
    A.timeout[e] = (uint8_t)timeout; A.success[e] = (uint8_t)success; A.reset[e] = (uint8_t)reset;
    const float ex = eex - cube[0], ey = eey - cube[1];
    const float eobj = sqrtf(ex * ex + ey * ey);
    const float r0 = (eobj < 0.1f) ? exp_spec(-(dgoal * dgoal) / 0.05f) : 0.0f;
    const float r1 = success ? 200.0f : 0.0f;
    float sums[2] = {A.rew_sums[e] + r0, A.rew_sums[(size_t)n + e] + r1};
    A.rew[e] = r0 + r1;
    float done[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (reset) {
      done[0] = sums[0]; done[1] = sums[1]; done[2] = success ? 1.0f : 0.0f; done[3] = 1.0f;
      sums[0] = sums[1] = 0.0f;
      abb_reset_env(tp, nd, nbx, A.env_off + e, (uint32_t)A.reset_count[e], L.dofb, rootl);
      ep = 0;
      A.reset_count[e] += 1;
    }
    A.rew_sums[e] = sums[0]; A.rew_sums[(size_t)n + e] = sums[1];
#pragma unroll
    for (int k = 0; k < 4; k++) A.done_sums[(size_t)k * n + e] = done[k];
    A.ep_len[e] = ep;
    {
      // extras["episode"] (env.py:149-158 + episode_log, a_prior_stage.py:94-95)
      const long long sv[4] = {stats_fix(done[0]), stats_fix(done[1]), (reset && success) ? 1ll : 0ll, reset ? 1ll : 0ll};
      const int first = (int)blockIdx.x * epb, eib = n - first < epb ? n - first : epb;
      stats_row = stats_contribute<4>(A.stats, stats_lds, sv, eib, stats_step);
    }
    const float co = tp.clip_obs;
    float* o = A.obs + (size_t)e * 6;
    o[0] = rclampf(rootl[13 * tp.cube_actor], -co, co); o[1] = rclampf(rootl[13 * tp.cube_actor + 1], -co, co);
    o[2] = rclampf(rootl[13 * tp.goal_actor], -co, co); o[3] = rclampf(rootl[13 * tp.goal_actor + 1], -co, co);
    o[4] = rclampf(eex, -co, co); o[5] = rclampf(eey, -co, co);
  }
  GROUP_SYNC();
  PHASE_MARK(15);
  unsigned long long stats_tk = 0ull;
  if (l == 0) stats_tk = stats_ticket(stats_row);
  for (int i = l; i < 2 * nd; i += G) dof[i] = L.dofb[(i >> 1) * DOF_STRIDE + (i & 1)];
  for (int i = l; i < 13 * actors; i += G) root[i] = rootl[i];
  if (l == 0) {
    const float Ts = tp.max_episode_length_s;
    stats_finish<4>(A.stats, stats_row, stats_tk, stats_step, [n, Ts](const long long* t, float* o) {
      const float c = (float)t[3];
      const float s0 = (float)t[0] * (1.0f / STATS_FIX), s1 = (float)t[1] * (1.0f / STATS_FIX);
      o[0] = s0; o[1] = s1; o[2] = (float)t[2]; o[3] = c;
      o[4] = c > 0.0f ? s0 / c / Ts : 0.0f;
      o[5] = c > 0.0f ? s1 / c / Ts : 0.0f;
      o[6] = c > 0.0f ? o[2] / c : 0.0f;
      o[7] = (float)n;
    });
  }
  PHASE_MARK(16);
}

// DM / SC: run-time model and scene (any arm, any boxes), or the shipped ABB scene fixed at compile time (ancestor-walk
// kinematics, compile-time level loops, ballot-driven box folds) -- the host picks the latter only when both match.
// ARM: number of links when the articulation is a fixed-base serial chain (ArmChain<ARM>::matches) in a compile-time
// scene -- its recursions then run on one lane (shf_arm.h); 0: the body-per-lane sub-step.
// LDS tail of an env: POS targets, this step's ee position, the arm's per-link records.
// (sized to the word: two workgroups of the two-wave kernel have to share a CU's 160 KiB)
#define SHF_ARM_MAX_LINKS 6
#define ABB_TGT_WORDS(nd) (((nd) + 2 + 3) & ~3)                     /* POS targets + the end effector's x, y */
#define WS_LINK_STASH_WORDS 32   /* k_abb_step_ws<512, true>: the free box's (IA, pA) and its corner ballots, parked per env */
#define ABB_TAIL_WORDS(nslots, nd) ((nslots) * PT_STRIDE + ABB_TGT_WORDS(nd) + ARM_KREC_WORDS(SHF_ARM_MAX_LINKS) + 4)
#define ABB_TAIL_WORDS_NOARM(nslots, nd) ((nslots) * PT_STRIDE + ABB_TGT_WORDS(nd) + 4)   /* (ARM = 0 and HARD: no link records of the arm recursions) */
// (WT threads per workgroup: 256, or 512 for k_abb_step_pgs_wide below)
template <int G, class DM, class SC, bool LINK, int ARM, bool HARD, int WT, bool EXT = false>
DEV void abb_step_body(const AbbArgs& A) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  PHASE_BEGIN();
  float* stats_lds = smem + MODEL_WORDS + SCENE_WORDS + ABB_WORDS;
  stats_block_init(stats_lds);
  const unsigned long long stats_step = stats_step_load(A.stats);
  {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(A.tp);
    uint32_t* dst = reinterpret_cast<uint32_t*>(smem + MODEL_WORDS + SCENE_WORDS);
    stage_block<(int)sizeof(ShfAbbTaskParams), WT>(src, reinterpret_cast<float*>(dst));
  }
  const ShfScene* scene = stage_scene<WT>(A.S.scene, smem + MODEL_WORDS);
  const ShfModel* m = stage_model<WT>(A.S.model, smem);
  const ShfAbbTaskParams& tp = *reinterpret_cast<const ShfAbbTaskParams*>(smem + MODEL_WORDS + SCENE_WORDS);
  const int epb = WT / G, es = threadIdx.x / G, l = threadIdx.x % G;
  const int e = blockIdx.x * epb + es;
  const int n = A.S.n;
  if (e >= n) return;
  const int nbx = SC::NBX > 0 ? SC::NBX : A.S.nboxes, actors = 1 + nbx;
  const int nb = DM::nb(m), nd = DM::nd(m), nbt = nb + nbx;
  const int nslots = DM::np(m) + box_slots(slot_lay<SC>(m, scene)) + (LINK ? 2 * SHF_MAX_LINK_CONTACTS : 0);
  const int nslots_all = HARD ? hard_total_slots(nslots, LINK) : nslots;   // HARD: the solve's records and response matrix inside the slot region
  const int env_words = env_lds_words(nbt, nd, nslots_all, HARD ? ABB_TAIL_WORDS_NOARM(nslots_all, nd) : ABB_TAIL_WORDS(nslots_all, nd), actors);
  EnvLds L = env_lds_carve(smem + MODEL_WORDS + SCENE_WORDS + ABB_WORDS + STATS_LDS_WORDS + es * env_words, nbt, nd, nslots_all, actors);
  float* tgtl = L.pt + nslots_all * PT_STRIDE;  // POS targets of this env step
  float* krec = tgtl + ABB_TGT_WORDS(nd);

  float* dof = A.S.dof + (size_t)e * nd * 2;
  float* root = A.S.root + (size_t)e * actors * 13;
  float* bstate = A.body_state + (size_t)e * nbt * 13;
  float* jac = A.jacobian + (size_t)e * (nb - 1) * 6 * nd;
  for (int i = l; i < 2 * nd; i += G) L.dofb[(i >> 1) * DOF_STRIDE + (i & 1)] = dof[i];
  for (int i = l; i < 13 * actors; i += G) L.root[i] = root[i];
  GROUP_SYNC();

  // AbbRobot.step: EE-delta -> clip -> damped least squares on the Jacobian tensor (possibly stale pose)
  if (l == 0) abb_ik_targets(A, tp, L.dofb, DOF_STRIDE, e, nd, bstate, jac, tgtl, stats_step);
  GROUP_SYNC();
  PHASE_MARK(11);

  StepCtx C;
  C.m = m; C.sp = A.S.sp; C.terr.t = A.S.terr; C.terr.h = A.S.heights; C.scene = scene;
  C.dropped = env_dropped(A.S.dropped, A.S.sp, e);
  C.mscale = A.S.mscale ? A.S.mscale + (size_t)e * nb : nullptr;
  C.hulls = A.S.hulls;
  const float mu = A.S.friction[e];
  const int nsub = tp.decimation + (tp.extra_substep ? 1 : 0);
  // (the 512-thread form has 256 registers at two wavefronts per SIMD and spills: its per-lane model constants stay in the
  // LDS model -- read where they are used -- instead of in registers that would go to scratch)
  typedef LaneModelT<!HARD> LM;
  LM M;
  lane_model_load<DM>(m, l, M);
  LanePoints<LANE_ROUNDS(G, DM)> P;
  if constexpr (DM::NPC > 0) lane_points_load<G>(m, DM::np(m), l, P);
  const BoxLane BL = SC::NBX > 0 ? box_lane_load(m, l) : BoxLane();
  // net contact forces are reported for the last sub-step only (what the refreshed tensor shows)
  for (int it = 0; it < nsub; it++) {
    if constexpr (ARM > 0)
      arm_substep<G, DM, SC, ARM>(C, L, krec, l, M, P, tgtl, mu, it == nsub - 1 ? L.xch : nullptr, BL);
    else
      substep<G, true, DM, false, LM, SC, false, LINK, HARD, EXT>(C, L, l, M, P, tgtl, nullptr, nullptr, mu, it == nsub - 1 ? L.xch : nullptr, BL);
  }
  abb_after_physics<G, DM>(A, tp, m, L, l, e, epb, nbx, tgtl, stats_lds, stats_step);
}
template <int G, class DM, class SC, bool LINK = false, int ARM = 0, bool HARD = false, bool EXT = false>
__global__ __launch_bounds__(256, ((G >= 32 && SC::NBX > 0) || (HARD && !LINK)) ? 2 : 1) void k_abb_step(AbbArgs A) {
  abb_step_body<G, DM, SC, LINK, ARM, HARD, 256, EXT>(A);
}
// The run-time-shaped step under the velocity-level solve with link contacts, 16 envs per workgroup of 512 threads: one staged
// model for sixteen envs leaves each 8.7 KB of LDS (csrc/shf_hard.h keeps the solve inside the contact-slot region), so that
// 4096 envs are resident at once -- two wavefronts per SIMD, 256 registers each -- instead of taking two rounds.
template <bool LINK>
__global__ __launch_bounds__(512) void k_abb_step_pgs_wide(AbbArgs A) {
  abb_step_body<32, DynDims, DynScene, LINK, 0, true, 512>(A);
}

// Wave-specialised form of the shipped ABB step (AbbDims arm, AbbScene boxes, 16 lanes per env).  A single wave issues
// one VALU instruction per ~4.6 clocks while its SIMD could issue one per ~2 (profiles/r03_valu_microbench.md), and at 4
// envs per wave 4096 envs are one wave per SIMD -- half the issue slots idle.  The arm's work (a serial chain) and the
// box actors' work (corner contacts, their fold, the boxes' 6x6 solves and integration) only meet at the rod's contact
// with the cube, so the workgroup's first WT/128 waves run the ARM of WT/32 envs and the other half the BOXES of the same
// envs, side by side, through the same LDS working set:
//     arm wave                                   |  box wave
//     joints, drives, chain composition,         |  box poses, corner slots, fold of the corners
//     inertias, terrain points                   |
//                                                |  the free box's folded (IA, pA) -> LDS
//   ---- S1 (workgroup barrier): box poses, the box's folded inertia and the arm's poses visible to both
//     rod-capsule slot vs the free box, its pair |
//     law (when it touches), pair fold,          |
//     hand-over, ABA inward / outward            |
//   ---- S4: the arm's accelerations, the slot's ballot and the pair records visible
//     (a four-barrier form with the pair law on the box wave measured the same: 0.0937 vs 0.0945 ms)
//     terrain-contact forces, arm contact rows,  |  pair forces, box solves, box contact rows,
//     joint integration                          |  box integration
// Same operations in the same order as the one-wave kernels (and the oracle): bit-identical results.
// LINK: with link contacts (ShfModel.link_collide: the arm's 59 sample points, its box volumes and the rod against every box
// actor) -- the box wave evaluates them (broad phase, candidate passes: csrc/shf_boxes.h link_contacts) between a barrier
// S0' behind the arm's chain composition and S1, while the arm wave computes inertias and the 59 points' terrain contacts;
// the arm wave then folds them with the rod slot.  512 threads = 16 envs per workgroup (one CU holds one: 9.2 KB of LDS per
// env), two waves per SIMD.
//
// HARD (k_abb_step_ws_hard): the same split under the velocity-level contact solve (ShfSimParams.solver = SHF_SOLVER_PGS / _TGS).
// The contact passes only record candidate constraints and the arm's articulated-body solve runs free, so the two waves need
// no exchange until both are done:
//     arm wave                                   |  box wave
//     joints, drives, chain composition          |  box poses, corner candidates; the free box's part of the solve's records
//                                                |  (LDL^T factors, velocity rate, free acceleration) while its inertia is in registers
//   ---- S0': the arm's poses and the boxes' visible to both
//     inertias, sample-point candidates,         |  rod-capsule candidate, link-contact candidates
//     free ABA inward / outward, the links' part |
//     of the solve's records (chain form)        |
//   ---- S1: candidates, free accelerations and the solve's body records visible
//     BOTH waves, regrouped at 32 lanes per env (two of the pair's four envs each): substep_hard_finish<.., RECORDS = false,
//     ARMNL = 6> -- gather, response matrix, sweeps, impulse passes (the arm's on the chain lane), integration (csrc/shf_hard.h,
//     the code the run-time-shaped kernels run: same values, same bits)
//   ---- S2
// S0', S1, S2 are barriers of that wave pair only (pair_barrier below); the workgroup's eight waves meet before and after the loop.
// Barrier of TWO wavefronts of a workgroup (both resident: same workgroup) on an LDS counter that only grows: the k-th meeting
// is over when the counter reaches 2 k.  k_abb_step_ws_hard's arm wave j and box wave j + 4 share four envs through all of a
// sub-step, so they only wait for each other -- not, as with s_barrier, for the slowest of the workgroup's eight waves in each
// of the three phases (the solve's time varies with the contact count: 14 % of the kernel was spent at the barrier behind it).
DEV void pair_barrier(int* ctr, int target) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  if ((threadIdx.x & 63u) == 0u) __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  // (bounded: a wave that lost its partner to a fault leaves after ~0.1 s instead of hanging the GPU)
  for (int spin = 0; spin < (1 << 22) && __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target; spin++)
    __builtin_amdgcn_s_sleep(1);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
// SIMONLY (k_sim_step_ws_hard): ONE gym.simulate of that scene for the hook path -- the same sub-step on the solver-side tensors
// (SimArgs in A.S: state in, state / net contact forces out; position targets and efforts from the command tensors), none of the
// task's code (inverse kinematics, statistics, state refresh, post_step).
template <int WT, bool LINK, bool HARD, bool SIMONLY = false>
DEV void abb_ws_body(const AbbArgs& A) {
  constexpr int G = 16, NL = 6, HALF = WT / 2, EPB = HALF / G;
  static_assert(!HARD || WT == 512, "the regrouped solve: sixteen envs x 32 lanes");
  static_assert(!SIMONLY || HARD, "the sub-step kernel of the hook path exists under the velocity-level solves");
  typedef typename std::conditional<LINK, AbbLinkDims, AbbDims>::type DM;
  typedef AbbScene SC;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  PHASE_BEGIN();
  PHASE_BEGIN_T();
  float* stats_lds = smem + MODEL_WORDS + SCENE_WORDS + ABB_WORDS;
  unsigned long long stats_step = 0ull;
  if constexpr (!SIMONLY) {
    stats_block_init(stats_lds);
    stats_step = stats_step_load(A.stats);
    stage_block<(int)sizeof(ShfAbbTaskParams), WT>(A.tp, smem + MODEL_WORDS + SCENE_WORDS);
  }
  const ShfScene* scene = stage_scene<WT>(A.S.scene, smem + MODEL_WORDS);
  const ShfModel* m = stage_model<WT>(A.S.model, smem);
  const ShfAbbTaskParams& tp = *reinterpret_cast<const ShfAbbTaskParams*>(smem + MODEL_WORDS + SCENE_WORDS);      // (SIMONLY: not staged, not read)
  const bool arm = (int)threadIdx.x < HALF;
  const int t = (int)threadIdx.x - (arm ? 0 : HALF), es = t / G, l = t % G;
  const int e = blockIdx.x * EPB + es;
  const int n = A.S.n;
  const bool live = e < n;                       // no early return: every wave meets every workgroup barrier
  constexpr int nbx = SC::NBX, actors = 1 + nbx, nb = NL + 1, nd = NL, nbt = nb + nbx;
  const int link_slot0 = DM::np(m) + box_slot_count(nbx, 1, m->nsph);   // the fixed scene has one free box
  const int nslots_own = link_slot0 + (LINK ? 2 * SHF_MAX_LINK_CONTACTS : 0);
  const int nslots = HARD ? hard_total_slots(nslots_own, LINK) : nslots_own;   // HARD: the solve's records and response matrix inside the slot region
  const int env_words = env_lds_words(nbt, nd, nslots, ABB_TAIL_WORDS(nslots, nd) + ((LINK || HARD) ? WS_LINK_STASH_WORDS : 0), actors);
  float* env_base = smem + MODEL_WORDS + SCENE_WORDS + ABB_WORDS + STATS_LDS_WORDS;
  EnvLds L = env_lds_carve(env_base + es * env_words, nbt, nd, nslots, actors);
  float* tgtl = L.pt + nslots * PT_STRIDE;       // POS targets of this env step
  float* krec = tgtl + ABB_TGT_WORDS(nd);
  unsigned* sphere_bits = reinterpret_cast<unsigned*>(krec + ARM_KREC_WORDS(NL));   // the rod slot's ballot, arm wave -> box wave
  int* link_count = reinterpret_cast<int*>(sphere_bits + 1);                          // active link slots, box wave -> arm wave
  // LINK: the free box's folded (IA, pA), parked while the box wave runs the link passes (the exchange slot the pair law reads
  // them from is overwritten by a joint-law record) -- 27 accumulators less in the box wave's registers across link_contacts
  float* box_stash = reinterpret_cast<float*>(sphere_bits + 4);

  if (arm && live) {
    const float* dof = A.S.dof + (size_t)e * nd * 2;
    const float* root = A.S.root + (size_t)e * actors * 13;
    for (int i = l; i < 2 * nd; i += G) L.dofb[(i >> 1) * DOF_STRIDE + (i & 1)] = dof[i];
    for (int i = l; i < 13 * actors; i += G) L.root[i] = root[i];
    if constexpr (SIMONLY) {
      if (l < nd) L.dofb[l * DOF_STRIDE + 5] = A.S.effort ? A.S.effort[(size_t)e * nd + l] : 0.0f;     // (EFFORT-mode dofs: sim_step_body)
    }
  }
  // HARD: the pair barrier's counter of (arm wave j, box wave j + 4): a spare word of their first env's tail
  int* pair_ctr = reinterpret_cast<int*>(env_base + (size_t)(es & ~3) * env_words + (tgtl - (env_base + es * env_words)) + ABB_TGT_WORDS(nd) + ARM_KREC_WORDS(NL) + 2);
  int pair_meet = 0;
  if constexpr (HARD) {
    if (arm && (t & 63) == 0) *pair_ctr = 0;
  }
  // AbbRobot.step's inverse kinematics on the box wave, beside the arm wave's loads (it reads the tensors directly)
  if constexpr (!SIMONLY) {
    if (!arm && live && l == 0)
      abb_ik_targets(A, tp, A.S.dof + (size_t)e * nd * 2, 2, e, nd, A.body_state + (size_t)e * nbt * 13,
                     A.jacobian + (size_t)e * (nb - 1) * 6 * nd, tgtl, stats_step);
  }
  __syncthreads();                               // S0: root rows and POS targets visible to both
  PHASE_MARK(11);

  StepCtx C;
  C.m = m; C.sp = A.S.sp; C.terr.t = A.S.terr; C.terr.h = A.S.heights; C.scene = scene;
  C.dropped = (LINK && live && !arm) ? env_dropped(A.S.dropped, A.S.sp, e) : nullptr;   // (the box wave counts the dropped link contacts)
  C.mscale = (live && A.S.mscale) ? A.S.mscale + (size_t)e * DM::nb(m) : nullptr;
  const float mu = live ? (A.S.friction ? A.S.friction[e] : 1.0f) : 0.0f;
  const int nsub = SIMONLY ? 1 : tp.decimation + (tp.extra_substep ? 1 : 0);
  // Per-lane model constants.  Without link contacts they are loaded once and stay in registers.  With them the kernel has 256
  // registers (two waves per SIMD) and the box wave's link passes need most: the arm wave re-reads its constants from the LDS
  // model at the head of each of its phases (WS_ARM_LOCALS), so that they are not live across the other wave's code.
  constexpr int NRP = LANE_ROUNDS(G, DM);
  // (HARD: the arm wave's phases read their model constants from the LDS model where they use them -- with them in registers the
  // phases' own peak, the free ABA recursions on the chain lane, spilled ~90 registers)
  typedef LaneModelT<!HARD> WsArmModel;
  LaneModel M;
  LanePoints<NRP> P;
  if constexpr (!LINK && !HARD) {
    lane_model_load<DM>(m, l, M);
    lane_points_load<G>(m, DM::np(m), l, P);
  }
#define WS_ARM_LOCALS_AT(lane)                           \
  WsArmModel Ml;                                         \
  lane_model_load<DM>(m, lane, Ml);                      \
  LanePoints<NRP> Pl;                                    \
  lane_points_load<G>(m, DM::np(m), lane, Pl);           \
  ArmLane<G, DM, NL, WsArmModel> ALl(C, L, krec, lane, Ml, Pl)
#define WS_ARM_LOCALS() WS_ARM_LOCALS_AT(l)
  const BoxLane BL = box_lane_load(m, l);
  ArmLane<G, DM, NL> AL(C, L, krec, l, M, P);    // (LINK: used for its gravity vector only)
  unsigned long long act[NRP];                   // LINK: the terrain-contact ballots, from the points phase to the force phase
#pragma unroll
  for (int k = 0; k < NRP; k++) act[k] = 0ull;
  BodyRegs B;                                    // arm wave: the lane's link; box wave: the lane's box
  BoxMasks BM;
  for (int it = 0; it < nsub; it++) {
    float* contact_out = it == nsub - 1 ? L.xch : nullptr;   // reported for the last sub-step only
    if constexpr (HARD) {
      // Everything lane-specific of a sub-step is derived here from the thread id passed through an empty asm: LDS addresses,
      // masks and per-lane constants are then recomputed where they are used instead of being hoisted out of the loop (~100
      // values, most of them spilled), and nothing but the loop counters has to stay in registers across the solve, whose own
      // peak is close to the 256 available.  The names shadow the kernel's.
      int tq = (int)threadIdx.x;
      asm volatile("" : "+v"(tq));
      {
        float dtq = C.sp.dt;                     // (likewise the step: 1 / dt, dt / n, ... are a dozen IEEE divisions' worth of registers)
        asm volatile("" : "+s"(dtq));
        C.sp.dt = dtq;
      }
      const bool arm = tq < HALF;
      const int t16 = tq - (arm ? 0 : HALF), esq = t16 / G, lq = t16 % G;
      const int e = blockIdx.x * EPB + esq;
      const bool live = e < n;
      const float mu = live ? (A.S.friction ? A.S.friction[e] : 1.0f) : 0.0f;
      C.dropped = (LINK && live && !arm) ? env_dropped(A.S.dropped, A.S.sp, e) : nullptr;
      C.mscale = (live && A.S.mscale) ? A.S.mscale + (size_t)e * DM::nb(m) : nullptr;
      const EnvLds L = env_lds_carve(env_base + esq * env_words, nbt, nd, nslots, actors);
      float* tgtl = L.pt + nslots * PT_STRIDE;
      float* krec = tgtl + ABB_TGT_WORDS(nd);
      int* link_count = reinterpret_cast<int*>(krec + ARM_KREC_WORDS(NL) + 1);
      float* box_stash = krec + ARM_KREC_WORDS(NL) + 4;
      int* pair_ctr = reinterpret_cast<int*>(env_base + (size_t)(esq & ~3) * env_words + (tgtl - (env_base + esq * env_words)) + ABB_TGT_WORDS(nd) + ARM_KREC_WORDS(NL) + 2);
      const float gq[3] = {AL.g[0], AL.g[1], AL.g[2]};
      // POS-drive targets: the fused step's are in LDS (inverse kinematics), gym.simulate's in the command tensor
      const float* ptq = SIMONLY ? ((live && A.S.pos_tgt) ? A.S.pos_tgt + (size_t)e * nd : nullptr) : tgtl;
      if (live) {
        if (arm) {
          WS_ARM_LOCALS_AT(lq);
          ALl.joints_and_drives(ptq);
          GROUP_SYNC();
          ALl.compose();
          GROUP_SYNC();
          PHASE_MARK(24);
        } else {
          // (body registers and slot ballots local to each phase: nothing of them is carried across the solve)
          BodyRegs Bb;
          BoxMasks BMb;
          boxes_pose<G>(C, L, lq, Bb);
          fixed_corner_slots<G, SC, true>(C, L, lq, BMb);
          // the free box's part of the solve's records here, where its rigid inertia is in registers (csrc/shf_hard.h: hard_records):
          // LDL^T factors and velocity rate into its exchange slot, its free acceleration and position on to its lane of the solve
          {
            const float a0[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
            float abox[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
            hard_records<G, true, DM, ArmSolveLane, 2>(C, L, lq, ArmSolveLane::load<NL>(m, lq), Bb, gq, a0, abox);
            if (lq == nb + SC::DYN) {
#pragma unroll
              for (int k = 0; k < 6; k++) L.acc[lq * 6 + k] = abox[k];
#pragma unroll
              for (int k = 0; k < 3; k++) box_stash[27 + k] = Bb.p[k];
            }
          }
        }
      }
      pair_barrier(pair_ctr, 2 * ++pair_meet);   // S0' (this wave and its partner only)
      PHASE_MARK(25);
      if (live) {
        if (arm) {
          WS_ARM_LOCALS_AT(lq);
          BodyRegs Ba;
          ALl.inertia_and_candidates(Ba, mu);
          ALl.hand_over(Ba);
          GROUP_SYNC();
          ALl.recursions();
          GROUP_SYNC();
          // the links' part of the solve's records (csrc/shf_hard.h: hard_records, PARTS 1) here, on the arm wave, which would
          // otherwise wait for the box wave's link passes -- for this arm in its chain form: link lanes copy (S, U, 1 / D) from the
          // chain lane's records, the chain lane zeroes the fixed root's rate and runs Dl_b = S_b qdd_b + Dl_(b-1) down the chain
          // (the same fma on the same values as the level-by-level loop; the free box's part is the box wave's, before S0')
          if (lq >= 1 && lq <= NL) {
            const float* kr = krec + (lq - 1) * KREC_STRIDE;
            float* rec = L.xch + lq * XCH_STRIDE;
#pragma unroll
            for (int k = 0; k < 6; k++) { rec[HB_S + k] = kr[k]; rec[HB_U + k] = kr[12 + k]; }
            rec[HB_INVD] = kr[18];
          }
          if (lq == 0) {
            float Dl[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int k = 0; k < 6; k++) L.xch[HB_FDL + k] = 0.0f;
#pragma unroll
            for (int b = 1; b <= NL; b++) {
              const float* kr = krec + (b - 1) * KREC_STRIDE;
              const float qdd = L.dofb[(b - 1) * DOF_STRIDE + 4];
              float* rec = L.xch + b * XCH_STRIDE;
#pragma unroll
              for (int k = 0; k < 6; k++) { Dl[k] = fmaf(kr[k], qdd, Dl[k]); rec[HB_DL + k] = Dl[k]; }
            }
          }
          PHASE_MARK(26);
        } else {
          BoxMasks BMb;
          fixed_sphere_slots<G, SC, true>(C, L, lq, mu, gq, BMb);
          int nl = 0;
          if constexpr (LINK) nl = link_contacts<G>(C, L, lq, link_slot0, mu, gq);
          if (lq == 0) *link_count = nl;
        }
      }
      pair_barrier(pair_ctr, 2 * ++pair_meet);   // S1
      PHASE_MARK(27);
      {
        // regrouped at 32 lanes per env: the four envs of (arm wave j, box wave j + 4) stay with these two waves -- the arm
        // wave solves its first two, the box wave the other two
        constexpr int G2 = 32;
        const int wq = tq >> 6, l2 = tq % G2;
        const int es2 = 4 * (wq & 3) + 2 * (wq >> 2) + ((tq >> 5) & 1);
        const int e2 = blockIdx.x * EPB + es2;
        if (e2 < n) {
          const EnvLds L2 = env_lds_carve(env_base + es2 * env_words, nbt, nd, nslots, actors);
          const float* krec2 = L2.pt + nslots * PT_STRIDE + ABB_TGT_WORDS(nd);
          const float* stash2 = krec2 + ARM_KREC_WORDS(NL) + 4;
          const int nlink2 = *reinterpret_cast<const int*>(krec2 + ARM_KREC_WORDS(NL) + 1);
          C.dropped = env_dropped(A.S.dropped, A.S.sp, e2);      // (the gather's drop counter and histogram: this lane's env of the solve)
          typedef ArmSolveLane LM2;
          const LM2 M2 = ArmSolveLane::load<NL>(m, l2);
          BodyRegs B2;                           // (the records are in LDS already: the solve needs the box's position only)
          if (l2 == nb + SC::DYN) {
#pragma unroll
            for (int k = 0; k < 3; k++) B2.p[k] = stash2[27 + k];
          }
          const float g2[3] = {gq[0], gq[1], gq[2]};     // (a local copy: the solve selects between pointers to it and to the boxes' gravity)
          float a2[6] = {0.0f, 0.0f, 0.0f, -g2[0], -g2[1], -g2[2]};
          substep_hard_finish<G2, true, DM, LM2, SC, false, LINK, false, NL>(C, L2, l2, M2, B2, g2, a2, 0, link_slot0, link_slot0, nlink2,
                                                                   it == nsub - 1 ? L2.xch : nullptr);
        }
      }
      PHASE_MARK(28);
      pair_barrier(pair_ctr, 2 * ++pair_meet);   // S2: the integrated state for the next sub-step's waves
      PHASE_MARK(29);
      continue;
    }
    if constexpr (!LINK) {
      if (live) {
        if (arm) {
          AL.joints_and_drives(tgtl);
          GROUP_SYNC();
          AL.compose();
          GROUP_SYNC();
          AL.inertia_and_points(B, mu);
        } else {
          boxes_pose<G>(C, L, l, B);
          fixed_corner_slots<G, SC>(C, L, l, BM);
          GROUP_SYNC();
          fixed_box_fold<G, SC>(C, L, l, B, BM);
        }
      }
    } else {
      if (live) {
        if (arm) {
          WS_ARM_LOCALS();
          ALl.joints_and_drives(tgtl);
          GROUP_SYNC();
          ALl.compose();
          GROUP_SYNC();
        } else {
          PHASE_MARK_T(17, HALF);                // (box wave's clock: time since its last mark = the finish of the sub-step before)
          boxes_pose<G>(C, L, l, B);             // (the boxes' own contacts need no arm pose: done beside the arm's composition)
          fixed_corner_slots<G, SC>(C, L, l, BM);
          PHASE_MARK_T(18, HALF);
          // the fold of the free box's own contacts is left to a spare lane of the ARM wave (which idles while this wave
          // runs the link passes): its rigid inertia and bias and the slots' ballots go to LDS
          if (l == nb + SC::DYN) {
#pragma unroll
            for (int k = 0; k < 21; k++) box_stash[k] = B.IA[k];
#pragma unroll
            for (int k = 0; k < 6; k++) box_stash[21 + k] = B.pA[k];
            unsigned* bs = reinterpret_cast<unsigned*>(box_stash + 27);
            bs[0] = BM.cplane; bs[1] = (unsigned)BM.cbox; bs[2] = (unsigned)(BM.cbox >> 32); bs[3] = BM.cedge;
          }
          PHASE_MARK_T(19, HALF);
        }
      }
      __syncthreads();                           // S0': the arm's poses and the boxes' are in LDS for both waves
      PHASE_MARK_T(20, HALF);                    // (box wave: waiting at S0')
      if (live) {
        if (arm) {
          WS_ARM_LOCALS();
          ALl.inertia_and_points(B, mu);
#pragma unroll
          for (int k = 0; k < NRP; k++) act[k] = ALl.active[k];
          // the rod's slot against the cube needs the poses only: evaluated here, while the box wave runs the link passes
          fixed_sphere_slots<G, SC>(C, L, l, mu, AL.g, BM);
          GROUP_SYNC();                          // the capsule's two slots come from two lanes
          // ... and the free box's own contacts folded into its inertia (fixed_box_fold's loop, on a lane that has no body):
          // corners ascending, terrain before boxes, the edge-edge slot at its place
          if (l == G - 1) {
            static_assert(G - 1 > NL + SC::NBX, "a lane without a body or a box");
            const SlotLay Q = slot_lay<SC>(m, scene);
            float IAf[21], pAf[6];
#pragma unroll
            for (int k = 0; k < 21; k++) IAf[k] = box_stash[k];
#pragma unroll
            for (int k = 0; k < 6; k++) pAf[k] = box_stash[21 + k];
            const unsigned* bs = reinterpret_cast<const unsigned*>(box_stash + 27);
            unsigned pl = bs[0], ed = bs[3];
            unsigned long long bx = (unsigned long long)bs[1] | ((unsigned long long)bs[2] << 32);
            int c, tg;
            while (corner_next<SC::NBX - 1, SC::DYN>(pl, bx, ed, &c, &tg))
              slot_accumulate(IAf, pAf, L.pt + corner_slot(Q, SC::DYN, c, tg) * PT_STRIDE, 1.0f, C.sp.dt, 1.0f);
            float* o = L.xch + (nb + SC::DYN) * XCH_STRIDE;   // for the pair law (after S1), and parked for the box wave's finish
#pragma unroll
            for (int k = 0; k < 21; k++) { o[k] = IAf[k]; box_stash[k] = IAf[k]; }
#pragma unroll
            for (int k = 0; k < 6; k++) { o[21 + k] = pAf[k]; box_stash[21 + k] = pAf[k]; }
          }
        } else {
          const int nl = link_contacts<G>(C, L, l, link_slot0, mu, AL.g);
          if (l == 0) *link_count = nl;
          PHASE_MARK_T(21, HALF);                // (box wave: the link passes)
        }
      }
    }
    if constexpr (!LINK) {
      if (live && !arm && l == nb + SC::DYN) {   // the free box with its own contacts folded in: what the pair law eliminates
        float* o = L.xch + l * XCH_STRIDE;
#pragma unroll
        for (int k = 0; k < 21; k++) o[k] = B.IA[k];
#pragma unroll
        for (int k = 0; k < 6; k++) o[21 + k] = B.pA[k];
      }
    }
    PHASE_MARK(24);
    __syncthreads();                             // S1
    PHASE_MARK(25);
    int nlink = 0;
    unsigned lb = 0u;                            // link slots on the free box
    if constexpr (LINK) {
      if (live) { nlink = *link_count; lb = link_box_bits(L, link_slot0, nlink, SC::DYN); }
    }
    BM.nlink = nlink;
    if (live && arm) {
      if constexpr (!LINK) {
        fixed_sphere_slots<G, SC>(C, L, l, mu, AL.g, BM);
        GROUP_SYNC();                            // the capsule's two slots come from two lanes
      }
      if (l == 0) {
        *sphere_bits = BM.spheres;
        if (BM.spheres || lb) {                  // rare: the pair laws on the lane that evaluated the slot
          const float* o = L.xch + (nb + SC::DYN) * XCH_STRIDE;
          float IAb[21], pAb[6];
#pragma unroll
          for (int k = 0; k < 21; k++) IAb[k] = o[k];
#pragma unroll
          for (int k = 0; k < 6; k++) pAb[k] = o[21 + k];
          fixed_pair_laws<G, SC>(C, L, true, IAb, pAb, BM.spheres, lb, link_slot0);
        }
      }
      GROUP_SYNC();
      fixed_arm_fold<G, SC>(C, L, l, B, BL, BM.spheres, nlink, lb, link_slot0);
      AL.hand_over(B);                           // (uses no model constants: AL serves both variants here)
      GROUP_SYNC();
      AL.recursions();
      GROUP_SYNC();
    }
    PHASE_MARK(26);
    __syncthreads();                             // S4
    PHASE_MARK(27);
    PHASE_MARK_T(22, HALF);                      // (box wave: from the end of its link passes through S1 to S4: idle)
    if (live) {
      if (arm) {
        if constexpr (LINK) {
          WS_ARM_LOCALS();
#pragma unroll
          for (int k = 0; k < NRP; k++) ALl.active[k] = act[k];
          if (contact_out) { ALl.point_forces(contact_out); GROUP_SYNC(); }
          boxes_finish<G, SC, 2>(C, L, l, B, contact_out, BL, BM, link_slot0);
          ALl.integrate();
          GROUP_SYNC();
        } else {
          if (contact_out) { AL.point_forces(contact_out); GROUP_SYNC(); }
          boxes_finish<G, SC, 2>(C, L, l, B, contact_out, BL, BM, link_slot0);
          AL.integrate();
          GROUP_SYNC();
        }
      } else {
        BM.spheres = *sphere_bits;
        if constexpr (LINK) {                     // the box lane's state back from LDS (it was not kept across the link passes)
          if (l == nb + SC::DYN) {
#pragma unroll
            for (int k = 0; k < 21; k++) B.IA[k] = box_stash[k];
#pragma unroll
            for (int k = 0; k < 6; k++) B.pA[k] = box_stash[21 + k];
            const float* pk = L.pose + l * POSE_STRIDE;
#pragma unroll
            for (int k = 0; k < 3; k++) B.p[k] = pk[9 + k];
          }
        }
        boxes_finish<G, SC, 1>(C, L, l, B, contact_out, BL, BM, link_slot0);
      }
    }
  }
  if constexpr (SIMONLY) {
    // gym.simulate ends here: the sub-step's state and net contact forces back to the solver-side tensors (sim_step_body's stores)
    __syncthreads();
    if (live && arm) {
      float* dof = A.S.dof + (size_t)e * nd * 2;
      float* root = A.S.root + (size_t)e * actors * 13;
      for (int i = l; i < 2 * nd; i += G) dof[i] = L.dofb[(i >> 1) * DOF_STRIDE + (i & 1)];
      for (int i = l; i < 13 * actors; i += G) root[i] = L.root[i];
      for (int i = l; i < 3 * nbt; i += G) A.S.contact[(size_t)e * nbt * 3 + i] = L.xch[i];
    }
    return;
  }
  PHASE_MARK(28);
  __syncthreads();                               // the boxes' final root rows and contact rows are in LDS
  PHASE_MARK(29);
  // State refresh and post_step, again side by side: the arm wave composes the end-of-step poses and goes on to post_step,
  // the box wave turns them into the rigid-body-state rows and the Jacobian tensor (gym.refresh_rigid_body_state_tensor /
  // refresh_jacobian_tensors: the values refresh_body_jac writes, from the same poses and motion subspaces).
  float* bstate = A.body_state + (size_t)(live ? e : 0) * nbt * 13;
  if (live && arm) {
    for (int i = l; i < 3 * nbt; i += G) A.S.contact[(size_t)e * nbt * 3 + i] = L.xch[i];
    GROUP_SYNC();
    if constexpr (LINK || HARD) {
      WS_ARM_LOCALS();
      ALl.joints();
      GROUP_SYNC();
      ALl.compose();
    } else {
      AL.joints();
      GROUP_SYNC();
      AL.compose();
    }
    if (l == 0) {
      const float* pe = L.pose + tp.ee_body * POSE_STRIDE;
      tgtl[nd] = L.root[0] + pe[9]; tgtl[nd + 1] = L.root[1] + pe[10];
    }
    GROUP_SYNC();
  }
  __syncthreads();
  if (!live) return;
  if (arm) {
    // a reset rewrites the root rows while the box wave still copies them out: post_step works on a private copy (the
    // contact slots are free by now)
    float* rootl = L.pt;
    for (int i = l; i < 13 * actors; i += G) rootl[i] = L.root[i];
    GROUP_SYNC();
    abb_post_step<G, DM>(A, tp, m, L, rootl, l, e, EPB, nbx, tgtl, stats_lds, stats_step);
    return;
  }
  if (l < nb) {
    const float* pb = L.pose + l * POSE_STRIDE;
    float Rw[9], p[3], v[6], t[3], q[4];
#pragma unroll
    for (int k = 0; k < 9; k++) Rw[k] = pb[k];
#pragma unroll
    for (int k = 0; k < 3; k++) p[k] = pb[9 + k];
#pragma unroll
    for (int k = 0; k < 6; k++) v[k] = pb[12 + k];
    float* o = L.xch + 13 * l;
#pragma unroll
    for (int k = 0; k < 3; k++) o[k] = L.root[k] + p[k];
    mat_to_quat(Rw, q);
#pragma unroll
    for (int k = 0; k < 4; k++) o[3 + k] = q[k];
    cross3(v, p, t);
#pragma unroll
    for (int k = 0; k < 3; k++) { o[7 + k] = v[3 + k] + t[k]; o[10 + k] = v[k]; }
  }
  GROUP_SYNC();
  for (int i = l; i < 13 * nb; i += G) bstate[i] = L.xch[i];
  for (int i = l; i < 13 * (actors - 1); i += G) bstate[nb * 13 + i] = L.root[13 + i];
  if (l >= 1 && l < nb) {
    float* J = A.jacobian + (size_t)e * (nb - 1) * 6 * nd + (size_t)(l - 1) * 6 * nd;
    const float* pb = L.pose + l * POSE_STRIDE;
    const float p[3] = {pb[9], pb[10], pb[11]};
    for (int k = 0; k < 6 * nd; k++) J[k] = 0.0f;
    for (int b = l; b > 0; b--) {
      const float* S = krec + (b - 1) * KREC_STRIDE;
      const float ax[3] = {S[0], S[1], S[2]};
      const int d = b - 1;
      float t[3];
      cross3(ax, p, t);
#pragma unroll
      for (int k = 0; k < 3; k++) { J[k * nd + d] = t[k] + S[3 + k]; J[(3 + k) * nd + d] = S[k]; }
    }
  }
}

#undef WS_ARM_LOCALS
#undef WS_ARM_LOCALS_AT
template <int WT, bool LINK = false>
__global__ __launch_bounds__(WT) void k_abb_step_ws(AbbArgs A) {
  abb_ws_body<WT, LINK, false>(A);
}
template <bool LINK>
__global__ __launch_bounds__(512) void k_abb_step_ws_hard(AbbArgs A) {
  abb_ws_body<512, LINK, true>(A);
}
// gym.simulate of the shipped arm in the shipped scene under the velocity-level solves (the hook path's sub-step: shf_sim_step
// with the split mapping): abb_ws_body's sub-step alone
template <bool LINK>
__global__ __launch_bounds__(512) void k_sim_step_ws_hard(SimArgs S) {
  AbbArgs A = {};
  A.S = S;
  abb_ws_body<512, LINK, true, true>(A);
}

#ifdef SHF_DEFINE_SMALL_KERNELS
__global__ void k_abb_reset_all(AbbArgs A) {
  // reset_idx(arange(N)): state is written straight into the tensors (one thread per env)
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = A.S.n;
  if (e >= n) return;
  const ShfAbbTaskParams& tp = *A.tp;
  const int nd = A.S.model->nd, nbx = A.S.nboxes, actors = 1 + nbx;
  float dofb[SHF_MAX_DOFS * DOF_STRIDE], rootl[13 * (SHF_MAX_BOXES + 1)];
  abb_reset_env(tp, nd, nbx, A.env_off + e, (uint32_t)A.reset_count[e], dofb, rootl);
  for (int d = 0; d < nd; d++) {
    A.S.dof[((size_t)e * nd + d) * 2] = dofb[d * DOF_STRIDE];
    A.S.dof[((size_t)e * nd + d) * 2 + 1] = 0.0f;
  }
  for (int k = 0; k < 13 * actors; k++) A.S.root[(size_t)e * actors * 13 + k] = rootl[k];
  A.done_sums[e] = A.rew_sums[e]; A.done_sums[(size_t)n + e] = A.rew_sums[(size_t)n + e];
  A.done_sums[(size_t)2 * n + e] = (float)A.success[e]; A.done_sums[(size_t)3 * n + e] = 1.0f;
  A.rew_sums[e] = 0.0f; A.rew_sums[(size_t)n + e] = 0.0f;
  A.ep_len[e] = 0;
  A.reset[e] = 1;
  A.reset_count[e] += 1;
}
#else
__global__ void k_abb_reset_all(AbbArgs A);
#endif
