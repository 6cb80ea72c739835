// shf_glue.hip -- the library glue of the hook-compatible path as single launches (K3 base_frame_state,
// K4 sample_heights, K7 episode log, K9 history shift).  On the path a user edits -- ShifuVecEnv with
// torch hooks -- these were ~60 small torch launches per vec-step of LIBRARY code (LeggedRobot.post_step,
// TerrainGymEnv.get_heights, HistoryRecorder, ShifuVecEnv.log_info / compute_reward); the hooks themselves stay torch.
// Same arithmetic as the fused kernels' in-kernel versions and as the oracle's glue_* functions (tests/test_gpu_glue.py).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "../../include/shifu_amd.h"

int shf_set_error(const std::string& msg);   // shf_api.hip: the message shf_last_error() returns
#define GLUE_LAUNCH_OK(who) (hipGetLastError() == hipSuccess ? 0 : shf_set_error(std::string(who) + ": launch failed"))

// isaacgym.torch_utils.quat_rotate_inverse (xyzw), the operations of the oracle's quat_rotate_inverse in its order
static __device__ __forceinline__ void qri(const float* q, const float* v, float* o) {
  const float w = q[3];
  const float s = 2.0f * (w * w) - 1.0f;
  const float cx = q[1] * v[2] - q[2] * v[1], cy = q[2] * v[0] - q[0] * v[2], cz = q[0] * v[1] - q[1] * v[0];
  const float d = q[0] * v[0] + q[1] * v[1] + q[2] * v[2];
  o[0] = v[0] * s - cx * w * 2.0f + q[0] * d * 2.0f;
  o[1] = v[1] * s - cy * w * 2.0f + q[1] * d * 2.0f;
  o[2] = v[2] * s - cz * w * 2.0f + q[2] * d * 2.0f;
}

// K3 -- LeggedRobot.post_step (shifu/units/robot.py:222-229)
__global__ void k_base_frame_state(const float* root_state, const int64_t* root_idx, int64_t num_rows, int n, int up_axis,
                                   float* lin, float* ang, float* pg, float* gvec) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  const int64_t row = root_idx ? root_idx[e] : (int64_t)e;
  if (row < 0 || row >= num_rows) return;
  const float* r = root_state + row * 13;
  const float q[4] = {r[3], r[4], r[5], r[6]}, lv[3] = {r[7], r[8], r[9]}, av[3] = {r[10], r[11], r[12]};
  float gv[3] = {0.0f, 0.0f, 0.0f}, a[3], b[3], c[3];
  gv[up_axis] = -1.0f;
  qri(q, lv, a); qri(q, av, b); qri(q, gv, c);
#pragma unroll
  for (int k = 0; k < 3; k++) { lin[3 * e + k] = a[k]; ang[3 * e + k] = b[k]; pg[3 * e + k] = c[k]; gvec[3 * e + k] = gv[k]; }
}

// K4 -- TerrainGymEnv.get_heights (shifu/gym/isaac_gym.py:412-433): one thread per (env, point)
__global__ void k_get_heights(ShfTerrain t, const int16_t* h, const float* root_state, const int64_t* root_idx, int64_t num_rows,
                              const float* hpoints, int n, int P, float* out) {
  const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= (long long)n * P) return;
  const int e = (int)(tid / P), i = (int)(tid % P);
  const int64_t row = root_idx ? root_idx[e] : (int64_t)e;
  if (row < 0 || row >= num_rows) return;
  const float* r = root_state + row * 13;
  float qz = r[5], qw = r[6];
  const float nr = sqrtf(qz * qz + qw * qw);
  const float nrm = nr > 1e-9f ? nr : 1e-9f;
  qz = qz / nrm; qw = qw / nrm;
  const float bx = hpoints[2 * i], by = hpoints[2 * i + 1];
  float hh = 0.0f;
  if (t.rows > 0) {
    const float tx = (-qz * by) * 2.0f, ty = (qz * bx) * 2.0f;
    float px = bx + qw * tx + (-qz * ty) + r[0];
    float py = by + qw * ty + (qz * tx) + r[1];
    px += t.border; py += t.border;
    // `points / horizontal_scale` as torch evaluates it on a GPU tensor with a Python-float divisor: x * (1 / s)
    // (ATen BinaryDivTrueKernel.cu); tests/test_gpu_glue.py holds this to the torch expression on cell-boundary points
    const float inv_hs = 1.0f / t.hscale;
    int ix = (int)truncf(px * inv_hs), iy = (int)truncf(py * inv_hs);
    ix = ix < 0 ? 0 : ix; ix = ix > t.rows - 2 ? t.rows - 2 : ix;
    iy = iy < 0 ? 0 : iy; iy = iy > t.cols - 2 ? t.cols - 2 : iy;
    const int16_t* p0 = h + (size_t)ix * t.cols + iy;
    const int16_t h1 = p0[0], h2 = p0[t.cols], h3 = p0[1];
    int16_t hm = h1 < h2 ? h1 : h2;
    hm = hm < h3 ? hm : h3;
    hh = (float)hm * t.vscale;
  }
  out[tid] = hh;
}

// K9 -- HistoryRecorder.add / reset_idx (shifu/utils/train.py:12-17): (rows, H) with the newest sample at column 0
__global__ void k_history_add(float* hist, const float* x, long long rows, int H) {
  const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= rows) return;
  float* o = hist + j * H;
  for (int hh = H - 1; hh >= 1; hh--) o[hh] = o[hh - 1];
  o[0] = x[j];
}
__global__ void k_rows_fill_indexed(float* buf, const int64_t* idx, int nidx, long long num_rows, int row_words, float value) {
  const int i = blockIdx.x;
  if (i >= nidx) return;
  const int64_t r = idx[i];
  if (r < 0 || r >= num_rows) return;
  for (int k = threadIdx.x; k < row_words; k += blockDim.x) buf[r * row_words + k] = value;
}

// K7 -- ShifuVecEnv.log_info (shifu/gym/env.py:149-158) and the buffer part of reset_idx (env.py:114-130) for the env
// ids of one reset: per key mean(sums[ids]) / T, sums[ids] = 0.  Exact: 2^-20 fixed-point integer sums (the fused
// kernels' statistics and oracle a1_stats), so the result does not depend on the order of the atomics.
#define GLUE_MAX_KEYS 16
struct GlueKeys { float* p[GLUE_MAX_KEYS]; };
// (ep_len / reset_buf / history: the rest of ShifuVecEnv.reset_idx's buffer writes for the same ids, shf_reset_bookkeeping; all NULL
// for shf_episode_log)
__global__ void k_episode_log(GlueKeys sums, int K, const int64_t* ids, int nids, long long n, float T, unsigned long long* acc,
                              float* out, int64_t* ep_len, void* reset_buf, int reset_bytes, float* history, int hist_words) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nids) {
    const int64_t e = ids[i];
    if (e >= 0 && e < n) {
      if (ep_len) ep_len[e] = 0;
      if (reset_buf) {
        if (reset_bytes == 8) reinterpret_cast<int64_t*>(reset_buf)[e] = 1;
        else reinterpret_cast<uint8_t*>(reset_buf)[e] = 1;
      }
      if (history)
        for (int k = 0; k < hist_words; k++) history[e * hist_words + k] = 0.0f;
      for (int k = 0; k < K; k++) {
        const float v = sums.p[k][e];
        sums.p[k][e] = 0.0f;
        const long long f = (long long)rintf(v * 1048576.0f);
        if (f != 0) atomicAdd(&acc[k], (unsigned long long)f);
      }
    }
  }
  __syncthreads();
  __shared__ unsigned ticket;
  if (threadIdx.x == 0) {
    __threadfence();
    ticket = (unsigned)atomicAdd(&acc[GLUE_MAX_KEYS], 1ull);
  }
  __syncthreads();
  if (ticket != gridDim.x - 1) return;
  // last block: every block's additions are visible (each fenced before taking its ticket)
  if ((int)threadIdx.x < K) {
    const long long tot = (long long)atomicExch(&acc[threadIdx.x], 0ull);
    const float sum = (float)tot * (1.0f / 1048576.0f), c = (float)nids;
    out[threadIdx.x] = c > 0.0f ? sum / c / T : 0.0f;
  }
  if (threadIdx.x == 0) atomicExch(&acc[GLUE_MAX_KEYS], 0ull);
}

// ShifuVecEnv.compute_reward's accumulation (env.py:180-185): rew = r_0 + r_1 + ... (that order), sums_k += r_k
struct GlueConstKeys { const float* p[GLUE_MAX_KEYS]; };
__global__ void k_reward_accumulate(GlueConstKeys r, GlueKeys sums, int K, long long n, float* rew) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  float acc = 0.0f;
  for (int k = 0; k < K; k++) {
    const float v = r.p[k][e];
    sums.p[k][e] += v;
    acc += v;
  }
  rew[e] = acc;
}

// Robot._reset_dof_state (shifu/units/robot.py:74-86) before its two indexed commits: targets and positions to the
// defaults, velocities to zero, and the int32 actor indices the commits take -- one block per env id
__global__ void k_reset_dof_rows(float* dof_state, float* dof_targets, const float* default_pos, const int64_t* env_ids, int nids,
                                 long long num_envs, int nd, const int64_t* root_idx, int32_t* actor_ids_out) {
  const int i = blockIdx.x;
  if (i >= nids) return;
  const int64_t e = env_ids[i];
  if (e < 0 || e >= num_envs) { if (threadIdx.x == 0 && actor_ids_out) actor_ids_out[i] = -1; return; }
  for (int d = threadIdx.x; d < nd; d += blockDim.x) {
    const float q0 = default_pos[d];
    dof_targets[e * nd + d] = q0;
    dof_state[(e * nd + d) * 2] = q0;
    dof_state[(e * nd + d) * 2 + 1] = 0.0f;
  }
  if (threadIdx.x == 0 && actor_ids_out) actor_ids_out[i] = (int32_t)(root_idx ? root_idx[e] : e);
}

extern "C" int shf_reset_dof_rows(float* dof_state, float* dof_targets, const float* default_dof_pos, const int64_t* env_ids,
                                  int32_t n_ids, int64_t num_envs, int32_t num_dof, const int64_t* root_idx_or_null,
                                  int32_t* actor_ids_out_or_null, void* stream) {
  if (!dof_state || !dof_targets || !default_dof_pos || (!env_ids && n_ids > 0)) return shf_set_error("shf_reset_dof_rows: null tensor");
  if (n_ids <= 0 || num_dof <= 0) return 0;
  hipLaunchKernelGGL(k_reset_dof_rows, dim3(n_ids), dim3(32), 0, (hipStream_t)stream, dof_state, dof_targets, default_dof_pos, env_ids,
                     (int)n_ids, (long long)num_envs, (int)num_dof, root_idx_or_null, actor_ids_out_or_null);
  return GLUE_LAUNCH_OK("shf_reset_dof_rows");
}

// K10 -- ArmRobot.inverse_kinematics (shifu/units/robot.py:162-182; shifu/utils/torch_utils.py:12-58): damped least
// squares u = J^T (J J^T + lambda^2 I)^-1 dpose on the end effector's (6, nd) Jacobian block, dof_pos + u; the 6x6
// system by LDL^T in packed storage -- the arithmetic of the fused ABB step and of the oracle's glue_ik_dls
#define GSYM(i, j) ((i) * 6 - ((i) * ((i) - 1)) / 2 + ((j) - (i)))
static __device__ __forceinline__ void glue_quat_mul(const float* a, const float* b, float* o) {   // torch_utils.py:12-33 (xyzw)
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
static __device__ __forceinline__ float glue_rcp_spec(float x) {   // the oracle's rcp_spec (csrc/shf_device.h has the same)
  float y = __uint_as_float(0x7EF311C7u - __float_as_uint(x));
#pragma unroll
  for (int k = 0; k < 3; k++) { const float e = fmaf(-x, y, 1.0f); y = fmaf(y, e, y); }
  return y;
}
static __device__ __forceinline__ void glue_ldlt_solve6(const float* IA, const float* pA, float* x) {   // IA x = -pA
  float Lm[6][6], Dg[6], iD[6], y[6];
#pragma unroll
  for (int j = 0; j < 6; j++) {
    float d = IA[GSYM(j, j)];
#pragma unroll
    for (int k = 0; k < j; k++) d = fmaf(-(Lm[j][k] * Lm[j][k]), Dg[k], d);
    Dg[j] = d;
    const float id = glue_rcp_spec(d);
    iD[j] = id;
#pragma unroll
    for (int i = j + 1; i < 6; i++) {
      float v = IA[GSYM(j, i)];
#pragma unroll
      for (int k = 0; k < j; k++) v = fmaf(-(Lm[i][k] * Lm[j][k]), Dg[k], v);
      Lm[i][j] = v * id;
    }
  }
#pragma unroll
  for (int i = 0; i < 6; i++) {
    float v = -pA[i];
#pragma unroll
    for (int k = 0; k < i; k++) v = fmaf(-Lm[i][k], y[k], v);
    y[i] = v;
  }
#pragma unroll
  for (int i = 0; i < 6; i++) y[i] = y[i] * iD[i];
#pragma unroll
  for (int i = 5; i >= 0; i--) {
    float v = y[i];
#pragma unroll
    for (int k = i + 1; k < 6; k++) v = fmaf(-Lm[k][i], x[k], v);
    x[i] = v;
  }
}
__global__ void k_ik_dls(const float* j_ee, long long j_stride, const float* dof_pos, int dof_stride, const float* ee_pose,
                         long long ee_stride, const float* goal_pose, int n, int nd, float damping, float* out) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  const float* J = j_ee + (long long)e * j_stride;
  const float* ee = ee_pose + (long long)e * ee_stride;
  const float* gp = goal_pose + (long long)e * 7;
  float dpose[6], cc[4], qr[4];
#pragma unroll
  for (int k = 0; k < 3; k++) dpose[k] = gp[k] - ee[k];
  cc[0] = -ee[3]; cc[1] = -ee[4]; cc[2] = -ee[5]; cc[3] = ee[6];
  const float tq[4] = {gp[3], gp[4], gp[5], gp[6]};
  glue_quat_mul(tq, cc, qr);
  const float sg = qr[3] > 0.0f ? 1.0f : (qr[3] < 0.0f ? -1.0f : 0.0f);
#pragma unroll
  for (int k = 0; k < 3; k++) dpose[3 + k] = qr[k] * sg;
  float Am[21], neg[6], x[6];
  const float lam2 = damping * damping;
#pragma unroll
  for (int i = 0; i < 6; i++)
#pragma unroll
    for (int j = i; j < 6; j++) {
      float acc = J[i * nd] * J[j * nd];
      for (int d = 1; d < nd; d++) acc = fmaf(J[i * nd + d], J[j * nd + d], acc);
      Am[GSYM(i, j)] = (i == j) ? acc + lam2 : acc;
    }
#pragma unroll
  for (int k = 0; k < 6; k++) neg[k] = -dpose[k];
  glue_ldlt_solve6(Am, neg, x);
  for (int d = 0; d < nd; d++) {
    float u = J[d] * x[0];
#pragma unroll
    for (int k = 1; k < 6; k++) u = fmaf(J[k * nd + d], x[k], u);
    out[(long long)e * nd + d] = dof_pos[(long long)e * nd * dof_stride + d * dof_stride] + u;
  }
}

extern "C" int shf_ik_dls(const float* j_ee, int64_t j_env_stride, const float* dof_pos, int32_t dof_elem_stride,
                          const float* ee_pose, int64_t ee_env_stride, const float* goal_pose, int32_t n, int32_t num_dof,
                          float damping, float* dof_targets_out, void* stream) {
  if (!j_ee || !dof_pos || !ee_pose || !goal_pose || !dof_targets_out) return shf_set_error("shf_ik_dls: null tensor");
  if (n <= 0 || num_dof <= 0) return 0;
  if (num_dof > SHF_MAX_DOFS) return shf_set_error("shf_ik_dls: too many dofs");
  hipLaunchKernelGGL(k_ik_dls, dim3((n + 127) / 128), dim3(128), 0, (hipStream_t)stream, j_ee, (long long)j_env_stride, dof_pos,
                     (int)dof_elem_stride, ee_pose, (long long)ee_env_stride, goal_pose, (int)n, (int)num_dof, damping,
                     dof_targets_out);
  return GLUE_LAUNCH_OK("shf_ik_dls");
}

extern "C" int shf_base_frame_state(const float* root_state, const int64_t* root_idx_or_null, int64_t num_root_rows, int32_t n,
                                    int32_t up_axis, float* base_lin_vel, float* base_ang_vel, float* projected_gravity,
                                    float* gravity_vec, void* stream) {
  if (!root_state || !base_lin_vel || !base_ang_vel || !projected_gravity || !gravity_vec) return shf_set_error("shf_base_frame_state: null tensor");
  if (n <= 0) return 0;
  if (up_axis < 0 || up_axis > 2) return shf_set_error("shf_base_frame_state: up_axis must be 0, 1 or 2");
  hipLaunchKernelGGL(k_base_frame_state, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, root_state, root_idx_or_null,
                     num_root_rows, (int)n, (int)up_axis, base_lin_vel, base_ang_vel, projected_gravity, gravity_vec);
  return GLUE_LAUNCH_OK("shf_base_frame_state");
}

extern "C" int shf_get_heights(const ShfTerrain* terrain, const int16_t* height_samples, const float* root_state,
                               const int64_t* root_idx_or_null, int64_t num_root_rows, const float* height_points_xy, int32_t n,
                               int32_t num_points, float* out, void* stream) {
  if (!terrain || !root_state || !height_points_xy || !out) return shf_set_error("shf_get_heights: null argument");
  if (terrain->rows > 0 && !height_samples) return shf_set_error("shf_get_heights: height samples not given");
  if (n <= 0 || num_points <= 0) return 0;
  const long long tot = (long long)n * num_points;
  hipLaunchKernelGGL(k_get_heights, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *terrain, height_samples,
                     root_state, root_idx_or_null, num_root_rows, height_points_xy, (int)n, (int)num_points, out);
  return GLUE_LAUNCH_OK("shf_get_heights");
}

extern "C" int shf_history_add(float* history, const float* x, int64_t rows, int32_t num_history, void* stream) {
  if (!history || !x) return shf_set_error("shf_history_add: null tensor");
  if (rows <= 0) return 0;
  if (num_history < 1) return shf_set_error("shf_history_add: num_history must be >= 1");
  hipLaunchKernelGGL(k_history_add, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, history, x, (long long)rows,
                     (int)num_history);
  return GLUE_LAUNCH_OK("shf_history_add");
}

extern "C" int shf_rows_fill_indexed(float* buf, const int64_t* idx, int32_t n_idx, int64_t num_rows, int32_t row_words,
                                     float value, void* stream) {
  if (!buf || (!idx && n_idx > 0)) return shf_set_error("shf_rows_fill_indexed: null tensor");
  if (n_idx <= 0 || row_words <= 0) return 0;
  hipLaunchKernelGGL(k_rows_fill_indexed, dim3(n_idx), dim3(64), 0, (hipStream_t)stream, buf, idx, (int)n_idx, (long long)num_rows,
                     (int)row_words, value);
  return GLUE_LAUNCH_OK("shf_rows_fill_indexed");
}

extern "C" int shf_episode_log(float* const* sums, int32_t num_keys, const int64_t* env_ids, int32_t n_ids, int64_t num_envs,
                               float episode_length_s, int64_t* workspace17, float* out_means, void* stream) {
  if (!sums || !env_ids || !workspace17 || !out_means) return shf_set_error("shf_episode_log: null argument");
  if (num_keys < 1 || num_keys > GLUE_MAX_KEYS) return shf_set_error("shf_episode_log: 1..16 keys");
  if (n_ids <= 0) return 0;
  GlueKeys G{};
  for (int k = 0; k < num_keys; k++) {
    if (!sums[k]) return shf_set_error("shf_episode_log: null sums tensor");
    G.p[k] = sums[k];
  }
  hipLaunchKernelGGL(k_episode_log, dim3((n_ids + 255) / 256), dim3(256), 0, (hipStream_t)stream, G, (int)num_keys, env_ids, (int)n_ids,
                     (long long)num_envs, episode_length_s, reinterpret_cast<unsigned long long*>(workspace17), out_means,
                     (int64_t*)nullptr, (void*)nullptr, 0, (float*)nullptr, 0);
  return GLUE_LAUNCH_OK("shf_episode_log");
}

extern "C" int shf_reset_bookkeeping(float* const* sums, int32_t num_keys, const int64_t* env_ids, int32_t n_ids, int64_t num_envs,
                                     float episode_length_s, int64_t* workspace17, float* out_means, int64_t* episode_length_or_null,
                                     void* reset_buf_or_null, int32_t reset_elem_bytes, float* history_or_null,
                                     int32_t history_row_words, void* stream) {
  if (!sums || !env_ids || !workspace17 || !out_means) return shf_set_error("shf_reset_bookkeeping: null argument");
  if (num_keys < 1 || num_keys > GLUE_MAX_KEYS) return shf_set_error("shf_reset_bookkeeping: 1..16 keys");
  if (reset_buf_or_null && reset_elem_bytes != 1 && reset_elem_bytes != 8)
    return shf_set_error("shf_reset_bookkeeping: reset_buf elements of 1 or 8 bytes");
  if (history_or_null && history_row_words < 1) return shf_set_error("shf_reset_bookkeeping: history_row_words must be >= 1");
  if (n_ids <= 0) return 0;
  GlueKeys G{};
  for (int k = 0; k < num_keys; k++) {
    if (!sums[k]) return shf_set_error("shf_reset_bookkeeping: null sums tensor");
    G.p[k] = sums[k];
  }
  hipLaunchKernelGGL(k_episode_log, dim3((n_ids + 255) / 256), dim3(256), 0, (hipStream_t)stream, G, (int)num_keys, env_ids, (int)n_ids,
                     (long long)num_envs, episode_length_s, reinterpret_cast<unsigned long long*>(workspace17), out_means,
                     episode_length_or_null, reset_buf_or_null, (int)reset_elem_bytes, history_or_null, (int)history_row_words);
  return GLUE_LAUNCH_OK("shf_reset_bookkeeping");
}

extern "C" int shf_reward_accumulate(const float* const* terms, float* const* sums, int32_t num_keys, int64_t num_envs,
                                     float* rew, void* stream) {
  if (!terms || !sums || !rew) return shf_set_error("shf_reward_accumulate: null argument");
  if (num_keys < 1 || num_keys > GLUE_MAX_KEYS) return shf_set_error("shf_reward_accumulate: 1..16 terms");
  if (num_envs <= 0) return 0;
  GlueConstKeys R{};
  GlueKeys S{};
  for (int k = 0; k < num_keys; k++) {
    if (!terms[k] || !sums[k]) return shf_set_error("shf_reward_accumulate: null tensor");
    R.p[k] = terms[k]; S.p[k] = sums[k];
  }
  hipLaunchKernelGGL(k_reward_accumulate, dim3((unsigned)((num_envs + 255) / 256)), dim3(256), 0, (hipStream_t)stream, R, S,
                     (int)num_keys, (long long)num_envs, rew);
  return GLUE_LAUNCH_OK("shf_reward_accumulate");
}
