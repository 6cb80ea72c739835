// shf_mlp.hip -- MFMA kernels of the PPO trainer's MLPs (the immediate caller of env.step; the
// reference gets them from the un-vendored rsl_rl package, shifu/runner/policy_runner.py:4,52-73, as stock
// nn.Linear + ELU: ActorCritic 259 -> 512 -> 256 -> 128 -> 12 / 1, shifu/configs/policy_config.py:8-16).
//
// One tiled GEMM, three uses per layer, everything else fused into its load / store paths:
//   forward           Y[M,N]  = act( X[M,K] W[N,K]^T + b[N] )
//   input gradient    dX[M,K] = G[M,N] W[N,K]               G = dY (.) act'(Y)   (ELU: act' = Y > 0 ? 1 : Y + 1)
//   weight gradient   dW[N,K] = G[M,N]^T X[M,K],  db[N] = colsum G      (split over M, deterministic two-pass sum)
// Operands are the trainer's fp32 tensors; tiles are converted to bf16 (round to nearest even) on their way into LDS
// and multiplied with v_mfma_f32_32x32x16_bf16, accumulating in fp32 (gfx950: 2.4 PFLOP/s dense bf16 against
// 0.157 PFLOP/s for fp32-input MFMA).  128 x 128 output tile per 256-thread block, K tile 32, four waves of 64 x 64
// (2 x 2 MFMA tiles).  No vendor BLAS.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "../../include/shifu_amd.h"

#define MLP_DEV __device__ __forceinline__

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

namespace {

constexpr int BM = 128, BN = 128, BK = 32, LDT = BK + 8;   // LDS row = 40 bf16 = 80 B: 16-byte aligned fragments, staggered banks

MLP_DEV uint16_t to_bf16(float x) {
  uint32_t u = __float_as_uint(x);
  u += 0x7FFFu + ((u >> 16) & 1u);   // round to nearest even (inputs are finite)
  return (uint16_t)(u >> 16);
}
MLP_DEV float elu_grad_from_output(float y) { return y > 0.0f ? 1.0f : y + 1.0f; }

// How the A ([rows x red]) and B ([cols x red]) operands of  C[rows, cols] = sum_red A[row, red] B[col, red]  sit in
// global memory: RED_CONTIG = element (r, k) at p[r * ld + k] (reduction index contiguous), else at p[k * ld + r].
struct Operand {
  const float* p;
  const float* mask_y;   // optional: multiply by act'(mask_y[...]) with the same indexing (G = dY (.) act'(Y))
  int ld;
  int rows;              // extent along the output index
};

// Stage a [128 x 32] operand tile (bf16, reduction index contiguous) into LDS.  256 threads, 4096 elements = 16 each.
template <bool RED_CONTIG>
MLP_DEV void stage_tile(const Operand& O, int r0, int k0, int red, uint16_t* lds) {
  const int t = (int)threadIdx.x;
  if (RED_CONTIG) {
    // thread -> (row = t / 2 + 0 | 64 ..., 16 consecutive k): two passes of 128 rows x 2 halves
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
      const int row = (t >> 1) + 0, half = t & 1;
      const int r = r0 + row + 0 * pass;
      (void)r;
    }
    // simple mapping: element e = t + 256 * i, row = e / 32, k = e % 32  (coalesced along k)
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const int e = t + 256 * i, row = e >> 5, k = e & 31;
      const int gr = r0 + row, gk = k0 + k;
      float v = 0.0f;
      if (gr < O.rows && gk < red) {
        const size_t idx = (size_t)gr * O.ld + gk;
        v = O.p[idx];
        if (O.mask_y) v *= elu_grad_from_output(O.mask_y[idx]);
      }
      lds[row * LDT + k] = to_bf16(v);
    }
  } else {
    // element e = t + 256 * i, k = e / 128, row = e % 128 (coalesced along the output index), transposed into LDS
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const int e = t + 256 * i, k = e >> 7, row = e & 127;
      const int gr = r0 + row, gk = k0 + k;
      float v = 0.0f;
      if (gr < O.rows && gk < red) {
        const size_t idx = (size_t)gk * O.ld + gr;
        v = O.p[idx];
        if (O.mask_y) v *= elu_grad_from_output(O.mask_y[idx]);
      }
      lds[row * LDT + k] = to_bf16(v);
    }
  }
}

struct Epilogue {
  float* c;            // output [rows, cols], row-major, ld = ldc
  int ldc;
  const float* bias;   // per column, or null
  int act;             // 0 none, 1 ELU
  float* colsum;       // optional [gridDim.z][cols] partial column sums of the A operand (db), written by blockIdx.y == 0
};

// C[rows, cols] (+)= A B^T over the reduction range [red0, red1) handled by this block (blockIdx.z slices for split-K).
template <bool A_RED_CONTIG, bool B_RED_CONTIG, bool COLSUM_A>
__global__ __launch_bounds__(256) void k_mlp_gemm(Operand A, Operand B, int red, int red_per_slice, Epilogue E, int rows, int cols) {
  __shared__ __attribute__((aligned(16))) uint16_t As[BM * LDT];
  __shared__ __attribute__((aligned(16))) uint16_t Bs[BN * LDT];
  const int r0 = (int)blockIdx.x * BM, c0 = (int)blockIdx.y * BN;
  const int red0 = (int)blockIdx.z * red_per_slice;
  const int red1 = red0 + red_per_slice < red ? red0 + red_per_slice : red;
  const int wave = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
  const int wr = (wave >> 1) * 64, wc = (wave & 1) * 64;       // this wave's 64 x 64 corner inside the block tile
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int k = 0; k < 16; k++) acc[i][j][k] = 0.0f;
  float csum = 0.0f;   // COLSUM_A: column (= A row index) sums, thread t < 128 owns A row t of the tile

  for (int k0 = red0; k0 < red1; k0 += BK) {
    __syncthreads();
    stage_tile<A_RED_CONTIG>(A, r0, k0, red1, As);
    stage_tile<B_RED_CONTIG>(B, c0, k0, red1, Bs);
    __syncthreads();
    if (COLSUM_A && blockIdx.y == 0 && threadIdx.x < BM) {
      // db: sum over the reduction index of the (bf16-rounded) G values of A row t -- the values the MFMA multiplies
      const uint16_t* row = As + threadIdx.x * LDT;
#pragma unroll
      for (int k = 0; k < BK; k++) csum += __uint_as_float((uint32_t)row[k] << 16);
    }
#pragma unroll
    for (int kk = 0; kk < BK; kk += 16) {
      bf16x8 a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; i++) a[i] = *reinterpret_cast<const bf16x8*>(As + (wr + 32 * i + (lane & 31)) * LDT + kk + 8 * (lane >> 5));
#pragma unroll
      for (int j = 0; j < 2; j++) b[j] = *reinterpret_cast<const bf16x8*>(Bs + (wc + 32 * j + (lane & 31)) * LDT + kk + 8 * (lane >> 5));
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }

  // epilogue: C/D layout of the 32 x 32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  float* Cout = E.c + (size_t)blockIdx.z * rows * E.ldc * (gridDim.z > 1 ? 1 : 0);   // split-K: slice z writes its own partial
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int col = c0 + wc + 32 * j + (lane & 31);
      const float bv = (E.bias && col < cols) ? E.bias[col] : 0.0f;
#pragma unroll
      for (int reg = 0; reg < 16; reg++) {
        const int row = r0 + wr + 32 * i + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
        if (row < rows && col < cols) {
          float v = acc[i][j][reg] + bv;
          if (E.act == 1) v = v > 0.0f ? v : expm1f(v);
          Cout[(size_t)row * E.ldc + col] = v;
        }
      }
    }
  if (COLSUM_A && blockIdx.y == 0 && threadIdx.x < BM) {
    const int row = r0 + (int)threadIdx.x;
    if (row < rows) E.colsum[(size_t)blockIdx.z * rows + row] = csum;
  }
}

// out[i] = sum_s part[s][i]  (fixed order: deterministic)
__global__ void k_mlp_reduce_slices(const float* part, float* out, int n, int slices) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (i >= n) return;
  float acc = 0.0f;
  for (int s = 0; s < slices; s++) acc += part[(size_t)s * n + i];
  out[i] = acc;
}

thread_local std::string g_mlp_err;
int mlp_fail(const std::string& m) { g_mlp_err = m; return 1; }

}  // namespace

extern "C" const char* shf_mlp_last_error(void) { return g_mlp_err.c_str(); }

extern "C" int shf_mlp_linear_forward(const float* x, const float* w, const float* b, float* y, int32_t M, int32_t K, int32_t N,
                                      int32_t act, void* stream) {
  if (!x || !w || !y) return mlp_fail("shf_mlp_linear_forward: null tensor");
  if (M <= 0 || K <= 0 || N <= 0 || act < 0 || act > 1) return mlp_fail("shf_mlp_linear_forward: bad shape / activation");
  Operand A{x, nullptr, K, M}, B{w, nullptr, K, N};
  Epilogue E{y, N, b, act, nullptr};
  dim3 grid((M + BM - 1) / BM, (N + BN - 1) / BN, 1);
  hipLaunchKernelGGL((k_mlp_gemm<true, true, false>), grid, dim3(256), 0, (hipStream_t)stream, A, B, K, K, E, M, N);
  return hipGetLastError() == hipSuccess ? 0 : mlp_fail("shf_mlp_linear_forward: launch failed");
}

extern "C" int shf_mlp_linear_backward_input(const float* dy, const float* y_or_null, const float* w, float* dx, int32_t M,
                                             int32_t K, int32_t N, void* stream) {
  if (!dy || !w || !dx) return mlp_fail("shf_mlp_linear_backward_input: null tensor");
  // dX[M,K] = G[M,N] W[N,K]: A = G (reduction index n contiguous), B[col = k][red = n] = W[n][k] (transposed access)
  Operand A{dy, y_or_null, N, M}, B{w, nullptr, K, K};
  Epilogue E{dx, K, nullptr, 0, nullptr};
  dim3 grid((M + BM - 1) / BM, (K + BN - 1) / BN, 1);
  hipLaunchKernelGGL((k_mlp_gemm<true, false, false>), grid, dim3(256), 0, (hipStream_t)stream, A, B, N, N, E, M, K);
  return hipGetLastError() == hipSuccess ? 0 : mlp_fail("shf_mlp_linear_backward_input: launch failed");
}

extern "C" int shf_mlp_backward_weight_workspace(int32_t M, int32_t K, int32_t N, int64_t* floats) {
  if (!floats) return mlp_fail("shf_mlp_backward_weight_workspace: null");
  const int slices = (M + 1023) / 1024;
  *floats = (int64_t)slices * ((int64_t)N * K + N);
  return 0;
}

extern "C" int shf_mlp_linear_backward_weight(const float* dy, const float* y_or_null, const float* x, float* dw, float* db,
                                              float* workspace, int32_t M, int32_t K, int32_t N, void* stream) {
  if (!dy || !x || !dw || !workspace) return mlp_fail("shf_mlp_linear_backward_weight: null tensor");
  // dW[N,K] = G^T X: rows = n, cols = k, reduction over m; both operands are stored with the reduction index as the row
  const int per = 1024, slices = (M + per - 1) / per;
  float* part_w = workspace;
  float* part_b = workspace + (size_t)slices * N * K;
  Operand A{dy, y_or_null, N, N}, B{x, nullptr, K, K};
  Epilogue E{part_w, K, nullptr, 0, part_b};
  dim3 grid((N + BM - 1) / BM, (K + BN - 1) / BN, slices);
  hipLaunchKernelGGL((k_mlp_gemm<false, false, true>), grid, dim3(256), 0, (hipStream_t)stream, A, B, M, per, E, N, K);
  const int nw = N * K;
  if (slices > 1) {
    hipLaunchKernelGGL(k_mlp_reduce_slices, dim3((nw + 255) / 256), dim3(256), 0, (hipStream_t)stream, part_w, dw, nw, slices);
  } else {
    if (hipMemcpyAsync(dw, part_w, (size_t)nw * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess)
      return mlp_fail("shf_mlp_linear_backward_weight: copy failed");
  }
  if (db) hipLaunchKernelGGL(k_mlp_reduce_slices, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, part_b, db, N, slices);
  return hipGetLastError() == hipSuccess ? 0 : mlp_fail("shf_mlp_linear_backward_weight: launch failed");
}
