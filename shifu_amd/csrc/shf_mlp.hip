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
constexpr int WGRAD_ROWS = 512;                           // rows of the batch per weight-gradient slice (split over M)

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

// One thread's share (16 elements) of a [128 x 32] operand tile, fetched from global memory into registers (fp32, with
// the optional ELU' mask applied), and later written to LDS as bf16 with the reduction index contiguous.  Fetch and
// commit are separate so that the next tile's loads are in flight while the MFMAs of the current one run.
//   RED_CONTIG : thread t -> row t / 2, 16 consecutive k (k half t % 2): four 16-byte loads, two 16-byte LDS stores
//   otherwise  : thread t -> 4 output rows (t % 32) * 4 .. + 3 x 4 k rows (t / 32) * 4 .. + 3: four 16-byte loads
//                along the output index (coalesced over t % 32), transposed in registers, four 8-byte LDS stores
// VEC = the operand's rows are 16-byte aligned (ld % 4 == 0, base aligned): else element-wise loads with the same mapping.
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;

struct TileRegs { float v[16]; };

template <bool RED_CONTIG, bool VEC>
MLP_DEV void fetch_tile(const Operand& O, int r0, int k0, int red, TileRegs& T) {
  const int t = (int)threadIdx.x;
  if (RED_CONTIG) {
    const int gr = r0 + (t >> 1), gk0 = k0 + (t & 1) * 16;
    const bool rok = gr < O.rows;
    const size_t base = (size_t)gr * O.ld + gk0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int gk = gk0 + 4 * q;
      if (VEC && rok && gk + 3 < red) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(O.p + base + 4 * q);
        f32x4 y = {1.0f, 1.0f, 1.0f, 1.0f};
        if (O.mask_y) {
          const f32x4 m = *reinterpret_cast<const f32x4*>(O.mask_y + base + 4 * q);
#pragma unroll
          for (int c = 0; c < 4; c++) y[c] = elu_grad_from_output(m[c]);
        }
#pragma unroll
        for (int c = 0; c < 4; c++) T.v[4 * q + c] = x[c] * y[c];
      } else {
#pragma unroll
        for (int c = 0; c < 4; c++) {
          float v = 0.0f;
          if (rok && gk + c < red) {
            v = O.p[base + 4 * q + c];
            if (O.mask_y) v *= elu_grad_from_output(O.mask_y[base + 4 * q + c]);
          }
          T.v[4 * q + c] = v;
        }
      }
    }
  } else {
    const int gr0 = r0 + (t & 31) * 4, gk0 = k0 + (t >> 5) * 4;
#pragma unroll
    for (int q = 0; q < 4; q++) {          // q: k row
      const int gk = gk0 + q;
      const size_t base = (size_t)gk * O.ld + gr0;
      if (VEC && gk < red && gr0 + 3 < O.rows) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(O.p + base);
        f32x4 y = {1.0f, 1.0f, 1.0f, 1.0f};
        if (O.mask_y) {
          const f32x4 m = *reinterpret_cast<const f32x4*>(O.mask_y + base);
#pragma unroll
          for (int c = 0; c < 4; c++) y[c] = elu_grad_from_output(m[c]);
        }
#pragma unroll
        for (int c = 0; c < 4; c++) T.v[4 * c + q] = x[c] * y[c];      // transposed: v[row c][k q]
      } else {
#pragma unroll
        for (int c = 0; c < 4; c++) {
          float v = 0.0f;
          if (gk < red && gr0 + c < O.rows) {
            v = O.p[base + c];
            if (O.mask_y) v *= elu_grad_from_output(O.mask_y[base + c]);
          }
          T.v[4 * c + q] = v;
        }
      }
    }
  }
}

template <bool RED_CONTIG>
MLP_DEV void commit_tile(const TileRegs& T, uint16_t* lds) {
  const int t = (int)threadIdx.x;
  if (RED_CONTIG) {
    uint16_t* dst = lds + (t >> 1) * LDT + (t & 1) * 16;
#pragma unroll
    for (int h = 0; h < 2; h++) {
      bf16x8 o;
#pragma unroll
      for (int c = 0; c < 8; c++) o[c] = (__bf16)T.v[8 * h + c];
      *reinterpret_cast<bf16x8*>(dst + 8 * h) = o;
    }
  } else {
#pragma unroll
    for (int c = 0; c < 4; c++) {          // c: output row
      bf16x4 o;
#pragma unroll
      for (int q = 0; q < 4; q++) o[q] = (__bf16)T.v[4 * c + q];
      *reinterpret_cast<bf16x4*>(lds + ((t & 31) * 4 + c) * LDT + (t >> 5) * 4) = o;
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
template <bool A_RED_CONTIG, bool B_RED_CONTIG, bool COLSUM_A, bool A_VEC, bool B_VEC>
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

  TileRegs ta, tb;
  fetch_tile<A_RED_CONTIG, A_VEC>(A, r0, red0, red1, ta);
  fetch_tile<B_RED_CONTIG, B_VEC>(B, c0, red0, red1, tb);
  for (int k0 = red0; k0 < red1; k0 += BK) {
    __syncthreads();                       // the previous tile's fragment reads are done
    commit_tile<A_RED_CONTIG>(ta, As);
    commit_tile<B_RED_CONTIG>(tb, Bs);
    __syncthreads();
    if (k0 + BK < red1) {                  // next tile's global loads fly under this tile's MFMAs
      fetch_tile<A_RED_CONTIG, A_VEC>(A, r0, k0 + BK, red1, ta);
      fetch_tile<B_RED_CONTIG, B_VEC>(B, c0, k0 + BK, red1, tb);
    }
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
bool vec_ok(const Operand& O) {
  return O.ld % 4 == 0 && ((uintptr_t)O.p & 15u) == 0 && (!O.mask_y || ((uintptr_t)O.mask_y & 15u) == 0);
}
template <bool AR, bool BR, bool CS>
void launch_gemm(dim3 grid, hipStream_t st, const Operand& A, const Operand& B, int red, int per, const Epilogue& E, int rows, int cols) {
  const bool av = vec_ok(A), bv = vec_ok(B);
  if (av && bv) hipLaunchKernelGGL((k_mlp_gemm<AR, BR, CS, true, true>), grid, dim3(256), 0, st, A, B, red, per, E, rows, cols);
  else if (av) hipLaunchKernelGGL((k_mlp_gemm<AR, BR, CS, true, false>), grid, dim3(256), 0, st, A, B, red, per, E, rows, cols);
  else if (bv) hipLaunchKernelGGL((k_mlp_gemm<AR, BR, CS, false, true>), grid, dim3(256), 0, st, A, B, red, per, E, rows, cols);
  else hipLaunchKernelGGL((k_mlp_gemm<AR, BR, CS, false, false>), grid, dim3(256), 0, st, A, B, red, per, E, rows, cols);
}

}  // namespace

extern "C" const char* shf_mlp_last_error(void) { return g_mlp_err.c_str(); }

extern "C" int shf_mlp_linear_forward(const float* x, const float* w, const float* b, float* y, int32_t M, int32_t K, int32_t N,
                                      int32_t act, void* stream) {
  if (!x || !w || !y) return mlp_fail("shf_mlp_linear_forward: null tensor");
  if (M <= 0 || K <= 0 || N <= 0 || act < 0 || act > 1) return mlp_fail("shf_mlp_linear_forward: bad shape / activation");
  Operand A{x, nullptr, K, M}, B{w, nullptr, K, N};
  Epilogue E{y, N, b, act, nullptr};
  dim3 grid((M + BM - 1) / BM, (N + BN - 1) / BN, 1);
  launch_gemm<true, true, false>(grid, (hipStream_t)stream, A, B, K, K, E, M, N);
  return hipGetLastError() == hipSuccess ? 0 : mlp_fail("shf_mlp_linear_forward: launch failed");
}

extern "C" int shf_mlp_linear_backward_input(const float* dy, const float* y_or_null, const float* w, float* dx, int32_t M,
                                             int32_t K, int32_t N, void* stream) {
  if (!dy || !w || !dx) return mlp_fail("shf_mlp_linear_backward_input: null tensor");
  // dX[M,K] = G[M,N] W[N,K]: A = G (reduction index n contiguous), B[col = k][red = n] = W[n][k] (transposed access)
  Operand A{dy, y_or_null, N, M}, B{w, nullptr, K, K};
  Epilogue E{dx, K, nullptr, 0, nullptr};
  dim3 grid((M + BM - 1) / BM, (K + BN - 1) / BN, 1);
  launch_gemm<true, false, false>(grid, (hipStream_t)stream, A, B, N, N, E, M, K);
  return hipGetLastError() == hipSuccess ? 0 : mlp_fail("shf_mlp_linear_backward_input: launch failed");
}

extern "C" int shf_mlp_backward_weight_workspace(int32_t M, int32_t K, int32_t N, int64_t* floats) {
  if (!floats) return mlp_fail("shf_mlp_backward_weight_workspace: null");
  const int slices = (M + WGRAD_ROWS - 1) / WGRAD_ROWS;
  *floats = (int64_t)slices * ((int64_t)N * K + N);
  return 0;
}

extern "C" int shf_mlp_linear_backward_weight(const float* dy, const float* y_or_null, const float* x, float* dw, float* db,
                                              float* workspace, int32_t M, int32_t K, int32_t N, void* stream) {
  if (!dy || !x || !dw || !workspace) return mlp_fail("shf_mlp_linear_backward_weight: null tensor");
  // dW[N,K] = G^T X: rows = n, cols = k, reduction over m; both operands are stored with the reduction index as the row
  const int per = WGRAD_ROWS, slices = (M + per - 1) / per;
  float* part_w = workspace;
  float* part_b = workspace + (size_t)slices * N * K;
  Operand A{dy, y_or_null, N, N}, B{x, nullptr, K, K};
  Epilogue E{part_w, K, nullptr, 0, part_b};
  dim3 grid((N + BM - 1) / BM, (K + BN - 1) / BN, slices);
  launch_gemm<false, false, true>(grid, (hipStream_t)stream, A, B, M, per, E, N, K);
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
