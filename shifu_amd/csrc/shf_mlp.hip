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

#include <atomic>
#include <cstdlib>
#include <string>

#include "../../include/shifu_amd.h"

#define MLP_DEV __device__ __forceinline__

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

namespace {

#ifdef SHF_MLP_PROBE_CLOCK
__device__ long long g_mlp_probe[4 * 8192];   // per block: start, loop end, epilogue end, (unused)
#define MLP_CLOCK(slot) do { const unsigned lb = blockIdx.x + gridDim.x * blockIdx.y; if (threadIdx.x == 0 && blockIdx.z == 0 && lb < 8192) g_mlp_probe[4 * lb + (slot)] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define MLP_CLOCK(slot)
#endif

#ifndef SHF_MLP_PREFETCH_DEPTH
#define SHF_MLP_PREFETCH_DEPTH 1
#endif
constexpr int PF = SHF_MLP_PREFETCH_DEPTH;   // operand tiles kept in flight in registers (1: measured faster than 2 -- the
                                             // registers of a second tile cost a resident block per CU, profiles/r02_mlp_probe.md)
constexpr int BN = 128, BK = 32, LDT = BK + 8;   // LDS row = 40 bf16 = 80 B: 16-byte aligned fragments, staggered banks
// rows of the batch per weight-gradient slice (split over M): 512, less for the small layers whose few output tiles would
// otherwise leave most CUs without a block
int wgrad_rows(int M, int K, int N) {
  const long tiles = (long)((N + 127) / 128) * ((K + 127) / 128);
  static const char* force = getenv("SHF_MLP_WGRAD_ROWS");       // experiments (tools/mlp_probe.py)
  if (force && atoi(force) >= 32) return atoi(force) / 32 * 32;
  int per = 512;
  while (per > 128 && tiles * ((M + per - 1) / per) < 384) per >>= 1;
  return per;
}

MLP_DEV uint16_t to_bf16(float x) {
  uint32_t u = __float_as_uint(x);
  u += 0x7FFFu + ((u >> 16) & 1u);   // round to nearest even (inputs are finite)
  return (uint16_t)(u >> 16);
}
// Workgroup barrier that orders LDS traffic only.  __syncthreads() is a workgroup-scope fence + barrier, and the fence
// drains every outstanding *global* load as well (s_waitcnt vmcnt(0)) -- which would serialise the K loop on HBM latency
// and defeat the register prefetch of the next tiles.  Here only the LDS counter is waited on.
MLP_DEV void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
MLP_DEV float elu_grad_from_output(float y) { return y > 0.0f ? 1.0f : y + 1.0f; }

// How the A ([rows x red]) and B ([cols x red]) operands of  C[rows, cols] = sum_red A[row, red] B[col, red]  sit in
// global memory: RED_CONTIG = element (r, k) at p[r * ld + k] (reduction index contiguous), else at p[k * ld + r].
struct Operand {
  const float* p;
  const float* mask_y;   // optional: multiply by act'(mask_y[...]) with the same indexing (G = dY (.) act'(Y))
  int ld;
  int rows;              // extent along the output index
};

// One thread's share (16 elements) of a [128 x 32] operand tile: fetch_tile only *issues* the global loads (raw fp32
// words, plus the raw words of the optional ELU' mask operand) -- no arithmetic on the loaded values and no branches, so no
// s_waitcnt lands between the loads and they stay in flight under the MFMAs of the tiles before; out-of-range elements
// are read from a clamped (valid) address and zeroed later.  commit_tile applies range mask and ELU' mask, converts to
// bf16 and writes LDS with the reduction index contiguous.  Element j of thread t sits at (row, k) of the tile:
//   RED_CONTIG, VEC  : row = t / 8 + 32 (j / 4), k = 4 (t % 8) + j % 4   -- eight lanes read one whole 128-byte row
//                      segment with 16-byte loads; four 8-byte LDS stores
//   RED_CONTIG, !VEC : row = t / 32 + 8 j,       k = t % 32             -- 32 lanes read one row segment word by word
//                      (rows not 16-byte aligned: the 259-wide observation); sixteen 2-byte LDS stores
//   !RED_CONTIG      : row = 4 (t % 32) + j % 4, k = 4 (t / 32) + j / 4  -- loads run along the output index
//                      (coalesced over t % 32), transposed in registers; four 8-byte LDS stores
// VEC = 16-byte chunks are whole (ld % 4 == 0, extents % 4 == 0, base aligned).
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;

template <bool MASK, int NE>   // NE elements per thread: 16 for a 128-row tile, 8 for a 64-row tile
struct TileRegs {
  float x[NE];
  float m[MASK ? NE : 1];
  uint32_t ok;            // bit j: element j is inside the operand
};

template <bool RED_CONTIG, bool VEC>
MLP_DEV void tile_elem(int t, int j, int* row, int* k) {
  if (RED_CONTIG && VEC) { *row = (t >> 3) + 32 * (j >> 2); *k = 4 * (t & 7) + (j & 3); }
  else if (RED_CONTIG) { *row = (t >> 5) + 8 * j; *k = t & 31; }
  else { *row = 4 * (t & 31) + (j & 3); *k = 4 * (t >> 5) + (j >> 2); }
}

template <bool RED_CONTIG, bool VEC, bool MASK, int NE>
MLP_DEV void fetch_tile(const Operand& O, int r0, int k0, int red, TileRegs<MASK, NE>& T) {
  static_assert(NE == 16 || RED_CONTIG, "64-row tiles only for operands with the reduction index contiguous");
  const int t = (int)threadIdx.x;
  uint32_t ok = 0u;
  uint32_t off[NE];        // element offsets fit 32 bits (the host refuses operands of 2^32 elements or more)
#pragma unroll
  for (int j = 0; j < NE; j++) {
    int row, k;
    tile_elem<RED_CONTIG, VEC>(t, j, &row, &k);
    const bool in = r0 + row < O.rows && k0 + k < red;
    ok |= (in ? 1u : 0u) << j;
    const uint32_t o = RED_CONTIG ? (uint32_t)(r0 + row) * (uint32_t)O.ld + (uint32_t)(k0 + k)
                                  : (uint32_t)(k0 + k) * (uint32_t)O.ld + (uint32_t)(r0 + row);
    off[j] = in ? o : 0u;
  }
  if (VEC) {
#pragma unroll
    for (int q = 0; q < NE / 4; q++) {                  // elements 4 q .. 4 q + 3 are contiguous: whole chunk in or out
      const f32x4 v = *reinterpret_cast<const f32x4*>(O.p + off[4 * q]);
#pragma unroll
      for (int c = 0; c < 4; c++) T.x[4 * q + c] = v[c];
      if (MASK) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(O.mask_y + off[4 * q]);
#pragma unroll
        for (int c = 0; c < 4; c++) T.m[4 * q + c] = w[c];
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < NE; j++) {
      T.x[j] = O.p[off[j]];
      if (MASK) T.m[j] = O.mask_y[off[j]];
    }
  }
  T.ok = ok;
}

template <bool RED_CONTIG, bool VEC, bool MASK, int NE>
MLP_DEV void commit_values(const float* v, uint16_t* lds);

// SPLIT: every value goes to LDS twice -- its bf16 head at `lds`, the bf16 of what the head lost at `lds + lo_off`
template <bool RED_CONTIG, bool VEC, bool MASK, int NE, bool SPLIT>
MLP_DEV void commit_tile(const TileRegs<MASK, NE>& T, uint16_t* lds, int lo_off) {
  float v[NE];
#pragma unroll
  for (int j = 0; j < NE; j++) {
    float x = T.x[j];
    if (MASK) x *= elu_grad_from_output(T.m[j]);
    v[j] = (T.ok >> j) & 1u ? x : 0.0f;
  }
  commit_values<RED_CONTIG, VEC, MASK, NE>(v, lds);
  if (SPLIT) {
    float w[NE];
#pragma unroll
    for (int j = 0; j < NE; j++) w[j] = v[j] - (float)(__bf16)v[j];
    commit_values<RED_CONTIG, VEC, MASK, NE>(w, lds + lo_off);
  }
}

template <bool RED_CONTIG, bool VEC, bool MASK, int NE>
MLP_DEV void commit_values(const float* v, uint16_t* lds) {
  const int t = (int)threadIdx.x;
  if (RED_CONTIG && VEC) {
#pragma unroll
    for (int q = 0; q < NE / 4; q++) {
      bf16x4 o;
#pragma unroll
      for (int c = 0; c < 4; c++) o[c] = (__bf16)v[4 * q + c];
      *reinterpret_cast<bf16x4*>(lds + ((t >> 3) + 32 * q) * LDT + 4 * (t & 7)) = o;
    }
  } else if (RED_CONTIG) {
#pragma unroll
    for (int j = 0; j < NE; j++) {
      const __bf16 o = (__bf16)v[j];
      lds[((t >> 5) + 8 * j) * LDT + (t & 31)] = __builtin_bit_cast(uint16_t, o);
    }
  } else {
#pragma unroll
    for (int c = 0; c < 4; c++) {          // c: output row; elements c, 4 + c, 8 + c, 12 + c are k .. k + 3
      bf16x4 o;
#pragma unroll
      for (int q = 0; q < 4; q++) o[q] = (__bf16)v[(4 * q + c) % NE];
      *reinterpret_cast<bf16x4*>(lds + (4 * (t & 31) + c) * LDT + 4 * (t >> 5)) = o;
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
// BMT x 128 output tile per block: BMT = 128 (four waves of 64 x 64) or 64 (four waves of 32 x 64) -- the smaller tile
// doubles the number of blocks for the skinny layers, whose 128-row grids leave CUs idle (384 blocks on 256 CUs run as
// two rounds), and halves the registers so that more blocks are resident to hide the HBM latency of the K loop.
template <int BMT, bool A_RED_CONTIG, bool B_RED_CONTIG, bool COLSUM_A, bool A_VEC, bool B_VEC, bool A_MASK, bool SPLIT>
__global__ __launch_bounds__(256) void k_mlp_gemm(Operand A, Operand B, int red, int red_per_slice, Epilogue E, int rows, int cols) {
  constexpr int TI = BMT / 64, NEA = BMT / 8;     // MFMA tiles per wave along the rows; A elements per thread and K tile
  constexpr int ALO = BMT * LDT, BLO = BN * LDT;  // SPLIT: offsets of the tail tiles behind the head tiles
  // one array: the operand tiles, and -- after the loop -- the epilogue's four 32 x 36 float patches (EPI_WORDS)
  constexpr int AS_WORDS = (SPLIT ? 2 : 1) * BMT * LDT, BS_WORDS = (SPLIT ? 2 : 1) * BN * LDT;
  constexpr int EPI_TS = 36, EPI_BYTES = 4 * 32 * EPI_TS * 4;
  constexpr int LDS_BYTES = (AS_WORDS + BS_WORDS) * 2 > EPI_BYTES ? (AS_WORDS + BS_WORDS) * 2 : EPI_BYTES;
  __shared__ __attribute__((aligned(16))) unsigned char lds_all[LDS_BYTES];
  uint16_t* As = reinterpret_cast<uint16_t*>(lds_all);
  uint16_t* Bs = As + AS_WORDS;
  MLP_CLOCK(0);
  // Workgroups go to the eight XCDs round-robin by linear id, and each XCD has its own L2.  With a split reduction
  // (weight gradient) the tiles of one slice share that slice's rows of both operands: renumber so that an XCD works
  // through whole slices -- consecutive tiles of a slice on the same L2 -- instead of every XCD fetching every slice.
  int bx = (int)blockIdx.x, by = (int)blockIdx.y, bz = (int)blockIdx.z;
#ifndef SHF_MLP_NO_XCD_SWIZZLE
  if (gridDim.z > 1) {
    const int gx = (int)gridDim.x, gy = (int)gridDim.y, nblk = gx * gy * (int)gridDim.z;
    if ((nblk & 7) == 0) {
      const int lin = bx + gx * (by + gy * bz);
      const int ren = (lin & 7) * (nblk >> 3) + (lin >> 3);
      bx = ren % gx; by = (ren / gx) % gy; bz = ren / (gx * gy);
    }
  }
#endif
  const int r0 = bx * BMT, c0 = by * BN;
  const int red0 = bz * red_per_slice;
#ifdef SHF_MLP_PROBE_NO_LOOP
  const int red1 = red0;                       // (probe build: epilogue only)
#else
  const int red1 = red0 + red_per_slice < red ? red0 + red_per_slice : red;
#endif
  const int wave = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
  const int wr = (wave >> 1) * (32 * TI), wc = (wave & 1) * 64;   // this wave's corner inside the block tile
  f32x16 acc[TI][2];
#pragma unroll
  for (int i = 0; i < TI; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int k = 0; k < 16; k++) acc[i][j][k] = 0.0f;
  float csum = 0.0f;   // COLSUM_A: column (= A row index) sums, thread t < BMT owns A row t of the tile

  // Two tiles per operand are kept in flight in registers: while tile k is multiplied, the global loads of tiles k + 1
  // and k + 2 are outstanding (the K loop of these skinny GEMMs is bound by HBM latency, not by the MFMAs).
  TileRegs<A_MASK, NEA> ta[PF];
  TileRegs<false, 16> tb[PF];
#pragma unroll
  for (int d = 0; d < PF; d++)
    if (d == 0 || red0 + d * BK < red1) {
      fetch_tile<A_RED_CONTIG, A_VEC, A_MASK, NEA>(A, r0, red0 + d * BK, red1, ta[d]);
      fetch_tile<B_RED_CONTIG, B_VEC, false, 16>(B, c0, red0 + d * BK, red1, tb[d]);
    }
  auto step = [&](int k0, TileRegs<A_MASK, NEA>& xa, TileRegs<false, 16>& xb) {
    lds_barrier();                         // the previous tile's fragment reads are done
    commit_tile<A_RED_CONTIG, A_VEC, A_MASK, NEA, SPLIT>(xa, As, ALO);
    commit_tile<B_RED_CONTIG, B_VEC, false, 16, SPLIT>(xb, Bs, BLO);
    lds_barrier();
    if (k0 + PF * BK < red1) {             // this register set is free again: tile k + PF
      fetch_tile<A_RED_CONTIG, A_VEC, A_MASK, NEA>(A, r0, k0 + PF * BK, red1, xa);
      fetch_tile<B_RED_CONTIG, B_VEC, false, 16>(B, c0, k0 + PF * BK, red1, xb);
    }
#ifndef SHF_MLP_PROBE_NO_COLSUM
    if (COLSUM_A && by == 0 && threadIdx.x < BMT) {
      // db: sum over the reduction index of the (bf16-rounded) G values of A row t -- the values the MFMA multiplies
      const uint16_t* row = As + threadIdx.x * LDT;
#pragma unroll
      for (int k = 0; k < BK; k++) {
        csum += __uint_as_float((uint32_t)row[k] << 16);
        if (SPLIT) csum += __uint_as_float((uint32_t)row[ALO + k] << 16);
      }
    }
#endif
#pragma unroll
    for (int kk = 0; kk < BK; kk += 16) {
      bf16x8 a[TI], b[2];
#pragma unroll
      for (int i = 0; i < TI; i++) a[i] = *reinterpret_cast<const bf16x8*>(As + (wr + 32 * i + (lane & 31)) * LDT + kk + 8 * (lane >> 5));
#pragma unroll
      for (int j = 0; j < 2; j++) b[j] = *reinterpret_cast<const bf16x8*>(Bs + (wc + 32 * j + (lane & 31)) * LDT + kk + 8 * (lane >> 5));
#ifndef SHF_MLP_PROBE_NO_MFMA
      if (SPLIT) {
        // the two cross terms first (small), then head * head; tail * tail (2^-18 relative) is dropped
        bf16x8 al[TI], bl[2];
#pragma unroll
        for (int i = 0; i < TI; i++) al[i] = *reinterpret_cast<const bf16x8*>(As + ALO + (wr + 32 * i + (lane & 31)) * LDT + kk + 8 * (lane >> 5));
#pragma unroll
        for (int j = 0; j < 2; j++) bl[j] = *reinterpret_cast<const bf16x8*>(Bs + BLO + (wc + 32 * j + (lane & 31)) * LDT + kk + 8 * (lane >> 5));
#pragma unroll
        for (int i = 0; i < TI; i++)
#pragma unroll
          for (int j = 0; j < 2; j++) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], bl[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], b[j], acc[i][j], 0, 0, 0);
          }
      }
#pragma unroll
      for (int i = 0; i < TI; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
#else
      acc[0][0][0] += (float)a[0][0] + (float)b[0][0] + (float)b[1][0];
#endif
    }
  };
  for (int k0 = red0; k0 < red1; k0 += PF * BK) {
#pragma unroll
    for (int d = 0; d < PF; d++)
      if (d == 0 || k0 + d * BK < red1) step(k0 + d * BK, ta[d], tb[d]);
  }

  MLP_CLOCK(1);
#ifdef SHF_MLP_PROBE_NO_EPILOGUE
  if (acc[0][0][0] == 12345.678f) E.c[0] = acc[TI - 1][1][3];   // (probe build: keep the accumulators alive, skip the stores)
  return;
#endif
  // epilogue: C/D layout of the 32 x 32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5) -- turned
  // through a wave-private LDS patch so that a lane stores 16 bytes and eight lanes cover a 128-byte row segment
  // (4-byte stores in the accumulator layout cost 5 - 10 us per call on the large layers, profiles/r04_mlp_panel.md)
  float* Cout = E.c + (size_t)bz * rows * E.ldc * (gridDim.z > 1 ? 1 : 0);   // split-K: slice z writes its own partial
  __syncthreads();                                                            // the operand tiles are dead
  float* T = reinterpret_cast<float*>(lds_all) + wave * 32 * EPI_TS;
  const bool vec_out = (E.ldc & 3) == 0 && ((uintptr_t)Cout & 15u) == 0;
#pragma unroll
  for (int i = 0; i < TI; i++)
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int col = c0 + wc + 32 * j + (lane & 31);
      const float bv = (E.bias && col < cols) ? E.bias[col] : 0.0f;
#pragma unroll
      for (int reg = 0; reg < 16; reg++) {
        float v = acc[i][j][reg] + bv;
#ifndef SHF_MLP_PROBE_NO_ELU
#ifdef SHF_MLP_PROBE_EXPM1
        if (E.act == 1) v = v > 0.0f ? v : expm1f(v);
#else
        if (E.act == 1) v = v > 0.0f ? v : __expf(v) - 1.0f;     // ELU, alpha = 1 (exp(x) - 1, as the stock elu kernel)
#endif
#endif
        T[((reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)) * EPI_TS + (lane & 31)] = v;
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const int tr = (lane >> 3) + 8 * g, row = r0 + wr + 32 * i + tr, c4 = c0 + wc + 32 * j + 4 * (lane & 7);
        const f32x4 o = *reinterpret_cast<const f32x4*>(T + tr * EPI_TS + 4 * (lane & 7));
        if (row < rows) {
          float* dst = Cout + (size_t)row * E.ldc + c4;
          if (vec_out && c4 + 3 < cols) *reinterpret_cast<f32x4*>(dst) = o;
          else {
#pragma unroll
            for (int c = 0; c < 4; c++)
              if (c4 + c < cols) dst[c] = o[c];
          }
        }
      }
      __builtin_amdgcn_wave_barrier();                                        // the patch is rewritten by the next tile
    }
  if (COLSUM_A && by == 0 && threadIdx.x < BMT) {
    const int row = r0 + (int)threadIdx.x;
    if (row < rows) E.colsum[(size_t)bz * rows + row] = csum;
  }
  MLP_CLOCK(2);
}

// out[i] = sum_s part[s][i], slices in ascending order (deterministic); dW and db of a layer in one launch.  The loads of
// eight slices go out together (the additions stay in slice order), so a thread pays slices / 8 round trips, not slices.
__global__ void k_mlp_reduce_slices(const float* part_w, float* dw, int nw, const float* part_b, float* db, int nb, int slices) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (i >= nw + nb) return;
  const bool isw = i < nw;
  const float* src = isw ? part_w + i : part_b + (i - nw);
  const size_t stride = isw ? (size_t)nw : (size_t)nb;
  float acc = 0.0f;
  for (int s0 = 0; s0 < slices; s0 += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; u++) v[u] = s0 + u < slices ? src[(size_t)(s0 + u) * stride] : 0.0f;
#pragma unroll
    for (int u = 0; u < 8; u++) acc += v[u];
  }
  if (isw) dw[i] = acc; else db[i - nw] = acc;
}

thread_local std::string g_mlp_err;
int mlp_fail(const std::string& m) { g_mlp_err = m; return 1; }
// bit of the current HIP device in the per-kernel "dynamic LDS limit raised" masks (hipFuncSetAttribute is per device)
static uint64_t mlp_device_bit() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  return 1ull << (dev & 63);
}
bool too_big(int64_t M, int64_t K, int64_t N) { return M * K >= (1ll << 32) || M * N >= (1ll << 32) || N * K >= (1ll << 32); }
bool vec_ok(const Operand& O, int red, bool red_contig) {
  // 16-byte chunks must be whole: aligned rows, and the extent along the contiguous index a multiple of 4
  const int contig_extent = red_contig ? red : O.rows;
  return O.ld % 4 == 0 && contig_extent % 4 == 0 && ((uintptr_t)O.p & 15u) == 0 && (!O.mask_y || ((uintptr_t)O.mask_y & 15u) == 0);
}
int g_mlp_precision = SHF_MLP_BF16X3;
template <int BMT, bool AR, bool BR, bool CS, bool AV, bool BV>
void launch_gemm_m(dim3 grid, hipStream_t st, const Operand& A, const Operand& B, int red, int per, const Epilogue& E, int rows, int cols) {
  // SHF_MLP_BF16X3_W1: the weight gradient (CS: the split reduction over the batch, whose 24 576 products average the
  // rounding out) takes its operands rounded once, forward and input gradient keep head + tail (profiles/r04_train.md)
  const bool split = CS ? g_mlp_precision == SHF_MLP_BF16X3 : g_mlp_precision != SHF_MLP_BF16;
  if (A.mask_y) {
    if (split) hipLaunchKernelGGL((k_mlp_gemm<BMT, AR, BR, CS, AV, BV, true, true>), grid, dim3(256), 0, st, A, B, red, per, E, rows, cols);
    else hipLaunchKernelGGL((k_mlp_gemm<BMT, AR, BR, CS, AV, BV, true, false>), grid, dim3(256), 0, st, A, B, red, per, E, rows, cols);
  } else {
    if (split) hipLaunchKernelGGL((k_mlp_gemm<BMT, AR, BR, CS, AV, BV, false, true>), grid, dim3(256), 0, st, A, B, red, per, E, rows, cols);
    else hipLaunchKernelGGL((k_mlp_gemm<BMT, AR, BR, CS, AV, BV, false, false>), grid, dim3(256), 0, st, A, B, red, per, E, rows, cols);
  }
}
// rows of the output tile: 64 when the A operand allows it (reduction index contiguous) and 128-row tiles would not give
// every CU at least three blocks
int pick_bm(bool a_red_contig, int rows, int cols, int slices) {
  if (!a_red_contig) return 128;
  static const char* force = getenv("SHF_MLP_FORCE_BM");     // experiments (tools/mlp_probe.py)
  if (force) return atoi(force) == 64 ? 64 : 128;
  const long tiles128 = (long)((rows + 127) / 128) * ((cols + BN - 1) / BN) * slices;
  return tiles128 >= 3 * 256 ? 128 : 64;
}
template <bool AR, bool BR, bool CS>
void launch_gemm(hipStream_t st, const Operand& A, const Operand& B, int red, int per, int slices, const Epilogue& E, int rows, int cols) {
  const bool av = vec_ok(A, red, AR) && per % 4 == 0, bv = vec_ok(B, red, BR) && per % 4 == 0;
  const int bm = pick_bm(AR, rows, cols, slices);
  const dim3 grid((rows + bm - 1) / bm, (cols + BN - 1) / BN, slices);
#define SHF_MLP_LAUNCH(BMV)                                                                               \
  do {                                                                                                    \
    if (av && bv) launch_gemm_m<BMV, AR, BR, CS, true, true>(grid, st, A, B, red, per, E, rows, cols);    \
    else if (av) launch_gemm_m<BMV, AR, BR, CS, true, false>(grid, st, A, B, red, per, E, rows, cols);    \
    else if (bv) launch_gemm_m<BMV, AR, BR, CS, false, true>(grid, st, A, B, red, per, E, rows, cols);    \
    else launch_gemm_m<BMV, AR, BR, CS, false, false>(grid, st, A, B, red, per, E, rows, cols);           \
  } while (0)
  if constexpr (AR) {
    if (bm == 64) { SHF_MLP_LAUNCH(64); return; }
  }
  SHF_MLP_LAUNCH(128);
#undef SHF_MLP_LAUNCH
}


// ---------------------------------------------------------------------------------------------------------------------
// Row-panel kernels (round 4): forward and input gradient of a layer as  C[M, cols] = A[M, red] B[cols, red]^T  with M as
// the only grid dimension.  These GEMMs are skinny (red, cols <= 512 against M = 24 576) and HBM-bound on A and C, so:
//   * a block owns BM whole rows of A: the panel [BM x red] is one contiguous range of memory, read ONCE with 16-byte
//     loads (many in flight, no K loop around them), converted to bf16 head (+ tail) and parked in LDS;
//   * B (the weights, <= 0.5 MB, L2-resident) is not staged through LDS at all: a tiny pack kernel lays it out once per
//     optimizer step in MFMA fragment order -- [column tile][k step][lane] -> 8 bf16 = 16 bytes -- so that a wave's B
//     fragment is one contiguous 1 KB load straight into registers (next k step's fragments in flight under this step's
//     MFMAs); head and tail arrays, plain and transposed (the input gradient multiplies by W, not W^T);
//   * after the one barrier behind the panel store there is no barrier in the loop: each wave owns column tiles and runs
//     its own reduction over the whole panel, epilogue (bias, ELU) straight from the accumulators.
// The k steps and the order of the three MFMAs per step are those of k_mlp_gemm, so results equal that kernel's bit for bit.
constexpr int PANEL_PAD = 8;     // bf16 of padding per LDS row: row stride = (redp + 8) / 2 words = 4 mod 8 -> conflict-free fragments

struct PanelArgs {
  const float* a;        // [M, red] row-major, ld = red
  const float* mask_y;   // optional, same shape: A is multiplied by act'(mask_y)
  int M, red, redp;      // redp = red rounded up to 16 (the pack's k extent)
  uint32_t red_magic;    // floor(2^32 / red) + 1:  e / red = umulhi(e, magic) for e * red < 2^32
  const uint4* bhi;      // [ceil(cols / 32)][redp / 16][64] fragments of the bf16 heads
  const uint4* blo;      // ... of the tails (bf16x3 only)
  float* c;              // [M, cols], ld = ldc
  int ldc, cols;
  const float* bias;     // per column or null
  int act;
};

// Streaming form: the block's BM rows of A pass through LDS in chunks of KC reduction indices, two buffers: the loads of
// chunk c + 1 are issued (into registers) before the MFMAs of chunk c and committed to the other buffer after them -- one
// barrier per chunk, HBM latency under a chunk's 50 - 200 MFMAs per wave.  Each wave keeps the accumulators of ALL its
// column tiles for the whole reduction (CT = ceil(column tiles / 4) <= 4), so A is read once.
constexpr int KC = 128, KCS = KC + PANEL_PAD;   // chunk row stride 136 bf16 = 68 words = 4 mod 64: conflict-free 16-byte fragments

template <int BM, int CT, bool MASK, bool SPLIT>
__global__ __launch_bounds__(256, 2) void k_mlp_panel(PanelArgs P) {
  extern __shared__ __attribute__((aligned(16))) uint16_t panel_lds[];
  constexpr int RT = BM / 32;
  constexpr int BUF = (SPLIT ? 2 : 1) * BM * KCS;            // one buffer: heads, then tails
  constexpr int NV = BM * KC / 4 / 256;                      // 16-byte chunks per thread and K chunk (8 for 64 rows)
  const int t = (int)threadIdx.x, wave = t >> 6, lane = t & 63;
  const int r0 = (int)blockIdx.x * BM;
  const int red = P.red;
  const bool vec = (red & 3) == 0;                           // rows 16-byte aligned (the host checked the base)
  const int nchunks = (P.redp + KC - 1) / KC;
  const int nks = P.redp >> 4, nct = (P.cols + 31) >> 5;

  // element (row, k) of 16-byte chunk u of this thread inside a K chunk: 32 lanes cover one row's 512 bytes
  f32x4 v[NV], m[MASK ? NV : 1];
  auto issue = [&](int kc0) {
#pragma unroll
    for (int u = 0; u < NV; u++) {
      const int idx = t + 256 * u, row = idx >> 5, k = kc0 + 4 * (idx & 31);
      const bool rin = r0 + row < P.M;
      const size_t off = (size_t)(r0 + row) * red + k;
      v[u] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      if (MASK) m[u] = v[u];
      if (vec) {
#ifdef SHF_PANEL_NO_ALOAD
        if (rin && k < red && P.act == 77) {
#else
        if (rin && k < red) {
#endif
          v[u] = *reinterpret_cast<const f32x4*>(P.a + off);
          if (MASK) m[u] = *reinterpret_cast<const f32x4*>(P.mask_y + off);
        }
      } else if (rin) {
#pragma unroll
        for (int c = 0; c < 4; c++)
          if (k + c < red) { v[u][c] = P.a[off + c]; if (MASK) m[u][c] = P.mask_y[off + c]; }
      }
    }
  };
  auto commit = [&](uint16_t* buf) {
#pragma unroll
    for (int u = 0; u < NV; u++) {
      const int idx = t + 256 * u, row = idx >> 5, k = 4 * (idx & 31);
      bf16x4 h, l;
#pragma unroll
      for (int c = 0; c < 4; c++) {
        const float x = MASK ? v[u][c] * elu_grad_from_output(m[u][c]) : v[u][c];
        h[c] = (__bf16)x;
        if (SPLIT) l[c] = (__bf16)(x - (float)h[c]);
      }
      *reinterpret_cast<bf16x4*>(buf + row * KCS + k) = h;
      if (SPLIT) *reinterpret_cast<bf16x4*>(buf + BM * KCS + row * KCS + k) = l;
    }
  };

  const int ct0 = wave * CT;
  const bool wave_on = ct0 < nct;                             // a wave without column tiles only helps moving A
  f32x16 acc[RT][CT];
#pragma unroll
  for (int i = 0; i < RT; i++)
#pragma unroll
    for (int j = 0; j < CT; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;
  size_t boff[CT];
#pragma unroll
  for (int j = 0; j < CT; j++) boff[j] = (size_t)(ct0 + j < nct ? ct0 + j : nct - 1) * nks * 64 + lane;
  // B fragments: a ring of D k steps in flight (straight from the pack in L2), refilled as each step is consumed
  constexpr int D = (MASK && SPLIT && BM == 64 && CT == 2) ? 2 : 8 / CT;   // (that form would spill 21 registers at depth 4)
  uint4 rh[D][CT], rl[D][SPLIT ? CT : 1];
  if (wave_on) {
#pragma unroll
    for (int d = 0; d < D; d++)
      if (d < nks) {
#pragma unroll
        for (int j = 0; j < CT; j++) { rh[d][j] = P.bhi[boff[j] + (size_t)d * 64]; if (SPLIT) rl[d][j] = P.blo[boff[j] + (size_t)d * 64]; }
      }
  }
  MLP_CLOCK(0);
  issue(0);
  commit(panel_lds);
  __syncthreads();
  MLP_CLOCK(1);
  const int frag = (lane & 31) * KCS + 8 * (lane >> 5);
  for (int c = 0; c < nchunks; c++) {
    const uint16_t* Ah = panel_lds + (c & 1) * BUF;
    const uint16_t* Al = Ah + BM * KCS;
    if (c + 1 < nchunks) issue((c + 1) * KC);
    if (wave_on) {
      const int ks_end = (c + 1) * (KC / 16) < nks ? (c + 1) * (KC / 16) : nks;
      static_assert((KC / 16) % D == 0, "the ring position of a chunk's first k step must be 0");
      for (int ks0 = c * (KC / 16); ks0 < ks_end; ks0 += D) {
#pragma unroll
        for (int d = 0; d < D; d++) {
          const int ks = ks0 + d;
          if (ks >= ks_end) break;
          bf16x8 bh[CT], bl[CT];
#pragma unroll
          for (int j = 0; j < CT; j++) { bh[j] = __builtin_bit_cast(bf16x8, rh[d][j]); if (SPLIT) bl[j] = __builtin_bit_cast(bf16x8, rl[d][j]); }
#ifdef SHF_PANEL_NO_BLOAD
          if (ks + D < nks && P.act == 77) {
#else
          if (ks + D < nks) {
#endif
#pragma unroll
            for (int j = 0; j < CT; j++) {
              rh[d][j] = P.bhi[boff[j] + (size_t)(ks + D) * 64];
              if (SPLIT) rl[d][j] = P.blo[boff[j] + (size_t)(ks + D) * 64];
            }
          }
          const int kl = 16 * (ks - c * (KC / 16));
          bf16x8 ah[RT], al[RT];
#pragma unroll
          for (int i = 0; i < RT; i++) {
            ah[i] = *reinterpret_cast<const bf16x8*>(Ah + 32 * i * KCS + frag + kl);
            if (SPLIT) al[i] = *reinterpret_cast<const bf16x8*>(Al + 32 * i * KCS + frag + kl);
          }
#ifdef SHF_PANEL_NO_MFMA
#pragma unroll
          for (int i = 0; i < RT; i++)
#pragma unroll
            for (int j = 0; j < CT; j++) acc[i][j][0] += (float)ah[i][0] + (float)bh[j][0] + (SPLIT ? (float)al[i][0] + (float)bl[j][0] : 0.0f);
#else
          if (SPLIT) {
#pragma unroll
            for (int i = 0; i < RT; i++)
#pragma unroll
              for (int j = 0; j < CT; j++) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
              }
          }
#pragma unroll
          for (int i = 0; i < RT; i++)
#pragma unroll
            for (int j = 0; j < CT; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
#endif
        }
      }
    }
    if (c + 1 < nchunks) {
      commit(panel_lds + ((c + 1) & 1) * BUF);
      lds_barrier();          // chunk c + 1 is in LDS, and every wave is done reading chunk c's buffer (rewritten next round)
    }
  }
  MLP_CLOCK(2);
  // Epilogue: the accumulator layout gives a lane one column and 16 rows -- stores of 4 bytes per lane, 128 bytes per row
  // segment.  Each wave turns its 32 x 32 tiles through a private LDS patch instead (the chunk buffers are idle now) and
  // writes 16 bytes per lane, eight lanes per row segment: 5 - 10 us per call on the 24 576-row layers
  // (profiles/r04_mlp_panel.md).  Same values, so the outputs still equal the tiled kernel's bit for bit.
  __syncthreads();                                            // every wave is done with the chunk buffers
  if (!wave_on) return;
  constexpr int TS = 36;                                      // patch row stride (floats): 16-byte aligned rows, staggered banks
  float* T = reinterpret_cast<float*>(panel_lds) + wave * 32 * TS;
  const bool vec_out = (P.ldc & 3) == 0 && ((uintptr_t)P.c & 15u) == 0;
#pragma unroll
  for (int i = 0; i < RT; i++)
#pragma unroll
    for (int j = 0; j < CT; j++) {
      if (ct0 + j >= nct) continue;                           // wave-uniform
      const int col = 32 * (ct0 + j) + (lane & 31);
      const float bv = (P.bias && col < P.cols) ? P.bias[col] : 0.0f;
#pragma unroll
      for (int reg = 0; reg < 16; reg++) {
        float v = acc[i][j][reg] + bv;
        if (P.act == 1) v = v > 0.0f ? v : __expf(v) - 1.0f;
        T[((reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)) * TS + (lane & 31)] = v;
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const int tr = (lane >> 3) + 8 * g, row = r0 + 32 * i + tr, c4 = 32 * (ct0 + j) + 4 * (lane & 7);
        const f32x4 o = *reinterpret_cast<const f32x4*>(T + tr * TS + 4 * (lane & 7));
#ifdef SHF_PANEL_NO_STORE
        if (row < P.M && o[0] == 12345.678f) {
#else
        if (row < P.M) {
#endif
          float* dst = P.c + (size_t)row * P.ldc + c4;
          if (vec_out && c4 + 3 < P.cols) *reinterpret_cast<f32x4*>(dst) = o;
          else {
#pragma unroll
            for (int c = 0; c < 4; c++)
              if (c4 + c < P.cols) dst[c] = o[c];
          }
        }
      }
      __builtin_amdgcn_wave_barrier();                        // the patch is rewritten by the next tile
    }
  MLP_CLOCK(3);
}

// ---------------------------------------------------------------------------------------------------------------------
// Chained forward: ALL layers of an MLP in one launch.  A block owns 32 rows of the batch from the observation to the
// output: layer 0 loads its 32 input rows into an LDS panel; every later layer takes its input from an LDS panel
// that the previous layer's epilogue wrote (bf16 head + tail of the activated fp32 value -- exactly what the per-layer
// kernel would have converted after reading it back from HBM), so the activations make one trip to HBM (written once, for
// the backward pass; not at all for inference) instead of two, and a network pass pays one launch floor instead of four.
// Two panels, A and B, alternate (layer l reads A when l is odd, B when even).  Same k steps,
// MFMAs and epilogue expressions as the per-layer kernels: every stored value is bit-identical to theirs.
constexpr int CHAIN_MAX = SHF_MLP_MAX_CHAIN;
struct ChainArgs {
  const float* x;
  int M, nl;
  int dims[CHAIN_MAX + 1];
  const uint4* bhi[CHAIN_MAX];
  const uint4* blo[CHAIN_MAX];
  const float* bias[CHAIN_MAX];
  int act[CHAIN_MAX];
  float* y[CHAIN_MAX];       // where layer l's output goes, or null (inference: only the last)
  int pa_words, pb_words;    // sizes (uint16) of panel A and of panel B / stream buffers; the epilogue patches follow
};
// One layer of the chain for a wave that owns CT column tiles (compile-time, so that the accumulators and the ring of B
// fragments are register arrays): ring depth 8 / CT k steps -- the block runs two waves per SIMD (its panels
// fill the LDS), so the L2 latency of the weight fragments has to be covered by the ring alone.
constexpr int CHAIN_THREADS = 512, CHAIN_WAVES = CHAIN_THREADS / 64;
template <bool SPLIT, int CT>
MLP_DEV void chain_layer(const ChainArgs& P, int l, uint16_t* PA, uint16_t* PB, float* T, int r0, int t, int wave, int lane) {
  constexpr int BM = 32, TS = 36;
  constexpr int D = 8 / CT;                                 // 64 registers of B fragments in flight (bf16x3)
  const int red = P.dims[l], cols = P.dims[l + 1];
  const int redp = (red + 15) & ~15, nks = redp >> 4, nct = (cols + 31) >> 5;
  const int ct0 = wave * CT;
  const bool wave_on = ct0 < nct;
  const uint4* bhi = P.bhi[l];
  const uint4* blo = P.blo[l];
  f32x16 acc[CT];
#pragma unroll
  for (int j = 0; j < CT; j++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[j][r] = 0.0f;
  size_t boff[CT];
#pragma unroll
  for (int j = 0; j < CT; j++) boff[j] = (size_t)(ct0 + j < nct ? ct0 + j : nct - 1) * nks * 64 + lane;
  uint4 rh[D][CT], rl[D][SPLIT ? CT : 1];
  if (wave_on) {
#pragma unroll
    for (int d = 0; d < D; d++)
      if (d < nks) {
#pragma unroll
        for (int j = 0; j < CT; j++) { rh[d][j] = bhi[boff[j] + (size_t)d * 64]; if (SPLIT) rl[d][j] = blo[boff[j] + (size_t)d * 64]; }
      }
  }
  auto kstep = [&](int ks, int d, const uint16_t* Ah, const uint16_t* Al) {
    bf16x8 bh[CT], bl[CT];
#pragma unroll
    for (int j = 0; j < CT; j++) { bh[j] = __builtin_bit_cast(bf16x8, rh[d][j]); if (SPLIT) bl[j] = __builtin_bit_cast(bf16x8, rl[d][j]); }
    if (ks + D < nks) {
#pragma unroll
      for (int j = 0; j < CT; j++) { rh[d][j] = bhi[boff[j] + (size_t)(ks + D) * 64]; if (SPLIT) rl[d][j] = blo[boff[j] + (size_t)(ks + D) * 64]; }
    }
    const bf16x8 ah = *reinterpret_cast<const bf16x8*>(Ah);
    bf16x8 al;
    if (SPLIT) al = *reinterpret_cast<const bf16x8*>(Al);
    // per accumulator the order is tail x head, head x tail, head x head as in the per-layer kernels; across the wave's
    // accumulators the three rounds are interleaved so that consecutive MFMAs are independent
    if (SPLIT) {
#pragma unroll
      for (int j = 0; j < CT; j++) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[j], acc[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < CT; j++) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[j], acc[j], 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < CT; j++) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[j], acc[j], 0, 0, 0);
  };

  // this wave's bias values: requested before the reduction so that their round trip is off the epilogue's path
  float bvs[CT];
#pragma unroll
  for (int j = 0; j < CT; j++) {
    const int col = 32 * (ct0 + j) + (lane & 31);
    bvs[j] = (P.bias[l] && ct0 + j < nct && col < cols) ? P.bias[l][col] : 0.0f;
  }
  const int S = ((red + 31) & ~31) + PANEL_PAD;
  uint16_t* Xp = (l & 1) ? PA : PB;
  if (l == 0) {
    // the block's 32 input rows -- one contiguous range of memory -- into panel B as bf16 head + tail: every load of the
    // panel is in flight at once (<= 8 sixteen-byte loads per thread at 512 columns)
    const int E = BM * red;
    const float* src = P.x + (size_t)r0 * red;
    const int Ein = (P.M - r0 < BM ? P.M - r0 : BM) * red;
    const uint32_t magic = (uint32_t)((1ull << 32) / (uint64_t)red) + 1u;
    constexpr int NVX = (BM * 512 / 4 + CHAIN_THREADS - 1) / CHAIN_THREADS;
    f32x4 v[NVX];
#pragma unroll
    for (int u = 0; u < NVX; u++) {
      const int e = 4 * (t + CHAIN_THREADS * u);
      v[u] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      if (e + 3 < Ein) v[u] = *reinterpret_cast<const f32x4*>(src + e);
      else if (e < Ein) {
        for (int c = 0; c < 4; c++)
          if (e + c < Ein) v[u][c] = src[e + c];
      }
    }
    const bool vec_store = (red & 3) == 0;
#pragma unroll
    for (int u = 0; u < NVX; u++) {
      const int e = 4 * (t + CHAIN_THREADS * u);
      if (e >= E) continue;
      int row = (int)__umulhi((uint32_t)e, magic), k = e - row * red;
      if (vec_store) {
        bf16x4 h, lo;
#pragma unroll
        for (int c = 0; c < 4; c++) { h[c] = (__bf16)v[u][c]; if (SPLIT) lo[c] = (__bf16)(v[u][c] - (float)h[c]); }
        *reinterpret_cast<bf16x4*>(Xp + row * S + k) = h;
        if (SPLIT) *reinterpret_cast<bf16x4*>(Xp + BM * S + row * S + k) = lo;
      } else {
#pragma unroll
        for (int c = 0; c < 4; c++) {
          if (e + c < E) {
            const __bf16 h = (__bf16)v[u][c];
            Xp[row * S + k] = __builtin_bit_cast(uint16_t, h);
            if (SPLIT) { const __bf16 lo = (__bf16)(v[u][c] - (float)h); Xp[BM * S + row * S + k] = __builtin_bit_cast(uint16_t, lo); }
          }
          if (++k == red) { k = 0; row++; }
        }
      }
    }
    const int padw = redp - red;                              // zero columns up to the pack's k extent
    for (int i = t; i < BM * padw; i += CHAIN_THREADS) {
      const int row = i / padw, k = red + (i - row * padw);
      Xp[row * S + k] = 0;
      if (SPLIT) Xp[BM * S + row * S + k] = 0;
    }
    __syncthreads();
  }
  if (wave_on) {
    const int frag = (lane & 31) * S + 8 * (lane >> 5);
    for (int ks0 = 0; ks0 < nks; ks0 += D) {
#pragma unroll
      for (int d = 0; d < D; d++) {
        const int ks = ks0 + d;
        if (ks >= nks) break;
        kstep(ks, d, Xp + frag + 16 * ks, Xp + BM * S + frag + 16 * ks);
      }
    }
  }

  // epilogue: bias, activation; through the wave's LDS patch to 16-byte global stores (if this layer's output is kept)
  // and to the next layer's input panel as bf16 head + tail
  const bool has_next = l + 1 < P.nl;
  const int Sn = ((cols + 31) & ~31) + PANEL_PAD;
  uint16_t* Xn = ((l + 1) & 1) ? PA : PB;
  float* Y = P.y[l];
  const bool vec_out = Y && (cols & 3) == 0 && ((uintptr_t)Y & 15u) == 0;
  if (wave_on) {
#pragma unroll
    for (int j = 0; j < CT; j++) {
      if (ct0 + j >= nct) continue;
#pragma unroll
      for (int reg = 0; reg < 16; reg++) {
        float v = acc[j][reg] + bvs[j];
        if (P.act[l] == 1) v = v > 0.0f ? v : __expf(v) - 1.0f;
        T[((reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)) * TS + (lane & 31)] = v;
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const int tr = (lane >> 3) + 8 * g, row = r0 + tr, c4 = 32 * (ct0 + j) + 4 * (lane & 7);
        const f32x4 o = *reinterpret_cast<const f32x4*>(T + tr * TS + 4 * (lane & 7));
        if (Y && row < P.M) {
          float* dst = Y + (size_t)row * cols + c4;
          if (vec_out && c4 + 3 < cols) *reinterpret_cast<f32x4*>(dst) = o;
          else {
#pragma unroll
            for (int c = 0; c < 4; c++)
              if (c4 + c < cols) dst[c] = o[c];
          }
        }
        if (has_next) {
          bf16x4 h, lo;
#pragma unroll
          for (int c = 0; c < 4; c++) { h[c] = (__bf16)o[c]; if (SPLIT) lo[c] = (__bf16)(o[c] - (float)h[c]); }
          *reinterpret_cast<bf16x4*>(Xn + tr * Sn + c4) = h;
          if (SPLIT) *reinterpret_cast<bf16x4*>(Xn + BM * Sn + tr * Sn + c4) = lo;
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
  __syncthreads();      // the next layer's input panel is complete; this layer's input may be overwritten from now on
}

template <bool SPLIT>
__global__ __launch_bounds__(CHAIN_THREADS) void k_mlp_chain(ChainArgs P) {
  extern __shared__ __attribute__((aligned(16))) uint16_t chain_lds[];
  uint16_t* PA = chain_lds;
  uint16_t* PB = chain_lds + P.pa_words;
  const int t = (int)threadIdx.x, wave = t >> 6, lane = t & 63;
  float* T = reinterpret_cast<float*>(chain_lds + P.pa_words + P.pb_words) + wave * 32 * 36;
  const int r0 = (int)blockIdx.x * 32;
  MLP_CLOCK(0);
  for (int l = 0; l < P.nl; l++) {
    const int nct = (P.dims[l + 1] + 31) >> 5;               // column tiles per wave: 2 from 9 tiles, else 1
    if (nct > 8) chain_layer<SPLIT, 2>(P, l, PA, PB, T, r0, t, wave, lane);
    else chain_layer<SPLIT, 1>(P, l, PA, PB, T, r0, t, wave, lane);
#ifdef SHF_MLP_PROBE_CLOCK
    if (l == 0) MLP_CLOCK(1);          // (probe: end of layers 0, 1 and of the last one)
    if (l == 1) MLP_CLOCK(2);
    if (l == P.nl - 1) MLP_CLOCK(3);
#endif
  }
}

// W[N, K] (fp32) -> four fragment-ordered bf16 arrays: plain (column = n, reduction = k) head / tail, then transposed
// (column = k, reduction = n) head / tail.  One thread per 16-byte fragment entry.
struct PackDims { int nct, nks; size_t entries; };
PackDims pack_dims(int cols, int red) { PackDims d; d.nct = (cols + 31) / 32; d.nks = (red + 15) / 16; d.entries = (size_t)d.nct * d.nks * 64; return d; }

__global__ __launch_bounds__(256) void k_mlp_pack(const float* __restrict__ w, uint4* __restrict__ out, int K, int N, int plain_entries, int nks_p,
                                                   int t_entries, int nks_t) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (i >= plain_entries + t_entries) return;
  const bool tr = i >= plain_entries;
  const int e = tr ? i - plain_entries : i, nks = tr ? nks_t : nks_p;
  const int lane = e & 63, ks = (e >> 6) % nks, ct = (e >> 6) / nks;
  const int col = 32 * ct + (lane & 31), k0 = 16 * ks + 8 * (lane >> 5);
  const int cols = tr ? K : N, red = tr ? N : K;
  uint16_t h[8], l[8];
#pragma unroll
  for (int c = 0; c < 8; c++) {
    float x = 0.0f;
    if (col < cols && k0 + c < red) x = tr ? w[(size_t)(k0 + c) * K + col] : w[(size_t)col * K + k0 + c];
    const __bf16 hb = (__bf16)x;
    const __bf16 lb = (__bf16)(x - (float)hb);
    h[c] = __builtin_bit_cast(uint16_t, hb);
    l[c] = __builtin_bit_cast(uint16_t, lb);
  }
  uint4 H, L;
  H.x = h[0] | ((uint32_t)h[1] << 16); H.y = h[2] | ((uint32_t)h[3] << 16); H.z = h[4] | ((uint32_t)h[5] << 16); H.w = h[6] | ((uint32_t)h[7] << 16);
  L.x = l[0] | ((uint32_t)l[1] << 16); L.y = l[2] | ((uint32_t)l[3] << 16); L.z = l[4] | ((uint32_t)l[5] << 16); L.w = l[6] | ((uint32_t)l[7] << 16);
  // layout: [plain head | plain tail | transposed head | transposed tail]
  uint4* head = tr ? out + 2 * (size_t)plain_entries : out;
  const size_t n = tr ? (size_t)t_entries : (size_t)plain_entries;
  head[e] = H;
  head[n + e] = L;
}

int panel_env(const char* name) { const char* v = getenv(name); return v ? atoi(v) : 0; }

template <int BM, int CT>
int launch_panel_bc(hipStream_t st, const PanelArgs& P, bool split, size_t lds) {
  const dim3 grid((P.M + BM - 1) / BM);
#define SHF_PANEL_GO(MASKV, SPLITV)                                                                              \
  do {                                                                                                           \
    auto fn = k_mlp_panel<BM, CT, MASKV, SPLITV>;                                                                \
    static std::atomic<uint64_t> attr{0};        /* one bit per device: the opt-in is per device, not per process */ \
    const uint64_t dbit = mlp_device_bit();                                                                      \
    if (!(attr.load(std::memory_order_relaxed) & dbit)) {                                                        \
      if (hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) \
        return mlp_fail("k_mlp_panel: cannot raise the dynamic LDS limit");                                      \
      attr.fetch_or(dbit, std::memory_order_relaxed);                                                            \
    }                                                                                                            \
    hipLaunchKernelGGL(fn, grid, dim3(256), lds, st, P);                                                         \
  } while (0)
  if (P.mask_y) { if (split) SHF_PANEL_GO(true, true); else SHF_PANEL_GO(true, false); }
  else { if (split) SHF_PANEL_GO(false, true); else SHF_PANEL_GO(false, false); }
#undef SHF_PANEL_GO
  return 0;
}

// CT = ceil(column tiles / 4) (1, 2 or 4: every wave holds all its column tiles' accumulators); BM: 64 rows, 32 when the
// accumulators of 64 would not leave room for two blocks per CU (CT = 4) or the grid would not cover the CUs.
int launch_panel(hipStream_t st, PanelArgs& P) {
  const bool split = g_mlp_precision != SHF_MLP_BF16;
  static const int force_bm = panel_env("SHF_MLP_PANEL_BM");
  const int nct = (P.cols + 31) / 32;
  if (nct > 16) return mlp_fail("k_mlp_panel: more than 512 output columns");
  const int ct = nct > 8 ? 4 : nct > 4 ? 2 : 1;
  int bm = (ct == 4 || (P.M + 63) / 64 < 256) ? 32 : 64;
  if ((force_bm == 32 || force_bm == 64) && !(force_bm == 64 && ct == 4)) bm = force_bm;
  size_t lds = (size_t)2 * (split ? 2 : 1) * bm * KCS * 2;
  if (lds < 4 * 32 * 36 * sizeof(float)) lds = 4 * 32 * 36 * sizeof(float);   // the epilogue's four 32 x 36 patches
  if (bm == 64) return ct == 2 ? launch_panel_bc<64, 2>(st, P, split, lds) : launch_panel_bc<64, 1>(st, P, split, lds);
  return ct == 4 ? launch_panel_bc<32, 4>(st, P, split, lds) : ct == 2 ? launch_panel_bc<32, 2>(st, P, split, lds) : launch_panel_bc<32, 1>(st, P, split, lds);
}
uint32_t div_magic(int d) { return (uint32_t)((1ull << 32) / (uint64_t)d) + 1u; }

}  // namespace


// Several contiguous buffers copied by one launch (the rollout's per-step writes into the RolloutStorage slots: ~8 small
// device-to-device copies otherwise).  blockIdx.y picks the buffer, 4-byte words.
constexpr int COPY_MAX = 16;
struct CopyMany {
  const unsigned* src[COPY_MAX];
  unsigned* dst[COPY_MAX];
  long long words[COPY_MAX];
};
__global__ __launch_bounds__(256) void k_copy_many(CopyMany C) {
  const int b = blockIdx.y;
  const unsigned* __restrict__ s = C.src[b];
  unsigned* __restrict__ d = C.dst[b];
  const long long n = C.words[b];
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) d[i] = s[i];
}

// The runner's per-step episode bookkeeping (rsl_rl OnPolicyRunner.learn: cur_reward_sum / cur_episode_length, and the
// sums over the episodes that ended in this step) as one single-block launch instead of ~16 element-wise / reduction
// launches.  The two running buffers get the same float32 values as the torch expressions; the three sums (logging
// only) are accumulated in double in a fixed order.
__global__ __launch_bounds__(1024) void k_episode_bookkeeping(const float* __restrict__ rew, const unsigned char* __restrict__ done,
                                                              int itemsize, long long N, float* __restrict__ crs,
                                                              float* __restrict__ cel, double* __restrict__ fin) {
  __shared__ double red[3][1024 / 64];
  double s0 = 0.0, s1 = 0.0, s2 = 0.0;
  for (long long i = threadIdx.x; i < N; i += 1024) {
    bool dn = false;
    for (int k = 0; k < itemsize; k++) dn |= done[i * itemsize + k] != 0;
    const float d = dn ? 1.0f : 0.0f;
    const float r = crs[i] + rew[i], l = cel[i] + 1.0f;
    s0 += (double)(r * d); s1 += (double)(l * d); s2 += (double)d;
    crs[i] = r * (1.0f - d);
    cel[i] = l * (1.0f - d);
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) { s0 += __shfl_down(s0, off, 64); s1 += __shfl_down(s1, off, 64); s2 += __shfl_down(s2, off, 64); }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) { red[0][wave] = s0; red[1][wave] = s1; red[2][wave] = s2; }
  __syncthreads();
  if (threadIdx.x < 3) {
    double t = 0.0;
    for (int w = 0; w < 1024 / 64; w++) t += red[threadIdx.x][w];
    fin[threadIdx.x] += t;
  }
}

// Rows idx[i] of several row-major tensors gathered by one launch (RolloutStorage.mini_batch: nine index kernels
// otherwise).  blockIdx.y picks the tensor; 4-byte words.
struct GatherMany {
  const unsigned* src[COPY_MAX];
  unsigned* dst[COPY_MAX];
  int row_words[COPY_MAX];
};
__global__ __launch_bounds__(256) void k_gather_rows(GatherMany G, const long long* __restrict__ idx, long long rows) {
  const int b = blockIdx.y, w = G.row_words[b];
  const unsigned* __restrict__ s = G.src[b];
  unsigned* __restrict__ d = G.dst[b];
  const long long total = rows * w;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const long long r = e / w;
    const int c = (int)(e - r * w);
    d[e] = s[idx[r] * w + c];
  }
}

// PPO's adaptive learning-rate rule on the device scalar (rl/ppo.py: adapt_learning_rate), the torch expressions' float32
// arithmetic: a tensor divided by a Python scalar is multiplied by the scalar's float32 reciprocal.
__global__ void k_adapt_lr(const float* __restrict__ kl, float* __restrict__ lr, float kl_high, float kl_low, float inv_down, float up,
                           float lr_min, float lr_max) {
  const float k = kl[0], l = lr[0];
  const float down = fmaxf(l * inv_down, lr_min), upv = fminf(l * up, lr_max);
  lr[0] = k > kl_high ? down : ((k > 0.0f && k < kl_low) ? upv : l);
}

// GAE(lambda) over a (T, N) rollout, one thread per env walking its T transitions backwards -- the same float32
// operations, in the same order, as RolloutStorage.compute_returns' torch loop (rl/storage.py), so the results are
// identical to the bit; it replaces that loop's ~9 launches per transition.
__global__ __launch_bounds__(256) void k_gae(const float* __restrict__ rew, const float* __restrict__ val,
                                             const unsigned char* __restrict__ done, const float* __restrict__ last, int T,
                                             long long N, float gamma, float lam, float* __restrict__ ret) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  float adv = 0.0f;
  for (int k = T - 1; k >= 0; k--) {
    const float nxt = k == T - 1 ? last[i] : val[(long long)(k + 1) * N + i];
    const float live = 1.0f - (float)done[(long long)k * N + i];
    const float v = val[(long long)k * N + i], lg = live * gamma;
    const float delta = (rew[(long long)k * N + i] + lg * nxt) - v;
    adv = delta + (lg * lam) * adv;
    ret[(long long)k * N + i] = adv + v;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// PPO mini-batch loss, forward and gradient in one pass (shf_ppo_loss).  One thread per sample walks its A action
// columns; block sums go through wave shuffles and LDS in a fixed order, the blocks' partial sums are added in block
// order by k_ppo_loss_finish: no atomics, the same bits on every run.  Gradients follow torch's conventions where the
// loss is not differentiable: maximum() splits the gradient evenly on a tie, clamp() passes it on the closed interval.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int PPO_MAX_ACTIONS = 32, PPO_BLOCK = 256, PPO_NRED = 3;

struct PpoArgs {
  const float *mu, *std, *value, *actions, *target_values, *advantages, *returns, *old_logp, *old_mu, *old_sigma;
  long long B;
  int A;
  float clip, value_coef, entropy_coef;
  int clipped_value;
  float *out, *dmu, *dstd, *dvalue, *partial;
};

MLP_DEV float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

__global__ __launch_bounds__(PPO_BLOCK) void k_ppo_loss(PpoArgs P) {
  __shared__ float red[PPO_BLOCK / 64][PPO_NRED + PPO_MAX_ACTIONS];
  const long long i = (long long)blockIdx.x * PPO_BLOCK + threadIdx.x;
  const bool live = i < P.B;
  const int A = P.A;
  const float invB = 1.0f / (float)P.B;
  float surr = 0.0f, vloss = 0.0f, kl = 0.0f, g_logp = 0.0f;
  if (live) {
    float logp = 0.0f;
    for (int j = 0; j < A; j++) {
      const float s = P.std[j], var = s * s, m = P.mu[i * A + j], d = P.actions[i * A + j] - m;
      logp += -(d * d) / (2.0f * var) - logf(s) - 0.91893853320467274178f;
      const float os = P.old_sigma[i * A + j], dm = P.old_mu[i * A + j] - m;
      kl += logf(s / os + 1.e-5f) + (os * os + dm * dm) / (2.0f * var) - 0.5f;
    }
    const float ratio = expf(logp - P.old_logp[i]), adv = P.advantages[i];
    const float lo = 1.0f - P.clip, hi = 1.0f + P.clip;
    const float rc = fminf(fmaxf(ratio, lo), hi);
    const float s1 = -adv * ratio, s2 = -adv * rc;
    const bool inside = ratio >= lo && ratio <= hi;
    surr = fmaxf(s1, s2);
    const float g_ratio = s1 > s2 ? -adv : (s1 < s2 ? 0.0f : -adv * 0.5f + (inside ? -adv * 0.5f : 0.0f));
    g_logp = g_ratio * ratio * invB;
    const float v = P.value[i], R = P.returns[i];
    float gv;
    if (P.clipped_value) {
      const float tv = P.target_values[i], dv = v - tv;
      const float vc = tv + fminf(fmaxf(dv, -P.clip), P.clip);
      const bool vin = dv >= -P.clip && dv <= P.clip;
      const float e1 = v - R, e2 = vc - R, l1 = e1 * e1, l2 = e2 * e2;
      vloss = fmaxf(l1, l2);
      const float g2 = vin ? 2.0f * e2 : 0.0f;
      gv = l1 > l2 ? 2.0f * e1 : (l1 < l2 ? g2 : 0.5f * (2.0f * e1) + 0.5f * g2);
    } else {
      const float e1 = R - v;
      vloss = e1 * e1;
      gv = -2.0f * e1;
    }
    P.dvalue[i] = P.value_coef * gv * invB;
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float r0 = wave_sum(surr), r1 = wave_sum(vloss), r2 = wave_sum(kl);
  if (lane == 0) { red[wave][0] = r0; red[wave][1] = r1; red[wave][2] = r2; }
  for (int j = 0; j < A; j++) {
    float ds = 0.0f;
    if (live) {
      const float s = P.std[j], var = s * s, d = P.actions[i * A + j] - P.mu[i * A + j];
      P.dmu[i * A + j] = g_logp * (d / var);               // d logp / d mu = (a - mu) / var
      ds = g_logp * ((d * d) / (var * s) - 1.0f / s);
    }
    ds = wave_sum(ds);
    if (lane == 0) red[wave][PPO_NRED + j] = ds;
  }
  __syncthreads();
  if (threadIdx.x < PPO_NRED + A) {
    float t = red[0][threadIdx.x];
#pragma unroll
    for (int w = 1; w < PPO_BLOCK / 64; w++) t += red[w][threadIdx.x];
    P.partial[(size_t)blockIdx.x * (PPO_NRED + PPO_MAX_ACTIONS) + threadIdx.x] = t;
  }
}

__global__ __launch_bounds__(64) void k_ppo_loss_finish(PpoArgs P, int nblocks) {
  __shared__ float tot[PPO_NRED];
  const int q = threadIdx.x, A = P.A;
  float t = 0.0f;
  if (q < PPO_NRED + A)
    for (int b = 0; b < nblocks; b++) t += P.partial[(size_t)b * (PPO_NRED + PPO_MAX_ACTIONS) + q];
  if (q < PPO_NRED) tot[q] = t;
  if (q >= PPO_NRED && q < PPO_NRED + A) {
    const float s = P.std[q - PPO_NRED];
    P.dstd[q - PPO_NRED] = t - P.entropy_coef * (1.0f / s);     // entropy of N(mu, s): 0.5 + 0.5 log(2 pi) + log s per column
  }
  __syncthreads();
  if (q == 0) {
    const float invB = 1.0f / (float)P.B;
    float ent = 0.0f;
    for (int j = 0; j < A; j++) ent += 1.41893853320467274178f + logf(P.std[j]);
    const float sm = tot[0] * invB, vm = tot[1] * invB, km = tot[2] * invB;
    P.out[0] = sm; P.out[1] = vm; P.out[2] = ent; P.out[3] = km;
    P.out[4] = sm + P.value_coef * vm - P.entropy_coef * ent;
  }
}

#ifdef SHF_MLP_PROBE_CLOCK
extern "C" int shf_mlp_probe_read(long long* out, int n) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mlp_probe), sizeof(long long) * n) == hipSuccess ? 0 : 1;
}
#endif
extern "C" const char* shf_mlp_last_error(void) { return g_mlp_err.c_str(); }
extern "C" int shf_mlp_set_precision(int32_t mode) {
  if (mode != SHF_MLP_BF16 && mode != SHF_MLP_BF16X3 && mode != SHF_MLP_BF16X3_W1)
    return mlp_fail("shf_mlp_set_precision: mode must be SHF_MLP_BF16, SHF_MLP_BF16X3 or SHF_MLP_BF16X3_W1");
  g_mlp_precision = mode;
  return 0;
}
extern "C" int shf_mlp_get_precision(void) { return g_mlp_precision; }

extern "C" int shf_mlp_linear_forward(const float* x, const float* w, const float* b, float* y, int32_t M, int32_t K, int32_t N,
                                      int32_t act, void* stream) {
  if (!x || !w || !y) return mlp_fail("shf_mlp_linear_forward: null tensor");
  if (M <= 0 || K <= 0 || N <= 0 || act < 0 || act > 1) return mlp_fail("shf_mlp_linear_forward: bad shape / activation");
  if (too_big(M, K, N)) return mlp_fail("shf_mlp_linear_forward: a tensor has 2^32 elements or more");
  Operand A{x, nullptr, K, M}, B{w, nullptr, K, N};
  Epilogue E{y, N, b, act, nullptr};
  launch_gemm<true, true, false>((hipStream_t)stream, A, B, K, K, 1, E, M, N);
  return hipGetLastError() == hipSuccess ? 0 : mlp_fail("shf_mlp_linear_forward: launch failed");
}

extern "C" int shf_mlp_linear_backward_input(const float* dy, const float* y_or_null, const float* w, float* dx, int32_t M,
                                             int32_t K, int32_t N, void* stream) {
  if (!dy || !w || !dx) return mlp_fail("shf_mlp_linear_backward_input: null tensor");
  if (M <= 0 || K <= 0 || N <= 0 || too_big(M, K, N)) return mlp_fail("shf_mlp_linear_backward_input: bad shape");
  // dX[M,K] = G[M,N] W[N,K]: A = G (reduction index n contiguous), B[col = k][red = n] = W[n][k] (transposed access)
  Operand A{dy, y_or_null, N, M}, B{w, nullptr, K, K};
  Epilogue E{dx, K, nullptr, 0, nullptr};
  launch_gemm<true, false, false>((hipStream_t)stream, A, B, N, N, 1, E, M, K);
  return hipGetLastError() == hipSuccess ? 0 : mlp_fail("shf_mlp_linear_backward_input: launch failed");
}

extern "C" int shf_mlp_pack_bytes(int32_t K, int32_t N, int64_t* bytes) {
  if (!bytes || K <= 0 || N <= 0) return mlp_fail("shf_mlp_pack_bytes: bad argument");
  *bytes = (int64_t)(2 * (pack_dims(N, K).entries + pack_dims(K, N).entries) * sizeof(uint4));
  return 0;
}
extern "C" int shf_mlp_pack_weights(const float* w, void* pack, int32_t K, int32_t N, void* stream) {
  if (!w || !pack || K <= 0 || N <= 0) return mlp_fail("shf_mlp_pack_weights: bad argument");
  if (((uintptr_t)pack & 15u) != 0) return mlp_fail("shf_mlp_pack_weights: pack must be 16-byte aligned");
  const PackDims p = pack_dims(N, K), q = pack_dims(K, N);
  const int total = (int)(p.entries + q.entries);
  hipLaunchKernelGGL(k_mlp_pack, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, (uint4*)pack, K, N, (int)p.entries, p.nks,
                     (int)q.entries, q.nks);
  return hipGetLastError() == hipSuccess ? 0 : mlp_fail("shf_mlp_pack_weights: launch failed");
}
extern "C" int shf_mlp_panel_forward(const float* x, const void* pack, const float* b, float* y, int32_t M, int32_t K, int32_t N,
                                     int32_t act, void* stream) {
  if (!x || !pack || !y) return mlp_fail("shf_mlp_panel_forward: null tensor");
  if (M <= 0 || K <= 0 || N <= 0 || act < 0 || act > 1) return mlp_fail("shf_mlp_panel_forward: bad shape / activation");
  if (too_big(M, K, N)) return mlp_fail("shf_mlp_panel_forward: a tensor has 2^32 elements or more");
  if (((uintptr_t)x & 15u) != 0) return mlp_fail("shf_mlp_panel_forward: x must be 16-byte aligned");
  const PackDims p = pack_dims(N, K);
  PanelArgs P{x, nullptr, M, K, p.nks * 16, div_magic(K), (const uint4*)pack, (const uint4*)pack + p.entries, y, N, N, b, act};
  if (launch_panel((hipStream_t)stream, P)) return 1;
  return hipGetLastError() == hipSuccess ? 0 : mlp_fail("shf_mlp_panel_forward: launch failed");
}
extern "C" int shf_mlp_panel_backward_input(const float* dy, const float* y_or_null, const void* pack, float* dx, int32_t M,
                                            int32_t K, int32_t N, void* stream) {
  if (!dy || !pack || !dx) return mlp_fail("shf_mlp_panel_backward_input: null tensor");
  if (M <= 0 || K <= 0 || N <= 0 || too_big(M, K, N)) return mlp_fail("shf_mlp_panel_backward_input: bad shape");
  if (((uintptr_t)dy & 15u) != 0 || ((uintptr_t)y_or_null & 15u) != 0) return mlp_fail("shf_mlp_panel_backward_input: dy / y must be 16-byte aligned");
  const PackDims p = pack_dims(N, K), q = pack_dims(K, N);
  const uint4* th = (const uint4*)pack + 2 * p.entries;
  PanelArgs P{dy, y_or_null, M, N, q.nks * 16, div_magic(N), th, th + q.entries, dx, K, K, nullptr, 0};
  if (launch_panel((hipStream_t)stream, P)) return 1;
  return hipGetLastError() == hipSuccess ? 0 : mlp_fail("shf_mlp_panel_backward_input: launch failed");
}

// LDS of one k_mlp_chain block for these widths at the current precision: panel A (inputs of the odd layers), panel B
// (inputs of the even layers, layer 0's rows of x among them), the waves' epilogue patches.  0 = the widths are not chainable at all.
static size_t chain_lds(const ShfMlpChain* c, size_t* pa_out, size_t* pb_out) {
  if (c->nlayers < 1 || c->nlayers > SHF_MLP_MAX_CHAIN) return 0;
  const int planes = g_mlp_precision != SHF_MLP_BF16 ? 2 : 1;
  size_t pa = 0, pb = 0;
  for (int l = 0; l <= c->nlayers; l++) {
    if (c->dims[l] <= 0 || c->dims[l] > 512) return 0;
    if (l < c->nlayers) {                                 // the input panel of layer l (layer 0: the block's rows of x)
      const size_t words = (size_t)planes * 32 * (((c->dims[l] + 31) & ~31) + PANEL_PAD);
      if (l & 1) pa = words > pa ? words : pa; else pb = words > pb ? words : pb;
    }
  }
  if (pa_out) *pa_out = pa;
  if (pb_out) *pb_out = pb;
  return (pa + pb) * 2 + CHAIN_WAVES * 32 * 36 * sizeof(float);
}
extern "C" int shf_mlp_chain_fits(const ShfMlpChain* c) {
  if (!c) return 0;
  const size_t lds = chain_lds(c, nullptr, nullptr);
  return lds > 0 && lds <= 160 * 1024 ? 1 : 0;
}
extern "C" int shf_mlp_chain_forward(const float* x, int32_t M, const ShfMlpChain* c, void* stream) {
  if (!x || !c || M <= 0) return mlp_fail("shf_mlp_chain_forward: bad argument");
  if (c->nlayers < 1 || c->nlayers > SHF_MLP_MAX_CHAIN) return mlp_fail("shf_mlp_chain_forward: 1 .. SHF_MLP_MAX_CHAIN layers");
  if (((uintptr_t)x & 15u) != 0) return mlp_fail("shf_mlp_chain_forward: x must be 16-byte aligned");
  const bool split = g_mlp_precision != SHF_MLP_BF16;
  ChainArgs P{};
  P.x = x; P.M = M; P.nl = c->nlayers;
  size_t pa = 0, pb = 0;
  const size_t lds = chain_lds(c, &pa, &pb);
  if (lds == 0) return mlp_fail("shf_mlp_chain_forward: layer widths must be 1 .. 512");
  if (lds > 160 * 1024) return mlp_fail("shf_mlp_chain_forward: the panels exceed the LDS (shf_mlp_chain_fits)");
  for (int l = 0; l <= c->nlayers; l++) P.dims[l] = c->dims[l];
  if (too_big(M, 512, 512)) return mlp_fail("shf_mlp_chain_forward: batch too large");
  for (int l = 0; l < c->nlayers; l++) {
    if (!c->pack[l]) return mlp_fail("shf_mlp_chain_forward: null pack");
    if (c->act[l] < 0 || c->act[l] > 1) return mlp_fail("shf_mlp_chain_forward: bad activation");
    const PackDims p = pack_dims(c->dims[l + 1], c->dims[l]);
    P.bhi[l] = (const uint4*)c->pack[l];
    P.blo[l] = (const uint4*)c->pack[l] + p.entries;
    P.bias[l] = c->bias[l]; P.act[l] = c->act[l]; P.y[l] = c->y[l];
  }
  if (!c->y[c->nlayers - 1]) return mlp_fail("shf_mlp_chain_forward: the last layer needs an output buffer");
  P.pa_words = (int)pa; P.pb_words = (int)pb;
  const dim3 grid((M + 31) / 32);
#define SHF_CHAIN_GO(SPLITV)                                                                                     \
  do {                                                                                                           \
    auto fn = k_mlp_chain<SPLITV>;                                                                               \
    static std::atomic<uint64_t> attr{0};        /* one bit per device */                                        \
    const uint64_t dbit = mlp_device_bit();                                                                      \
    if (!(attr.load(std::memory_order_relaxed) & dbit)) {                                                        \
      if (hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) \
        return mlp_fail("k_mlp_chain: cannot raise the dynamic LDS limit");                                      \
      attr.fetch_or(dbit, std::memory_order_relaxed);                                                            \
    }                                                                                                            \
    hipLaunchKernelGGL(fn, grid, dim3(CHAIN_THREADS), lds, (hipStream_t)stream, P);                              \
  } while (0)
  if (split) SHF_CHAIN_GO(true); else SHF_CHAIN_GO(false);
#undef SHF_CHAIN_GO
  return hipGetLastError() == hipSuccess ? 0 : mlp_fail("shf_mlp_chain_forward: launch failed");
}

extern "C" int shf_mlp_backward_weight_workspace(int32_t M, int32_t K, int32_t N, int64_t* floats) {
  if (!floats) return mlp_fail("shf_mlp_backward_weight_workspace: null");
  const int per = wgrad_rows(M, K, N), slices = (M + per - 1) / per;
  *floats = (int64_t)slices * ((int64_t)N * K + N);
  return 0;
}

extern "C" int shf_mlp_linear_backward_weight(const float* dy, const float* y_or_null, const float* x, float* dw, float* db,
                                              float* workspace, int32_t M, int32_t K, int32_t N, void* stream) {
  if (!dy || !x || !dw || !workspace) return mlp_fail("shf_mlp_linear_backward_weight: null tensor");
  if (M <= 0 || K <= 0 || N <= 0 || too_big(M, K, N)) return mlp_fail("shf_mlp_linear_backward_weight: bad shape");
  // dW[N,K] = G^T X: rows = n, cols = k, reduction over m; both operands are stored with the reduction index as the row
  const int per = wgrad_rows(M, K, N), slices = (M + per - 1) / per;
  float* part_w = workspace;
  float* part_b = workspace + (size_t)slices * N * K;
  Operand A{dy, y_or_null, N, N}, B{x, nullptr, K, K};
  Epilogue E{part_w, K, nullptr, 0, part_b};
  launch_gemm<false, false, true>((hipStream_t)stream, A, B, M, per, slices, E, N, K);
  const int nw = N * K, nb = db ? N : 0;
  hipLaunchKernelGGL(k_mlp_reduce_slices, dim3((nw + nb + 255) / 256), dim3(256), 0, (hipStream_t)stream, part_w, dw, nw, part_b, db, nb,
                     slices);
  return hipGetLastError() == hipSuccess ? 0 : mlp_fail("shf_mlp_linear_backward_weight: launch failed");
}

extern "C" int shf_ppo_loss_workspace(int64_t B, int32_t A, int64_t* floats) {
  if (!floats || B <= 0 || A <= 0 || A > PPO_MAX_ACTIONS) return mlp_fail("shf_ppo_loss_workspace: bad argument (1 <= A <= 32)");
  *floats = ((B + PPO_BLOCK - 1) / PPO_BLOCK) * (int64_t)(PPO_NRED + PPO_MAX_ACTIONS);
  return 0;
}

extern "C" int shf_ppo_loss(const float* mu, const float* std, const float* value, const float* actions,
                            const float* target_values, const float* advantages, const float* returns, const float* old_logp,
                            const float* old_mu, const float* old_sigma, int64_t B, int32_t A, float clip, float value_coef,
                            float entropy_coef, int32_t clipped_value, float* out5, float* dmu, float* dstd, float* dvalue,
                            float* workspace, void* stream) {
  if (!mu || !std || !value || !actions || !advantages || !returns || !old_logp || !old_mu || !old_sigma || !out5 || !dmu ||
      !dstd || !dvalue || !workspace || (clipped_value && !target_values))
    return mlp_fail("shf_ppo_loss: null tensor");
  if (B <= 0 || A <= 0 || A > PPO_MAX_ACTIONS || B * (int64_t)A >= (1ll << 31)) return mlp_fail("shf_ppo_loss: bad shape (1 <= A <= 32)");
  const int nblocks = (int)((B + PPO_BLOCK - 1) / PPO_BLOCK);
  PpoArgs P{mu, std, value, actions, target_values, advantages, returns, old_logp, old_mu, old_sigma, (long long)B, A, clip,
            value_coef, entropy_coef, clipped_value, out5, dmu, dstd, dvalue, workspace};
  hipLaunchKernelGGL(k_ppo_loss, dim3(nblocks), dim3(PPO_BLOCK), 0, (hipStream_t)stream, P);
  hipLaunchKernelGGL(k_ppo_loss_finish, dim3(1), dim3(64), 0, (hipStream_t)stream, P, nblocks);
  return hipGetLastError() == hipSuccess ? 0 : mlp_fail("shf_ppo_loss: launch failed");
}

extern "C" int shf_gae(const float* rewards, const float* values, const unsigned char* dones, const float* last_values, int32_t T,
                       int64_t N, float gamma, float lam, float* returns, void* stream) {
  if (!rewards || !values || !dones || !last_values || !returns) return mlp_fail("shf_gae: null tensor");
  if (T <= 0 || N <= 0) return mlp_fail("shf_gae: bad shape");
  hipLaunchKernelGGL(k_gae, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rewards, values, dones, last_values,
                     (int)T, (long long)N, gamma, lam, returns);
  return hipGetLastError() == hipSuccess ? 0 : mlp_fail("shf_gae: launch failed");
}

extern "C" int shf_copy_many(const void* const* src, void* const* dst, const int64_t* bytes, int32_t n, void* stream) {
  if (!src || !dst || !bytes || n <= 0 || n > COPY_MAX) return mlp_fail("shf_copy_many: 1..16 buffers");
  CopyMany C{};
  long long most = 0;
  for (int i = 0; i < n; i++) {
    if (!src[i] || !dst[i] || bytes[i] < 0 || (bytes[i] & 3) || ((uintptr_t)src[i] & 3) || ((uintptr_t)dst[i] & 3))
      return mlp_fail("shf_copy_many: buffers must be non-null, 4-byte aligned, a multiple of 4 bytes");
    C.src[i] = (const unsigned*)src[i]; C.dst[i] = (unsigned*)dst[i]; C.words[i] = bytes[i] / 4;
    most = C.words[i] > most ? C.words[i] : most;
  }
  if (most == 0) return 0;
  long long bx = (most + 1023) / 1024;                  // about four words per thread
  if (bx > 4096) bx = 4096;
  hipLaunchKernelGGL(k_copy_many, dim3((unsigned)bx, (unsigned)n), dim3(256), 0, (hipStream_t)stream, C);
  return hipGetLastError() == hipSuccess ? 0 : mlp_fail("shf_copy_many: launch failed");
}

extern "C" int shf_episode_bookkeeping(const float* rewards, const void* dones, int32_t done_itemsize, int64_t N,
                                       float* cur_reward_sum, float* cur_episode_length, double* fin3, void* stream) {
  if (!rewards || !dones || !cur_reward_sum || !cur_episode_length || !fin3) return mlp_fail("shf_episode_bookkeeping: null tensor");
  if (N <= 0 || (done_itemsize != 1 && done_itemsize != 2 && done_itemsize != 4 && done_itemsize != 8))
    return mlp_fail("shf_episode_bookkeeping: bad shape / dones item size");
  hipLaunchKernelGGL(k_episode_bookkeeping, dim3(1), dim3(1024), 0, (hipStream_t)stream, rewards, (const unsigned char*)dones,
                     (int)done_itemsize, (long long)N, cur_reward_sum, cur_episode_length, fin3);
  return hipGetLastError() == hipSuccess ? 0 : mlp_fail("shf_episode_bookkeeping: launch failed");
}

extern "C" int shf_gather_rows(const void* const* src, void* const* dst, const int32_t* row_bytes, int32_t n, const int64_t* idx_dev,
                               int64_t rows, void* stream) {
  if (!src || !dst || !row_bytes || !idx_dev || n <= 0 || n > COPY_MAX) return mlp_fail("shf_gather_rows: 1..16 tensors");
  if (rows <= 0) return 0;
  GatherMany G{};
  long long most = 0;
  for (int i = 0; i < n; i++) {
    if (!src[i] || !dst[i] || row_bytes[i] <= 0 || (row_bytes[i] & 3) || ((uintptr_t)src[i] & 3) || ((uintptr_t)dst[i] & 3))
      return mlp_fail("shf_gather_rows: tensors must be non-null, 4-byte aligned, rows a multiple of 4 bytes");
    G.src[i] = (const unsigned*)src[i]; G.dst[i] = (unsigned*)dst[i]; G.row_words[i] = row_bytes[i] / 4;
    const long long t = rows * G.row_words[i];
    most = t > most ? t : most;
  }
  long long bx = (most + 1023) / 1024;
  if (bx > 8192) bx = 8192;
  hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)bx, (unsigned)n), dim3(256), 0, (hipStream_t)stream, G, (const long long*)idx_dev,
                     (long long)rows);
  return hipGetLastError() == hipSuccess ? 0 : mlp_fail("shf_gather_rows: launch failed");
}

extern "C" int shf_adapt_lr(const float* kl_dev, float* lr_dev, float kl_high, float kl_low, float inv_down, float up, float lr_min,
                            float lr_max, void* stream) {
  if (!kl_dev || !lr_dev) return mlp_fail("shf_adapt_lr: null tensor");
  hipLaunchKernelGGL(k_adapt_lr, dim3(1), dim3(1), 0, (hipStream_t)stream, kl_dev, lr_dev, kl_high, kl_low, inv_down, up, lr_min, lr_max);
  return hipGetLastError() == hipSuccess ? 0 : mlp_fail("shf_adapt_lr: launch failed");
}
