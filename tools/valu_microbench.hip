// tools/valu_microbench.hip -- what one wave / one SIMD of gfx950 issues per clock.
//
//   hipcc --offload-arch=gfx950 -O2 tools/valu_microbench.hip -o tools/valu_microbench   (here, no GPU)
//   tools/valu_microbench > gpurun_out/valu_microbench.md                                   (on the MI355X box)
//
// Every test is one kernel: 256 blocks (one per CU) of 256*W threads, so every SIMD hosts exactly W waves
// (W = 8: 512 blocks of 1024).  Each wave runs ITER x 64 copies of one instruction pattern written in inline
// assembly and brackets them with s_memtime; reported: shader clocks per wave-instruction as seen by one wave
// (latency view) and per SIMD (throughput view = the former / W), plus the wall-clock rate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <string>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define R64(x) R4(R16(x))

constexpr int ITER = 64;
typedef float v4f __attribute__((ext_vector_type(4)));

struct Out { unsigned long long ticks; float sink; };

template <int TEST>
__global__ void k_bench(Out* out, float seed, int lanes) {
  __shared__ float lds[4096];
  float a0 = seed + threadIdx.x, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
  float b = 1.000001f, c = 1e-7f;
  unsigned idx = (threadIdx.x * 4u) & 16383u;
  lds[threadIdx.x & 4095] = (float)((threadIdx.x * 4) & 16383);
  if (blockDim.x <= 1024) for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = __int_as_float((i * 4) & 16383);
  __syncthreads();
  // optional partial exec mask: lanes < `lanes` of each wave stay active for the timed region
  const bool on = (int)(threadIdx.x & 63u) < lanes;
  unsigned long long t0 = 0, t1 = 0;
  if (on) {
    t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITER; it++) {
      if constexpr (TEST == 0) {        // dependent v_fma_f32
        asm volatile(R64("v_fma_f32 %0, %0, %1, %2\n") : "+v"(a0) : "v"(b), "v"(c));
      } else if constexpr (TEST == 1) { // 2 independent chains
        asm volatile(R16("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n" "v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n")
                     : "+v"(a0), "+v"(a1) : "v"(b), "v"(c));
      } else if constexpr (TEST == 2) { // 4 independent chains
        asm volatile(R16("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n")
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));
      } else if constexpr (TEST == 3) { // 8 independent chains
        asm volatile(R4(R4("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n")
                        R4("v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"))
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
      } else if constexpr (TEST == 4) { // 8 independent v_mul_f32
        asm volatile(R4(R4("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n")
                        R4("v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"))
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
      } else if constexpr (TEST == 5) { // 8 independent v_cndmask_b32 (vcc)
        asm volatile("v_cmp_lt_f32 vcc, %8, %9\n"
                     R4(R4("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n")
                        R4("v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n"))
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
      } else if constexpr (TEST == 6) { // 4 independent v_pk_fma_f32 (register pairs)
        double p0 = __hiloint2double(__float_as_int(a0), __float_as_int(a1)), p1 = __hiloint2double(__float_as_int(a2), __float_as_int(a3)),
               p2 = __hiloint2double(__float_as_int(a4), __float_as_int(a5)), p3 = __hiloint2double(__float_as_int(a6), __float_as_int(a7));
        double pb = __hiloint2double(__float_as_int(b), __float_as_int(b)), pc = __hiloint2double(__float_as_int(c), __float_as_int(c));
        asm volatile(R16("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n")
                     : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc));
        a0 += (float)__double2loint(p0) + (float)__double2loint(p1) + (float)__double2loint(p2) + (float)__double2loint(p3);
      } else if constexpr (TEST == 7) { // dependent v_rcp_f32
        asm volatile(R64("v_rcp_f32 %0, %0\n") : "+v"(a0));
      } else if constexpr (TEST == 8) { // 4 independent v_rcp_f32
        asm volatile(R16("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
      } else if constexpr (TEST == 9) { // dependent IEEE division as the compiler emits it (64 per iteration)
#pragma unroll
        for (int k = 0; k < 64; k++) a0 = b / a0;
      } else if constexpr (TEST == 10) { // dependent sqrtf (IEEE)
#pragma unroll
        for (int k = 0; k < 64; k++) a0 = sqrtf(a0) + b;
      } else if constexpr (TEST == 11) { // dependent ds_read_b32 chain (pointer chasing inside LDS)
        asm volatile(R64("ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)\n") : "+v"(idx));
      } else if constexpr (TEST == 12) { // ds_write_b32 -> ds_read_b32 of the same word, dependent (a hand-off round trip)
        asm volatile(R64("ds_write_b32 %1, %0\n ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)\n") : "+v"(a0) : "v"(idx));
      } else if constexpr (TEST == 13) { // 8 independent ds_read_b128 then one wait (batched reads)
        v4f q0, q1, q2, q3;
        asm volatile(R16("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:16\n ds_read_b128 %2, %4 offset:32\n ds_read_b128 %3, %4 offset:48\n s_waitcnt lgkmcnt(0)\n")
                     : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3) : "v"(idx));
        a0 += q0.x + q1.y + q2.z + q3.w;
      } else if constexpr (TEST == 14) { // dependent ds_bpermute_b32
        asm volatile(R64("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)\n") : "+v"(a0) : "v"(idx & 255u));
      } else if constexpr (TEST == 15) { // dependent DPP move (row_shr:1)
        asm volatile(R64("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n") : "+v"(a0));
      } else if constexpr (TEST == 16) { // divergent-branch skeleton around one fma: saveexec / branch / restore
        asm volatile(R64("v_cmp_lt_f32 vcc, %1, %0\n s_and_saveexec_b64 s[20:21], vcc\n s_cbranch_execz 1f\n v_fma_f32 %0, %0, %1, %2\n1:\n s_or_b64 exec, exec, s[20:21]\n")
                     : "+v"(a0) : "v"(b), "v"(c) : "vcc", "s20", "s21");
      } else if constexpr (TEST == 19) { // 8 independent v_cndmask_b32_e64 with an SGPR-pair mask
        asm volatile("v_cmp_lt_f32 s[20:21], %8, %9\n"
                     R4(R4("v_cndmask_b32 %0, %0, %8, s[20:21]\n v_cndmask_b32 %1, %1, %8, s[20:21]\n v_cndmask_b32 %2, %2, %8, s[20:21]\n v_cndmask_b32 %3, %3, %8, s[20:21]\n")
                        R4("v_cndmask_b32 %4, %4, %8, s[20:21]\n v_cndmask_b32 %5, %5, %8, s[20:21]\n v_cndmask_b32 %6, %6, %8, s[20:21]\n v_cndmask_b32 %7, %7, %8, s[20:21]\n"))
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "s20", "s21");
      } else if constexpr (TEST == 20) { // v_cndmask_b32 (vcc) with distinct destinations: no read-after-write on the selected operand
        float d0, d1, d2, d3;
        asm volatile("v_cmp_lt_f32 vcc, %4, %5\n"
                     R16("v_cndmask_b32 %0, %4, %5, vcc\n v_cndmask_b32 %1, %5, %4, vcc\n v_cndmask_b32 %2, %4, %5, vcc\n v_cndmask_b32 %3, %5, %4, vcc\n")
                     : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3) : "v"(b), "v"(c) : "vcc");
        a0 += d0 + d1 + d2 + d3;
      } else if constexpr (TEST == 21) { // v_max_f32 / v_min_f32, 4 independent (what clamps compile to)
        asm volatile(R16("v_max_f32 %0, %0, %4\n v_min_f32 %1, %1, %4\n v_max_f32 %2, %2, %4\n v_min_f32 %3, %3, %4\n")
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));
      } else if constexpr (TEST == 22) { // ds_write_b128 + ds_read_b128 of the lane's own 16 bytes (a 4-float hand-off round trip)
        const unsigned ad = (threadIdx.x * 16u) & 16383u;
        v4f qi = {a0, a1, a2, a3}, qo;
        asm volatile(R64("ds_write_b128 %1, %2\n ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)\n") : "=&v"(qo) : "v"(ad), "v"(qi));
        a4 += qo.x;
      } else if constexpr (TEST == 23) { // 7 x ds_read_b128 (28 floats: one ABA hand-off record) in flight, one wait
        v4f q0, q1, q2, q3, q4, q5, q6;
        const unsigned ad = (threadIdx.x * 112u) % 16000u & ~15u;
        asm volatile(R64("ds_read_b128 %0, %7\n ds_read_b128 %1, %7 offset:16\n ds_read_b128 %2, %7 offset:32\n ds_read_b128 %3, %7 offset:48\n"
                         "ds_read_b128 %4, %7 offset:64\n ds_read_b128 %5, %7 offset:80\n ds_read_b128 %6, %7 offset:96\n s_waitcnt lgkmcnt(0)\n")
                     : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3), "=&v"(q4), "=&v"(q5), "=&v"(q6) : "v"(ad));
        a0 += q0.x + q1.y + q2.z + q3.w + q4.x + q5.y + q6.z;
      } else if constexpr (TEST == 24) { // dependent global_load_dword, L2-resident pointer chase
        const unsigned* gp = reinterpret_cast<const unsigned*>(out + 512 * 16);   // scratch words behind the results
        unsigned off = idx;
#pragma unroll
        for (int k = 0; k < 64; k++) off = __builtin_nontemporal_load(gp + (off >> 2)) & 16383u;
        idx = off;
      } else if constexpr (TEST == 25) { // v_cndmask_b32 in the VOP3 encoding with vcc as the mask operand
        asm volatile("v_cmp_lt_f32 vcc, %8, %9\n"
                     R4(R4("v_cndmask_b32_e64 %0, %0, %8, vcc\n v_cndmask_b32_e64 %1, %1, %8, vcc\n v_cndmask_b32_e64 %2, %2, %8, vcc\n v_cndmask_b32_e64 %3, %3, %8, vcc\n")
                        R4("v_cndmask_b32_e64 %4, %4, %8, vcc\n v_cndmask_b32_e64 %5, %5, %8, vcc\n v_cndmask_b32_e64 %6, %6, %8, vcc\n v_cndmask_b32_e64 %7, %7, %8, vcc\n"))
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
      } else if constexpr (TEST == 26) { // v_cmp_lt_f32 (VOPC, writes vcc) + v_cndmask_b32_e32 pairs: what `a < b ? a : b` compiles to
        asm volatile(R16("v_cmp_lt_f32 vcc, %0, %4\n v_cndmask_b32 %0, %4, %0, vcc\n v_cmp_lt_f32 vcc, %1, %4\n v_cndmask_b32 %1, %4, %1, vcc\n")
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc");
      } else if constexpr (TEST == 27) { // the same pairs through an SGPR pair (VOP3 compare + VOP3 select)
        asm volatile(R16("v_cmp_lt_f32 s[20:21], %0, %4\n v_cndmask_b32 %0, %4, %0, s[20:21]\n v_cmp_lt_f32 s[22:23], %1, %4\n v_cndmask_b32 %1, %4, %1, s[22:23]\n")
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "s20", "s21", "s22", "s23");
      } else if constexpr (TEST == 28) { // v_addc_co_u32 reading vcc (64-bit address arithmetic), 4 independent
        unsigned u0 = idx, u1 = idx + 1, u2 = idx + 2, u3 = idx + 3;
        asm volatile(R16("v_addc_co_u32 %0, vcc, %0, %4, vcc\n v_addc_co_u32 %1, vcc, %1, %4, vcc\n v_addc_co_u32 %2, vcc, %2, %4, vcc\n v_addc_co_u32 %3, vcc, %3, %4, vcc\n")
                     : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(idx) : "vcc");
        idx = u0 + u1 + u2 + u3;
      } else if constexpr (TEST == 17) { // SALU dependent chain
        unsigned s = (unsigned)it;
        asm volatile(R64("s_add_u32 %0, %0, 3\n") : "+s"(s));
        a0 += (float)s;
      } else if constexpr (TEST == 18) { // dependent v_readlane -> v_mov (VALU -> SALU -> VALU hop)
        unsigned s;
        asm volatile(R64("v_readlane_b32 %1, %0, 3\n s_nop 3\n v_mov_b32 %0, %1\n") : "+v"(a0), "=s"(s));
      }
    }
    t1 = __builtin_readcyclecounter();
  }
  const int w = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
  if ((threadIdx.x & 63) == 0) { out[w].ticks = t1 - t0; out[w].sink = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + __uint_as_float(idx); }
}

struct Test { int id; const char* name; int per_iter; };
static const Test TESTS[] = {
    {0, "v_fma_f32, 1 dependent chain", 64}, {1, "v_fma_f32, 2 independent chains", 64}, {2, "v_fma_f32, 4 independent chains", 64},
    {3, "v_fma_f32, 8 independent chains", 128}, {4, "v_mul_f32, 8 independent", 128}, {5, "v_cndmask_b32 (vcc), 8 independent", 128},
    {6, "v_pk_fma_f32, 4 independent (2 fma each)", 64}, {7, "v_rcp_f32, dependent", 64}, {8, "v_rcp_f32, 4 independent", 64},
    {9, "IEEE a/b (div_scale..div_fixup), dependent", 64}, {10, "IEEE sqrtf + add, dependent", 64},
    {11, "ds_read_b32 pointer chase", 64}, {12, "ds_write_b32 + ds_read_b32 round trip", 64}, {13, "ds_read_b128 x4 batched + wait", 64},
    {14, "ds_bpermute_b32, dependent", 64}, {15, "v_mov_b32_dpp row_shr:1, dependent", 64},
    {16, "saveexec / cbranch_execz / fma / restore", 64}, {17, "s_add_u32, dependent", 64}, {18, "v_readlane -> v_mov hop", 64},
    {19, "v_cndmask_b32_e64 (sgpr mask), 8 independent", 128}, {20, "v_cndmask_b32 (vcc), fresh destinations", 64},
    {21, "v_max_f32 / v_min_f32, 4 independent", 64}, {22, "ds_write_b128 + ds_read_b128 round trip", 64},
    {23, "7 x ds_read_b128 in flight + wait (per 7 reads)", 64}, {24, "global_load_dword pointer chase (L2)", 64},
    {25, "v_cndmask_b32_e64 with vcc as mask, 8 independent", 128}, {26, "v_cmp (vcc) + v_cndmask_e32 pair", 32},
    {27, "v_cmp_e64 (sgpr) + v_cndmask_e64 pair", 32}, {28, "v_addc_co_u32 (vcc in/out), 4 independent", 64}};

template <int T> static void launch(Out* d, int blocks, int threads, int lanes) { hipLaunchKernelGGL(k_bench<T>, dim3(blocks), dim3(threads), 0, 0, d, 1.0f, lanes); }
typedef void (*Launcher)(Out*, int, int, int);
static const Launcher LAUNCH[] = {launch<0>, launch<1>, launch<2>, launch<3>, launch<4>, launch<5>, launch<6>, launch<7>, launch<8>, launch<9>,
                                  launch<10>, launch<11>, launch<12>, launch<13>, launch<14>, launch<15>, launch<16>, launch<17>, launch<18>,
                                  launch<19>, launch<20>, launch<21>, launch<22>, launch<23>, launch<24>, launch<25>, launch<26>, launch<27>, launch<28>};

int main(int argc, char** argv) {
  hipDeviceProp_t p;
  CHECK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  int clock_khz = 0;
  CHECK(hipDeviceGetAttribute(&clock_khz, hipDeviceAttributeClockRate, 0));
  printf("# VALU / LDS issue microbenchmark (%s, %d CUs, clock attribute %.0f MHz)\n\n", p.gcnArchName, cus, clock_khz / 1000.0);
  printf("One kernel per row and occupancy: every SIMD of the chip hosts W waves, each wave runs %d x 64 copies of the pattern\n"
         "between two s_memtime reads.  `clk/instr (wave)` = median ticks per pattern instance as one wave sees it;\n"
         "`clk/instr (SIMD)` = that / W = what the SIMD spends per wave-instruction; `GHz` = ticks / wall time of the launch.\n\n", ITER);
  Out* d;
  const int maxw = 512 * 16;
  CHECK(hipMalloc(&d, sizeof(Out) * maxw + 4 * 32768));
  {
    std::vector<unsigned> chase(16384 + 16384);
    for (unsigned i = 0; i < chase.size(); i++) chase[i] = (i * 4u * 17u + 64u) & 16383u;
    CHECK(hipMemcpy(reinterpret_cast<unsigned*>(d + maxw), chase.data() + 0, 4 * 16384, hipMemcpyHostToDevice));
  }
  std::vector<Out> h(maxw);
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int WS[] = {1, 2, 4};
  printf("| pattern | lanes | W=1 wave | W=1 SIMD | W=2 wave | W=2 SIMD | W=4 wave | W=4 SIMD | GHz (W=1) |\n|---|---:|---:|---:|---:|---:|---:|---:|---:|\n");
  for (const Test& t : TESTS) {
    if (argc > 1) { bool want = false; for (int a = 1; a < argc; a++) want |= atoi(argv[a]) == t.id; if (!want) continue; }
    for (int lanes : {64, 17}) {
      if (lanes != 64 && !(t.id == 0 || t.id == 3 || t.id == 12)) continue;
      printf("| %s | %d |", t.name, lanes);
      double ghz1 = 0;
      for (int W : WS) {
        const int threads = W <= 4 ? 256 * W : 1024, blocks = W <= 4 ? cus : cus * 2;
        const int nw = blocks * threads / 64;
        LAUNCH[t.id](d, blocks, threads, lanes);   // warm-up
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        LAUNCH[t.id](d, blocks, threads, lanes);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        CHECK(hipMemcpy(h.data(), d, sizeof(Out) * nw, hipMemcpyDeviceToHost));
        std::vector<unsigned long long> tk(nw);
        for (int i = 0; i < nw; i++) tk[i] = h[i].ticks;
        std::sort(tk.begin(), tk.end());
        const double med = (double)tk[nw / 2], n = (double)ITER * t.per_iter;
        printf(" %.2f | %.2f |", med / n, med / n / W);
        if (W == 1) ghz1 = (double)tk[nw - 1] / (ms * 1e6);
      }
      printf(" %.2f |\n", ghz1);
    }
  }
  printf("\nNotes: a pattern instance of the round-trip / branch rows is several instructions (see the source); "
         "`lanes` = 17 runs the same stream with 17 of 64 lanes enabled in EXEC.  s_memtime ticks: see the GHz column -- if it reads ~0.1 the counter is the 100 MHz constant clock and the tick columns must be scaled by (shader clock / 100 MHz).\n");
  return 0;
}
