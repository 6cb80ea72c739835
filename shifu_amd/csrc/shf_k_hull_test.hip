// shf_k_hull_test.hip -- the convex narrow phase (csrc/shf_hull.h) for given pairs, outside any env step: the entry point
// tests/test_gpu_hull.py holds to the oracle's convex_manifold on thousands of random pairs at 16 / 32 / 64 lanes per pair.
#include "shf_device.h"

int shf_set_error(const std::string& msg);

template <int G>
__global__ __launch_bounds__(64) void k_convex_manifold(int n, const float* in, const ShfHull* hull_a, float offset, float* out) {
  const int l = threadIdx.x % G;
  const int e = blockIdx.x * (64 / G) + threadIdx.x / G;
  // (every lane runs: the shuffles of a group need all of its lanes; a group beyond n works on the last pair and stores nothing)
  const float* a = in + (size_t)(e < n ? e : n - 1) * 30;
  PolyDev PA, PB;
  PA.h = hull_a; PB.h = nullptr;
#pragma unroll
  for (int k = 0; k < 9; k++) { PA.R[k] = a[k]; PB.R[k] = a[15 + k]; }
#pragma unroll
  for (int k = 0; k < 3; k++) { PA.p[k] = a[9 + k]; PA.hx[k] = a[12 + k]; PB.p[k] = a[24 + k]; PB.hx[k] = a[27 + k]; }
  float r[4][3], nn[3] = {0.0f, 0.0f, 0.0f}, phi[4];
  const int nc = convex_manifold_dev<G>(PA, PB, offset, l, r, nn, phi);
  if (e < n && l == 0) {
    float* o = out + (size_t)e * 20;
    for (int k = 0; k < 20; k++) o[k] = 0.0f;
    o[0] = (float)nc;
    for (int k = 0; k < 3; k++) o[1 + k] = nn[k];
    for (int q = 0; q < nc; q++) { for (int k = 0; k < 3; k++) o[4 + 4 * q + k] = r[q][k]; o[4 + 4 * q + 3] = phi[q]; }
  }
}

extern "C" int shf_convex_manifold(int32_t n, const float* pairs_dev, const ShfHull* hull_a_dev_or_null, float offset, int32_t lanes,
                                   float* out_dev, void* stream) {
  if (n <= 0 || !pairs_dev || !out_dev) return shf_set_error("shf_convex_manifold: bad argument");
  hipStream_t st = (hipStream_t)stream;
  switch (lanes) {
    case 64: hipLaunchKernelGGL(k_convex_manifold<64>, dim3(n), dim3(64), 0, st, n, pairs_dev, hull_a_dev_or_null, offset, out_dev); break;
    case 32: hipLaunchKernelGGL(k_convex_manifold<32>, dim3((n + 1) / 2), dim3(64), 0, st, n, pairs_dev, hull_a_dev_or_null, offset, out_dev); break;
    case 16: hipLaunchKernelGGL(k_convex_manifold<16>, dim3((n + 3) / 4), dim3(64), 0, st, n, pairs_dev, hull_a_dev_or_null, offset, out_dev); break;
    default: return shf_set_error("shf_convex_manifold: lanes must be 16, 32 or 64");
  }
  return hipGetLastError() == hipSuccess ? 0 : shf_set_error("shf_convex_manifold: launch failed");
}
