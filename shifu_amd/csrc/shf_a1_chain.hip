// shf_a1_chain.hip -- instantiations of the chain-mapped fused A1 step (shf_chain.h); its own translation unit so that
// the two mappings compile side by side.  shf_api.hip owns the C ABI and launches these through their function pointers.
#include <hip/hip_runtime.h>

#include "shf_chain.h"
#include "shf_chain_hard.h"

template <int G, bool TW, bool SELF = false>
__global__ __launch_bounds__(256, (G == 32 ? 2 : 1)) void k_a1_chain(A1Args A) { a1_chain_step_body<G, A1Chain, TW, SELF>(A); }

// the same step under the velocity-level contact solve (ShfSimParams.solver == SHF_SOLVER_PGS; csrc/shf_chain_hard.h)
template <bool TW>
__global__ __launch_bounds__(256, 2) void k_a1_chain_pgs(A1Args A) { a1_chain_step_body<32, A1Chain, TW, false, true>(A); }
const void* shf_a1_chain_pgs_kernel(bool warped) {
  return warped ? reinterpret_cast<const void*>(k_a1_chain_pgs<true>) : reinterpret_cast<const void*>(k_a1_chain_pgs<false>);
}
int shf_a1_chain_pgs_max_contacts(void) { return HCK; }

bool shf_a1_chain_matches(const ShfModel& m) { return A1Chain::matches(m); }
// dynamic LDS of one 256-thread block at G lanes per env
size_t shf_a1_chain_lds_bytes(int G, int nobs, bool self) {
  const int epb = 256 / G;
  return ((size_t)CHAIN_MODEL_WORDS + TASK_WORDS + STATS_LDS_WORDS + (size_t)epb * chain_lds_words<A1Chain>(SCR_OBS + nobs, self)) * 4;
}
// self: with the capsule-pair self-collision pass (32 lanes per env only: the pair tests want the lanes)
const void* shf_a1_chain_kernel(int G, bool warped, bool self) {
  if (self) {
    if (G != 32) return nullptr;
    return warped ? reinterpret_cast<const void*>(k_a1_chain<32, true, true>) : reinterpret_cast<const void*>(k_a1_chain<32, false, true>);
  }
  switch (G) {
    case 16: return warped ? reinterpret_cast<const void*>(k_a1_chain<16, true>) : reinterpret_cast<const void*>(k_a1_chain<16, false>);
    case 32: return warped ? reinterpret_cast<const void*>(k_a1_chain<32, true>) : reinterpret_cast<const void*>(k_a1_chain<32, false>);
    default: return nullptr;
  }
}
#ifdef SHF_PHASE_CLOCK
int shf_a1_chain_phase_cycles(unsigned long long* out, int n, int reset) {
  unsigned long long h[32];
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_phase_cycles), sizeof h) != hipSuccess) return 1;
  for (int k = 0; k < n && k < 32; k++) out[k] = h[k];
  if (reset) {
    unsigned long long z[32] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_phase_cycles), z, sizeof z) != hipSuccess) return 1;
  }
  return 0;
}
#endif
