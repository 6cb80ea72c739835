// shf_a1_chain.hip -- instantiations of the chain-mapped fused A1 step (shf_chain.h); its own translation unit so that
// the two mappings compile side by side.  shf_api.hip owns the C ABI and launches these through their function pointers.
#include <hip/hip_runtime.h>

#include "shf_chain.h"
#include "shf_chain_hard.h"

template <int G, bool TW, bool SELF = false>
__global__ __launch_bounds__(256, (G == 32 ? 2 : 1)) void k_a1_chain(A1Args A) { a1_chain_step_body<G, A1Chain, TW, SELF>(A); }

// the same step under the velocity-level contact solve (ShfSimParams.solver == SHF_SOLVER_PGS / SHF_SOLVER_TGS; csrc/shf_chain_hard.h):
// k_a1_chain_pgs / _tgs hold 8 constraints per env, ..16 up to 16 (ShfSimParams.max_contacts > 8: the response matrix's upper triangle, packed)
template <bool TW, bool SELF>
__global__ __launch_bounds__(256, 2) void k_a1_chain_pgs(A1Args A) { a1_chain_step_body<32, A1Chain, TW, SELF, true>(A); }
template <bool TW, bool SELF>
__global__ __launch_bounds__(256, 2) void k_a1_chain_pgs16(A1Args A) { a1_chain_step_body<32, A1Chain, TW, SELF, true, 16>(A); }
template <bool TW, bool SELF>
__global__ __launch_bounds__(256, 2) void k_a1_chain_tgs(A1Args A) { a1_chain_step_body<32, A1Chain, TW, SELF, true, 8, true>(A); }
template <bool TW, bool SELF>
__global__ __launch_bounds__(256, 2) void k_a1_chain_tgs16(A1Args A) { a1_chain_step_body<32, A1Chain, TW, SELF, true, 16, true>(A); }
#define SHF_PICK4(K, warped, self) ((self) ? ((warped) ? reinterpret_cast<const void*>(K<true, true>) : reinterpret_cast<const void*>(K<false, true>)) \
                                           : ((warped) ? reinterpret_cast<const void*>(K<true, false>) : reinterpret_cast<const void*>(K<false, false>)))
const void* shf_a1_chain_pgs_kernel(bool warped, bool self, bool k16, bool tgs) {
  if (tgs) return k16 ? SHF_PICK4(k_a1_chain_tgs16, warped, self) : SHF_PICK4(k_a1_chain_tgs, warped, self);
  return k16 ? SHF_PICK4(k_a1_chain_pgs16, warped, self) : SHF_PICK4(k_a1_chain_pgs, warped, self);
}
int shf_a1_chain_pgs_max_contacts(void) { return 16; }

// gym.simulate (examples/a1_conditional/a1_conditional.py:69, shifu/gym/isaac_gym.py:140) of the hook path under the
// velocity-level solve, for an A1-shaped articulation on its own (no box actors): one chain_substep_hard per call.
// Forces at the centres of mass.
template <bool TW, bool SELF, int KC, bool TGS>
DEV void sim_step_chain_pgs_body(const SimArgs& A) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  typedef A1Chain CD;
  constexpr int G = 32, nb = CD::NB, nd = CD::ND, NR = (CD::NEV + G - 1) / G;
  const int es = threadIdx.x / G, l = threadIdx.x % G;
  const int e = blockIdx.x * (256 / G) + es;
  stage_block<CHAIN_MODEL_BYTES>(A.model, smem);
  __syncthreads();
  const ShfModel* m = reinterpret_cast<const ShfModel*>(smem);
  if (e >= A.n) return;
  ChainLds L = chain_lds_carve<CD>(smem + CHAIN_MODEL_WORDS + es * chain_lds_words<CD>(0, SELF));
  const float* dof = A.dof + (size_t)e * nd * 2;
  const float* root = A.root + (size_t)e * 13;
  for (int i = l; i < 2 * nd; i += G) L.dofb[(i >> 1) * DOF_STRIDE + (i & 1)] = dof[i];
  if (l < 13) L.root[l] = root[l];
  GROUP_SYNC();
  StepCtx C;
  C.m = m; C.sp = A.sp; C.terr.t = A.terr; C.terr.h = A.heights; C.scene = nullptr;
  C.dropped = env_dropped(A.dropped, A.sp, e);
  C.mscale = A.mscale ? A.mscale + (size_t)e * nb : nullptr;
  const float mu = A.friction ? A.friction[e] : 1.0f;
  const int dl = l < nd ? l : 0;
  DofLane X = {L.dofb[dl * DOF_STRIDE], L.dofb[dl * DOF_STRIDE + 1], (A.effort && l < nd) ? A.effort[(size_t)e * nd + l] : 0.0f,
               (A.pos_tgt && l < nd) ? A.pos_tgt[(size_t)e * nd + l] : 0.0f, (A.vel_tgt && l < nd) ? A.vel_tgt[(size_t)e * nd + l] : 0.0f};
  ChainPoints<NR> LP;
  chain_points_load<G>(m, CD::NEV, l, C.sp.contact_offset + C.sp.rest_offset, LP);
  const RowLane RL = row_lane_load<CD>(l);
  chain_substep_hard<G, CD, TW, SELF, KC, TGS>(C, L, l, X, LP, RL, A.body_force ? A.body_force + (size_t)e * nb * 3 : nullptr, mu, L.xch);
  if (l < nd) {
    A.dof[((size_t)e * nd + l) * 2] = X.q;
    A.dof[((size_t)e * nd + l) * 2 + 1] = X.qd;
  }
  if (l < 13) A.root[(size_t)e * 13 + l] = L.root[l];
  for (int i = l; i < 3 * nb; i += G) A.contact[(size_t)e * nb * 3 + i] = L.xch[i];
}
template <bool TW, bool SELF>
__global__ __launch_bounds__(256, 2) void k_sim_step_chain_pgs(SimArgs A) { sim_step_chain_pgs_body<TW, SELF, 8, false>(A); }
template <bool TW, bool SELF>
__global__ __launch_bounds__(256, 2) void k_sim_step_chain_pgs16(SimArgs A) { sim_step_chain_pgs_body<TW, SELF, 16, false>(A); }
template <bool TW, bool SELF>
__global__ __launch_bounds__(256, 2) void k_sim_step_chain_tgs(SimArgs A) { sim_step_chain_pgs_body<TW, SELF, 8, true>(A); }
template <bool TW, bool SELF>
__global__ __launch_bounds__(256, 2) void k_sim_step_chain_tgs16(SimArgs A) { sim_step_chain_pgs_body<TW, SELF, 16, true>(A); }
const void* shf_sim_step_chain_pgs_kernel(bool warped, bool self, bool k16, bool tgs) {
  if (tgs) return k16 ? SHF_PICK4(k_sim_step_chain_tgs16, warped, self) : SHF_PICK4(k_sim_step_chain_tgs, warped, self);
  return k16 ? SHF_PICK4(k_sim_step_chain_pgs16, warped, self) : SHF_PICK4(k_sim_step_chain_pgs, warped, self);
}
size_t shf_sim_step_chain_pgs_lds_bytes(bool self) { return ((size_t)CHAIN_MODEL_WORDS + 8 * (size_t)chain_lds_words<A1Chain>(0, self)) * 4; }

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
  unsigned long long h[48];
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_phase_cycles), sizeof h) != hipSuccess) return 1;
  for (int k = 0; k < n && k < 48; k++) out[k] = h[k];
  if (reset) {
    unsigned long long z[48] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_phase_cycles), z, sizeof z) != hipSuccess) return 1;
  }
  return 0;
}
#endif
