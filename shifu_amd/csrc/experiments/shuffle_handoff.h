// experiments/shuffle_handoff.h -- EXPERIMENT (not part of the product build): the child -> parent hand-off of the ABA
// inward pass through registers instead of LDS, which is what BASELINE's north-star wording ("wavefront shuffles for the
// per-body pass") describes.  Built only with -DSHF_EXP_SHUFFLE_HANDOFF (tools/experiment.py shuffle_handoff).
//
// Bodies are numbered depth first, so a moving body's first moving child is the next lane: its (IA, pA) -- 27 floats --
// arrive with one DPP `wave_shl:1` each; further children (only the floating base has any: the four hips) come through
// ds_bpermute, executed wave-wide only at the level where some lane has such a child.  Children are added in
// child_list order, like the LDS path, so results are bit-identical (tests/test_gpu_parity.py under SHIFU_AMD_LIB).
//
// Measured on the MI355X (4096 envs, config 3, this round's kernel): see profiles/r02_experiments.md.  It is slower: the
// 27 cross-lane moves per level are issued by the whole wave whether or not a lane is a parent, and each predicated add
// waits on its own move, where the LDS path issues 7 ds_write_b128 + 7 ds_read_b128 that the scheduler batches.
#pragma once

DEV float exp_dpp_wave_shl1(float v) {
  // lane l receives lane l + 1's value (GFX9 DPP wave shift; the last lane keeps `old` = 0)
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, false));
}
DEV float exp_bpermute(float v, int src_lane_in_wave) {
  return __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane_in_wave << 2, __float_as_int(v)));
}

// Called by every lane of the wave once per level `lev` after the lanes at that level have updated (B.IA, B.pA).
// `isparent`: this lane is a moving body of level lev - 1.
template <int G, class LM>
DEV void exp_shuffle_handoff(const LM& M, int l, bool isparent, BodyRegs& B) {
  const int lane0 = (int)(threadIdx.x & 63u) - l;   // first lane of this env's group within the wavefront
  float v[27];
#pragma unroll
  for (int k = 0; k < 21; k++) v[k] = B.IA[k];
#pragma unroll
  for (int k = 0; k < 6; k++) v[21 + k] = B.pA[k];
  // first child: the next lane (depth-first numbering; checked on the host for the models this build is run with)
  const bool has0 = isparent && M.nchild > 0;
#pragma unroll
  for (int k = 0; k < 27; k++) {
    const float c = exp_dpp_wave_shl1(v[k]);
    if (has0) { if (k < 21) B.IA[k] += c; else B.pA[k - 21] += c; }
  }
#pragma unroll
  for (int kk = 1; kk < LANE_CHILDREN; kk++) {
    const bool hask = isparent && M.nchild > kk;
    if (__ballot(hask) == 0ull) continue;           // wave-uniform: nobody has a kk-th child at this level
    const int src = lane0 + (hask ? M.child[kk] : l);
#pragma unroll
    for (int k = 0; k < 27; k++) {
      const float c = exp_bpermute(v[k], src);
      if (hask) { if (k < 21) B.IA[k] += c; else B.pA[k - 21] += c; }
    }
  }
}
