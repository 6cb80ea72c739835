// shf_kernel_list.h -- every explicit instantiation of the kernel templates in shf_kernels.h, grouped by the translation unit
// (shf_k_<family>.hip) that compiles it.  The includer defines SHF_KERNEL(...) and, to select one family, SHF_KERNEL_FAMILY_<name>;
// with no family selected every entry is listed (shf_api.hip: extern template declarations).
#if defined(SHF_KERNEL_FAMILY_sim) || defined(SHF_KERNEL_FAMILY_sim_link) || defined(SHF_KERNEL_FAMILY_sim_hard) || defined(SHF_KERNEL_FAMILY_sim_hard_wide) || defined(SHF_KERNEL_FAMILY_a1) || defined(SHF_KERNEL_FAMILY_abb) || defined(SHF_KERNEL_FAMILY_abb_link) || defined(SHF_KERNEL_FAMILY_abb_hard) || defined(SHF_KERNEL_FAMILY_abb_ws)
#define SHF_KERNEL_ONE_FAMILY
#endif
#if !defined(SHF_KERNEL_ONE_FAMILY) || defined(SHF_KERNEL_FAMILY_sim)
SHF_KERNEL(k_sim_step<64, true, true>(SimArgs))
SHF_KERNEL(k_sim_step<64, true, false>(SimArgs))
SHF_KERNEL(k_sim_step<64, false, true>(SimArgs))
SHF_KERNEL(k_sim_step<64, false, false>(SimArgs))
SHF_KERNEL(k_sim_step<32, true, true>(SimArgs))
SHF_KERNEL(k_sim_step<32, true, false>(SimArgs))
SHF_KERNEL(k_sim_step<32, false, true>(SimArgs))
SHF_KERNEL(k_sim_step<32, false, false>(SimArgs))
SHF_KERNEL(k_sim_step<16, true, true>(SimArgs))
SHF_KERNEL(k_sim_step<16, true, false>(SimArgs))
SHF_KERNEL(k_sim_step<16, false, true>(SimArgs))
SHF_KERNEL(k_sim_step<16, false, false>(SimArgs))
#endif
#if !defined(SHF_KERNEL_ONE_FAMILY) || defined(SHF_KERNEL_FAMILY_sim_link)
SHF_KERNEL(k_sim_step<64, true, true, true>(SimArgs))
SHF_KERNEL(k_sim_step<64, true, false, true>(SimArgs))
SHF_KERNEL(k_sim_step<32, true, true, true>(SimArgs))
SHF_KERNEL(k_sim_step<32, true, false, true>(SimArgs))
SHF_KERNEL(k_sim_step<16, true, true, true>(SimArgs))
SHF_KERNEL(k_sim_step<16, true, false, true>(SimArgs))
#endif
#if !defined(SHF_KERNEL_ONE_FAMILY) || defined(SHF_KERNEL_FAMILY_sim_hard)
SHF_KERNEL(k_sim_step<32, true, true, true, true>(SimArgs))
SHF_KERNEL(k_sim_step<32, true, false, true, true>(SimArgs))
SHF_KERNEL(k_sim_step<32, true, true, false, true>(SimArgs))
SHF_KERNEL(k_sim_step<32, true, false, false, true>(SimArgs))
SHF_KERNEL(k_sim_step<32, false, true, false, true>(SimArgs))
SHF_KERNEL(k_sim_step<32, false, false, false, true>(SimArgs))
#endif
#if !defined(SHF_KERNEL_ONE_FAMILY) || defined(SHF_KERNEL_FAMILY_sim_hard_wide)
SHF_KERNEL(k_sim_step_pgs_wide<true>(SimArgs))
SHF_KERNEL(k_sim_step_pgs_wide<false>(SimArgs))
#endif
#if !defined(SHF_KERNEL_ONE_FAMILY) || defined(SHF_KERNEL_FAMILY_a1)
SHF_KERNEL(k_a1_step<64, DynDims>(A1Args))
SHF_KERNEL(k_a1_step<64, A1Dims>(A1Args))
SHF_KERNEL(k_a1_step<32, DynDims>(A1Args))
SHF_KERNEL(k_a1_step<32, A1Dims>(A1Args))
SHF_KERNEL(k_a1_step<16, DynDims>(A1Args))
SHF_KERNEL(k_a1_step_self<32, DynDims>(A1Args))
SHF_KERNEL(k_a1_step_self<32, A1Dims>(A1Args))
SHF_KERNEL(k_a1_step_self<16, DynDims>(A1Args))
#endif
#if !defined(SHF_KERNEL_ONE_FAMILY) || defined(SHF_KERNEL_FAMILY_abb)
SHF_KERNEL(k_abb_step<64, DynDims, DynScene>(AbbArgs))
SHF_KERNEL(k_abb_step<64, AbbDims, AbbScene>(AbbArgs))
SHF_KERNEL(k_abb_step<32, DynDims, DynScene>(AbbArgs))
SHF_KERNEL(k_abb_step<32, AbbDims, AbbScene>(AbbArgs))
SHF_KERNEL(k_abb_step<16, DynDims, DynScene>(AbbArgs))
SHF_KERNEL(k_abb_step<16, AbbDims, AbbScene>(AbbArgs))
SHF_KERNEL(k_abb_step<32, AbbDims, AbbScene, false, 6>(AbbArgs))
SHF_KERNEL(k_abb_step<16, AbbDims, AbbScene, false, 6>(AbbArgs))
#endif
#if !defined(SHF_KERNEL_ONE_FAMILY) || defined(SHF_KERNEL_FAMILY_abb_link)
SHF_KERNEL(k_abb_step<64, DynDims, DynScene, true>(AbbArgs))
SHF_KERNEL(k_abb_step<64, AbbLinkDims, AbbScene, true>(AbbArgs))
SHF_KERNEL(k_abb_step<32, DynDims, DynScene, true>(AbbArgs))
SHF_KERNEL(k_abb_step<32, AbbLinkDims, AbbScene, true>(AbbArgs))
SHF_KERNEL(k_abb_step<16, DynDims, DynScene, true>(AbbArgs))
SHF_KERNEL(k_abb_step<16, AbbLinkDims, AbbScene, true>(AbbArgs))
#endif
#if !defined(SHF_KERNEL_ONE_FAMILY) || defined(SHF_KERNEL_FAMILY_abb_hard)
SHF_KERNEL(k_abb_step<32, DynDims, DynScene, true, 0, true>(AbbArgs))
SHF_KERNEL(k_abb_step<32, DynDims, DynScene, false, 0, true>(AbbArgs))
SHF_KERNEL(k_abb_step_pgs_wide<true>(AbbArgs))
SHF_KERNEL(k_abb_step_pgs_wide<false>(AbbArgs))
#endif
#if !defined(SHF_KERNEL_ONE_FAMILY) || defined(SHF_KERNEL_FAMILY_abb_ws)
SHF_KERNEL(k_abb_step_ws<512, true>(AbbArgs))
SHF_KERNEL(k_abb_step_ws<256>(AbbArgs))
#endif
#undef SHF_KERNEL_ONE_FAMILY
