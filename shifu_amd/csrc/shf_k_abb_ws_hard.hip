// shf_k_abb_ws_hard.hip -- explicit instantiations of one kernel family of shf_kernels.h (see shf_kernel_list.h), so that the families
// compile side by side.  No host logic here: the launches are in shf_api.hip.
#include "shf_kernels.h"
#define SHF_KERNEL_FAMILY_abb_ws_hard
#define SHF_KERNEL(...) template __global__ void __VA_ARGS__;
#include "shf_kernel_list.h"
