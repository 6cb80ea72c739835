// shf_k_a1.hip -- explicit instantiations of one kernel family of shf_kernels.h (see shf_kernel_list.h), so that the families
// compile side by side.  No host logic here: the launches are in shf_api.hip.
#define SHF_DEFINE_A1_KERNELS
#include "shf_kernels.h"
#define SHF_KERNEL_FAMILY_a1
#define SHF_KERNEL(...) template __global__ void __VA_ARGS__;
#include "shf_kernel_list.h"
