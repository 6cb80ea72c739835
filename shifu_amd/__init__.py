"""shifu_amd -- MI355X-native vectorised-env backend behind shifu's ShifuVecEnv / Unit API.

Layout (only what the step path needs):
  csrc/        hand-written HIP kernels + the C ABI (include/shifu_amd.h)
  backend.py   torch-allocated buffers bound to the C ABI
  isaacgym/    naming facade for the `isaacgym` names shifu user code imports
  configs/ units/ gym/ runner/ utils/   host-side mirror of the reference's interface
  gym/a1_fused.py   A1Conditional with the whole env step in one launch
  parallel.py  env sharding + RCCL all-gather of episode stats
"""
__version__ = "0.1.0"
