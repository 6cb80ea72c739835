"""Pure-Python namespace standing in for the `isaacgym` package on the MI355X
backend: `from shifu_amd.isaacgym import gymapi, gymtorch, gymutil`,
`from shifu_amd.isaacgym.torch_utils import *`, `terrain_utils`.
shifu_amd.compat.install() additionally registers it as `isaacgym` in sys.modules
so unmodified reference user code imports it."""
from . import gymapi, gymtorch, gymutil, terrain_utils, torch_utils  # noqa: F401
