"""`gymtorch` facade: wrap_tensor / unwrap_tensor (10 + 13 call sites in the
reference, e.g. shifu/gym/isaac_gym.py:127-130, shifu/units/robot.py:58-86).

In Isaac Gym, wrap_tensor turns a sim-owned device buffer into a zero-copy torch
view and unwrap_tensor hands torch storage down.  Here device memory is allocated
by torch and *bound* to the C-ABI library (shifu_amd/backend.py), so wrapping is
just handing out that tensor -- still zero-copy, still pointer-stable."""
import torch

from .gymapi import _TensorHandle


def wrap_tensor(handle) -> torch.Tensor:
    if handle is None:
        return None
    if isinstance(handle, _TensorHandle):
        return handle.tensor
    if isinstance(handle, torch.Tensor):
        return handle
    raise TypeError(f"wrap_tensor: expected an acquire_*_tensor handle, got {type(handle)}")


def unwrap_tensor(t: torch.Tensor):
    """Contiguous device tensor whose data_ptr the backend reads during the call."""
    if t is None:
        return None
    return t if t.is_contiguous() else t.contiguous()
