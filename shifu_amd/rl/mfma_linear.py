"""nn.Linear (+ ELU) whose forward and backward run on the hand-written MFMA kernels of csrc/shf_mlp.hip.

Operands stay the trainer's fp32 tensors; the kernels convert tiles to bf16 on the fly and accumulate in fp32
(v_mfma_f32_32x32x16_bf16), with bias, ELU and the ELU derivative fused into the GEMMs' load / store paths -- one launch
per layer forward, two per layer backward (+ the deterministic split-M reduction of dW).  `MfmaLinear` has nn.Linear's
parameters and state_dict keys, so checkpoints are interchangeable with the stock layer.

Precision: bf16 operand rounding (relative 2^-9) with fp32 sums; tests/test_gpu_mlp.py holds every output to the fp32
torch reference within 2e-2 of the tensor's scale, and the A1 schedule trains to the same tracking rewards."""
import ctypes as C

import torch
import torch.nn as nn

from .._lib import BackendError, lib


def _check(rc):
    if rc != 0:
        raise BackendError(lib().shf_mlp_last_error().decode())


def _ptr(t):
    return C.c_void_p(t.data_ptr())


def _stream(t):
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


class _MfmaLinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, act):
        x = x.contiguous()
        M, K = x.shape
        N = weight.shape[0]
        y = torch.empty(M, N, device=x.device, dtype=torch.float32)
        with torch.cuda.device(x.device):
            _check(lib().shf_mlp_linear_forward(_ptr(x), _ptr(weight), _ptr(bias), _ptr(y), M, K, N, act, _stream(x)))
        ctx.save_for_backward(x, weight, y)
        ctx.act = act
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, y = ctx.saved_tensors
        gy = gy.contiguous()
        M, K = x.shape
        N = weight.shape[0]
        yp = _ptr(y) if ctx.act == 1 else None
        gx = gw = gb = None
        with torch.cuda.device(x.device):
            if ctx.needs_input_grad[0]:
                gx = torch.empty_like(x)
                _check(lib().shf_mlp_linear_backward_input(_ptr(gy), yp, _ptr(weight), _ptr(gx), M, K, N, _stream(x)))
            if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
                n = C.c_int64()
                _check(lib().shf_mlp_backward_weight_workspace(M, K, N, C.byref(n)))
                ws = torch.empty(n.value, device=x.device, dtype=torch.float32)
                gw, gb = torch.empty_like(weight), torch.empty(N, device=x.device, dtype=torch.float32)
                _check(lib().shf_mlp_linear_backward_weight(_ptr(gy), yp, _ptr(x), _ptr(gw), _ptr(gb), _ptr(ws), M, K, N,
                                                            _stream(x)))
        return gx, gw, gb, None


class MfmaLinear(nn.Linear):
    """y = act(x W^T + b) on the MFMA kernels (CUDA fp32 2-D inputs); anything else falls back to the stock ops of
    nn.Linear -- same maths in fp32 -- so the module also works on CPU (tests, checkpoints)."""

    def __init__(self, in_features, out_features, elu: bool = False):
        super().__init__(in_features, out_features)
        self.elu = bool(elu)

    def forward(self, x):
        if x.is_cuda and x.dim() == 2 and x.dtype == torch.float32:
            return _MfmaLinearFn.apply(x, self.weight, self.bias, 1 if self.elu else 0)
        y = super().forward(x)
        return nn.functional.elu(y) if self.elu else y
