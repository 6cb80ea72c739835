"""nn.Linear (+ ELU) whose forward and backward run on the hand-written MFMA kernels of csrc/shf_mlp.hip.

Operands stay the trainer's fp32 tensors; the kernels convert tiles to bf16 on the fly and accumulate in fp32
(v_mfma_f32_32x32x16_bf16), with bias, ELU and the ELU derivative fused into the GEMMs' load / store paths -- one launch
per layer forward, two per layer backward (+ the deterministic split-M reduction of dW).  `MfmaLinear` has nn.Linear's
parameters and state_dict keys, so checkpoints are interchangeable with the stock layer.

Precision (`set_precision`, env SHIFU_AMD_MFMA_PRECISION): "bf16x3" (default) splits every operand value into a bf16
head and tail and accumulates three MFMAs per tile pair -- products good to 2^-16, outputs within 1e-4 of the fp32 torch
reference's scale (tests/test_gpu_mlp.py); "bf16" rounds operands once (2^-9 relative, 2e-2 of scale), one MFMA.  With
"bf16" two of three 3000-iteration A1 runs lost return late in training (the action-noise std grew faster than with fp32
layers); "bf16x3" costs a few per cent of the layer time, the layers being HBM-bound."""
import os
import ctypes as C

import torch
import torch.nn as nn

from .._lib import BackendError, lib


PRECISIONS = {"bf16": 0, "bf16x3": 1}


def set_precision(mode: str) -> None:
    """Process-wide operand precision of the MFMA layers: "bf16x3" or "bf16" (include/shifu_amd.h SHF_MLP_*)."""
    _check(lib().shf_mlp_set_precision(PRECISIONS[mode]))


def get_precision() -> str:
    v = lib().shf_mlp_get_precision()
    return next(k for k, x in PRECISIONS.items() if x == v)


_env_precision_applied = False


def _apply_env_precision():
    global _env_precision_applied
    if not _env_precision_applied:
        _env_precision_applied = True
        mode = os.environ.get("SHIFU_AMD_MFMA_PRECISION")
        if mode:
            set_precision(mode)


def _check(rc):
    if rc != 0:
        raise BackendError(lib().shf_mlp_last_error().decode())


def _ptr(t):
    return C.c_void_p(t.data_ptr())


def _stream(t):
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


# Rows from which the row-panel kernels (k_mlp_panel: one block per 32 / 64 whole rows, at most 512 columns) replace the
# tiled GEMM for forward and input gradient -- same bits, 10-25 % less time per call at 24 576 rows (profiles/r04_mlp_panel.md);
# below it their grid does not fill the 256 CUs (the 4096-row rollout batches stay on the tiled kernel).
# SHIFU_AMD_MLP_PANEL_ROWS overrides (0 = never).
PANEL_MIN_ROWS = int(os.environ.get("SHIFU_AMD_MLP_PANEL_ROWS", "8192"))


def _pack_weights(weight):
    """The layer's weights in MFMA fragment order (bf16 heads and tails, plain and transposed): shf_mlp_pack_weights.
    Packed on every forward that uses the panel kernels -- the weights change every optimizer step, and a cached pack
    inside a captured graph would go stale without notice; the launch is a few microseconds."""
    N, K = weight.shape
    n = C.c_int64()
    _check(lib().shf_mlp_pack_bytes(K, N, C.byref(n)))
    pack = torch.empty(n.value, device=weight.device, dtype=torch.uint8)
    _check(lib().shf_mlp_pack_weights(_ptr(weight), _ptr(pack), K, N, _stream(weight)))
    return pack


class _MfmaLinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, act):
        x = x.contiguous()
        M, K = x.shape
        N = weight.shape[0]
        y = torch.empty(M, N, device=x.device, dtype=torch.float32)
        pack = None
        with torch.cuda.device(x.device):
            if PANEL_MIN_ROWS and M >= PANEL_MIN_ROWS and N <= 512 and K <= 512 and x.data_ptr() % 16 == 0:
                pack = _pack_weights(weight)
                _check(lib().shf_mlp_panel_forward(_ptr(x), _ptr(pack), _ptr(bias), _ptr(y), M, K, N, act, _stream(x)))
            else:
                _check(lib().shf_mlp_linear_forward(_ptr(x), _ptr(weight), _ptr(bias), _ptr(y), M, K, N, act, _stream(x)))
        ctx.save_for_backward(x, weight, y)
        ctx.pack = pack
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
                if ctx.pack is not None and gy.data_ptr() % 16 == 0:
                    _check(lib().shf_mlp_panel_backward_input(_ptr(gy), yp, _ptr(ctx.pack), _ptr(gx), M, K, N, _stream(x)))
                else:
                    _check(lib().shf_mlp_linear_backward_input(_ptr(gy), yp, _ptr(weight), _ptr(gx), M, K, N, _stream(x)))
            if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
                n = C.c_int64()
                _check(lib().shf_mlp_backward_weight_workspace(M, K, N, C.byref(n)))
                ws = torch.empty(n.value, device=x.device, dtype=torch.float32)
                gw, gb = torch.empty_like(weight), torch.empty(N, device=x.device, dtype=torch.float32)
                _check(lib().shf_mlp_linear_backward_weight(_ptr(gy), yp, _ptr(x), _ptr(gw), _ptr(gb), _ptr(ws), M, K, N,
                                                            _stream(x)))
        return gx, gw, gb, None


# Inference batches (no autograd) from this many rows use the row-panel kernel with the layer's KEPT pack (see
# MfmaLinear.refresh_pack): at 4096 rows it is 13 - 50 % faster per layer than the tiled kernel once the pack launch is
# not paid per call (profiles/r04_mlp_panel.md).
PANEL_INFER_MIN_ROWS = int(os.environ.get("SHIFU_AMD_MLP_PANEL_INFER_ROWS", "1024"))


class MfmaLinear(nn.Linear):
    """y = act(x W^T + b) on the MFMA kernels (CUDA fp32 2-D inputs); anything else falls back to the stock ops of
    nn.Linear -- same maths in fp32 -- so the module also works on CPU (tests, checkpoints).

    A rollout calls the layer dozens of times between two optimizer steps: `refresh_pack()` lays the weights out once
    (shf_mlp_pack_weights into a buffer the layer keeps) and marks it valid; until `invalidate_pack()` every no-grad
    forward uses it.  The flag is the CALLER's promise that the weights have not changed since (OnPolicyRunner refreshes
    at the start of every rollout -- inside the captured rollout graph, so a replay re-packs -- and invalidates before
    the update); without it every call packs for itself, which is always correct."""

    def __init__(self, in_features, out_features, elu: bool = False):
        super().__init__(in_features, out_features)
        self.elu = bool(elu)
        self._pack = None
        self._pack_valid = False

    def refresh_pack(self):
        if not self.weight.is_cuda or self.in_features > 512 or self.out_features > 512:
            return
        with torch.no_grad(), torch.cuda.device(self.weight.device):
            if self._pack is None or self._pack.device != self.weight.device:
                n = C.c_int64()
                _check(lib().shf_mlp_pack_bytes(self.in_features, self.out_features, C.byref(n)))
                self._pack = torch.empty(n.value, device=self.weight.device, dtype=torch.uint8)
            _check(lib().shf_mlp_pack_weights(_ptr(self.weight), _ptr(self._pack), self.in_features, self.out_features,
                                              _stream(self.weight)))
        self._pack_valid = True

    def invalidate_pack(self):
        self._pack_valid = False

    def forward(self, x):
        if x.is_cuda and x.dim() == 2 and x.dtype == torch.float32:
            _apply_env_precision()
            if self._pack_valid and not torch.is_grad_enabled() and x.shape[0] >= PANEL_INFER_MIN_ROWS:
                x = x.contiguous()
                if x.data_ptr() % 16 == 0:
                    y = torch.empty(x.shape[0], self.out_features, device=x.device, dtype=torch.float32)
                    with torch.cuda.device(x.device):
                        _check(lib().shf_mlp_panel_forward(_ptr(x), _ptr(self._pack), _ptr(self.bias), _ptr(y), x.shape[0],
                                                           self.in_features, self.out_features, 1 if self.elu else 0, _stream(x)))
                    return y
            return _MfmaLinearFn.apply(x, self.weight, self.bias, 1 if self.elu else 0)
        y = super().forward(x)
        return nn.functional.elu(y) if self.elu else y


def refresh_packs(module: nn.Module) -> None:
    """MfmaLinear.refresh_pack on every such layer of `module` (a no-op for other layers)."""
    for m in module.modules():
        if isinstance(m, MfmaLinear):
            m.refresh_pack()


def invalidate_packs(module: nn.Module) -> None:
    for m in module.modules():
        if isinstance(m, MfmaLinear):
            m.invalidate_pack()


def mark_packs_valid(module: nn.Module) -> None:
    """After replaying a captured graph that contains the refresh: the pack kernels ran, only the Python flag is stale."""
    for m in module.modules():
        if isinstance(m, MfmaLinear) and m._pack is not None:
            m._pack_valid = True
