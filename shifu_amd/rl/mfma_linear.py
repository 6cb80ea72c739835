"""nn.Linear (+ ELU) whose forward and backward run on the hand-written MFMA kernels of csrc/shf_mlp.hip.

Operands stay the trainer's fp32 tensors; the kernels convert tiles to bf16 on the fly and accumulate in fp32
(v_mfma_f32_32x32x16_bf16), with bias, ELU and the ELU derivative fused into the GEMMs' load / store paths -- one launch
per layer forward, two per layer backward (+ the deterministic split-M reduction of dW).  `MfmaLinear` has nn.Linear's
parameters and state_dict keys, so checkpoints are interchangeable with the stock layer.

Precision (`set_precision`, env SHIFU_AMD_MFMA_PRECISION): "bf16x3" (default) splits every operand value into a bf16
head and tail and accumulates three MFMAs per tile pair -- products good to 2^-16, outputs within 1e-4 of the fp32 torch
reference's scale (tests/test_gpu_mlp.py); "bf16" rounds operands once (2^-9 relative, 2e-2 of scale), one MFMA.  With
"bf16" two of three 3000-iteration A1 runs lost return late in training (the action-noise std grew faster than with fp32
layers); "bf16x3" costs a few per cent of the layer time, the layers being HBM-bound.  "bf16x3-w1" keeps
head + tail operands for forward and input gradient and rounds the weight gradient's operands once (its sum over the batch
rows averages the rounding): same outcomes as "bf16x3" over 30 seeds, learn -6 % (profiles/r04_train.md)."""
import os
import ctypes as C

import torch
import torch.nn as nn

from .._lib import BackendError, lib


PRECISIONS = {"bf16": 0, "bf16x3": 1, "bf16x3-w1": 2}


def set_precision(mode: str) -> None:
    """Process-wide operand precision of the MFMA layers: "bf16x3", "bf16x3-w1" (weight gradient from once-rounded operands)
    or "bf16" (include/shifu_amd.h SHF_MLP_*)."""
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

    def _load_from_state_dict(self, *args, **kwargs):       # a checkpoint load changes the weights behind a kept pack
        self._pack_valid = False
        return super()._load_from_state_dict(*args, **kwargs)

    def _apply(self, fn, *args, **kwargs):                  # .to(device) / .float() ...: the pack does not follow
        self._pack_valid = False
        return super()._apply(fn, *args, **kwargs)

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


MAX_CHAIN = 6      # SHF_MLP_MAX_CHAIN


class _ShfMlpChain(C.Structure):        # include/shifu_amd.h ShfMlpChain
    _fields_ = [("nlayers", C.c_int32), ("dims", C.c_int32 * (MAX_CHAIN + 1)), ("pack", C.c_void_p * MAX_CHAIN),
                ("bias", C.c_void_p * MAX_CHAIN), ("act", C.c_int32 * MAX_CHAIN), ("y", C.c_void_p * MAX_CHAIN)]


def _chain_call(x, layers, packs, ys):
    """shf_mlp_chain_forward over `layers` (MfmaLinear), their packs and per-layer output tensors (None = not kept)."""
    c = _ShfMlpChain()
    c.nlayers = len(layers)
    c.dims[0] = layers[0].in_features
    for i, (m, pk, y) in enumerate(zip(layers, packs, ys)):
        c.dims[i + 1] = m.out_features
        c.pack[i] = pk.data_ptr()
        c.bias[i] = m.bias.data_ptr() if m.bias is not None else None
        c.act[i] = 1 if m.elu else 0
        c.y[i] = y.data_ptr() if y is not None else None
    with torch.cuda.device(x.device):
        _check(lib().shf_mlp_chain_forward(_ptr(x), x.shape[0], C.byref(c), _stream(x)))


class _MfmaChainFn(torch.autograd.Function):
    """The whole MLP as one autograd node: forward = ONE launch (k_mlp_chain: the activations stay in LDS between layers
    and are written once for the backward pass), backward = the per-layer input-gradient / weight-gradient kernels in
    reverse order.  Same values as the chain of _MfmaLinearFn nodes, bit for bit (tests/test_gpu_mlp.py)."""

    @staticmethod
    def forward(ctx, x, acts, *params):
        x = x.contiguous()
        n = len(params) // 2
        ws, bs = params[0::2], params[1::2]
        M = x.shape[0]
        packs = [_pack_weights(w) for w in ws]
        ys = [torch.empty(M, w.shape[0], device=x.device, dtype=torch.float32) for w in ws]
        c = _ShfMlpChain()
        c.nlayers = n
        c.dims[0] = ws[0].shape[1]
        for i in range(n):
            c.dims[i + 1] = ws[i].shape[0]
            c.pack[i] = packs[i].data_ptr(); c.bias[i] = bs[i].data_ptr(); c.act[i] = acts[i]; c.y[i] = ys[i].data_ptr()
        with torch.cuda.device(x.device):
            _check(lib().shf_mlp_chain_forward(_ptr(x), M, C.byref(c), _stream(x)))
        ctx.save_for_backward(x, *ws, *ys)
        ctx.packs, ctx.acts, ctx.n = packs, acts, n
        return ys[-1]

    @staticmethod
    def backward(ctx, gy):
        n = ctx.n
        saved = ctx.saved_tensors
        x, ws, ys = saved[0], saved[1:1 + n], saved[1 + n:]
        g = gy.contiguous()
        grads = [None] * (2 * n)
        L = lib()
        with torch.cuda.device(x.device):
            for i in range(n - 1, -1, -1):
                inp = x if i == 0 else ys[i - 1]
                M, K = inp.shape
                N = ws[i].shape[0]
                yp = _ptr(ys[i]) if ctx.acts[i] == 1 else None
                gx = None
                if i > 0 or ctx.needs_input_grad[0]:
                    gx = torch.empty_like(inp)
                    if PANEL_MIN_ROWS and M >= PANEL_MIN_ROWS and g.data_ptr() % 16 == 0:
                        _check(L.shf_mlp_panel_backward_input(_ptr(g), yp, _ptr(ctx.packs[i]), _ptr(gx), M, K, N, _stream(x)))
                    else:
                        _check(L.shf_mlp_linear_backward_input(_ptr(g), yp, _ptr(ws[i]), _ptr(gx), M, K, N, _stream(x)))
                if ctx.needs_input_grad[2 + 2 * i] or ctx.needs_input_grad[3 + 2 * i]:
                    nw = C.c_int64()
                    _check(L.shf_mlp_backward_weight_workspace(M, K, N, C.byref(nw)))
                    wsp = torch.empty(nw.value, device=x.device, dtype=torch.float32)
                    gw, gb = torch.empty_like(ws[i]), torch.empty(N, device=x.device, dtype=torch.float32)
                    _check(L.shf_mlp_linear_backward_weight(_ptr(g), yp, _ptr(inp), _ptr(gw), _ptr(gb), _ptr(wsp), M, K, N,
                                                            _stream(x)))
                    grads[2 * i], grads[2 * i + 1] = gw, gb
                g = gx
        return (g if ctx.needs_input_grad[0] else None, None, *grads)


# Rows from which an MfmaMLP runs a no-grad forward (kept packs valid) as ONE chained launch (SHIFU_AMD_MLP_CHAIN_ROWS;
# 0 = never): the rollout's 4096-row inference passes take 29 us per network instead of 41 (profiles/r04_mlp_panel.md).
CHAIN_MIN_ROWS = int(os.environ.get("SHIFU_AMD_MLP_CHAIN_ROWS", "1024"))
# The autograd form (_MfmaChainFn: forward chained, every activation kept; backward layer by layer) is bit-identical too but
# not faster than the row-panel kernels at 24 576 rows with bf16x3 operands (107 vs 103 us; 66 vs 75 with bf16), and it
# fills the LDS, which costs the update's two streams their overlap: off unless SHIFU_AMD_MLP_CHAIN_TRAIN=1.
CHAIN_TRAIN = os.environ.get("SHIFU_AMD_MLP_CHAIN_TRAIN", "0") == "1"


class MfmaMLP(nn.Sequential):
    """nn.Sequential of MfmaLinear layers (nn.Identity between them keeps rsl_rl's parameter names) whose forward is one
    chained launch when every layer is an MfmaLinear of at most 512 units on a CUDA fp32 batch; otherwise -- CPU tensors,
    small batches, wider layers -- the layers run one by one as in any nn.Sequential."""

    def _chain_layers(self):
        key = (len(self), get_precision())        # the fit depends on the widths and on the precision mode's LDS planes
        cached = getattr(self, "_chain_cache", None)
        if cached is None or cached[0] != key:
            self._chain_cache = (key, self._chain_layers_uncached())
        return self._chain_cache[1]

    def _chain_layers_uncached(self):
        ls = [m for m in self if not isinstance(m, nn.Identity)]
        if not ls or len(ls) > MAX_CHAIN or not all(isinstance(m, MfmaLinear) and m.bias is not None for m in ls):
            return None
        if any(m.in_features > 512 or m.out_features > 512 for m in ls):
            return None
        c = _ShfMlpChain()
        c.nlayers = len(ls)
        c.dims[0] = ls[0].in_features
        for i, m in enumerate(ls):
            c.dims[i + 1] = m.out_features
        return ls if lib().shf_mlp_chain_fits(C.byref(c)) else None      # (depends on the precision mode: asked per call)

    def forward(self, x):
        # the cheap tensor checks first: a CPU / small / non-fp32 batch takes the layers one by one (stock ops on CPU) and
        # must not touch the native library, which _chain_layers() does (get_precision, shf_mlp_chain_fits)
        if not (CHAIN_MIN_ROWS and x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and x.shape[0] >= CHAIN_MIN_ROWS):
            return super().forward(x)
        ls = self._chain_layers()
        if ls is None:
            return super().forward(x)
        _apply_env_precision()
        x = x.contiguous()
        if x.data_ptr() % 16 != 0:
            return super().forward(x)
        if not torch.is_grad_enabled():
            if all(m._pack_valid for m in ls):          # inference between refresh_packs and invalidate_packs
                y = torch.empty(x.shape[0], ls[-1].out_features, device=x.device, dtype=torch.float32)
                _chain_call(x, ls, [m._pack for m in ls], [None] * (len(ls) - 1) + [y])
                return y
            return super().forward(x)
        if not CHAIN_TRAIN:
            return super().forward(x)
        params = []
        for m in ls:
            params += [m.weight, m.bias]
        return _MfmaChainFn.apply(x, tuple(1 if m.elu else 0 for m in ls), *params)


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
