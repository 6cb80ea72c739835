"""The PPO mini-batch loss and its gradient as one HIP pass (csrc/shf_mlp.hip: k_ppo_loss, include/shifu_amd.h:
shf_ppo_loss) instead of the ~100 element-wise / reduction launches autograd makes of PPO.losses' torch expressions.
Same formulas (rsl_rl's PPO.update loss block, which shifu's runner drives: shifu/runner/policy_runner.py:52-73), sums
in a fixed order; there is no CPU form of it -- PPO falls back to the torch expressions off the GPU."""
import ctypes as C

import torch

from .._lib import BackendError, lib


def _ptr(t):
    return C.c_void_p(t.data_ptr())


def _flat(t, n, name):
    if t.dtype != torch.float32 or not t.is_cuda:
        raise BackendError(f"ppo_loss: {name} must be a float32 tensor on the GPU")
    t = t.detach()
    if not t.is_contiguous():
        t = t.contiguous()
    if t.numel() != n:
        raise BackendError(f"ppo_loss: {name} has {t.numel()} elements, expected {n}")
    return t


class _PpoLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mu, std, value, actions, target_values, advantages, returns, old_logp, old_mu, old_sigma, clip,
                value_coef, entropy_coef, clipped_value):
        B, A = mu.shape
        dev = mu.device
        t = [_flat(mu, B * A, "mu"), _flat(std, A, "std"), _flat(value, B, "value"), _flat(actions, B * A, "actions"),
             _flat(target_values, B, "target_values"), _flat(advantages, B, "advantages"), _flat(returns, B, "returns"),
             _flat(old_logp, B, "old_logp"), _flat(old_mu, B * A, "old_mu"), _flat(old_sigma, B * A, "old_sigma")]
        n = C.c_int64()
        L = lib()
        if L.shf_ppo_loss_workspace(B, A, C.byref(n)) != 0:
            raise BackendError(L.shf_mlp_last_error().decode())
        out = torch.empty(5, device=dev, dtype=torch.float32)
        dmu = torch.empty(B, A, device=dev, dtype=torch.float32)
        dstd = torch.empty(A, device=dev, dtype=torch.float32)
        dvalue = torch.empty(B, device=dev, dtype=torch.float32)
        ws = torch.empty(n.value, device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            rc = L.shf_ppo_loss(*[_ptr(x) for x in t], B, A, float(clip), float(value_coef), float(entropy_coef),
                                int(bool(clipped_value)), _ptr(out), _ptr(dmu), _ptr(dstd), _ptr(dvalue), _ptr(ws),
                                C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        if rc != 0:
            raise BackendError(L.shf_mlp_last_error().decode())
        ctx.save_for_backward(dmu, dstd, dvalue)
        ctx.value_shape = value.shape
        loss, stats = out[4], out[:4]
        ctx.mark_non_differentiable(stats)
        return loss, stats

    @staticmethod
    def backward(ctx, g_loss, _g_stats):
        dmu, dstd, dvalue = ctx.saved_tensors
        return (dmu * g_loss, dstd * g_loss, (dvalue * g_loss).view(ctx.value_shape)) + (None,) * 11


def ppo_loss(mu, std, value, actions, target_values, advantages, returns, old_logp, old_mu, old_sigma, clip, value_coef,
             entropy_coef, clipped_value=True):
    """-> (loss, stats) with stats = [surrogate, value loss, entropy, KL(old || new)] (no gradient through stats)."""
    return _PpoLossFn.apply(mu, std, value, actions, target_values, advantages, returns, old_logp, old_mu, old_sigma, clip,
                            value_coef, entropy_coef, clipped_value)
