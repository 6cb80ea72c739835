"""Linear layer whose weight gradient is computed as a split-K batched GEMM.

PPO mini-batches are tall: dW = dY^T X has K = 24 576 rows against a 512 x 259 (or smaller) output, so a
single GEMM fills 18 of the MI355X's 256 CUs (hipBLASLt picks a 32x256 macro-tile without split-K here:
328 us = 20 TFLOP/s for the first layer, rocprofv3 of tools/train_a1.py).  Cutting the batch into C chunks
turns it into C independent GEMMs -- one bmm -- plus a small sum; same maths, different fp32 summation order.
"""
import torch
import torch.nn as nn


def _chunks(batch: int) -> int:
    c = 1
    while c < 64 and batch % (2 * c) == 0 and batch // (2 * c) >= 256:
        c *= 2
    return c


class _SplitKLinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        return torch.addmm(bias, x, weight.t())

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gx = gy @ weight if ctx.needs_input_grad[0] else None
        B, c = x.shape[0], _chunks(x.shape[0])
        if c > 1:
            gw = torch.bmm(gy.reshape(c, B // c, -1).transpose(1, 2), x.reshape(c, B // c, -1)).sum(0)
        else:
            gw = gy.t() @ x
        return gx, gw, gy.sum(0)


class SplitKLinear(nn.Linear):
    """Drop-in nn.Linear (same parameters / state_dict keys)."""

    def forward(self, x):
        if x.dim() == 2 and x.shape[0] >= 4096 and torch.is_grad_enabled() and self.weight.requires_grad:
            return _SplitKLinearFn.apply(x, self.weight, self.bias)
        return super().forward(x)
