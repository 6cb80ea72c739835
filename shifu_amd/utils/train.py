"""HistoryRecorder (reference shifu/utils/train.py:4-35): an (N, A, H) ring with the
newest sample at index 0.  `flatten()` lays it out history-major, newest first, as
(N, H*A) -- the 36 action-history observation columns of A1Conditional.  The fused HIP
step keeps the same (N, A, H) buffer (SHF_A1_HISTORY)."""
import torch


class HistoryRecorder:
    def __init__(self, shape, num_history, device):
        assert isinstance(num_history, int) and num_history > 0, "num_history must be a positive int"
        self.dshape = shape
        self.num_history = num_history
        self.device = device
        self.history_buf = torch.zeros(*shape, num_history, device=device)

    def add(self, x):
        # (anything else -- a broadcastable or differently typed x, as `history_buf[..., 0] = x` accepts -- takes the torch lines)
        if self.history_buf.is_cuda and torch.is_tensor(x) and x.is_cuda and x.dtype == torch.float32 \
                and self.history_buf.is_contiguous() and tuple(x.shape) == tuple(self.history_buf.shape[:-1]):
            from shifu_amd import glue              # the two statements below as one launch (shf_history_add)
            glue.history_add(self.history_buf, x)
            return
        # shift towards the past, then store the newest at slot 0
        self.history_buf[..., 1:] = self.history_buf[..., :-1].clone()
        self.history_buf[..., 0] = x

    def reset_idx(self, idx):
        if self.history_buf.is_cuda and torch.is_tensor(idx) and idx.is_cuda and idx.dtype == torch.int64 \
                and self.history_buf.is_contiguous():
            from shifu_amd import glue
            glue.rows_fill_indexed(self.history_buf, idx, 0.0)
            return
        self.history_buf.index_fill_(0, idx, 0.)

    def get_last(self, idx):
        """idx 0 = newest stored sample, 1 = the one before, ..."""
        return self.history_buf[..., idx]

    def flatten(self):
        nd = self.history_buf.dim()
        p = self.history_buf.permute(0, *reversed(range(1, nd)))
        return p.reshape(*self.dshape[:-1], self.dshape[-1] * self.num_history)
