"""Rollout buffer: (T, N, ...) tensors on the training device, GAE(lambda), shuffled mini-batches."""
import torch


class RolloutStorage:
    class Transition:
        def __init__(self):
            self.clear()

        def clear(self):
            self.observations = self.critic_observations = self.actions = self.rewards = self.dones = None
            self.values = self.actions_log_prob = self.action_mean = self.action_sigma = None

    def __init__(self, num_envs, num_transitions_per_env, obs_shape, privileged_obs_shape, actions_shape, device="cpu"):
        self.device = device
        T, N = num_transitions_per_env, num_envs
        self.num_envs, self.num_transitions_per_env = N, T
        z = lambda *s: torch.zeros(T, N, *s, device=device)
        self.observations = z(*obs_shape)
        self.privileged_observations = z(*privileged_obs_shape) if privileged_obs_shape[0] is not None else None
        self.actions = z(*actions_shape)
        self.rewards, self.values, self.returns, self.advantages, self.actions_log_prob = z(1), z(1), z(1), z(1), z(1)
        self.dones = torch.zeros(T, N, 1, device=device, dtype=torch.uint8)
        self.mu, self.sigma = z(*actions_shape), z(*actions_shape)
        self.step = 0

    def add_transitions(self, t: "RolloutStorage.Transition"):
        if self.step >= self.num_transitions_per_env:
            raise AssertionError("Rollout buffer overflow")
        k = self.step
        pairs = [(self.observations[k], t.observations), (self.actions[k], t.actions), (self.rewards[k], t.rewards),
                 (self.values[k], t.values), (self.actions_log_prob[k], t.actions_log_prob), (self.mu[k], t.action_mean),
                 (self.sigma[k], t.action_sigma)]
        if self.privileged_observations is not None:
            pairs.append((self.privileged_observations[k], t.critic_observations))
        self.dones[k].copy_(t.dones.view(-1, 1))                 # (a dtype conversion: stays a torch copy)
        if self.observations.is_cuda and all(src.dtype == torch.float32 and src.is_cuda and src.is_contiguous() and src.numel() == dst.numel()
                                              for dst, src in pairs):
            self._copy_many(pairs)                               # one launch (shf_copy_many) instead of one per tensor
        else:
            for dst, src in pairs:
                dst.copy_(src.view(dst.shape))
        self.step += 1

    def _copy_many(self, pairs):
        import ctypes as C
        from .._lib import BackendError, lib
        n = len(pairs)
        src = (C.c_void_p * n)(*[s.data_ptr() for _, s in pairs])
        dst = (C.c_void_p * n)(*[d.data_ptr() for d, _ in pairs])
        nbytes = (C.c_int64 * n)(*[4 * d.numel() for d, _ in pairs])
        dev = self.observations.device
        with torch.cuda.device(dev):
            rc = lib().shf_copy_many(src, dst, nbytes, n, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        if rc != 0:
            raise BackendError(lib().shf_mlp_last_error().decode())

    def clear(self):
        self.step = 0

    def compute_returns(self, last_values, gamma, lam):
        """delta_t = r_t + gamma (1-d_t) V_{t+1} - V_t;  A_t = delta_t + gamma lam (1-d_t) A_{t+1};  R_t = A_t + V_t;
        advantages are then standardised over the whole buffer."""
        if self.returns.is_cuda:
            self._gae_kernel(last_values, gamma, lam)
        else:
            adv = 0
            for k in reversed(range(self.num_transitions_per_env)):
                nxt = last_values if k == self.num_transitions_per_env - 1 else self.values[k + 1]
                live = 1.0 - self.dones[k].float()
                delta = self.rewards[k] + live * gamma * nxt - self.values[k]
                adv = delta + live * gamma * lam * adv
                self.returns[k] = adv + self.values[k]
        # in place: a captured update graph reads this buffer at a fixed address
        adv = self.returns - self.values
        self.advantages.copy_((adv - adv.mean()) / (adv.std() + 1e-8))

    def _gae_kernel(self, last_values, gamma, lam):
        """The loop above as one launch (shf_gae, csrc/shf_mlp.hip): the same float32 operations in the same order, so
        the same bits (tests/test_gpu_mlp.py) -- 1 launch instead of ~9 per transition."""
        import ctypes as C
        from .._lib import BackendError, lib
        T, N = self.num_transitions_per_env, self.num_envs
        last = last_values.detach().reshape(-1).contiguous()
        if last.numel() != N or last.dtype != torch.float32 or not all(t.is_contiguous() for t in (self.rewards, self.values, self.dones, self.returns)):
            raise BackendError("shf_gae: rollout buffers must be contiguous float32 (T, N, 1) and last_values (N, 1)")
        p = lambda t: C.c_void_p(t.data_ptr())
        dev = self.returns.device
        with torch.cuda.device(dev):
            rc = lib().shf_gae(p(self.rewards), p(self.values), p(self.dones), p(last), T, N, float(gamma), float(lam), p(self.returns),
                               C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        if rc != 0:
            raise BackendError(lib().shf_mlp_last_error().decode())

    def get_statistics(self):
        done = self.dones.clone()
        done[-1] = 1
        flat = done.permute(1, 0, 2).reshape(-1, 1)
        idx = torch.cat((flat.new_tensor([-1], dtype=torch.int64), flat.nonzero(as_tuple=False)[:, 0]))
        lengths = idx[1:] - idx[:-1]
        return lengths.float().mean(), self.rewards.mean()

    def mini_batch(self, idx):
        """The rollout rows `idx` (flat (t, env) indices) as
        (obs, critic_obs, actions, values, advantages, returns, log_prob, mu, sigma)."""
        srcs = [self.observations, self.actions, self.values, self.advantages, self.returns, self.actions_log_prob, self.mu, self.sigma]
        if self.privileged_observations is not None:
            srcs.append(self.privileged_observations)
        if self.observations.is_cuda and idx.dtype == torch.int64 and idx.is_contiguous():
            out = self._gather_rows(srcs, idx)                       # one launch (shf_gather_rows) instead of one per tensor
        else:
            out = [t.flatten(0, 1)[idx] for t in srcs]
        obs, actions, values, adv, ret, logp, mu, sigma = out[:8]
        cobs = out[8] if self.privileged_observations is not None else obs
        return (obs, cobs, actions, values, adv, ret, logp, mu, sigma)

    def _gather_rows(self, srcs, idx):
        import ctypes as C
        from .._lib import BackendError, lib
        n, rows = len(srcs), idx.numel()
        flat = [t.flatten(0, 1) for t in srcs]
        out = [torch.empty((rows,) + tuple(t.shape[1:]), device=t.device, dtype=t.dtype) for t in flat]
        src = (C.c_void_p * n)(*[t.data_ptr() for t in flat])
        dst = (C.c_void_p * n)(*[t.data_ptr() for t in out])
        rb = (C.c_int32 * n)(*[t[0].numel() * t.element_size() for t in flat])
        dev = idx.device
        with torch.cuda.device(dev):
            rc = lib().shf_gather_rows(src, dst, rb, n, C.c_void_p(idx.data_ptr()), rows, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        if rc != 0:
            raise BackendError(lib().shf_mlp_last_error().decode())
        return out
