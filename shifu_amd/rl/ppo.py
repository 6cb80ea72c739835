"""PPO update (PPOConfig.algorithm, shifu/configs/policy_config.py:18-32)."""
import os
import torch
import torch.nn as nn
import torch.optim as optim

from ..parallel import average_, average_gradients, world_size, collectives_on
from .storage import RolloutStorage


class PPO:
    def __init__(self, actor_critic, num_learning_epochs=1, num_mini_batches=1, clip_param=0.2, gamma=0.998, lam=0.95,
                 value_loss_coef=1.0, entropy_coef=0.0, learning_rate=1e-3, max_grad_norm=1.0, use_clipped_value_loss=True,
                 schedule="fixed", desired_kl=0.01, device="cpu", graph_update=False, fused_loss=True):
        self.device = device
        # graph_update: replay one captured hipGraph per mini-batch step (gather, losses, backward, clip, Adam, lr
        # schedule: ~280 launches) instead of launching it eagerly; bit-identical to the eager update with either layer
        # backend (tests/test_gpu_mlp.py), given the host wait after each replay (update()).  One rank only.
        self.graph_update = bool(graph_update)
        self._upd_graph = self._upd_idx = self._upd_sums = None
        # the critic's forward / backward on a second stream (bit-identical results, learn -25 % with the MFMA layers,
        # profiles/r02_mlp_probe.md); SHIFU_AMD_TWO_STREAM_UPDATE=0 switches it off
        # (streams are created on and asked of THIS trainer's device, which need not be the current one)
        self._side = (torch.cuda.Stream(device=torch.device(device))
                      if (torch.device(device).type == "cuda" and os.environ.get("SHIFU_AMD_TWO_STREAM_UPDATE", "1") == "1") else None)
        self._updates_done = 0
        # what follows each replay of the captured update: 'wait' (device-wide host wait, the only mode that reproduces the
        # eager update on this stack; update()) or one of the experiment modes of profiles/r02_mlp_probe.md -- read once,
        # and anything but 'wait' is announced because it is known to drift
        self._replay_mode = os.environ.get("SHIFU_AMD_REPLAY_MODE", "wait")
        if self._replay_mode != "wait" and self.graph_update:
            import warnings
            warnings.warn(f"SHIFU_AMD_REPLAY_MODE={self._replay_mode}: the captured PPO update is NOT equivalent to the eager one "
                          "in this mode (profiles/r02_mlp_probe.md); experiments only")
        self.desired_kl, self.schedule = desired_kl, schedule
        self.actor_critic = actor_critic.to(device)
        # fused_loss: the loss block and its gradient as one HIP pass (rl/fused_loss.py) instead of ~100 small autograd
        # launches: learn 24.5 -> 17.5 ms per iteration on the A1 schedule.  Same formulas -- gradients equal to autograd's to
        # 2e-5 (tests/test_gpu_mlp.py) -- sums in another (fixed) order.  The default on a GPU since round 5: 44 seeds of the
        # 3000-iteration schedule per arm (38 under the compliant contact law, rounds 2-4; 6 under PGS, round 5) ended with 41
        # walking policies against 43 for the torch expressions, mean return 529 vs 516 under PGS (profiles/r05_train.md).
        # fused_loss=False / SHIFU_AMD_FUSED_PPO_LOSS=0 / tools/train_a1.py --torch-loss select the torch expressions (always
        # used for CPU tensors).
        want = os.environ.get("SHIFU_AMD_FUSED_PPO_LOSS")
        self.fused_loss = torch.device(device).type == "cuda" and (want == "1" if want in ("0", "1") else bool(fused_loss))
        self.storage = None
        # The learning rate is a device tensor shared with the optimizer: the adaptive schedule needs no host
        # round trip per mini-batch.
        # One fused multi-tensor Adam kernel on the GPU (18 parameter tensors), the stock loop on CPU.
        cuda = torch.device(device).type == "cuda"
        self.lr = torch.tensor(float(learning_rate), dtype=torch.float32, device=device)
        self.optimizer = optim.Adam(self.actor_critic.parameters(), lr=self.lr, fused=cuda, capturable=cuda)
        self.transition = RolloutStorage.Transition()
        self.clip_param, self.num_learning_epochs, self.num_mini_batches = clip_param, num_learning_epochs, num_mini_batches
        self.value_loss_coef, self.entropy_coef = value_loss_coef, entropy_coef
        self.gamma, self.lam, self.max_grad_norm = gamma, lam, max_grad_norm
        self.use_clipped_value_loss = use_clipped_value_loss

    def init_storage(self, num_envs, num_transitions_per_env, actor_obs_shape, critic_obs_shape, action_shape):
        self.storage = RolloutStorage(num_envs, num_transitions_per_env, actor_obs_shape, critic_obs_shape, action_shape,
                                      self.device)

    def test_mode(self):
        self.actor_critic.eval()

    def train_mode(self):
        self.actor_critic.train()

    def act(self, obs, critic_obs):
        t = self.transition
        t.actions = self.actor_critic.act(obs).detach()
        t.values = self.actor_critic.evaluate(critic_obs).detach()
        t.actions_log_prob = self.actor_critic.get_actions_log_prob(t.actions).detach()
        t.action_mean = self.actor_critic.action_mean.detach()
        t.action_sigma = self.actor_critic.action_std.detach()
        t.observations = obs                # copied into the rollout buffer by add_transitions
        t.critic_observations = critic_obs
        return t.actions

    def process_env_step(self, rewards, dones, infos):
        t = self.transition
        t.rewards = rewards.clone()
        t.dones = dones
        # an episode cut by the time limit is not a failure: bootstrap its tail with V(s_t)
        if "time_outs" in infos:
            t.rewards += self.gamma * torch.squeeze(t.values * infos["time_outs"].unsqueeze(1).to(self.device), 1)
        self.storage.add_transitions(t)
        t.clear()
        self.actor_critic.reset(dones)

    def compute_returns(self, last_critic_obs):
        last_values = self.actor_critic.evaluate(last_critic_obs).detach()
        self.storage.compute_returns(last_values, self.gamma, self.lam)

    def losses(self, obs, cobs, actions, target_values, advantages, returns, old_logp, old_mu, old_sigma):
        """Loss terms of one mini-batch under the current parameters (also what tests/test_rl.py checks)."""
        ac = self.actor_critic
        if self.fused_loss:
            return self._fused_losses(obs, cobs, actions, target_values, advantages, returns, old_logp, old_mu, old_sigma)
        if self._side is not None:
            # actor and critic are independent networks: the critic's forward (and, through autograd's stream bookkeeping,
            # its backward) runs on a second stream, so the small layers of one fill the CUs the other leaves idle
            cur = torch.cuda.current_stream(torch.device(self.device))
            self._side.wait_stream(cur)
            with torch.cuda.stream(self._side):
                value = ac.evaluate(cobs)
            value.record_stream(cur)             # consumed on the main stream after the join
            ac.update_distribution(obs)
            logp = ac.get_actions_log_prob(actions)
            cur.wait_stream(self._side)
        else:
            ac.update_distribution(obs)
            logp = ac.get_actions_log_prob(actions)
            value = ac.evaluate(cobs)
        mu, sigma, entropy = ac.action_mean, ac.action_std, ac.entropy
        with torch.no_grad():
            # KL(old || new) of diagonal Gaussians, averaged over the mini-batch
            kl = torch.sum(torch.log(sigma / old_sigma + 1.e-5)
                           + (old_sigma.square() + (old_mu - mu).square()) / (2.0 * sigma.square()) - 0.5, dim=-1).mean()
        ratio = torch.exp(logp - torch.squeeze(old_logp))
        adv = torch.squeeze(advantages)
        surrogate = -adv * ratio
        surrogate_clipped = -adv * torch.clamp(ratio, 1.0 - self.clip_param, 1.0 + self.clip_param)
        surrogate_loss = torch.max(surrogate, surrogate_clipped).mean()
        if self.use_clipped_value_loss:
            value_clipped = target_values + (value - target_values).clamp(-self.clip_param, self.clip_param)
            value_loss = torch.max((value - returns).square(), (value_clipped - returns).square()).mean()
        else:
            value_loss = (returns - value).square().mean()
        ent = entropy.mean()
        return {"surrogate": surrogate_loss, "value": value_loss, "entropy": ent, "kl": kl,
                "loss": surrogate_loss + self.value_loss_coef * value_loss - self.entropy_coef * ent}

    def _fused_losses(self, obs, cobs, actions, target_values, advantages, returns, old_logp, old_mu, old_sigma):
        """losses() with the loss block as one kernel: the two networks as before (critic on the second stream), then
        shf_ppo_loss on (action means, std, values)."""
        from .fused_loss import ppo_loss
        ac = self.actor_critic
        if self._side is not None:
            cur = torch.cuda.current_stream(torch.device(self.device))
            self._side.wait_stream(cur)
            with torch.cuda.stream(self._side):
                value = ac.evaluate(cobs)
            value.record_stream(cur)
            mu = ac.actor(obs)
            cur.wait_stream(self._side)
        else:
            mu, value = ac.actor(obs), ac.evaluate(cobs)
        loss, stats = ppo_loss(mu, ac.std, value, actions, target_values, advantages, returns, old_logp, old_mu, old_sigma,
                               self.clip_param, self.value_loss_coef, self.entropy_coef, self.use_clipped_value_loss)
        return {"surrogate": stats[0], "value": stats[1], "entropy": stats[2], "kl": stats[3], "loss": loss}

    @property
    def learning_rate(self) -> float:
        return float(self.lr)

    def relink_learning_rate(self):
        """After optimizer.load_state_dict the param groups hold the checkpoint's lr value AND its implementation flags
        (a stock rsl_rl checkpoint: python-float lr, no fused / capturable; one of ours saved on the GPU: both set):
        share one lr tensor again and put this trainer's own flags for the current device back, so that a checkpoint from
        either side resumes on either device without falling into the per-parameter loop or the capturable assertion."""
        self.lr.fill_(float(self.optimizer.param_groups[0]["lr"]))
        # a captured update holds the ADDRESSES of the optimizer state it was captured with; load_state_dict has just
        # replaced those tensors (the old ones went back to the allocator): drop the graph, run the next update eagerly
        # and capture again against the new state
        self._upd_graph = self._upd_idx = self._upd_sums = None
        self._updates_done = 0
        cuda = torch.device(self.device).type == "cuda"
        for g in self.optimizer.param_groups:
            g["lr"] = self.lr
            g["fused"], g["capturable"], g["foreach"] = cuda, cuda, None
        for st in self.optimizer.state.values():
            if "step" in st and torch.is_tensor(st["step"]):
                st["step"] = st["step"].to(self.lr.device if cuda else "cpu", dtype=torch.float32)

    def optimizer_state_dict(self):
        """optimizer.state_dict() with the learning rate as a python float: interchangeable with the stock trainer's."""
        sd = self.optimizer.state_dict()
        for g in sd["param_groups"]:
            g["lr"] = float(g["lr"])
        return sd

    def adapt_learning_rate(self, kl_mean):
        """schedule='adaptive': keep the policy step near desired_kl (x1.5 / /1.5, clamped to [1e-5, 1e-2]).
        `kl_mean` may be a device scalar; the decision is taken on the device."""
        kl = torch.as_tensor(kl_mean, dtype=torch.float32, device=self.lr.device)
        lr = self.lr
        if lr.is_cuda:
            # the expressions below as one launch (shf_adapt_lr), same float32 arithmetic (tests/test_gpu_mlp.py)
            import ctypes as C
            from .._lib import BackendError, lib
            f32 = lambda x: float(torch.tensor(x, dtype=torch.float32))
            with torch.cuda.device(lr.device):
                rc = lib().shf_adapt_lr(C.c_void_p(kl.data_ptr()), C.c_void_p(lr.data_ptr()), f32(self.desired_kl * 2.0),
                                        f32(self.desired_kl / 2.0), f32(1.0) / f32(1.5), 1.5, 1e-5, 1e-2,
                                        C.c_void_p(torch.cuda.current_stream(lr.device).cuda_stream))
            if rc != 0:
                raise BackendError(lib().shf_mlp_last_error().decode())
            return
        down, up = torch.clamp(lr / 1.5, min=1e-5), torch.clamp(lr * 1.5, max=1e-2)
        lr.copy_(torch.where(kl > self.desired_kl * 2.0, down,
                             torch.where((kl > 0.0) & (kl < self.desired_kl / 2.0), up, lr)))

    def _minibatch_step(self, batch, sums):
        """One optimizer step on one mini-batch (static shapes, no host sync)."""
        ac = self.actor_critic
        obs, cobs, actions, target_values, advantages, returns, old_logp, old_mu, old_sigma = batch
        L = self.losses(obs, cobs, actions, target_values, advantages, returns, old_logp, old_mu, old_sigma)
        if self.desired_kl is not None and self.schedule == "adaptive":
            # the KL is averaged over ranks first so that every rank takes the same decision
            self.adapt_learning_rate(average_(L["kl"].detach().clone()))
        L["loss"].backward()
        if collectives_on():
            average_gradients(ac.parameters())
        nn.utils.clip_grad_norm_(ac.parameters(), self.max_grad_norm)
        self.optimizer.step()
        sums[0] += L["value"].detach()
        sums[1] += L["surrogate"].detach()

    def _graph_update_ok(self) -> bool:
        return self.graph_update and torch.device(self.device).type == "cuda" and not collectives_on()

    def _capture_update(self, mb):
        self._upd_idx = torch.zeros(mb, dtype=torch.long, device=self.device)
        self._upd_sums = torch.zeros(2, device=self.device)
        self.optimizer.zero_grad(set_to_none=True)
        g = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            self.optimizer.zero_grad(set_to_none=True)
            self._minibatch_step(self.storage.mini_batch(self._upd_idx), self._upd_sums)
        self._upd_graph = g

    def update(self):
        st = self.storage
        B = st.num_envs * st.num_transitions_per_env
        mb = B // self.num_mini_batches
        perm = torch.randperm(self.num_mini_batches * mb, device=self.device)
        # the first update runs eagerly (lazy initialisations, optimizer state); the second is captured
        use_graph = self._graph_update_ok() and self._updates_done >= 1
        if use_graph and self._upd_graph is None:
            self._capture_update(mb)
        sums = self._upd_sums.zero_() if use_graph else torch.zeros(2, device=self.device)   # loss sums stay on the device
        for _ in range(self.num_learning_epochs):
            for i in range(self.num_mini_batches):
                if use_graph:
                    if os.environ.get("SHIFU_AMD_IDX_COPY", "memcpy") == "kernel":
                        torch.add(perm[i * mb:(i + 1) * mb], 0, out=self._upd_idx)      # a compute kernel, not a DMA copy
                    else:
                        self._upd_idx.copy_(perm[i * mb:(i + 1) * mb])
                    self._upd_graph.replay()
                    # Replayed back to back, the captured update is NOT equivalent to the eager one on this stack
                    # (ROCm 7.2 / torch 2.10): parameters drift from the first iterations on, differently from run to
                    # run, with the MFMA layers and with the stock library GEMMs alike (round 1's "corrupted captured
                    # update").  A device-wide wait after every replay makes it identical to the bit
                    # (tests/test_gpu_mlp.py; the 3000-iteration schedule reproduces the eager run's final return digit
                    # for digit); waiting on an event recorded on the launch stream behind the replay does NOT -- so part
                    # of this graph's work is not ordered before later work on the launch stream.  Minimal graphs do not
                    # show it (tools/hipgraph_order_probe.py).  SHIFU_AMD_REPLAY_MODE=none|event|kernel reproduces the
                    # experiments of profiles/r02_mlp_probe.md.  The wait costs no throughput: the host has nothing else
                    # to do here.
                    mode = self._replay_mode
                    if mode == "wait":
                        torch.cuda.synchronize()
                    elif mode == "event":
                        ev = torch.cuda.Event(); ev.record(); ev.synchronize()
                    elif mode == "kernel":
                        self._upd_sums.add_(0.0)
                else:
                    self.optimizer.zero_grad(set_to_none=True)
                    self._minibatch_step(st.mini_batch(perm[i * mb:(i + 1) * mb]), sums)
        n = self.num_learning_epochs * self.num_mini_batches
        st.clear()
        self._updates_done += 1
        mean_value_loss, mean_surrogate_loss = (sums / n).tolist()
        return mean_value_loss, mean_surrogate_loss
