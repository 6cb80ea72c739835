"""ShifuVecEnv (reference shifu/gym/env.py:18-193): the task layer users subclass.

User hooks keep their reference signatures: build_reward_functions() -> list of bound
methods, compute_observations(), compute_termination(), optional episode_log(env_ids).
The upward contract is the rsl_rl.env.VecEnv duck type (attributes num_envs, num_obs,
..., step -> (obs, priv_obs, rew, dones, infos)); rsl_rl itself is not imported, so the
class works with or without it.

This is the hook-compatible path: physics runs on the HIP kernels, the hooks run as
ordinary torch code, and `reset_buf.nonzero()` costs the same host sync as in the
reference (env.py:101,115).  The sync-free, single-launch path for the A1 task is
shifu_amd.gym.a1_fused.FusedA1Env.
"""
import typing

import numpy as np
import torch

from shifu_amd._lib import BackendError
from shifu_amd.configs import TerrainEnvConfig
from shifu_amd.utils.torch_utils import free_tensor_attrs
from shifu_amd.utils.train import HistoryRecorder

from .isaac_gym import IsaacGymEnv, TerrainGymEnv


class ShifuVecEnv:
    def __init__(self, cfg):
        self.cfg = cfg
        self.isg_env = TerrainGymEnv(cfg) if isinstance(cfg, TerrainEnvConfig) else IsaacGymEnv(cfg)
        n, dev = self.isg_env.num_envs, self.isg_env.device
        self.num_envs, self.device = n, dev
        self.num_obs = cfg.num_obs
        self.num_privileged_obs = cfg.num_privileged_obs
        self.num_actions = cfg.num_actions
        self.clip_obs = cfg.normalization.clip_observations
        self.clip_actions = cfg.normalization.clip_actions
        self.max_episode_length_s = cfg.episode_length_s
        self.max_episode_length = np.ceil(self.max_episode_length_s / self.isg_env.dt)

        self.actions = torch.zeros(n, self.num_actions, device=dev, dtype=torch.float, requires_grad=False)
        self.obs_buf = torch.zeros(n, self.num_obs, device=dev, dtype=torch.float)
        self.rew_buf = torch.zeros(n, device=dev, dtype=torch.float)
        self.reset_buf = torch.ones(n, device=dev, dtype=torch.long)      # int64 ones until the first step (Q7)
        self.episode_length_buf = torch.zeros(n, device=dev, dtype=torch.long)
        self.time_out_buf = torch.zeros(n, device=dev, dtype=torch.bool)
        self.extras = {}
        if cfg.num_actions_history:
            self.actions_recorder = HistoryRecorder(self.actions.shape, cfg.num_actions_history, device=dev)
        self.privileged_obs_buf = None if self.num_privileged_obs is None else \
            torch.zeros(n, self.num_privileged_obs, device=dev, dtype=torch.float)
        self.common_step_counter = 0
        self.reward_functions = self.build_reward_functions()
        self._prepare_reward_functions()

    # -- hooks -------------------------------------------------------------------
    def build_reward_functions(self) -> typing.List:
        raise NotImplementedError

    def compute_observations(self):
        raise NotImplementedError

    def compute_termination(self):
        """Set self.time_out_buf and self.reset_buf."""
        raise NotImplementedError

    def episode_log(self, env_ids) -> typing.Dict:
        """Optional extra scalars for extras["episode"], e.g. a success rate over env_ids."""
        return None

    # -- VecEnv surface -------------------------------------------------------------
    def step(self, actions: torch.Tensor):
        assert self.isg_env.robot, "add robot before step"
        if getattr(self, "_hook_graphs", None) is not None:
            return self._step_replayed(actions)
        self.actions = torch.clip(actions, -self.clip_actions, self.clip_actions)
        self.isg_env.step(self.actions)
        self.post_step()
        self.obs_buf = torch.clip(self.obs_buf, -self.clip_obs, self.clip_obs)
        return self.obs_buf, self.privileged_obs_buf, self.rew_buf, self.reset_buf, self.extras

    def post_step(self):
        self._post_step_before_reset()
        env_ids = self.reset_buf.nonzero(as_tuple=False).flatten()
        self.reset_idx(env_ids)
        self._post_step_after_reset()

    def _post_step_before_reset(self):
        self.episode_length_buf += 1
        self.common_step_counter += 1
        self.compute_termination()
        self.compute_reward()            # rewards of terminating envs are taken BEFORE their reset (Q15)

    def _post_step_after_reset(self):
        self.compute_observations()      # ... observations AFTER it
        self.isg_env.refresh_sensors()
        if self.cfg.num_actions_history:
            self.actions_recorder.add(self.actions)   # Q12: the history lags the obs by one step

    # -- optional execution mode: the shape-static parts of a step replayed from two hipGraphs -----------------------
    def enable_graph_hooks(self, warmup: int = 2):
        """Opt in: capture everything of `step` that does not depend on which envs reset -- the action clip, the robot's
        `step` (the user's control loop over gym.simulate), termination, rewards (graph 1); observations, sensors, action
        history, observation clip (graph 2) -- into two hipGraphs and replay them, with `reset_buf.nonzero()` and `reset_idx`
        in between as ordinary eager code.  The user's hooks are not modified, only launched differently: on this path a
        step is ~150 eager torch launches of 5-8 us of host time each (profiles/r04_hook_path.md).

        Contract for the hooks (the shipped examples keep it): between steps they may update tensors IN PLACE (index
        assignment, `+=`, ...) but must not rebind an attribute that a captured hook reads to a NEW tensor (`self.x =
        torch.zeros(...)` inside `reset_idx`), draw no random numbers and take no data-dependent Python branches inside the
        captured hooks; and an attribute that the second graph's hooks (observations, sensors, history) REBIND must not be
        read by the first graph's hooks (control loop, termination, rewards) of the next step -- the capture holds the
        address of the tensor the warm-up bound, not the one the last replay wrote.  `step` returns copies of the
        observation and reset tensors (they are fixed graph buffers underneath, overwritten by the next replay).  Call after construction and before `reset()`: the capture runs the hooks a few times on whatever
        state is there.  Not the default: a misbehaving hook fails silently under replay, and graph replays have misbehaved
        on this stack before (profiles/r03_graph_replay.md) -- tests/test_gpu_env.py holds this mode to the fused kernel."""
        assert self.obs_buf.is_cuda, "graph replay needs the GPU path"
        dev = self.obs_buf.device
        self._static_actions = torch.zeros(self.num_envs, self.num_actions, device=dev, dtype=torch.float)

        def before():
            self.actions = torch.clip(self._static_actions, -self.clip_actions, self.clip_actions)
            self.isg_env.step(self.actions)
            self._post_step_before_reset()

        def after():
            self._post_step_after_reset()
            self.obs_buf = torch.clip(self.obs_buf, -self.clip_obs, self.clip_obs)

        counter = self.common_step_counter
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(warmup):
                before()
                after()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(g1):
                before()
            with torch.cuda.graph(g2):      # its own memory pool: g1's transient buffers must not alias g2's live outputs
                after()
        except Exception as exc:      # a hook copied from the host, synchronised or branched on device data
            self.common_step_counter = counter
            self._hook_graphs = None
            raise BackendError("enable_graph_hooks: this env's hooks cannot be captured into a hipGraph -- they must be pure "
                               f"tensor code with fixed shapes (no host-to-device copies, .item(), Python branches on device "
                               f"values); the env stays on the eager path.  Cause: {exc}") from exc
        self.common_step_counter = counter
        self._hook_graphs = (g1, g2)

    def _step_replayed(self, actions: torch.Tensor):
        g1, g2 = self._hook_graphs
        self._static_actions.copy_(actions)
        g1.replay()
        self.common_step_counter += 1
        env_ids = self.reset_buf.nonzero(as_tuple=False).flatten()       # (the host sync of env.py:101, as in eager mode)
        self.reset_idx(env_ids)
        g2.replay()
        # obs_buf / reset_buf / time_out_buf are fixed graph-pool buffers here, overwritten in place by the next replay; the
        # eager path binds fresh tensors every step (compute_observations, torch.clip, compute_termination), and callers rely
        # on it -- rsl_rl's PPO.act keeps `transition.observations = obs` by reference until after the next env.step.  So the
        # step hands out copies; the attributes stay bound to the static buffers (what get_observations() returns is
        # overwritten by the next step, as rew_buf is on the eager path too).
        obs = self.obs_buf.clone()
        priv = None if self.privileged_obs_buf is None else self.privileged_obs_buf.clone()
        if torch.is_tensor(self.extras.get("time_outs")):
            self.extras["time_outs"] = self.extras["time_outs"].clone()
        return obs, priv, self.rew_buf, self.reset_buf.clone(), self.extras

    def reset(self):
        self.reset_idx(torch.arange(self.num_envs, device=self.device))
        obs, priv, _, _, _ = self.step(torch.zeros(self.num_envs, self.num_actions, device=self.device,
                                                   requires_grad=False))
        return obs, priv

    def reset_idx(self, env_ids):
        if len(env_ids) == 0:
            return
        self.isg_env.reset_idx(env_ids)
        if self._reset_buffers_in_one_launch(env_ids):
            pass        # the five statements of the else-branch as one launch (csrc/shf_glue.hip: shf_reset_bookkeeping)
        else:
            self.episode_length_buf[env_ids] = 0
            self.reset_buf[env_ids] = 1
            if self.cfg.num_actions_history:
                self.actions_recorder.reset_idx(env_ids)
            self.extras["episode"] = {}
            self.log_info(env_ids)
        if self.cfg.send_timeouts:
            self.extras["time_outs"] = self.time_out_buf

    def _reset_buffers_in_one_launch(self, env_ids) -> bool:
        """episode_length_buf[ids] = 0, reset_buf[ids] = 1, the action history's rows zeroed and log_info(ids) as ONE launch --
        when every tensor involved is what the library's kernel takes (CUDA, contiguous, the dtypes this class creates: an int64
        length buffer, a bool or int64 reset buffer, float32 sums and history); False: nothing was done."""
        rec = self.actions_recorder.history_buf if self.cfg.num_actions_history else None
        ok = (torch.is_tensor(env_ids) and env_ids.is_cuda and env_ids.dtype == torch.int64 and 0 < len(self.episode_rewards) <= 16
              and all(v.is_cuda and v.dtype == torch.float32 and v.is_contiguous() for v in self.episode_rewards.values())
              and self.episode_length_buf.is_cuda and self.episode_length_buf.dtype == torch.int64 and self.episode_length_buf.is_contiguous()
              and self.reset_buf.is_cuda and self.reset_buf.is_contiguous() and self.reset_buf.element_size() in (1, 8)
              and self.reset_buf.dtype in (torch.bool, torch.uint8, torch.int64)
              and (rec is None or (rec.is_cuda and rec.dtype == torch.float32 and rec.is_contiguous() and rec.shape[0] == self.num_envs))
              and type(self).log_info is ShifuVecEnv.log_info)          # (a subclass that overrides log_info keeps the statement-by-statement path)
        if not ok:
            return False
        from shifu_amd import glue
        if getattr(self, "_episode_log", None) is None:
            self._episode_log = glue.EpisodeLog(env_ids.device)
        keys = list(self.episode_rewards.keys())
        means = self._episode_log([self.episode_rewards[k] for k in keys], env_ids, self.max_episode_length_s,
                                  episode_length=self.episode_length_buf, reset_buf=self.reset_buf, history=rec)
        self.extras["episode"] = {key: means[i] for i, key in enumerate(keys)}
        info = self.episode_log(env_ids)
        if info:
            self.extras["episode"].update(info)
        return True

    def log_info(self, env_ids):
        if torch.is_tensor(env_ids) and env_ids.is_cuda and env_ids.dtype == torch.int64 and 0 < len(self.episode_rewards) <= 16 \
                and all(v.is_cuda for v in self.episode_rewards.values()):
            # the loop below as one launch (csrc/shf_glue.hip: shf_episode_log; exact fixed-point sums)
            from shifu_amd import glue
            if getattr(self, "_episode_log", None) is None:
                self._episode_log = glue.EpisodeLog(env_ids.device)
            keys = list(self.episode_rewards.keys())
            means = self._episode_log([self.episode_rewards[k] for k in keys], env_ids, self.max_episode_length_s)
            for i, key in enumerate(keys):
                self.extras["episode"][key] = means[i]
            info = self.episode_log(env_ids)
            if info:
                self.extras["episode"].update(info)
            return
        for key, sums in self.episode_rewards.items():
            self.extras["episode"][key] = torch.mean(sums[env_ids]) / self.max_episode_length_s
            sums[env_ids] = 0.
        info = self.episode_log(env_ids)
        if info:
            self.extras["episode"].update(info)

    def _prepare_reward_functions(self):
        assert len(self.reward_functions) > 0
        self.episode_rewards = {f.__name__: torch.zeros(self.num_envs, device=self.device, dtype=torch.float)
                                for f in self.reward_functions}

    def compute_reward(self):
        if getattr(self, "rew_buf", None) is not None and self.rew_buf.is_cuda and len(self.reward_functions) <= 16:
            from shifu_amd import glue
            def as_term(r):      # a hook may return a Python number or a 0-d / broadcastable tensor (`rew_buf += r` took those)
                if not torch.is_tensor(r):
                    r = torch.as_tensor(r, dtype=torch.float32, device=self.rew_buf.device)
                if r.dtype == torch.float32 and r.is_contiguous() and r.shape == self.rew_buf.shape and r.device == self.rew_buf.device:
                    return r
                return (r.to(device=self.rew_buf.device, dtype=torch.float32) + torch.zeros_like(self.rew_buf)).contiguous()
            terms = [as_term(f()) for f in self.reward_functions]        # the user's hooks, untouched
            # rew_buf = sum of the terms in their order, episode sums += term: one launch (shf_reward_accumulate)
            glue.reward_accumulate(terms, [self.episode_rewards[f.__name__] for f in self.reward_functions], self.rew_buf)
            return
        self.rew_buf[:] = 0.
        for f in self.reward_functions:
            r = f()
            self.episode_rewards[f.__name__] += r
            self.rew_buf[:] += r

    def get_observations(self):
        return self.obs_buf

    def get_privileged_observations(self):
        return self.privileged_obs_buf

    def destroy(self):
        self.isg_env.destroy()
        free_tensor_attrs(self)
