"""OnPolicyRunner: rollout collection + PPO updates + logging / checkpoints.

Bound by the reference at shifu/runner/policy_runner.py:7-14 (`load`), :21-22 (`learn`),
:47-48 (`get_inference_policy`), :62 (constructor).  `train_cfg` is `class_to_dict(PPOConfig)`:
keys "policy", "algorithm", "runner".  The env is any rsl_rl-style VecEnv (ShifuVecEnv, FusedA1Env, ...).
"""
import json
import os
import time
from collections import deque

import torch

from ..parallel import broadcast_parameters, rank, world_size
from .actor_critic import ActorCritic
from .mfma_linear import invalidate_packs, mark_packs_valid, refresh_packs
from .ppo import PPO


class OnPolicyRunner:
    def __init__(self, env, train_cfg, log_dir=None, device="cpu"):
        self.cfg = train_cfg["runner"]
        self.alg_cfg = train_cfg["algorithm"]
        self.policy_cfg = train_cfg["policy"]
        self.device = device
        self.env = env
        num_critic_obs = env.num_privileged_obs if env.num_privileged_obs is not None else env.num_obs
        name = self.cfg.get("policy_class_name", "ActorCritic")
        if name != "ActorCritic":
            raise NotImplementedError(f"policy_class_name '{name}': only the feed-forward ActorCritic is implemented")
        if self.cfg.get("algorithm_class_name", "PPO") != "PPO":
            raise NotImplementedError("algorithm_class_name: only PPO is implemented")
        actor_critic = ActorCritic(env.num_obs, num_critic_obs, env.num_actions, **self.policy_cfg).to(device)
        broadcast_parameters(actor_critic)
        self.alg = PPO(actor_critic, device=device, **self.alg_cfg)
        self.num_steps_per_env = self.cfg["num_steps_per_env"]
        self.save_interval = self.cfg["save_interval"]
        self.alg.init_storage(env.num_envs, self.num_steps_per_env, [env.num_obs], [env.num_privileged_obs],
                              [env.num_actions])
        self.log_dir = log_dir
        self.tot_timesteps = 0
        self.tot_time = 0.0
        self.current_learning_iteration = 0
        self.history = []          # one dict per iteration (also appended to <log_dir>/progress.jsonl)
        self.env.reset()

    # ------------------------------------------------------------------ learn
    def learn(self, num_learning_iterations, init_at_random_ep_len=False):
        env, alg, dev = self.env, self.alg, self.device
        if self.log_dir is not None and rank() == 0:
            os.makedirs(self.log_dir, exist_ok=True)
        if init_at_random_ep_len:
            env.episode_length_buf.copy_(torch.randint_like(env.episode_length_buf, high=int(env.max_episode_length)))
        obs = env.get_observations()
        priv = env.get_privileged_observations()
        critic_obs = priv if priv is not None else obs
        obs, critic_obs = obs.to(dev), critic_obs.to(dev)
        alg.actor_critic.train()

        # Episode statistics stay on the device during the rollout (no nonzero()/cpu() per step: a host sync
        # per env step would expose every launch latency of the policy's small kernels); one read per iteration.
        # The printed means cover the most recent >= 100 finished episodes, like the deque of the original runner.
        rewbuffer, lenbuffer = deque(maxlen=100), deque(maxlen=100)     # entries: (sum, count) per iteration
        R = {"ep_infos": [], "cur_reward_sum": torch.zeros(env.num_envs, dtype=torch.float, device=dev),
             "cur_episode_length": torch.zeros(env.num_envs, dtype=torch.float, device=dev),
             "fin": torch.zeros(3, dtype=torch.float64, device=dev)}     # return sum, length sum, count
        fin = R["fin"]
        graph = None
        use_graph = bool(self.cfg.get("graph_rollout", False)) and torch.device(dev).type == "cuda"

        first, last = self.current_learning_iteration, self.current_learning_iteration + num_learning_iterations
        for it in range(first, last):
            start = time.time()
            with torch.no_grad():     # (not inference_mode: graph-captured RNG state must stay an ordinary tensor)
                if use_graph and graph is None and it > first:
                    # the first iteration ran eagerly (warm-up: lazy inits, LDS opt-ins); capture the second and
                    # replay it from then on.  Everything in the rollout writes static buffers (rollout storage, env
                    # tensors, the accumulators in R), so one hipGraph of T x (policy + env step) replaces ~2000 launches.
                    torch.cuda.synchronize()
                    graph = torch.cuda.CUDAGraph()
                    R["ep_infos"].clear()
                    alg.storage.clear()
                    with torch.cuda.graph(graph):
                        self._rollout(R)
                    alg.storage.clear()
                if graph is not None:
                    graph.replay()
                    torch.cuda.synchronize()      # same precaution as after the captured update's replays (rl/ppo.py)
                    mark_packs_valid(alg.actor_critic)     # (the replay re-ran _rollout's refresh_packs)
                else:
                    R["ep_infos"].clear()
                    self._rollout(R)
                ep_infos = R["ep_infos"]
                obs = env.get_observations()
                priv = env.get_privileged_observations()
                critic_obs = (priv if priv is not None else obs).to(dev)
                stop = time.time()
                collection_time = stop - start
                start = stop
                alg.compute_returns(critic_obs.clone())
            invalidate_packs(alg.actor_critic)             # the update changes the weights
            mean_value_loss, mean_surrogate_loss = alg.update()
            learn_time = time.time() - start
            if self.log_dir is not None:
                rs, ls, cnt = fin.tolist()
                fin.zero_()
                if cnt > 0:
                    rewbuffer.append((rs, cnt))
                    lenbuffer.append((ls, cnt))
                self.log(locals())
            if it % self.save_interval == 0 and self.log_dir is not None and rank() == 0:
                self.save(os.path.join(self.log_dir, f"model_{it}.pt"))
        self.current_learning_iteration += num_learning_iterations
        if self.log_dir is not None and rank() == 0:
            self.save(os.path.join(self.log_dir, f"model_{self.current_learning_iteration}.pt"))

    def _rollout(self, R):
        """num_steps_per_env x (policy, env.step, bookkeeping).  Sync-free and shape-static, so it can be captured."""
        env, alg, dev = self.env, self.alg, self.device
        refresh_packs(alg.actor_critic)      # MFMA layers: the weights laid out once for the rollout's 2 x T inference passes
        for _ in range(self.num_steps_per_env):
            obs = env.get_observations().to(dev)
            priv = env.get_privileged_observations()
            # the env rewrites its observation buffer in place: the rollout keeps a copy
            obs_in = obs.clone()
            cobs_in = obs_in if priv is None else priv.to(dev).clone()
            actions = alg.act(obs_in, cobs_in)
            _, _, rewards, dones, infos = env.step(actions)
            rewards, dones = rewards.to(dev), dones.to(dev)
            alg.process_env_step(rewards, dones, infos)
            if self.log_dir is not None:
                if "episode" in infos:
                    ep = infos["episode"]
                    vals = list(ep.values())
                    if vals and all(torch.is_tensor(v) and v.shape == vals[0].shape and v.dtype == vals[0].dtype
                                    and v.device == vals[0].device for v in vals):
                        R["ep_infos"].append((tuple(ep.keys()), torch.stack(vals)))      # one launch, not one clone per key
                    else:
                        R["ep_infos"].append({k: (v.clone() if torch.is_tensor(v) else v) for k, v in ep.items()})
                if "episode_sums" in infos:          # (sums..., count) of the episodes that finished in this step
                    if R.get("ep_sums") is None:
                        R["ep_sums"] = torch.zeros_like(infos["episode_sums"], dtype=torch.float64)
                    R["ep_sums"] += infos["episode_sums"]
                if rewards.is_cuda and rewards.dtype == torch.float32 and rewards.is_contiguous() and dones.is_contiguous() \
                        and not dones.dtype.is_floating_point and dones.numel() == rewards.numel():
                    self._bookkeeping_kernel(R, rewards, dones)      # one launch (shf_episode_bookkeeping) instead of ~16
                else:
                    R["cur_reward_sum"] += rewards
                    R["cur_episode_length"] += 1
                    d = (dones > 0).to(torch.float32)
                    R["fin"][0] += (R["cur_reward_sum"] * d).sum()
                    R["fin"][1] += (R["cur_episode_length"] * d).sum()
                    R["fin"][2] += d.sum()
                    R["cur_reward_sum"] *= 1.0 - d
                    R["cur_episode_length"] *= 1.0 - d

    @staticmethod
    def _bookkeeping_kernel(R, rewards, dones):
        """The block above as one launch: the running buffers get the same float32 values; the three sums (logging only)
        are accumulated in double in a fixed order.  dones: bool / integer flags (nonzero = episode ended)."""
        import ctypes as C
        from .._lib import BackendError, lib
        p = lambda t: C.c_void_p(t.data_ptr())
        dev = rewards.device
        with torch.cuda.device(dev):
            rc = lib().shf_episode_bookkeeping(p(rewards), p(dones), dones.element_size(), rewards.numel(), p(R["cur_reward_sum"]),
                                               p(R["cur_episode_length"]), p(R["fin"]), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        if rc != 0:
            raise BackendError(lib().shf_mlp_last_error().decode())

    # -------------------------------------------------------------------- log
    def log(self, locs, width=80, pad=35):
        n_samples = self.num_steps_per_env * self.env.num_envs * world_size()
        self.tot_timesteps += n_samples
        it_time = locs["collection_time"] + locs["learn_time"]
        self.tot_time += it_time
        rec = {"iteration": locs["it"], "fps": n_samples / it_time, "collection_time": locs["collection_time"],
               "learn_time": locs["learn_time"], "value_loss": locs["mean_value_loss"],
               "surrogate_loss": locs["mean_surrogate_loss"], "learning_rate": self.alg.learning_rate,
               "mean_noise_std": float(self.alg.actor_critic.std.detach().mean()), "total_timesteps": self.tot_timesteps,
               "total_time": self.tot_time}
        raw = locs["ep_infos"]
        if raw and all(isinstance(e, tuple) and e[0] == raw[0][0] for e in raw):
            # stacked per-step values with one key set: every key's rollout mean from one reduction and one host read
            keys = raw[0][0]
            means = torch.stack([e[1].to(torch.float32).reshape(len(keys), -1) for e in raw]).mean(dim=(0, 2)).tolist()
            rec.update({"episode/" + k: m for k, m in zip(keys, means)})
            ep_infos = []
        else:
            ep_infos = [dict(zip(e[0], e[1].unbind(0))) if isinstance(e, tuple) else e for e in raw]
        if ep_infos:
            for key in ep_infos[0]:
                vals = [torch.as_tensor(e[key], dtype=torch.float32, device=self.device).reshape(-1) for e in ep_infos]
                rec["episode/" + key] = float(torch.cat(vals).mean())
        sums, names = locs["R"].get("ep_sums"), getattr(self.env, "reward_names", None)
        if sums is not None and names is not None:
            # An env that reports (sum, count) per step gets the exact rollout mean, sum of sums / sum of counts.  (The
            # per-step means above weigh every step alike; the fused envs report 0 for a step in which no episode
            # ended, where the reference keeps the previous value, so that average would be biased towards 0.)
            tot = sums.tolist()
            cnt = tot[len(names) + 1] if len(tot) > len(names) + 1 else tot[-1]
            if cnt > 0:
                for k, name in enumerate(names):
                    rec["episode/" + name] = tot[k] / cnt / float(self.env.max_episode_length_s)
            sums.zero_()
        if len(locs["rewbuffer"]) > 0:
            rec["mean_reward"] = self._recent_mean(locs["rewbuffer"])
            rec["mean_episode_length"] = self._recent_mean(locs["lenbuffer"])
        self.history.append(rec)
        if rank() != 0:
            return
        if self.log_dir is not None:
            with open(os.path.join(self.log_dir, "progress.jsonl"), "a") as f:
                f.write(json.dumps(rec) + "\n")
        head = f" Learning iteration {locs['it']}/{locs['last']} "
        lines = ["#" * width, head.center(width), ""]
        show = [("Computation:", f"{rec['fps']:.0f} steps/s (collection: {rec['collection_time']:.3f}s, learning {rec['learn_time']:.3f}s)"),
                ("Value function loss:", f"{rec['value_loss']:.4f}"), ("Surrogate loss:", f"{rec['surrogate_loss']:.4f}"),
                ("Mean action noise std:", f"{rec['mean_noise_std']:.2f}")]
        if "mean_reward" in rec:
            show += [("Mean reward:", f"{rec['mean_reward']:.2f}"), ("Mean episode length:", f"{rec['mean_episode_length']:.2f}")]
        show += [(f"Mean episode {k[8:]}:", f"{v:.4f}") for k, v in rec.items() if k.startswith("episode/")]
        show += [("Total timesteps:", str(self.tot_timesteps)), ("Iteration time:", f"{it_time:.2f}s"),
                 ("Total time:", f"{self.tot_time:.2f}s")]
        lines += [f"{k:>{pad}} {v}" for k, v in show]
        print("\n".join(lines))

    @staticmethod
    def _recent_mean(buf, at_least=100):
        """Mean over the newest iterations that together hold >= `at_least` finished episodes."""
        s = c = 0.0
        for rs, cnt in reversed(buf):
            s, c = s + rs, c + cnt
            if c >= at_least:
                break
        return s / c

    # ------------------------------------------------------------ checkpoints
    def save(self, path, infos=None):
        # the contact solver the policy was trained under travels with it (a policy trained on the compliant law plays differently
        # under the velocity-level solve): tools/play_a1.py warns on a mismatch
        solver = getattr(self.env, "solver", None)
        if solver is not None:
            infos = dict(infos or {})
            infos.setdefault("contact_solver", solver)
        torch.save({"model_state_dict": self.alg.actor_critic.state_dict(),
                    "optimizer_state_dict": self.alg.optimizer_state_dict(),
                    "iter": self.current_learning_iteration, "infos": infos}, path)

    def load(self, path, load_optimizer=True):
        loaded = torch.load(path, map_location=self.device)      # shifu/runner/policy_runner.py:8-14
        self.alg.actor_critic.load_state_dict(loaded["model_state_dict"])
        if load_optimizer:
            self.alg.optimizer.load_state_dict(loaded["optimizer_state_dict"])
            self.alg.relink_learning_rate()
        self.current_learning_iteration = loaded["iter"]
        return loaded["infos"]

    def get_inference_policy(self, device=None):
        self.alg.actor_critic.eval()
        if device is not None:
            self.alg.actor_critic.to(device)
        return self.alg.actor_critic.act_inference
