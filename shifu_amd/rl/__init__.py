"""On-policy trainer behind `run_policy('train' | 'play')`.

The reference delegates training to the un-vendored `rsl_rl` package
(shifu/runner/policy_runner.py:4, README.md:35-37: "rsl_rl v1.0.2").  It is not part of the
reference tree, so this is a restatement of the published algorithm -- PPO with a clipped
surrogate and clipped value loss, GAE(lambda), time-out bootstrapping, an adaptive
learning rate driven by the analytic KL between successive Gaussian policies -- behind the
names the reference binds: `OnPolicyRunner(env, train_cfg_dict, log_dir, device)`,
`.learn(num_learning_iterations, init_at_random_ep_len)`, `.load(path)`,
`.get_inference_policy(device)`, and the `PPOConfig` keys of shifu/configs/policy_config.py:4-47.

Multi-GPU: one process per GPU, each stepping its own env shard; gradients and the KL
estimate are averaged with one bucketed all-reduce per mini-batch (RCCL on MI355X).
"""
from .actor_critic import ActorCritic
from .on_policy_runner import OnPolicyRunner
from .ppo import PPO
from .storage import RolloutStorage

__all__ = ["ActorCritic", "OnPolicyRunner", "PPO", "RolloutStorage"]
