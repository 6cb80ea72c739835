"""run_policy (reference shifu/runner/policy_runner.py:17-73).

`run_mode='random'` -- the benchmark driver shape (:33-41: reset, then
`2*rand(N, A) - 1` per step) -- and 'train' / 'play' all run on this backend.  The
reference's trainer is the un-vendored rsl_rl package (`OnPolicyRunner`); here the same
interface is provided by shifu_amd.rl."""
import torch

from .utils import class_to_dict, datetime_logdir, get_load_path, set_seed


def _on_policy_runner_class():
    """The one trainer of this backend (shifu_amd.rl); its load() already has the semantics of the reference's
    OnPolicyRunner subclass (policy_runner.py:7-14).  INTEGRATION.md shows how a site with rsl_rl installed binds it."""
    from ..rl import OnPolicyRunner
    return OnPolicyRunner


def run_policy(run_mode, env_class, env_cfg, policy_cfg, log_root="./logs", play_num_envs=50, play_iterations=3000):
    if run_mode == 'train':
        env = env_class(env_cfg)
        runner = build_policy_runner(env, policy_cfg, log_root, device=_device_of(env))
        runner.learn(num_learning_iterations=policy_cfg.runner.max_iterations, init_at_random_ep_len=True)
    elif run_mode == 'play':
        env_cfg.num_envs = play_num_envs
        env_cfg.debug.headless = False
        env = env_class(env_cfg)
        policy = load_policy(env, policy_cfg, log_root, device=_device_of(env))
        env.reset()
        obs = env.get_observations()
        for _ in range(play_iterations):
            obs, _, rews, dones, infos = env.step(policy(obs.detach()).detach())
    elif run_mode == 'random':
        env_cfg.num_envs = play_num_envs
        env_cfg.debug.headless = False
        env = env_class(env_cfg)
        env.reset()
        for _ in range(play_iterations):
            actions = 2 * torch.rand(env.num_envs, env.num_actions, device=env.device) - 1
            obs, _, rews, dones, infos = env.step(actions.detach())
        return env
    else:
        raise NotImplementedError


def _device_of(env):
    """The reference hard-codes 'cuda:0' (:45, :56); a sharded run trains on the env's own device."""
    return str(getattr(env, "device", "cuda:0"))


def load_policy(env, policy_cfg, log_root, device='cuda:0'):
    return build_policy_runner(env, policy_cfg, log_root, resume=True, device=device).get_inference_policy()


def build_policy_runner(env, train_cfg, log_root="./logs", device="cuda:0", resume=False):
    log_dir = datetime_logdir(log_root, train_cfg.runner.run_name)
    cfg_dict = class_to_dict(train_cfg)
    set_seed(train_cfg.seed)
    runner = _on_policy_runner_class()(env, cfg_dict, log_dir, device=device)
    if resume:
        path = get_load_path(log_root, load_run=train_cfg.runner.load_run, checkpoint=train_cfg.runner.checkpoint)
        print(f"Loading model from: {path}")
        runner.load(path)
    return runner
