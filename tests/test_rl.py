"""Trainer row: GAE and the PPO loss against the NumPy loop oracle, the adaptive-LR rule, a toy
problem that must actually be learnt, checkpoints, and data-parallel gradient averaging on two gloo ranks."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ppo_oracle
from shifu_amd.rl import PPO, ActorCritic, OnPolicyRunner, RolloutStorage

CFG = {"policy": {"init_noise_std": 1.0, "actor_hidden_dims": [32, 32], "critic_hidden_dims": [32, 32], "activation": "elu"},
       "algorithm": {"value_loss_coef": 1.0, "use_clipped_value_loss": True, "clip_param": 0.2, "entropy_coef": 0.01,
                     "num_learning_epochs": 5, "num_mini_batches": 4, "learning_rate": 1e-3, "schedule": "adaptive",
                     "gamma": 0.99, "lam": 0.95, "desired_kl": 0.01, "max_grad_norm": 1.0},
       "runner": {"policy_class_name": "ActorCritic", "algorithm_class_name": "PPO", "num_steps_per_env": 16,
                  "max_iterations": 10, "save_interval": 50, "experiment_name": "t", "run_name": ""}}


def test_gae_matches_loop_oracle():
    rng = np.random.default_rng(0)
    T, N = 24, 7
    st = RolloutStorage(N, T, [3], [None], [2])
    r = rng.normal(size=(T, N)).astype(np.float32)
    v = rng.normal(size=(T, N)).astype(np.float32)
    d = (rng.random((T, N)) < 0.15)
    lv = rng.normal(size=(N,)).astype(np.float32)
    st.rewards[:, :, 0] = torch.from_numpy(r)
    st.values[:, :, 0] = torch.from_numpy(v)
    st.dones[:, :, 0] = torch.from_numpy(d.astype(np.uint8))
    st.compute_returns(torch.from_numpy(lv).view(N, 1), 0.99, 0.95)
    ret, adv = ppo_oracle.gae(r.astype(np.float64), v.astype(np.float64), d, lv.astype(np.float64), 0.99, 0.95)
    np.testing.assert_allclose(st.returns[:, :, 0].numpy(), ret, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(st.advantages[:, :, 0].numpy(), adv, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("clipped_value", [True, False])
def test_ppo_loss_terms_match_loop_oracle(clipped_value):
    torch.manual_seed(1)
    ac = ActorCritic(5, 5, 3, actor_hidden_dims=[16], critic_hidden_dims=[16], init_noise_std=0.7)
    alg = PPO(ac, clip_param=0.2, value_loss_coef=0.8, entropy_coef=0.02, use_clipped_value_loss=clipped_value)
    B = 33
    obs = torch.randn(B, 5)
    actions = torch.randn(B, 3)
    old_mu, old_sigma = torch.randn(B, 3) * 0.3, torch.rand(B, 3) * 0.5 + 0.4
    old_logp = torch.distributions.Normal(old_mu, old_sigma).log_prob(actions).sum(-1, keepdim=True) + 0.3 * torch.randn(B, 1)
    old_val, adv, ret = torch.randn(B, 1), torch.randn(B, 1), torch.randn(B, 1)
    L = alg.losses(obs, obs, actions, old_val, adv, ret, old_logp, old_mu, old_sigma)
    with torch.no_grad():
        mu, sigma, value = ac.actor(obs).double().numpy(), ac.std.detach().double().expand(B, 3).numpy(), ac.critic(obs).double().numpy()[:, 0]
    o = ppo_oracle.ppo_loss(actions.double().numpy(), mu, sigma, value, old_logp.double().numpy()[:, 0], old_val.double().numpy()[:, 0],
                            adv.double().numpy()[:, 0], ret.double().numpy()[:, 0], 0.2, 0.8, 0.02, clipped_value)
    for k in ("surrogate", "value", "entropy", "loss"):
        assert abs(float(L[k].detach()) - o[k]) < 2e-5 * max(1.0, abs(o[k])), (k, float(L[k].detach()), o[k])
    kl = ppo_oracle.gaussian_kl(old_mu.double().numpy(), old_sigma.double().numpy(), mu, sigma)
    assert abs(float(L["kl"]) - kl) < 1e-4 * max(1.0, abs(kl))


def test_adaptive_learning_rate_rule():
    alg = PPO(ActorCritic(2, 2, 1), learning_rate=1e-3, schedule="adaptive", desired_kl=0.01)
    alg.adapt_learning_rate(0.05)
    assert alg.learning_rate == pytest.approx(1e-3 / 1.5) and float(alg.optimizer.param_groups[0]["lr"]) == alg.learning_rate
    alg.adapt_learning_rate(0.001)
    alg.adapt_learning_rate(0.001)
    assert alg.learning_rate == pytest.approx(1e-3 * 1.5)
    alg.adapt_learning_rate(0.01)       # inside the band: unchanged
    assert alg.learning_rate == pytest.approx(1e-3 * 1.5)
    for _ in range(40):
        alg.adapt_learning_rate(1.0)
    assert alg.learning_rate == pytest.approx(1e-5)


class ReachEnv:
    """VecEnv duck type: the observation is a target in [-1,1]^2, the reward -|action - target|^2; 8-step episodes
    with time-outs; buffers are rewritten in place like the fused envs do."""

    def __init__(self, n=64, device="cpu", seed=0):
        self.num_envs, self.num_obs, self.num_privileged_obs, self.num_actions = n, 2, None, 2
        self.device = torch.device(device)
        self.max_episode_length = 8
        self.g = torch.Generator(device="cpu").manual_seed(seed)
        self.obs_buf = torch.zeros(n, 2)
        self.rew_buf = torch.zeros(n)
        self.reset_buf = torch.zeros(n, dtype=torch.bool)
        self.episode_length_buf = torch.zeros(n, dtype=torch.long)
        self.extras = {}
        self._draw(torch.arange(n))

    def _draw(self, ids):
        self.obs_buf[ids] = torch.rand(len(ids), 2, generator=self.g) * 2 - 1

    def reset(self):
        self.episode_length_buf.zero_()
        self._draw(torch.arange(self.num_envs))
        return self.obs_buf, None

    def get_observations(self):
        return self.obs_buf

    def get_privileged_observations(self):
        return None

    def step(self, actions):
        self.rew_buf.copy_(-((actions - self.obs_buf) ** 2).sum(-1))
        self.episode_length_buf += 1
        self.reset_buf.copy_(self.episode_length_buf >= self.max_episode_length)
        ids = self.reset_buf.nonzero()[:, 0]
        self.extras = {"time_outs": self.reset_buf.clone(), "episode": {"reach": self.rew_buf.mean()}}
        self.episode_length_buf[ids] = 0
        self._draw(torch.arange(self.num_envs))
        return self.obs_buf, None, self.rew_buf, self.reset_buf, self.extras


def test_trainer_learns_the_toy_problem(tmp_path):
    torch.manual_seed(0)
    runner = OnPolicyRunner(ReachEnv(), CFG, log_dir=str(tmp_path), device="cpu")
    runner.learn(60, init_at_random_ep_len=True)
    first = np.mean([h["mean_reward"] for h in runner.history[1:6]])
    last = np.mean([h["mean_reward"] for h in runner.history[-5:]])
    assert last > 0.35 * first and last > first + 4.0, (first, last)     # returns are negative: closer to 0 is better
    assert os.path.exists(tmp_path / "progress.jsonl") and os.path.exists(tmp_path / "model_60.pt")
    # checkpoint round trip (reference load(): policy_runner.py:7-14)
    other = OnPolicyRunner(ReachEnv(seed=1), CFG, log_dir=None, device="cpu")
    other.load(str(tmp_path / "model_60.pt"))
    assert other.current_learning_iteration == 60
    obs = torch.rand(5, 2)
    torch.testing.assert_close(other.get_inference_policy()(obs), runner.get_inference_policy()(obs))
    # the learnt mean tracks the target
    assert float((runner.get_inference_policy()(obs) - obs).abs().mean()) < 0.25


def _ddp_worker(rank, world, port, out):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from shifu_amd.parallel import average_gradients, broadcast_parameters
    torch.manual_seed(100 + rank)                 # different init per rank on purpose
    net = ActorCritic(4, 4, 2, actor_hidden_dims=[8], critic_hidden_dims=[8])
    broadcast_parameters(net)
    torch.manual_seed(7)
    x = torch.randn(16, 4)
    y = torch.randn(16, 2)
    half = slice(rank * 8, (rank + 1) * 8)
    loss = ((net.actor(x[half]) - y[half]) ** 2).mean() + net.critic(x[half]).square().mean() + net.std.square().sum()
    loss.backward()
    average_gradients(net.parameters(), bucket_bytes=256)     # tiny buckets: exercises the flush path
    torch.save({"grads": [p.grad.clone() for p in net.parameters()], "params": [p.detach().clone() for p in net.parameters()]},
               os.path.join(out, f"r{rank}.pt"))
    dist.destroy_process_group()


def test_gradient_averaging_two_ranks(tmp_path):
    import torch.multiprocessing as mp
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_ddp_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    for a, b in zip(r0["params"], r1["params"]):
        torch.testing.assert_close(a, b, rtol=0, atol=0)            # broadcast made them identical
    for a, b in zip(r0["grads"], r1["grads"]):
        torch.testing.assert_close(a, b, rtol=0, atol=0)            # every rank holds the same averaged gradient
    # ... and it is the gradient of the full-batch loss
    net = ActorCritic(4, 4, 2, actor_hidden_dims=[8], critic_hidden_dims=[8])
    with torch.no_grad():
        for p, q in zip(net.parameters(), r0["params"]):
            p.copy_(q)
    torch.manual_seed(7)
    x, y = torch.randn(16, 4), torch.randn(16, 2)
    (((net.actor(x) - y) ** 2).mean() + net.critic(x).square().mean() + net.std.square().sum()).backward()
    for p, g in zip(net.parameters(), r0["grads"]):
        torch.testing.assert_close(p.grad, g, rtol=1e-5, atol=1e-6)


def test_split_k_linear_matches_nn_linear():
    """The weight-gradient re-association (shifu_amd/rl/linear.py) against plain nn.Linear."""
    from shifu_amd.rl.linear import SplitKLinear
    torch.manual_seed(3)
    a, b = torch.nn.Linear(37, 19), SplitKLinear(37, 19)
    b.load_state_dict(a.state_dict())
    x = torch.randn(8192, 37, requires_grad=True)
    x2 = x.detach().clone().requires_grad_(True)
    g = torch.randn(8192, 19)
    (a(x) * g).sum().backward()
    (b(x2) * g).sum().backward()
    torch.testing.assert_close(b(x2), a(x), rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(x2.grad, x.grad, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(b.weight.grad, a.weight.grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(b.bias.grad, a.bias.grad, rtol=1e-4, atol=1e-4)
    # small batches take the stock path
    assert torch.equal(b(x2[:64]), torch.nn.functional.linear(x2[:64], b.weight, b.bias))


def test_run_policy_train_then_play(tmp_path, capsys):
    """The reference entry point (shifu/runner/policy_runner.py:17-32) end to end on the in-tree trainer:
    'train' writes <log_root>/<timestamp>_<run_name>/model_*.pt, 'play' finds the newest one and rolls the policy."""
    from types import SimpleNamespace
    from shifu_amd.configs.policy_config import PPOConfig
    from shifu_amd.runner.policy_runner import run_policy

    class Cfg(PPOConfig):
        seed = 3

        class policy(PPOConfig.policy):
            actor_hidden_dims = [16]
            critic_hidden_dims = [16]

        class runner(PPOConfig.runner):
            num_steps_per_env = 8
            max_iterations = 3
            save_interval = 2
            run_name = "reach"

    made = []

    class Env(ReachEnv):
        def __init__(self, cfg):
            super().__init__(n=cfg.num_envs)
            made.append(self)

    env_cfg = SimpleNamespace(num_envs=32, debug=SimpleNamespace(headless=True))
    run_policy("train", Env, env_cfg, Cfg(), log_root=str(tmp_path))
    runs = os.listdir(tmp_path)
    assert len(runs) == 1 and runs[0].endswith("_reach")
    assert {"model_0.pt", "model_2.pt", "model_3.pt", "progress.jsonl"} <= set(os.listdir(tmp_path / runs[0]))
    run_policy("play", Env, env_cfg, Cfg(), log_root=str(tmp_path), play_num_envs=5, play_iterations=4)
    assert made[-1].num_envs == 5 and env_cfg.debug.headless is False
    assert "model_3.pt" in capsys.readouterr().out          # "Loading model from: ..."


def test_optimizer_checkpoints_are_interchangeable_with_the_stock_trainer(tmp_path):
    """The checkpoint keys are rsl_rl's (policy_runner.py:7-14); the optimizer part must round-trip too: the saved lr is a
    python float without implementation flags baked in from the saving device, and loading -- our own file or a stock
    Adam state dict (float lr, no fused / capturable) -- leaves the trainer with ITS flags and one shared lr tensor."""
    from shifu_amd.rl.ppo import PPO
    torch.manual_seed(0)
    net = ActorCritic(4, 4, 2, actor_hidden_dims=[8], critic_hidden_dims=[8])
    alg = PPO(net, learning_rate=3e-4, device="cpu")
    x = torch.randn(32, 4)
    (net.actor(x).square().mean() + net.critic(x).square().mean()).backward()
    alg.optimizer.step()
    sd = alg.optimizer_state_dict()
    assert isinstance(sd["param_groups"][0]["lr"], float) and abs(sd["param_groups"][0]["lr"] - 3e-4) < 1e-9
    torch.save(sd, tmp_path / "opt.pt")
    # a stock Adam (what rsl_rl builds) loads it, and its own state dict loads back into ours
    stock = torch.optim.Adam(ActorCritic(4, 4, 2, actor_hidden_dims=[8], critic_hidden_dims=[8]).parameters(), lr=1e-3)
    stock.load_state_dict(torch.load(tmp_path / "opt.pt"))
    assert abs(stock.param_groups[0]["lr"] - 3e-4) < 1e-9
    stock_sd = stock.state_dict()
    stock_sd["param_groups"][0]["lr"] = 5e-4
    alg.optimizer.load_state_dict(stock_sd)
    alg.relink_learning_rate()
    g = alg.optimizer.param_groups[0]
    assert g["lr"] is alg.lr and abs(alg.learning_rate - 5e-4) < 1e-9
    assert g["fused"] is False and g["capturable"] is False          # this trainer's flags for the CPU, whatever the file said
    (net.actor(x).square().mean()).backward()
    alg.optimizer.step()                                               # and it still steps


def test_mfma_mlp_keeps_parameter_names_and_drops_kept_packs_on_load():
    """MfmaMLP is an nn.Sequential (rsl_rl's parameter names actor.0, actor.2, ... stay), runs layer by layer on CPU, and
    a checkpoint load or a device move invalidates a kept weight pack (rl/mfma_linear.py: refresh_pack)."""
    import torch
    from shifu_amd.rl.actor_critic import ActorCritic
    from shifu_amd.rl.mfma_linear import MfmaLinear, MfmaMLP
    ac = ActorCritic(7, 7, 3, actor_hidden_dims=(16, 8), critic_hidden_dims=(16, 8), mlp_backend="mfma")
    ref = ActorCritic(7, 7, 3, actor_hidden_dims=(16, 8), critic_hidden_dims=(16, 8), mlp_backend="torch")
    assert isinstance(ac.actor, MfmaMLP) and list(ac.state_dict().keys()) == list(ref.state_dict().keys())
    ref.load_state_dict(ac.state_dict())
    x = torch.randn(5, 7)
    assert torch.allclose(ac.act_inference(x), ref.act_inference(x), atol=1e-6)
    layer = next(m for m in ac.actor if isinstance(m, MfmaLinear))
    layer._pack_valid = True
    ac.load_state_dict(ref.state_dict())
    assert not layer._pack_valid
    layer._pack_valid = True
    ac.to("cpu")
    assert not layer._pack_valid


def test_mfma_actor_critic_runs_on_cpu_without_the_native_library(monkeypatch):
    """a CPU forward of an mlp_backend='mfma' ActorCritic must take the stock layer-by-layer path without loading
    libshifu_amd.so (a host without the built library can still evaluate a checkpoint): with the library path pointing at a
    missing file the forward works, large batches included, and nothing raises BackendError."""
    import torch
    from shifu_amd import _lib
    from shifu_amd.rl.actor_critic import ActorCritic
    monkeypatch.setattr(_lib, "_PATH", "/nonexistent/libshifu_amd.so")
    monkeypatch.setattr(_lib, "_lib", None)
    ac = ActorCritic(7, 7, 3, actor_hidden_dims=(16, 8), critic_hidden_dims=(16, 8), mlp_backend="mfma")
    for rows in (5, 4096):
        y = ac.act_inference(torch.randn(rows, 7))
        assert y.shape == (rows, 3) and torch.isfinite(y).all()
    with pytest.raises(_lib.BackendError):
        _lib.lib()                       # (the library really is out of reach in this test)
