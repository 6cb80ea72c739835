"""Gaussian MLP actor + MLP critic (PPOConfig.policy, shifu/configs/policy_config.py:8-16)."""
import os
from typing import Sequence

import torch
import torch.nn as nn
from torch.distributions import Normal

from .linear import SplitKLinear

_ACTIVATIONS = {"elu": nn.ELU, "selu": nn.SELU, "relu": nn.ReLU, "lrelu": nn.LeakyReLU, "tanh": nn.Tanh,
                "sigmoid": nn.Sigmoid}


def _mfma_output_layer() -> bool:
    """mlp_backend='mfma': whether the output layers (action means / value, 128 -> 12 / 1: 0.1 % of the flops) run on the
    MFMA kernels too (bf16 operands) or stay fp32 library GEMMs.  Default on: then the whole update is free of library
    GEMMs and can be replayed from a hipGraph (PPO.graph_update); SHIFU_AMD_MFMA_OUTPUT_LAYER=0 keeps them in fp32."""
    return os.environ.get("SHIFU_AMD_MFMA_OUTPUT_LAYER", "1") == "1"


def _mlp(n_in: int, hidden: Sequence[int], n_out: int, act: str, backend: str = "torch") -> nn.Sequential:
    if act not in _ACTIVATIONS:
        raise ValueError(f"unknown activation '{act}' (one of {sorted(_ACTIVATIONS)})")
    if backend == "mfma":
        if act != "elu":
            raise ValueError("mlp_backend='mfma' implements ELU only (the activation of every reference config)")
        # csrc/shf_mlp.hip: bias + ELU fused into the GEMM; an Identity keeps rsl_rl's parameter names (actor.0, actor.2, ...)
        from .mfma_linear import MfmaLinear, MfmaMLP
        layers, last = [], n_in
        for h in hidden:
            layers += [MfmaLinear(last, h, elu=True), nn.Identity()]
            last = h
        # the output layer (action means / value: 128 -> 12 / 1, 0.1 % of the flops) stays in fp32
        layers.append(MfmaLinear(last, n_out, elu=False) if _mfma_output_layer() else SplitKLinear(last, n_out))
        return MfmaMLP(*layers)      # one chained launch per forward where every layer qualifies (rl/mfma_linear.py)
    layers, last = [], n_in
    for h in hidden:
        layers += [SplitKLinear(last, h), _ACTIVATIONS[act]()]
        last = h
    layers.append(SplitKLinear(last, n_out))
    return nn.Sequential(*layers)


class ActorCritic(nn.Module):
    """A1 default: 259 -> 512 -> 256 -> 128 -> 12 (actor) / -> 1 (critic), ELU; state-independent std."""
    is_recurrent = False

    def __init__(self, num_actor_obs, num_critic_obs, num_actions, actor_hidden_dims=(256, 256, 256),
                 critic_hidden_dims=(256, 256, 256), activation="elu", init_noise_std=1.0, mlp_backend=None, **kwargs):
        if kwargs:
            print("ActorCritic: ignoring unknown policy keys " + ", ".join(kwargs))
        super().__init__()
        # mlp_backend: "mfma" = the hand-written MFMA layers (bf16 operands, fp32 accumulation), "torch" = stock fp32
        # library GEMMs; default from SHIFU_AMD_MLP, else "torch"
        import os
        self.mlp_backend = mlp_backend or os.environ.get("SHIFU_AMD_MLP", "torch")
        self.all_layers_mfma = self.mlp_backend == "mfma" and _mfma_output_layer()
        self.actor = _mlp(num_actor_obs, actor_hidden_dims, num_actions, activation, self.mlp_backend)
        self.critic = _mlp(num_critic_obs, critic_hidden_dims, 1, activation, self.mlp_backend)
        self.std = nn.Parameter(init_noise_std * torch.ones(num_actions))
        self.distribution = None
        Normal.set_default_validate_args(False)

    def reset(self, dones=None):
        pass

    def forward(self):
        raise NotImplementedError

    @property
    def action_mean(self):
        return self.distribution.mean

    @property
    def action_std(self):
        return self.distribution.stddev

    @property
    def entropy(self):
        return self.distribution.entropy().sum(dim=-1)

    def update_distribution(self, observations):
        mean = self.actor(observations)
        self.distribution = Normal(mean, mean * 0.0 + self.std)

    def act(self, observations, **kwargs):
        self.update_distribution(observations)
        # mean + std * N(0, 1): the same draw as distribution.sample(), but torch.normal(tensor, tensor) cannot be
        # captured into a hipGraph on this ROCm build and randn_like can
        with torch.no_grad():
            return self.distribution.mean + self.distribution.stddev * torch.randn_like(self.distribution.mean)

    def get_actions_log_prob(self, actions):
        return self.distribution.log_prob(actions).sum(dim=-1)

    def act_inference(self, observations):
        return self.actor(observations)

    def evaluate(self, critic_observations, **kwargs):
        return self.critic(critic_observations)
