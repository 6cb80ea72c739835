"""TEST INFRASTRUCTURE (checker for shifu_amd.rl; imported only by tests/).

NumPy float64 restatement, written as explicit loops, of the two pieces of arithmetic the trainer
must get right: GAE(lambda) returns/advantages with episode boundaries, and the PPO mini-batch loss
(clipped surrogate, clipped value loss, entropy bonus) with the analytic Gaussian KL that drives the
adaptive learning rate.  Source: Schulman et al. 2017 (PPO), Schulman et al. 2016 (GAE), and the
conventions of rsl_rl v1.0.2, the trainer the reference binds (shifu/runner/policy_runner.py:4; not
vendored in the reference tree -- parity for this row is against the published algorithm).
"""
import numpy as np


def gae(rewards, values, dones, last_values, gamma, lam):
    """rewards/values/dones: (T, N); last_values: (N,).  Returns (returns, standardised advantages)."""
    T, N = rewards.shape
    returns = np.zeros((T, N))
    for n in range(N):
        adv = 0.0
        for t in reversed(range(T)):
            nxt = last_values[n] if t == T - 1 else values[t + 1, n]
            live = 1.0 - float(dones[t, n])
            delta = rewards[t, n] + live * gamma * nxt - values[t, n]
            adv = delta + live * gamma * lam * adv
            returns[t, n] = adv + values[t, n]
    adv = returns - values
    adv = (adv - adv.mean()) / (adv.std(ddof=1) + 1e-8)
    return returns, adv


def gaussian_logp(a, mu, sigma):
    return float(np.sum(-0.5 * ((a - mu) / sigma) ** 2 - np.log(sigma) - 0.5 * np.log(2.0 * np.pi)))


def ppo_loss(actions, mu, sigma, value, old_logp, old_value, advantages, returns, clip, value_coef, entropy_coef,
             clipped_value=True):
    """Per-sample loops over a mini-batch: actions/mu/sigma (B, A); the rest (B,).  -> dict of scalars."""
    B = actions.shape[0]
    sur = val = ent = 0.0
    for b in range(B):
        logp = gaussian_logp(actions[b], mu[b], sigma[b])
        ratio = np.exp(logp - old_logp[b])
        s1 = -advantages[b] * ratio
        s2 = -advantages[b] * min(max(ratio, 1.0 - clip), 1.0 + clip)
        sur += max(s1, s2)
        if clipped_value:
            vc = old_value[b] + min(max(value[b] - old_value[b], -clip), clip)
            val += max((value[b] - returns[b]) ** 2, (vc - returns[b]) ** 2)
        else:
            val += (returns[b] - value[b]) ** 2
        ent += float(np.sum(0.5 + 0.5 * np.log(2.0 * np.pi) + np.log(sigma[b])))
    sur, val, ent = sur / B, val / B, ent / B
    return {"surrogate": sur, "value": val, "entropy": ent, "loss": sur + value_coef * val - entropy_coef * ent}


def gaussian_kl(old_mu, old_sigma, mu, sigma):
    """mean over the batch of KL(N(old) || N(new)), diagonal covariance."""
    B = mu.shape[0]
    tot = 0.0
    for b in range(B):
        tot += float(np.sum(np.log(sigma[b] / old_sigma[b]) + (old_sigma[b] ** 2 + (old_mu[b] - mu[b]) ** 2) / (2.0 * sigma[b] ** 2) - 0.5))
    return tot / B
