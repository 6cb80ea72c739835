#!/usr/bin/env python3
"""Bisect of the captured PPO update's back-to-back replay fault (rl/ppo.py waits on the host after
every replay because without the wait the parameters drift from the eager update's).

For each variant: U updates of the same synthetic rollouts, (a) launched eagerly, (b) replayed from the captured graph
back to back (SHIFU_AMD_REPLAY_MODE=none), (c) replayed with the host wait.  Reported: whether (b) and (c) reproduce
(a) bit for bit, over R repetitions of (b) (the fault is intermittent).  Variants peel the update apart: critic on the
side stream or not, adaptive schedule (the KL clone + shf_adapt_lr), gradient clipping, Adam vs SGD, the gather kernel
vs torch indexing, layer backend.

    python tools/graph_bisect.py [updates=6] [reps=3]      # on the MI355X
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make(seed, graph_update, v):
    from shifu_amd.rl.actor_critic import ActorCritic
    from shifu_amd.rl.ppo import PPO
    os.environ["SHIFU_AMD_TWO_STREAM_UPDATE"] = "1" if v.get("two_stream", True) else "0"
    torch.manual_seed(seed)
    dev = "cuda:0"
    ac = ActorCritic(259, 259, 12, actor_hidden_dims=[512, 256, 128], critic_hidden_dims=[512, 256, 128],
                     mlp_backend=v.get("backend", "torch"))
    alg = PPO(ac, num_learning_epochs=v.get("epochs", 2), num_mini_batches=4, schedule=v.get("schedule", "adaptive"), desired_kl=0.01,
              learning_rate=1e-3, entropy_coef=0.01, device=dev, graph_update=graph_update, fused_loss=v.get("fused_loss", False),
              max_grad_norm=v.get("max_grad_norm", 1.0))
    if v.get("sgd"):
        alg.optimizer = torch.optim.SGD(alg.actor_critic.parameters(), lr=1e-3)
    if v.get("no_clip"):
        import torch.nn as nn
        alg._clip = False
        orig = alg._minibatch_step

        def step(batch, sums, _alg=alg):
            saved = nn.utils.clip_grad_norm_
            nn.utils.clip_grad_norm_ = lambda *a, **k: None
            try:
                orig(batch, sums)
            finally:
                nn.utils.clip_grad_norm_ = saved
        alg._minibatch_step = step
    N = v.get("envs", 256)
    alg.init_storage(N, 24, [259], [259], [12])
    if v.get("torch_gather"):
        st = alg.storage
        st._gather_rows = lambda srcs, idx: [t.flatten(0, 1)[idx] for t in srcs]

    def fill(gen_seed):
        g = torch.Generator(device=dev).manual_seed(gen_seed)
        obs = torch.randn(N, 259, device=dev, generator=g)
        for _ in range(24):
            with torch.no_grad():
                alg.act(obs, obs)
            rew = torch.randn(N, device=dev, generator=g)
            done = torch.rand(N, device=dev, generator=g) < 0.05
            alg.process_env_step(rew, done, {})
            obs = torch.randn(N, 259, device=dev, generator=g)
        alg.compute_returns(obs)
    return alg, fill


def run(v, graph, mode, updates):
    os.environ["SHIFU_AMD_REPLAY_MODE"] = mode
    alg, fill = make(3, graph, v)
    for it in range(updates):
        fill(100 + it)
        torch.manual_seed(7 + it)
        alg.update()
    torch.cuda.synchronize()
    return torch.cat([p.detach().flatten() for p in alg.actor_critic.parameters()]).clone()


VARIANTS = [
    ("training shape: 4096 envs, 5 epochs x 4 mini-batches = 20 replays per update, side stream", {"envs": 4096, "epochs": 5}),
    ("training shape, critic on the main stream", {"envs": 4096, "epochs": 5, "two_stream": False}),
    ("training shape, main stream, fixed lr, no clipping, SGD", {"envs": 4096, "epochs": 5, "two_stream": False, "schedule": "fixed", "no_clip": True, "sgd": True}),
    ("256 envs, 40 epochs x 4 = 160 replays per update, main stream", {"epochs": 40, "two_stream": False}),
    ("baseline: torch layers, critic on side stream, adaptive lr, clip, Adam", {}),
    ("critic on the main stream", {"two_stream": False}),
    ("main stream, fixed lr (no KL clone / shf_adapt_lr)", {"two_stream": False, "schedule": "fixed"}),
    ("main stream, fixed lr, no gradient clipping", {"two_stream": False, "schedule": "fixed", "no_clip": True}),
    ("main stream, fixed lr, no clipping, SGD", {"two_stream": False, "schedule": "fixed", "no_clip": True, "sgd": True}),
    ("main stream, torch indexing instead of shf_gather_rows", {"two_stream": False, "torch_gather": True}),
    ("side stream, torch indexing instead of shf_gather_rows", {"torch_gather": True}),
    ("MFMA layers, main stream", {"two_stream": False, "backend": "mfma"}),
    ("MFMA layers + one-pass loss, main stream, fixed lr", {"two_stream": False, "backend": "mfma", "fused_loss": True, "schedule": "fixed"}),
]


def main():
    updates = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    out = []
    for name, v in VARIANTS:
        try:
            ref = run(v, False, "wait", updates)
            waited = torch.equal(run(v, True, "wait", updates), ref)
            b2b = [run(v, True, "none", updates) for _ in range(reps)]
            rec = {"variant": name, "wait_equals_eager": waited, "back_to_back_equals_eager": [bool(torch.equal(x, ref)) for x in b2b],
                   "back_to_back_max_abs_diff": [float((x - ref).abs().max()) for x in b2b]}
        except Exception as e:                                   # keep going: a variant that cannot be captured is a data point too
            rec = {"variant": name, "error": repr(e)[:300]}
        out.append(rec)
        print(json.dumps(rec), flush=True)
    print(json.dumps({"updates": updates, "reps": reps, "n_variants": len(out)}))


if __name__ == "__main__":
    main()
