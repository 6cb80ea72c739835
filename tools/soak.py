#!/usr/bin/env python3
"""Long random-action run of the fused envs: finiteness, reset rates, episode statistics."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shifu_amd.gym.a1_fused import FusedA1Env
from shifu_amd.gym.abb_fused import FusedAbbEnv

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
for name, env in (("a1", FusedA1Env(num_envs=4096, group=32)), ("a1-trimesh", FusedA1Env(num_envs=4096, group=32, terrain="trimesh")),
                  ("a1-selfcollision", FusedA1Env(num_envs=4096, group=32, self_collision=True)),
                  ("abb", FusedAbbEnv(num_envs=4096))):
    env.reset()
    t0 = time.time()
    bad = 0
    ep_sum = None
    for k in range(steps):
        a = 2 * torch.rand(env.num_envs, env.num_actions, device=env.device) - 1
        obs, _, rew, done, ex = env.step(a)
        if k % 500 == 499:
            fin = torch.isfinite(obs).all() & torch.isfinite(rew).all() & torch.isfinite(env.root_state).all() & torch.isfinite(env.dof_state).all()
            print(name, k + 1, "finite", bool(fin), "resets/step", float(done.float().mean()), "mean rew", float(rew.mean()),
                  "max|root|", float(env.root_state[:, :3].abs().max()), "max|qd|", float(env.dof_state[:, 1].abs().max()),
                  {k2: round(float(v), 4) for k2, v in ex["episode"].items()})
            bad += int(not fin)
    print(name, "done", steps, "steps in", round(time.time() - t0, 2), "s; non-finite checks:", bad)
