#!/usr/bin/env python3
"""Throughput of the hook-compatible (non-fused) envs through the gym facade, for the
BASELINE.md table: A1Conditional (config 3) and AbbPushBox (config 5), random actions.
    python tools/bench_hook_envs.py [--envs 4096] [--steps 200]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def run(env, steps, warmup=20):
    env.reset()
    n, a = env.num_envs, env.num_actions
    for _ in range(warmup):
        env.step(2 * torch.rand(n, a, device=env.device) - 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        env.step(2 * torch.rand(n, a, device=env.device) - 1)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    return n * steps / el, el / steps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--graph-hooks", action="store_true",
                    help="also time the hook envs with ShifuVecEnv.enable_graph_hooks(): the shape-static hooks replayed from hipGraphs")
    ap.add_argument("--only-hooks", action="store_true", help="skip the fused envs")
    args = ap.parse_args()
    np.random.seed(0); torch.manual_seed(0)
    from examples.abb_pushbox_vision.a_prior_stage import AbbPushBox
    from examples.abb_pushbox_vision.task_config import PriorStageEnvConfig
    cfg = PriorStageEnvConfig(); cfg.num_envs = args.envs
    v, ms = run(AbbPushBox(cfg), args.steps)
    print(json.dumps({"env": "AbbPushBox (config 5, hook path, 6 sub-steps of 20 ms)", "envs": args.envs,
                      "env_steps_per_s": v, "ms_per_step": ms}))
    if args.graph_hooks:
        # (this repo's example builds its constant tensors once and re-spawns the boxes with one vectorised draw, so its hooks
        # are pure tensor code and can be replayed; the reference's own file rebuilds a tensor from a Python list per step)
        try:
            env = AbbPushBox(cfg)
            env.enable_graph_hooks()
            v, ms = run(env, args.steps)
            print(json.dumps({"env": "AbbPushBox (config 5, hook path, hooks replayed from hipGraphs)", "envs": args.envs,
                              "env_steps_per_s": v, "ms_per_step": ms}))
        except Exception as exc:          # noqa: BLE001 -- report and go on with the other envs
            print(json.dumps({"env": "AbbPushBox graph hooks", "error": str(exc)[:300]}))
    from shifu_amd.gym.abb_fused import FusedAbbEnv
    for g in (() if args.only_hooks else (64, 32, 16)):
        v, ms = run(FusedAbbEnv(num_envs=args.envs, group=g, link_contacts=False, solver="compliant"), args.steps * 5)
        print(json.dumps({"env": "FusedAbbEnv (config 5, fused single launch, lanes/env=%d)" % g, "envs": args.envs,
                          "env_steps_per_s": v, "ms_per_step": ms}))
    from examples.a1_conditional.a1_conditional import A1Conditional
    from examples.a1_conditional.task_config import A1EnvConfig
    cfg = A1EnvConfig(); cfg.num_envs = args.envs
    v, ms = run(A1Conditional(cfg), args.steps)
    print(json.dumps({"env": "A1Conditional (config 3, hook path)", "envs": args.envs, "env_steps_per_s": v,
                      "ms_per_step": ms}))
    if args.graph_hooks:
        env = A1Conditional(cfg)
        env.enable_graph_hooks()
        v, ms = run(env, args.steps)
        print(json.dumps({"env": "A1Conditional (config 3, hook path, hooks replayed from hipGraphs)", "envs": args.envs,
                          "env_steps_per_s": v, "ms_per_step": ms}))


if __name__ == "__main__":
    main()
