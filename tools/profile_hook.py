#!/usr/bin/env python3
"""Where the hook path's host time goes: cProfile over N ShifuVecEnv.step calls of A1Conditional (config 3).
    python tools/profile_hook.py [--envs 4096] [--steps 200] [--abb]
"""
import argparse
import cProfile
import io
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--abb", action="store_true")
    ap.add_argument("--top", type=int, default=45)
    ap.add_argument("--graph-hooks", action="store_true", help="env.enable_graph_hooks(): the hooks replayed from two hipGraphs")
    args = ap.parse_args()
    np.random.seed(0); torch.manual_seed(0)
    if args.abb:
        from examples.abb_pushbox_vision.a_prior_stage import AbbPushBox as Env
        from examples.abb_pushbox_vision.task_config import PriorStageEnvConfig as Cfg
    else:
        from examples.a1_conditional.a1_conditional import A1Conditional as Env
        from examples.a1_conditional.task_config import A1EnvConfig as Cfg
    cfg = Cfg(); cfg.num_envs = args.envs
    env = Env(cfg)
    env.reset()
    if args.graph_hooks:
        env.enable_graph_hooks()
    n, a = env.num_envs, env.num_actions
    acts = [2 * torch.rand(n, a, device=env.device) - 1 for _ in range(8)]
    for i in range(20):
        env.step(acts[i % 8])
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for i in range(args.steps):
        env.step(acts[i % 8])
    torch.cuda.synchronize()
    pr.disable()
    s = io.StringIO()
    st = pstats.Stats(pr, stream=s).sort_stats("cumulative")
    st.print_stats(args.top)
    print(s.getvalue())
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(25)
    print(s.getvalue())


if __name__ == "__main__":
    main()
