#!/usr/bin/env python3
"""Train the A1 velocity-tracking policy with the in-tree PPO trainer on the fused env.

    python tools/train_a1.py [--iters 300] [--envs 4096] [--hook] [--log gpurun_out/train_a1]
    python -m torch.distributed.run --nproc-per-node N tools/train_a1.py ...      # env shards + gradient all-reduce

Prints one JSON line: samples/s (whole job, rollout + update), the reward curve at a few iterations and the
final episode statistics.  `--hook` trains on the hook-compatible A1Conditional instead (same observations).
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=300)
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--hook", action="store_true")
    ap.add_argument("--log", "--out", dest="log", default=None, help="log / checkpoint directory (--out under torchrun, whose own parser claims --log*)")
    ap.add_argument("--quiet", action="store_true")
    ap.add_argument("--graph", action="store_true", help="capture the rollout in one hipGraph (runner.graph_rollout)")
    ap.add_argument("--mlp", choices=["torch", "mfma"], default=None,
                    help="ActorCritic layers: stock fp32 library GEMMs, or the hand-written MFMA kernels (csrc/shf_mlp.hip)")
    ap.add_argument("--eager-update", action="store_true", help="with --graph: capture the rollout only, launch the PPO update eagerly")
    ap.add_argument("--graph-update", action="store_true", help="capture the PPO update only (debugging)")
    ap.add_argument("--fused-loss", action="store_true", help="(the default since round 5; accepted for old command lines)")
    ap.add_argument("--torch-loss", action="store_true", help="the PPO loss as torch expressions + autograd instead of the one HIP pass "
                                                              "(rl/fused_loss.py, shf_ppo_loss): rounds 1-4's default")
    ap.add_argument("--terrain", default="heightfield", choices=["heightfield", "trimesh", "flat"],
                    help="terrain collision form (trimesh = what the reference's A1EnvConfig effectively builds, SURVEY Q5)")
    ap.add_argument("--seed", type=int, default=None, help="override A1PPOConfig.seed")
    ap.add_argument("--self-collision", action="store_true", help="collide the robot's own links (reference collision filter 0)")
    ap.add_argument("--solver", choices=["pgs", "tgs", "compliant"], default=None, help="contact solver (FusedA1Env(solver=...)); default: the env's")
    ap.add_argument("--gpus", type=int, default=1, help="started bare with --gpus N: becomes the launcher of N ranks (one per GPU, RCCL), like bench.py")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # data-parallel PPO on a node: one child process per GPU (nothing here has touched the GPU yet); a failing rank fails the job
        from shifu_amd.parallel import launch_ranks
        raise SystemExit(launch_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus))
    import torch.distributed as dist
    world, rank, local = int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0))
    backend = os.environ.get("SHIFU_AMD_DIST_BACKEND", "nccl")     # gloo: several ranks on one GPU (testing only)
    local = local % torch.cuda.device_count() if backend != "nccl" else local
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1 or os.environ.get("SHIFU_AMD_FORCE_DIST", "0") == "1":    # FORCE_DIST: one-rank RCCL walk-through (testing)
        from shifu_amd.parallel import init_ranks
        init_ranks(dev, backend)
    from examples.a1_conditional.task_config import A1PPOConfig
    from shifu_amd.rl import OnPolicyRunner
    from shifu_amd.runner.utils import class_to_dict, set_seed
    cfg = class_to_dict(A1PPOConfig())
    cfg["runner"]["graph_rollout"] = args.graph
    cfg["algorithm"]["graph_update"] = (args.graph or args.graph_update) and not args.eager_update   # PPO.graph_update
    cfg["algorithm"]["fused_loss"] = not args.torch_loss
    if args.mlp:
        cfg["policy"]["mlp_backend"] = args.mlp
    set_seed((A1PPOConfig.seed if args.seed is None else args.seed) + rank)
    if args.hook:
        from examples.a1_conditional.a1_conditional import A1Conditional
        from examples.a1_conditional.task_config import A1EnvConfig
        ec = A1EnvConfig()
        ec.num_envs = args.envs
        env = A1Conditional(ec)
    else:
        from shifu_amd.gym.a1_fused import FusedA1Env
        env = FusedA1Env(num_envs=args.envs, device=dev, rank=rank, world_size=world, self_collision=args.self_collision,
                         terrain=args.terrain, **({} if args.solver is None else {"solver": args.solver}),
                         **({} if args.seed is None else {"seed": args.seed}))
    log_dir = args.log or os.path.join("gpurun_out", "train_a1")
    runner = OnPolicyRunner(env, cfg, log_dir=log_dir, device=str(dev))
    if args.quiet:
        import builtins
        _print, builtins.print = builtins.print, (lambda *a, **k: None)
    t0 = time.time()
    runner.learn(args.iters, init_at_random_ep_len=True)
    torch.cuda.synchronize()
    el = time.time() - t0
    if args.quiet:
        builtins.print = _print
    # every rank must hold the same parameters after synchronised updates: compare a checksum across ranks
    flat = torch.cat([p.detach().reshape(-1).double() for p in runner.alg.actor_critic.parameters()])
    checksum = torch.stack([flat.sum(), flat.abs().sum(), (flat * torch.arange(flat.numel(), device=flat.device) % 7).sum()])
    in_sync = True
    if dist.is_initialized():
        mine = checksum if backend == "nccl" else checksum.cpu()
        gathered = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
        dist.all_gather(gathered, mine)
        in_sync = all(torch.equal(g, gathered[0]) for g in gathered)
    if rank == 0:
        H = runner.history
        pick = sorted(set([0, len(H) // 8, len(H) // 4, len(H) // 2, 3 * len(H) // 4, len(H) - 1]))
        out = {"env": "A1Conditional (hook path)" if args.hook else "FusedA1Env", "envs_per_gpu": args.envs, "n_gpus": world,
               "mlp_backend": runner.alg.actor_critic.mlp_backend, "ranks_in_sync": in_sync, "param_checksum": checksum.tolist(),
               "iterations": args.iters, "steps_per_env_per_iter": cfg["runner"]["num_steps_per_env"],
               "samples_per_s": args.iters * cfg["runner"]["num_steps_per_env"] * args.envs * world / el, "seconds": el,
               "mean_collection_s": sum(h["collection_time"] for h in H) / len(H), "mean_learn_s": sum(h["learn_time"] for h in H) / len(H),
               "curve": [{"it": H[i]["iteration"], "mean_reward": H[i].get("mean_reward"), "len": H[i].get("mean_episode_length"),
                          "track_lin": H[i].get("episode/tracking_lin_vel"), "levels": H[i].get("episode/terrain_levels"),
                          "std": H[i]["mean_noise_std"]} for i in pick]}
        print(json.dumps(out))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
