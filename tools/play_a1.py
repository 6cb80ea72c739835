#!/usr/bin/env python3
"""Roll a trained A1 policy (tools/train_a1.py checkpoint) on the fused env: the reference's run_mode='play'
(shifu/runner/policy_runner.py:23-32) without a viewer.  Prints tracking errors and the fall rate, and can dump
a trajectory for an offline viewer (shifu_amd/checkpoint.py).

    python tools/play_a1.py gpurun_out/train_a1/model_300.pt [--envs 1024] [--steps 500] [--traj out.npz]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("checkpoint")
    ap.add_argument("--envs", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--terrain", default="heightfield", choices=["heightfield", "trimesh", "flat"])
    ap.add_argument("--traj", default=None)
    ap.add_argument("--self-collision", action="store_true", help="collide the robot's own links, as in training with the same flag")
    ap.add_argument("--solver", choices=["pgs", "tgs", "compliant"], default=None, help="contact solver (FusedA1Env(solver=...)); default: the env's")
    ap.add_argument("--max-contacts", type=int, default=None, help="ShfSimParams.max_contacts (pgs): 8 (default) .. 16")
    args = ap.parse_args()
    from examples.a1_conditional.task_config import A1PPOConfig
    from shifu_amd.checkpoint import TrajectoryRecorder
    from shifu_amd.gym.a1_fused import FusedA1Env
    from shifu_amd.rl import OnPolicyRunner
    from shifu_amd.runner.utils import class_to_dict
    env = FusedA1Env(num_envs=args.envs, terrain=args.terrain, seed=7, self_collision=args.self_collision,
                     **({} if args.solver is None else {"solver": args.solver}),
                     **({} if args.max_contacts is None else {"solver_kw": {"max_contacts": args.max_contacts}}))
    runner = OnPolicyRunner(env, class_to_dict(A1PPOConfig()), log_dir=None, device="cuda:0")
    infos = runner.load(args.checkpoint, load_optimizer=False)
    trained = (infos or {}).get("contact_solver") if isinstance(infos, dict) else None
    if trained is None:
        print(f"play_a1: the checkpoint does not record its contact solver (rounds 1-4: compliant); playing under {env.solver}", file=sys.stderr)
    elif trained != env.solver:
        print(f"play_a1: WARNING the policy was trained under solver={trained}, this env runs solver={env.solver} (pass --solver {trained})", file=sys.stderr)
    policy = runner.get_inference_policy()
    rec = TrajectoryRecorder(env, num_envs=8, bodies=True) if args.traj else None
    env.reset()
    obs = env.get_observations()
    # candidate constraints per env and sub-step before the max_contacts cap, and what the cap drops (SHF_T_CONTACT_HIST)
    hist = env.sim.bind_contact_hist(True) if env.solver in ("pgs", "tgs") else None
    lin_err = ang_err = along = speed = cmdn = 0.0
    falls = 0
    with torch.no_grad():
        for _ in range(args.steps):
            obs, _, rew, done, extras = env.step(policy(obs.clone()))
            bv = env.task.tensors[_abi.A1_BASE_VEL]               # base-frame lin (0:3) / ang (3:6) velocity
            cmd = env.command_buf
            c = cmd[:, :2]
            cn = c.norm(dim=1).clamp(min=1e-6)
            along += float(((bv[:, :2] * c).sum(1) / cn).mean())  # base speed along the commanded direction
            speed += float(bv[:, :2].norm(dim=1).mean())
            cmdn += float(cn.mean())
            lin_err += float((bv[:, :2] - c).norm(dim=1).mean())
            ang_err += float((bv[:, 5] - cmd[:, 2]).abs().mean())
            falls += int((done & ~env.time_out_buf).sum())
            if rec:
                rec.record()
    n = args.steps
    out = {"checkpoint": args.checkpoint, "envs": args.envs, "steps": args.steps, "terrain": args.terrain,
           "mean_cmd_speed_m_s": cmdn / n, "mean_base_speed_m_s": speed / n, "mean_speed_along_cmd_m_s": along / n,
           "speed_along_cmd_over_cmd": along / max(cmdn, 1e-9),
           "mean_lin_vel_error_m_s": lin_err / n, "mean_yaw_rate_error_rad_s": ang_err / n,
           "falls_per_env_per_1000_steps": falls / args.envs / n * 1000.0,
           "mean_reward_per_step": float(rew.mean()), "terrain_levels_mean": float(env.terrain_levels.float().mean())}
    if hist is not None:
        import numpy as np
        h = hist.sum(0).cpu().numpy().astype(np.float64)
        bins, tot = h[:-1], max(h[:-1].sum(), 1.0)
        cap = int(env.sim_params.max_contacts) or 8
        out.update({"max_contacts": cap, "candidates_per_substep_hist": [round(float(x), 5) for x in bins / tot],
                    "candidates_per_substep_mean": float((bins * np.arange(len(bins))).sum() / tot),
                    "substeps_above_cap_frac": float(bins[cap + 1:].sum() / tot),
                    "dropped_constraints_per_env_step": float(h[-1]) / float(args.envs * args.steps)})
    if rec:
        out["trajectory"] = rec.save(args.traj)
    print(json.dumps(out))


if __name__ == "__main__":
    from shifu_amd import _abi
    main()
