#!/usr/bin/env python3
"""Why does PPO park the A1 at the standing optimum?  Trains short schedules under one-factor variations of the
contact model / actuation / trainer and reports, per variant, what the deterministic policy does afterwards
(mean base speed along the command vs |command|) plus trainer health (noise std, learning-rate trace, KL).

    python tools/diag_walk.py --iters 600 --variants base,flat,fv002 --out gpurun_out/diag_walk.json
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def joint_edit(d, arm):
    def edit(cm):
        for k in range(cm.blob.nd):
            cm.blob.damping[k] = d
            cm.blob.armature[k] = arm
    return edit


def variants():
    from shifu_amd.backend import default_sim_params
    damp = joint_edit
    V = {
        "base": {},
        "flat": {"env": {"terrain": "flat"}},
        "fv002": {"env": {"sim_params": default_sim_params(friction_vel=0.002)}},
        "stiff": {"env": {"sim_params": default_sim_params(contact_k=2e5, contact_d=400.0)}},
        "noent": {"alg": {"entropy_coef": 0.0}},
        "noclip": {"env": {"task_overrides": {"clip_actions": 100.0}}},
        "jdamp": {"env": {"model_edit": damp(0.1, 0.01)}},
        "nopush": {"env": {"task_overrides": {"max_push_force": 0.0}}},
        "lr3e4": {"alg": {"schedule": "fixed", "learning_rate": 3e-4}},
        "g64": {"env": {"group": 64}},
    }
    return V


@torch.no_grad()
def evaluate(env, policy, steps=400):
    from shifu_amd import _abi
    env.reset()
    obs = env.get_observations()
    T = env.task.tensors
    along = speed = cmdn = yaw_err = lin_err = 0.0
    falls = 0
    dofv = 0.0
    for k in range(steps):
        obs, _, rew, done, _ = env.step(policy(obs.clone()))
        bv, cmd = T[_abi.A1_BASE_VEL], env.command_buf
        c = cmd[:, :2]
        cn = c.norm(dim=1).clamp(min=1e-6)
        along += float(((bv[:, :2] * c).sum(1) / cn).mean())
        speed += float(bv[:, :2].norm(dim=1).mean())
        cmdn += float(cn.mean())
        lin_err += float((bv[:, :2] - c).norm(dim=1).mean())
        yaw_err += float((bv[:, 5] - cmd[:, 2]).abs().mean())
        dofv += float(env.dof_state.view(env.num_envs, -1, 2)[..., 1].abs().mean())
        falls += int((done & ~env.time_out_buf).sum())
    return {"speed_along_cmd": along / steps, "speed": speed / steps, "cmd_norm": cmdn / steps,
            "ratio_along": along / max(cmdn, 1e-9), "lin_err": lin_err / steps, "yaw_err": yaw_err / steps,
            "falls_per_env_per_1000": falls / env.num_envs / steps * 1000.0, "mean_abs_dof_vel": dofv / steps}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=600)
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--variants", default="base,flat,fv002,stiff,noent,noclip,jdamp")
    ap.add_argument("--out", default="gpurun_out/diag_walk.json")
    ap.add_argument("--save", default=None, help="directory for final checkpoints")
    args = ap.parse_args()
    from examples.a1_conditional.task_config import A1PPOConfig
    from shifu_amd.gym.a1_fused import FusedA1Env
    from shifu_amd.rl import OnPolicyRunner
    from shifu_amd.runner.utils import class_to_dict, set_seed
    V = variants()
    results = {}
    import builtins
    for name in args.variants.split(","):
        if name.startswith("jd:"):          # jd:<damping>:<armature>[:seed]
            f = name.split(":")
            spec = {"env": {"model_edit": joint_edit(float(f[1]), float(f[2]))}}
            if len(f) > 3:
                spec["seed"] = int(f[3])
        else:
            spec = V[name]
        cfg = class_to_dict(A1PPOConfig())
        cfg["algorithm"].update(spec.get("alg", {}))
        cfg["runner"]["graph_rollout"] = True
        set_seed(spec.get("seed", A1PPOConfig.seed))
        env = FusedA1Env(num_envs=args.envs, device="cuda:0", seed=spec.get("seed", 42), **spec.get("env", {}))
        log_dir = os.path.join("/tmp", "diag_walk", name.replace(":", "_"))
        runner = OnPolicyRunner(env, cfg, log_dir=log_dir, device="cuda:0")
        _print, builtins.print = builtins.print, (lambda *a, **k: None)
        t0 = time.time()
        try:
            runner.learn(args.iters, init_at_random_ep_len=True)
        finally:
            builtins.print = _print
        torch.cuda.synchronize()
        el = time.time() - t0
        H = runner.history
        pick = sorted(set(int(x) for x in np.linspace(0, len(H) - 1, 9)))
        ev = evaluate(env, runner.get_inference_policy())
        lrs = np.array([h["learning_rate"] for h in H])
        results[name] = {"seconds": el, "eval": ev, "lr_at_floor_frac": float((lrs < 2e-5).mean()),
                         "lr_median": float(np.median(lrs)),
                         "curve": [{"it": H[i]["iteration"], "rew": H[i].get("mean_reward"), "len": H[i].get("mean_episode_length"),
                                    "lin": H[i].get("episode/tracking_lin_vel"), "ang": H[i].get("episode/tracking_ang_vel"),
                                    "std": H[i]["mean_noise_std"], "lr": H[i]["learning_rate"]} for i in pick]}
        print(name, json.dumps(results[name]), flush=True)
        os.makedirs(os.path.dirname(args.out), exist_ok=True)
        json.dump(results, open(args.out, "w"), indent=1)
        env.destroy()
        del runner, env
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
