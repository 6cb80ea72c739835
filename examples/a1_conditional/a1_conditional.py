"""A1 conditional walking on procedural terrain -- the hook-based version, source
compatible with the reference example (examples/a1_conditional/a1_conditional.py:22-245):
user code is torch, physics is the MI355X backend behind the `gym` facade.

    python -m examples.a1_conditional.a1_conditional -r random [--fused]

`--fused` runs the same task through shifu_amd.gym.a1_fused.FusedA1Env (one HIP launch
per vec-step); tests/test_gpu_env.py checks both produce the same trajectories."""
import typing

import numpy as np
import torch

from shifu_amd.gym import ShifuVecEnv
from shifu_amd.isaacgym import gymapi
from shifu_amd.isaacgym.torch_utils import to_torch, torch_rand_float
from shifu_amd.runner import run_policy
from shifu_amd.units import LeggedRobot

from examples.a1_conditional.task_config import A1ActorConfig, A1EnvConfig, A1PPOConfig

LOG_ROOT = "./logs/a1_conditional"


class A1Robot(LeggedRobot):
    def __init__(self, cfg):
        super().__init__(cfg)
        self.p_gains = to_torch(self.cfg.dof_stiffness, device=getattr(cfg, "device", "cpu"))
        self.d_gains = to_torch(self.cfg.dof_damping, device=getattr(cfg, "device", "cpu"))

    def random_rigid_shape_props(self, env_ids, rigid_shape_props):
        for prop in rigid_shape_props:
            prop.friction = np.random.uniform(0.5, 1.25)
        return rigid_shape_props

    def _reset_root_state(self, env_ids):
        rows = self.root_indices[env_ids]
        self.env.root_state[rows, :3] = self.default_base_pose[:3] + self.env.env_origins[env_ids]
        self.env.root_state[rows, :2] += torch_rand_float(-1, 1, shape=(len(env_ids), 2), device=self.device)
        self.env.root_state[rows, 3:7] = self.default_base_pose[3:7]
        self.env.root_state[rows, 7:] = 0.

    def init_buffers(self):
        super().init_buffers()
        self.p_gains, self.d_gains = self.p_gains.to(self.device), self.d_gains.to(self.device)
        n = self.env.num_envs
        self.torques = torch.zeros(n, self.num_dof, dtype=torch.float, device=self.device, requires_grad=False)
        legs = [i for name, i in self.rigid_body_dict.items() if "thigh" in name or "calf" in name]
        self.leg_indices = to_torch(legs, dtype=torch.long, device=self.device)
        self.rand_force_buf = torch.zeros(n, self.num_bodies, 3, device=self.device)

    def step(self, actions):
        """Explicit PD in EFFORT mode, `decimation` sub-steps; the push force is applied
        AFTER the loop, so it acts during refresh_state's extra sub-step only (Q4)."""
        for _ in range(self.env.decimation):
            self.torques = self.p_gains * (actions + self.default_dof_pos - self.dof_pos) - self.d_gains * self.dof_vel
            self.torques = torch.clip(self.torques, -self.torque_limits, self.torque_limits)
            self._internal_motor_step(self.torques)
            self.gym.simulate(self.sim)
            if self.device == 'cpu':
                self.gym.fetch_results(self.sim, True)
            self.gym.refresh_dof_state_tensor(self.sim)
        self.post_step()
        self.apply_force_on_base(self.rand_force_buf.view(-1, 3))

    def reset_idx(self, env_ids):
        super().reset_idx(env_ids)
        self.update_rand_force_buf(env_ids)

    def update_rand_force_buf(self, env_ids):
        max_force = 5.
        base = self.rigid_body_dict['base']
        self.rand_force_buf[env_ids, base] = torch_rand_float(-max_force, max_force, (len(env_ids), 3),
                                                              device=self.device)


class A1Conditional(ShifuVecEnv):
    def __init__(self, cfg):
        super().__init__(cfg)
        self.robot = A1Robot(A1ActorConfig())
        self.isg_env.create_envs(robot=self.robot)
        self._init_command()
        self.contact_terminate_indices = self.isg_env.gym.find_actor_rigid_body_handle(
            self.isg_env.env_handles[0], self.robot.actor_handle, 'base')
        n, dev = self.num_envs, self.device
        self.swing_time = torch.zeros(n, self.robot.ee_indices.shape[0], dtype=torch.float, device=dev)
        self.last_contacts = torch.zeros(n, len(self.robot.ee_indices), dtype=torch.bool, device=dev)
        self.contact_terminate_buf = torch.zeros(n, dtype=torch.long, device=dev)
        self.terrain_levels = torch.zeros(n, dtype=torch.long, device=dev)    # Q13

    def _init_command(self):
        self.cmd_lin_vel_x = [-1., 1.]
        self.cmd_lin_vel_y = [-1., 1.]
        self.cmd_ang_vel_yaw = [-1., 1.]
        self.num_commands = 3
        self.command_buf = torch.zeros(self.num_envs, self.num_commands, dtype=torch.float32, device=self.device)

    def reset_idx(self, env_ids):
        if self.cfg.terrain.curriculum:
            self.update_terrain_curriculum(env_ids)
        super().reset_idx(env_ids)
        self.sample_command(env_ids)

    def step(self, actions: torch.Tensor):
        return super().step(actions * 0.5)        # scaled BEFORE the +-1 clip (Q8)

    def episode_log(self, env_ids) -> typing.Dict:
        return {"terrain_levels": torch.mean(self.terrain_levels.to(torch.float))}

    def compute_observations(self):
        heights = torch.clip(self.robot.base_pose[:, 2].unsqueeze(1) - 0.5 - self.isg_env.measured_heights, -1, 1.)
        self.obs_buf = torch.cat([
            self.command_buf,
            self.robot.base_lin_vel,
            self.robot.base_ang_vel,
            self.robot.gravity_vec,                    # the constant (0,0,-1), not projected gravity (Q3)
            self.robot.dof_pos - self.robot.default_dof_pos,
            self.robot.dof_vel,
            self.actions_recorder.flatten(),
            heights,
        ], dim=1)

    def compute_termination(self):
        self.contact_terminate_buf = torch.norm(
            self.robot.contact_forces[:, self.contact_terminate_indices, :], dim=-1) > 1.
        self.time_out_buf = self.episode_length_buf > self.max_episode_length      # '>' : 501 steps (Q6)
        self.reset_buf = self.time_out_buf | self.contact_terminate_buf

    def build_reward_functions(self) -> typing.List:
        return [self.tracking_lin_vel, self.tracking_ang_vel, self.stabilizing_base, self.smoothing_action,
                self.leg_collision, self.torques_penalize]

    def tracking_lin_vel(self):
        err = torch.sum(torch.square(self.command_buf[:, :2] - self.robot.base_lin_vel[:, :2]), dim=1)
        return 1.0 * torch.exp(-err / 0.25)

    def tracking_ang_vel(self):
        err = torch.square(self.command_buf[:, 2] - self.robot.base_ang_vel[:, 2])
        return 0.5 * torch.exp(-err / 0.25)

    def stabilizing_base(self):
        z_vel = -2.0 * torch.square(self.robot.base_lin_vel[:, 2])
        ang_vel = -0.005 * torch.sum(torch.square(self.robot.base_ang_vel[:, :2]), dim=1)
        return z_vel + ang_vel

    def leg_collision(self):
        touch = torch.norm(self.robot.contact_forces[:, self.robot.leg_indices, :], dim=-1) > 0.1
        return -1. * torch.sum(touch.to(torch.float), dim=1)

    def smoothing_action(self):
        a0, a1, a2 = (self.actions_recorder.get_last(k) for k in range(3))
        first = torch.sum(torch.square(a1 - a0), dim=1)
        second = torch.sum(torch.square(a2 - 2 * a1 + a0), dim=1)
        return -0.005 * (first + second)

    def torques_penalize(self):
        return -2e-5 * torch.sum(torch.square(self.robot.torques), dim=1)

    def sample_command(self, env_ids):
        for k, rng in enumerate((self.cmd_lin_vel_x, self.cmd_lin_vel_y, self.cmd_ang_vel_yaw)):
            self.command_buf[env_ids, k] = torch_rand_float(rng[0], rng[1], (len(env_ids), 1),
                                                            device=self.device).squeeze(1)

    def update_terrain_curriculum(self, env_ids):
        if not self.isg_env.init_done:
            return
        dist = torch.norm(self.robot.base_pose[env_ids, :2] - self.isg_env.env_origins[env_ids, :2], dim=1)
        move_up = dist > self.isg_env.terrain.env_length / 2
        move_down = (dist < torch.norm(self.command_buf[env_ids, :2], dim=1) * self.max_episode_length_s * 0.5) * ~move_up
        self.terrain_levels[env_ids] += 1 * move_up - 1 * move_down
        self.terrain_levels[env_ids] = torch.where(
            self.terrain_levels[env_ids] >= self.isg_env.max_terrain_level,
            torch.randint_like(self.terrain_levels[env_ids], self.isg_env.max_terrain_level),
            torch.clip(self.terrain_levels[env_ids], 0))
        self.isg_env.update_terrain_level(env_ids, self.terrain_levels)


def get_args():
    import argparse
    parser = argparse.ArgumentParser("A1 conditional walking")
    parser.add_argument("--run-mode", '-r', type=str, choices=['train', 'play', 'random'], default='random')
    parser.add_argument("--fused", action="store_true", help="single-launch HIP env step (FusedA1Env)")
    parser.add_argument("--num-envs", type=int, default=50)
    parser.add_argument("--iterations", type=int, default=300)
    return parser.parse_args()


if __name__ == '__main__':
    args = get_args()
    if args.fused:
        from shifu_amd.gym.a1_fused import FusedA1Env
        env = FusedA1Env(num_envs=args.num_envs)
        env.reset()
        for _ in range(args.iterations):
            env.step(2 * torch.rand(env.num_envs, env.num_actions, device=env.device) - 1)
        print("mean reward", float(env.rew_buf.mean()))
    else:
        run_policy(run_mode=args.run_mode, env_class=A1Conditional, env_cfg=A1EnvConfig(), policy_cfg=A1PPOConfig(),
                   log_root=LOG_ROOT, play_num_envs=args.num_envs, play_iterations=args.iterations)
