"""ABB arm pushes a cube to a goal pad using privileged state (BASELINE config 5; reference
examples/abb_pushbox_vision/a_prior_stage.py:24-160), on the MI355X backend.

    python -m examples.abb_pushbox_vision.a_prior_stage -r random

Per env step: EE-delta action -> workspace clip -> damped-least-squares IK on the EE Jacobian ->
POS targets -> 5 x simulate (+1 in refresh_state) at dt = 20 ms; the scene is arm + table + cube
+ goal pad (4 actors, root_state rows env-major in creation order)."""
import argparse

import numpy as np
import torch

from shifu_amd.gym import ShifuVecEnv
from shifu_amd.isaacgym.torch_utils import quat_from_euler_xyz, to_torch
from shifu_amd.runner import run_policy
from shifu_amd.units import ArmRobot, Box

from examples.abb_pushbox_vision.task_config import (AbbRobotConfig, GoalBoxConfig, PriorStageEnvConfig,
                                                     PriorStagePPOConfig, PushBoxConfig, TableConfig)

LOG_ROOT = './logs/abb_pushbox_vision'


class RandPosBox(Box):
    """Box re-spawned at a random xy / yaw on reset (a_prior_stage.py:24-47)."""

    def __init__(self, cfg):
        super().__init__(cfg)
        self.pos_range = {"low": [-0.1, -0.1, 0.125], "high": [0.1, 0.1, 0.125]}
        self.euler_range = {"low": [0, 0, -np.pi], "high": [0, 0, np.pi]}

    def _reset_root_state(self, env_ids):
        # The reference draws one np.random.uniform per env in a Python loop (a_prior_stage.py:39-51: 1.8 of AbbPushBox's
        # 2.1 ms per vec-step on this backend's hook path, profiles/r04_hook_path.md); this example draws the same
        # distributions -- U(low, high) per component -- for all reset envs at once, on the device.  (Not the same numbers:
        # torch's generator instead of NumPy's.)
        rows = self.root_indices[env_ids]
        if getattr(self, "_pos_lo", None) is None:
            self._pos_lo = to_torch(self.pos_range["low"], dtype=torch.float, device=self.device)
            self._pos_hi = to_torch(self.pos_range["high"], dtype=torch.float, device=self.device)
            self._eul_lo = to_torch(self.euler_range["low"], dtype=torch.float, device=self.device)
            self._eul_hi = to_torch(self.euler_range["high"], dtype=torch.float, device=self.device)
        n = len(env_ids)
        pos = self._pos_lo + (self._pos_hi - self._pos_lo) * torch.rand(n, 3, device=self.device)
        eul = self._eul_lo + (self._eul_hi - self._eul_lo) * torch.rand(n, 3, device=self.device)
        self.env.root_state[rows, :3] = pos
        self.env.root_state[rows, 3:7] = quat_from_euler_xyz(eul[:, 0], eul[:, 1], eul[:, 2])
        self.env.root_state[rows, 7:13] = 0.
        return rows


class GoalBox(RandPosBox):
    def __init__(self, cfg):
        super().__init__(cfg)
        self.pos_range['low'][2] = self.cfg.default_pos[2]
        self.pos_range['high'][2] = self.cfg.default_pos[2]


class AbbRobot(ArmRobot):
    def init_buffers(self):
        super().init_buffers()
        self.min_ee_pos = to_torch(self.cfg.min_ee_pos, device=self.device)
        self.max_ee_pos = to_torch(self.cfg.max_ee_pos, device=self.device)
        # the constant target orientation, built once (the reference rebuilds it from a Python list every step,
        # a_prior_stage.py:70: a host-to-device copy per step, which also keeps the hook out of a hipGraph capture)
        self.tar_quat = torch.tensor([0., 1., 0., 0.], device=self.device).repeat((self.env.num_envs, 1))

    def step(self, actions):
        tar_pos = self.ee_pose[:, 0, :3] + actions * self.end_effector_velocity * self.env.dt
        tar_pos = torch.clip(tar_pos, self.min_ee_pos, self.max_ee_pos)
        tar_quat = self.tar_quat
        self.dof_targets[:] = self.inverse_kinematics(torch.cat([tar_pos, tar_quat], dim=1))
        self.apply_dof_targets(self.dof_targets)


class AbbPushBox(ShifuVecEnv):
    def __init__(self, cfg):
        super().__init__(cfg)
        self.robot = AbbRobot(AbbRobotConfig())
        self.table = Box(TableConfig())
        self.cube = RandPosBox(PushBoxConfig())
        self.goal = GoalBox(GoalBoxConfig())
        self.isg_env.create_envs(robot=self.robot, objects=[self.table, self.cube, self.goal])
        self.success_buf = torch.zeros(self.num_envs, device=self.device, dtype=torch.float)

    def episode_log(self, env_ids):
        return {'success_rate': torch.mean(self.success_buf.to(torch.float)[env_ids])}

    def compute_observations(self):
        self.obs_buf = torch.cat([self.cube.base_pose[:, :2], self.goal.base_pose[:, :2],
                                  self.robot.ee_pose[:, 0, :2]], dim=1)

    def compute_termination(self):
        self.time_out_buf = self.episode_length_buf > self.max_episode_length
        self.success_buf = self.is_success().to(torch.bool)
        lo, hi = self.robot.min_ee_pos[:2], self.robot.max_ee_pos[:2]
        cube, ee = self.cube.base_pose[:, :2], self.robot.ee_pose[:, 0, :2]
        outbound = (torch.any(cube < lo, dim=1) | torch.any(cube > hi, dim=1) |
                    torch.any(ee < lo, dim=1) | torch.any(ee > hi, dim=1))
        self.reset_buf = self.time_out_buf | outbound | self.success_buf

    def build_reward_functions(self):
        return [self.reward_reaching, self.reward_success]

    def reward_reaching(self):
        goal_dist = torch.linalg.norm(self.goal.base_pose[:, :2] - self.cube.base_pose[:, :2], axis=1)
        ee_dist = torch.linalg.norm(self.robot.ee_pose[:, 0, :2] - self.cube.base_pose[:, :2], axis=1)
        in_ws = (ee_dist < 0.1).to(torch.long)
        return in_ws * torch.exp(-torch.square(goal_dist) / 0.05)

    def reward_success(self):
        return self.is_success().to(torch.float) * 200

    def is_success(self):
        d = torch.linalg.norm(self.goal.base_pose[:, :2] - self.cube.base_pose[:, :2], axis=1)
        return (d < 0.02).to(torch.long)

    def in_ws(self):
        d = torch.linalg.norm(self.robot.ee_pose[:, 0, :3] - self.cube.base_pose[:, :3], axis=1)
        return (d < 0.1).to(torch.long)


def get_args():
    parser = argparse.ArgumentParser("Abb Robot push box task")
    parser.add_argument("--run-mode", '-r', type=str, choices=['train', 'play', 'random'], default='random')
    parser.add_argument("--num-envs", type=int, default=50)
    parser.add_argument("--iterations", type=int, default=300)
    return parser.parse_args()


if __name__ == '__main__':
    args = get_args()
    run_policy(run_mode=args.run_mode, env_class=AbbPushBox, env_cfg=PriorStageEnvConfig(),
               policy_cfg=PriorStagePPOConfig(), log_root=f"{LOG_ROOT}/Prior", play_num_envs=args.num_envs,
               play_iterations=args.iterations)
