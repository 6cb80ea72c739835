"""Builds ShfA1TaskParams and the height-point grid for the A1Conditional task
from the same numbers the reference hard-codes (examples/a1_conditional/
task_config.py:11-69, a1_conditional.py:22-221, shifu/configs/env_config.py:78-102)."""
from __future__ import annotations

import numpy as np

from . import _abi
from .model import CompiledModel

A1_DEFAULT_DOF_POS = [0.1, 0.8, -1.5, 0.1, 0.8, -1.5, -0.1, 0.8, -1.5, -0.1, 0.8, -1.5]  # task_config.py:17-20
MEASURED_POINTS_X = [-0.8, -0.7, -0.6, -0.5, -0.4, -0.3, -0.2, -0.1, 0., 0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8]
MEASURED_POINTS_Y = [-0.5, -0.4, -0.3, -0.2, -0.1, 0., 0.1, 0.2, 0.3, 0.4, 0.5]


def height_points(xs=MEASURED_POINTS_X, ys=MEASURED_POINTS_Y) -> np.ndarray:
    """(P,2) base-frame sample grid in the order of TerrainGymEnv._init_height_points
    (isaac_gym.py:304-318): meshgrid(x, y, indexing='xy') flattened."""
    gx, gy = np.meshgrid(np.asarray(xs, np.float32), np.asarray(ys, np.float32), indexing="xy")
    return np.stack([gx.reshape(-1), gy.reshape(-1)], 1).astype(np.float32)


def a1_task_params(cm: CompiledModel, *, dt=0.005, decimation=4, episode_length_s=10.0, extra_substep=True,
                   curriculum=True, num_rows=10, num_cols=20, env_length=8.0, seed=42,
                   default_pos=(0.0, 0.0, 0.42), default_quat=(0.0, 0.0, 0.0, 1.0),
                   default_dof_pos=A1_DEFAULT_DOF_POS, kp=20.0, kd=0.5, num_height_points=187,
                   clip_obs=100.0, clip_actions=1.0, action_scale=0.5) -> _abi.ShfA1TaskParams:
    tp = _abi.ShfA1TaskParams()
    tp.decimation = decimation
    tp.extra_substep = int(extra_substep)                      # Q1 (isaac_gym.py:140)
    tp.num_history = 3
    tp.num_height_points = num_height_points
    names = cm.rigid_body_dict
    tp.base_body = names["base"]
    tp.curriculum = int(curriculum)
    tp.max_terrain_level = num_rows
    tp.num_terrain_cols = num_cols
    tp.action_scale = action_scale                             # a1_conditional.py:123
    tp.clip_actions = clip_actions
    tp.clip_obs = clip_obs
    tp.max_episode_length = float(np.ceil(episode_length_s / (dt * decimation)))  # env.py:42
    tp.max_episode_length_s = episode_length_s
    tp.env_length = env_length
    tp.max_push_force = 5.0                                    # a1_conditional.py:83
    tp.spawn_xy = 1.0                                          # a1_conditional.py:47
    tp.default_pos[:] = default_pos
    tp.default_quat[:] = default_quat
    for d in range(cm.blob.nd):
        tp.default_dof_pos[d] = default_dof_pos[d]
        tp.p_gain[d] = kp
        tp.d_gain[d] = kd
    legs = [i for n, i in names.items() if "thigh" in n or "calf" in n]  # a1_conditional.py:59-60
    tp.num_leg_bodies = len(legs)
    for k, i in enumerate(legs):
        tp.leg_bodies[k] = i
    tp.seed = seed
    return tp
