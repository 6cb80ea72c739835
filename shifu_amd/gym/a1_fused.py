"""A1Conditional with the whole ShifuVecEnv.step fused into one HIP launch.

Same observable behaviour as the hook-based examples/a1_conditional env running on
the `gym` facade (quirks Q1-Q15 of SURVEY.md 3.1 preserved), but nothing between
`step(actions)` and the returned tensors runs in torch: physics x5, get_heights,
termination, rewards, on-device reset, observations and the episode logging
reduction are shf_a1_step.  Buffers keep the reference's
names (obs_buf, rew_buf, reset_buf, episode_length_buf, extras, ...: env.py:34-58)
so rsl_rl-style callers work unchanged.

Sharding: rank r of W owns global envs [r*N, (r+1)*N); everything that
depends on the env index (terrain type column, RNG streams, friction) uses the
global id, so a sharded run reproduces the unsharded one env for env.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Optional

import numpy as np
import torch

from .. import _abi
from ..a1_task import MEASURED_POINTS_X, MEASURED_POINTS_Y, a1_task_params, height_points
from ..backend import A1Task, Sim, default_sim_params
from ..model import asset_path, compile_urdf
from ..utils.terrain import Terrain

REWARD_NAMES = ["tracking_lin_vel", "tracking_ang_vel", "stabilizing_base", "smoothing_action", "leg_collision",
                "torques_penalize"]  # build_reward_functions order, a1_conditional.py:152-160


def default_terrain_cfg(mesh_type="heightfield", **kw):
    """TerrainEnvConfig.terrain defaults (shifu/configs/env_config.py:78-102)."""
    d = dict(mesh_type=mesh_type, horizontal_scale=0.1, vertical_scale=0.005, border_size=25, static_friction=1.0,
             dynamic_friction=1.0, restitution=0., measure_heights=True, measured_points_x=MEASURED_POINTS_X,
             measured_points_y=MEASURED_POINTS_Y, selected=False, terrain_kwargs=None, terrain_length=8.,
             terrain_width=8., num_rows=10, num_cols=20, terrain_proportions=[0.1, 0.1, 0.35, 0.25, 0.2],
             slope_treshold=0.75, curriculum=True, max_init_terrain_level=5)
    d.update(kw)
    return SimpleNamespace(**d)


def _philox_uniform(global_ids: np.ndarray, seed: int, stream: int) -> np.ndarray:
    """Host mirror of the kernels' counter-based draws, for init-time per-env data."""
    out = np.empty(len(global_ids), np.float64)
    M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
    for n, g in enumerate(global_ids):
        c = [int(g) & 0xFFFFFFFF, 0xFFFFFFFF, stream, (int(g) >> 32) & 0xFFFFFFFF]
        k0, k1 = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
        for _ in range(10):
            p0, p1 = M0 * c[0], M1 * c[2]
            c = [((p1 >> 32) ^ c[1] ^ k0) & 0xFFFFFFFF, p1 & 0xFFFFFFFF, ((p0 >> 32) ^ c[3] ^ k1) & 0xFFFFFFFF,
                 p0 & 0xFFFFFFFF]
            k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
        out[n] = (c[0] >> 8) * 2.0 ** -24
    return out


class FusedA1Env:
    """rsl_rl.env.VecEnv duck type (SURVEY 8b 'upward contract')."""

    def __init__(self, num_envs: int = 4096, device="cuda:0", terrain: str = "heightfield", seed: int = 42,
                 rank: int = 0, world_size: int = 1, terrain_cfg=None, sim_params: Optional[_abi.ShfSimParams] = None,
                 group: Optional[int] = None, episode_length_s: float = 10.0, dt: float = 0.005, decimation: int = 4,
                 terrain_seed: int = 42, send_timeouts: bool = True, extra_substep: bool = True,
                 model_edit=None, task_overrides: Optional[dict] = None, dof_stiffness: float = 20.0,
                 dof_damping: float = 0.5, self_collision: bool = False, mapping: Optional[str] = None,
                 solver: Optional[str] = None, solver_kw: Optional[dict] = None):
        self.device = torch.device(device)
        self.num_envs = num_envs
        self.rank, self.world_size = rank, world_size
        self.env_id_offset = rank * num_envs
        total = num_envs * world_size
        # self_collision: capsule pairs of the robot's own links (the reference creates actors with collision filter 0,
        # units.py:68 -- on in Isaac Gym; off by default here: BASELINE's config names height-field contact only, and
        # the pair tests cost kernel time)
        self.cm = compile_urdf(asset_path("a1.urdf"), default_dof_drive_mode=_abi.DOF_MODE_EFFORT,
                               self_collision=self_collision)
        # A1ActorConfig.dof_damping (task_config.py:22-23) reaches the simulator twice in the reference: as the explicit
        # PD's d gain (a1_conditional.py:66) and, through dof_props['damping'] (robot.py:35-37), as the joint's passive
        # damping, which stays active in EFFORT mode ([EXT], see isaacgym/gymapi.py prepare_sim)
        for d in range(self.cm.blob.nd):
            self.cm.blob.damping[d] = dof_damping
        if model_edit is not None:        # experiments: e.g. other joint damping / armature
            model_edit(self.cm)
        # solver: "tgs" (the default where it is built) = physx.solver_type = 1 as the reference sets it (env_config.py:50): the
        # velocity-level solve as sub-stepped sweeps (SHF_SOLVER_TGS, round 6); "pgs" = solver_type = 0: the same sweeps against the
        # step's own gaps (round 5's default); both with the reference's other PhysX settings (env_config.py:50-58;
        # csrc/shf_chain_hard.h: chain mapping at 32 lanes per env) -- the default where it is built;
        # "compliant" = rounds 1-4's spring-damper law (the default on another lane mapping / width)
        if solver is None:
            solver = "tgs" if (mapping in (None, "chain") and group in (None, 32)) else "compliant"
        self.solver = solver if sim_params is None else {_abi.SOLVER_PGS: "pgs", _abi.SOLVER_TGS: "tgs"}.get(sim_params.solver, "compliant")
        self.sim_params = sim_params or default_sim_params(dt=dt, solver=solver, **(solver_kw or {}))
        self.dt = dt * decimation                                         # isaac_gym.py:26
        self.sim = Sim(self.sim_params, self.device)
        self.cfg_terrain = terrain_cfg or default_terrain_cfg()
        ct = self.cfg_terrain
        if terrain == "flat":
            # BASELINE config 2: an all-zero height map of the same size (the reference cannot
            # build a 'plane' TerrainGymEnv: isaac_gym.py:327-334)
            ct.curriculum_layout_flat = True
            rows = int(ct.num_rows * ct.terrain_length / ct.horizontal_scale) + 2 * int(ct.border_size / ct.horizontal_scale)
            cols = int(ct.num_cols * ct.terrain_width / ct.horizontal_scale) + 2 * int(ct.border_size / ct.horizontal_scale)
            samples = np.zeros((rows, cols), np.int16)
            origins = np.zeros((ct.num_rows, ct.num_cols, 3))
            for i in range(ct.num_rows):
                for j in range(ct.num_cols):
                    origins[i, j] = [(i + 0.5) * ct.terrain_length, (j + 0.5) * ct.terrain_width, 0.0]
        elif terrain in ("heightfield", "trimesh"):
            state = np.random.get_state()
            np.random.seed(terrain_seed)          # every rank builds the identical replica
            self.terrain = Terrain(ct, total)
            np.random.set_state(state)
            samples, origins = self.terrain.heightsamples, self.terrain.env_origins
        else:
            raise ValueError("terrain must be 'flat', 'heightfield' or 'trimesh'")
        warp = None
        if terrain == "trimesh":
            # the reference's effective A1 terrain (Q5): the same samples as a triangle mesh whose steep steps have
            # vertical risers (terrain.py:75-79 -> gym.add_triangle_mesh); get_heights still reads the raw samples
            from ..isaacgym.terrain_utils import trimesh_warp_map
            warp = trimesh_warp_map(samples, ct.horizontal_scale, ct.vertical_scale, ct.slope_treshold)
        self.sim.set_heightfield(np.ascontiguousarray(samples), ct.horizontal_scale, ct.vertical_scale, ct.border_size,
                                 ct.static_friction, warp=warp)
        self.sim.set_articulation(self.cm.blob)
        # kernel selection (same results bit for bit): lane = kinematic chain (mapping="chain", csrc/shf_chain.h: A1-shaped
        # trees at 16 or 32 lanes per env, no self-collision -- the measured-fastest at 32 lanes: 71.0 vs 72.6 us per
        # vec-step, profiles/r03_bench_terrain*.json) or lane = rigid body (mapping="body": any articulation, 32 lanes
        # for the A1, also 64; self-collision)
        if mapping is None:
            mapping = "chain" if (group in (None, 32) or (group == 16 and not self_collision)) else "body"
        if group is None:
            group = 32
        self.mapping, self.group = mapping, group
        self.sim.finalize(num_envs, self.env_id_offset, group=group, mapping=mapping)

        self.max_episode_length_s = episode_length_s
        self.task_params = a1_task_params(self.cm, dt=dt, decimation=decimation, episode_length_s=episode_length_s,
                                          extra_substep=extra_substep, curriculum=ct.curriculum, num_rows=ct.num_rows, num_cols=ct.num_cols,
                                          env_length=ct.terrain_length, seed=seed, kp=dof_stiffness, kd=dof_damping,
                                          num_height_points=len(ct.measured_points_x) * len(ct.measured_points_y))
        for k, v in (task_overrides or {}).items():
            if not hasattr(self.task_params, k):
                raise AttributeError(f"ShfA1TaskParams has no field '{k}'")
            setattr(self.task_params, k, v)
        self.max_episode_length = np.ceil(episode_length_s / self.dt)     # env.py:42
        self.task = A1Task(self.sim, self.task_params)
        T, S = self.task.tensors, self.sim.tensors
        gids = np.arange(self.env_id_offset, self.env_id_offset + num_envs)
        # terrain_types = floor(i / (N/num_cols)) on GLOBAL ids (isaac_gym.py:342-344); levels start at
        # zero because A1Conditional replaces the sim-side random levels on the first reset (Q13)
        # (same float32 torch.div as the reference, so e.g. 8 / 1.6 floors to 4, not 5)
        types = torch.div(torch.from_numpy(gids), (total / ct.num_cols), rounding_mode='floor').to(torch.long).numpy()
        T[_abi.A1_TYPES].copy_(torch.from_numpy(types))
        T[_abi.A1_TORIGINS].copy_(torch.from_numpy(origins.astype(np.float32)))
        T[_abi.A1_ORIGINS].copy_(torch.from_numpy(origins[0, types].astype(np.float32)))
        T[_abi.A1_HPOINTS].copy_(torch.from_numpy(height_points(ct.measured_points_x, ct.measured_points_y)))
        # per-env shape friction U(0.5, 1.25) (a1_conditional.py:28-31), keyed by global id
        fr = 0.5 + 0.75 * _philox_uniform(gids, seed, 7)
        S[_abi.T_FRICTION].copy_(torch.from_numpy(fr.astype(np.float32)))

        self.num_obs = T[_abi.A1_OBS].shape[1]
        self.num_privileged_obs = None
        self.num_actions = self.cm.blob.nd
        self.obs_buf = T[_abi.A1_OBS]
        self.privileged_obs_buf = None
        self.rew_buf = T[_abi.A1_REW]
        self.reset_buf = T[_abi.A1_RESET].view(torch.bool)
        self.time_out_buf = T[_abi.A1_TIMEOUT].view(torch.bool)
        self.episode_length_buf = T[_abi.A1_EP_LEN]
        self.actions = T[_abi.A1_ACTIONS]
        self.command_buf = T[_abi.A1_COMMAND]
        self.episode_rewards = {n: T[_abi.A1_REW_SUMS][k] for k, n in enumerate(REWARD_NAMES)}
        self.terrain_levels = T[_abi.A1_LEVELS]
        self.dof_state, self.root_state = S[_abi.T_DOF_STATE], S[_abi.T_ROOT_STATE]
        self.body_state, self.contact_state = S[_abi.T_BODY_STATE], S[_abi.T_CONTACT]
        self.measured_heights = T[_abi.A1_HEIGHTS]
        self.send_timeouts = send_timeouts
        self.reward_names = REWARD_NAMES
        self.extras = {}
        self.common_step_counter = 0
        # actors are created at default_pos + env origin (units.py:57-70 with Q14 fixed), identity pose, at rest: the
        # first reset_idx(all) then sees distance 0 and zero commands, so no env changes its terrain level (Q13)
        spawn = torch.zeros(num_envs, 13, device=self.device)
        spawn[:, :3] = T[_abi.A1_ORIGINS] + torch.tensor(list(self.task_params.default_pos), device=self.device)
        spawn[:, 3:7] = torch.tensor(list(self.task_params.default_quat), device=self.device)
        S[_abi.T_ROOT_STATE].copy_(spawn)
        # place every env (ShifuVecEnv.__init__ leaves reset_buf at ones: env.py:48)
        self.task.reset_all()

    # -- VecEnv surface ------------------------------------------------------
    def step(self, actions: torch.Tensor):
        self.task.step(actions)
        self.common_step_counter += 1
        self._fill_extras()
        return self.obs_buf, self.privileged_obs_buf, self.rew_buf, self.reset_buf, self.extras

    def step_random(self):
        """run_policy('random') (policy_runner.py:38-41) in one launch: the U(-1, 1) actions are drawn inside the fused
        step from the task's counter-based generator (csrc/shf_task.h: random_action); same return as step()."""
        self.task.step_random()
        self.common_step_counter += 1
        self._fill_extras()
        return self.obs_buf, self.privileged_obs_buf, self.rew_buf, self.reset_buf, self.extras

    def _fill_extras(self):
        # the last row of the statistics tensor always holds the step that ran last (views: consume before the next
        # step, like the reference's extras, which are overwritten every step)
        st = self.task.tensors[_abi.A1_STATS][-1]
        ep = {n: st[8 + k] for k, n in enumerate(REWARD_NAMES)}
        ep["terrain_levels"] = st[14]                                  # episode_log, a1_conditional.py:126-129
        self.extras["episode"] = ep
        self.extras["episode_sums"] = st[:8]                           # (sum, count) form for the all-gather
        if self.send_timeouts:
            self.extras["time_outs"] = self.time_out_buf

    def reset(self):
        self.task.reset_all()
        obs, priv, _, _, _ = self.step(torch.zeros(self.num_envs, self.num_actions, device=self.device))
        return obs, priv

    def get_observations(self):
        return self.obs_buf

    def get_privileged_observations(self):
        return self.privileged_obs_buf

    def state_dict(self):
        """Checkpoint of the whole simulation + task state (shifu_amd/checkpoint.py)."""
        from ..checkpoint import env_state_dict
        return env_state_dict(self)

    def load_state_dict(self, sd):
        from ..checkpoint import load_env_state_dict
        load_env_state_dict(self, sd)

    def destroy(self):
        self.task.destroy()
        self.sim.destroy()
