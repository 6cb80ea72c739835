"""Unit / Actor / Sensor (reference shifu/units/units.py:10-160).

A Unit is anything that lives in every env; an Actor owns one row of the sim's
root-state tensor per env (`root_indices`) and knows how to reset it.  The gym calls
are the same as the reference's and land on shifu_amd.isaacgym.gymapi.Gym.

Deviation: the reference adds each env's origin onto ONE shared Transform
(`self._init_root_pose.p += origin`, units.py:58-59), so env k is created at
default_pos + sum_{i<=k} origin_i.  Actors here are created at default_pos + origin_k,
the evident intent; after the first reset_idx both agree."""
import torch

from shifu_amd.isaacgym import gymapi
from shifu_amd.isaacgym.torch_utils import to_torch


class Unit:
    def __init__(self, cfg):
        self.cfg = cfg
        self.name = cfg.name

    def set_env(self, env):
        self.env = env
        self.gym = env.gym
        self.sim = env.sim
        self.device = env.device
        self._init_props()

    def _init_props(self):
        raise NotImplementedError

    def reset_idx(self, env_ids):
        raise NotImplementedError

    def load_to(self, env_id, env_handle, seg_id):
        raise NotImplementedError

    def init_buffers(self):
        raise NotImplementedError


class Actor(Unit):
    def __init__(self, cfg):
        super().__init__(cfg)
        self.asset_options = cfg.asset_options
        self.root_indices = []
        self.rigid_body_dict = {}

    # -- construction ---------------------------------------------------------
    def create_asset(self):
        self.asset = self.gym.load_asset(self.sim, self.cfg.root_dir, self.cfg.urdf_filename, self.asset_options)

    def _init_props(self):
        self._default_p = tuple(self.cfg.default_pos)
        self._default_r = tuple(self.cfg.default_quat)
        self.create_asset()
        self.num_bodies = self.gym.get_asset_rigid_body_count(self.asset)
        self.default_rigid_shape_props = self.gym.get_asset_rigid_shape_properties(self.asset)
        self.num_dof = self.gym.get_asset_dof_count(self.asset)
        self.dof_props = self.gym.get_asset_dof_properties(self.asset)

    def load_to(self, env_id, env_handle, seg_id):
        origin = self.env.env_origins[env_id]
        pose = gymapi.Transform()
        pose.p = gymapi.Vec3(*(float(self._default_p[k]) + float(origin[k]) for k in range(3)))
        pose.r = gymapi.Quat(*self._default_r)
        try:  # per-env shape randomisation hook (units.py:61-66)
            props = self.random_rigid_shape_props(env_id, self.default_rigid_shape_props)
            self.gym.set_asset_rigid_shape_properties(self.asset, props)
        except NotImplementedError:
            pass
        self.actor_handle = self.gym.create_actor(env_handle, self.asset, pose, self.name, env_id, 0)
        self.root_indices.append(self.gym.get_actor_index(env_handle, self.actor_handle, gymapi.DOMAIN_SIM))
        self.set_segmentation_id(env_handle, seg_id)

    def set_segmentation_id(self, env_handle, seg_id):
        self.segmentation_id = seg_id
        self.rigid_body_dict = self.gym.get_actor_rigid_body_dict(env_handle, self.actor_handle)
        for rigid_id in self.rigid_body_dict.values():
            self.gym.set_rigid_body_segmentation_id(env_handle, self.actor_handle, rigid_id, seg_id)

    def set_asset_rigid_properties(self, env_handle, mass=None, friction=None):
        if friction is not None:
            props = self.gym.get_actor_rigid_shape_properties(env_handle, self.actor_handle)
            for p in props:
                p.friction = friction
            self.gym.set_actor_rigid_shape_properties(env_handle, self.actor_handle, props)
        if mass is not None:
            props = self.gym.get_actor_rigid_body_properties(env_handle, self.actor_handle)
            for p in props:
                p.mass = mass
            self.gym.set_actor_rigid_body_properties(env_handle, self.actor_handle, props, recomputeInertia=True)

    def random_rigid_shape_props(self, env_ids, rigid_shape_props):
        """Override to randomise friction/restitution per env; return the edited list."""
        raise NotImplementedError

    # -- runtime ----------------------------------------------------------------
    def init_buffers(self):
        self.root_indices = to_torch(self.root_indices, dtype=torch.long, device=self.device)
        self.rigid_body_dict = self.gym.get_asset_rigid_body_dict(self.asset)
        self.default_base_pose = to_torch(list(self.cfg.default_pos) + list(self.cfg.default_quat), device=self.device)

    def reset_idx(self, env_ids):
        self._reset_root_state(env_ids)

    def _reset_root_state(self, env_ids):
        rows = self.root_indices[env_ids]
        self.env.root_state[rows, :3] = self.default_base_pose[:3] + self.env.env_origins[env_ids]
        self.env.root_state[rows, 3:7] = self.default_base_pose[3:7]
        self.env.root_state[rows, 7:] = 0.

    @property
    def base_pose(self):
        return self.env.root_state[self.root_indices, :7]


class Sensor(Unit):
    def refresh(self):
        raise NotImplementedError
