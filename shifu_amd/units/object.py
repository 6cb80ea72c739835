"""Object / Box actors (reference shifu/units/object.py:11-39)."""
from shifu_amd.isaacgym import gymapi

from .units import Actor


class Object(Actor):
    """A single rigid body actor."""


class Box(Object):
    def create_asset(self):
        x, y, z = self.cfg.box_dim
        self.asset = self.gym.create_box(self.sim, x, y, z, self.asset_options)

    def load_to(self, env_id, env_handle, seg_id):
        super().load_to(env_id, env_handle, seg_id)
        self.set_asset_rigid_properties(env_handle, mass=self.cfg.mass, friction=self.cfg.friction)
        self.gym.set_rigid_body_color(env_handle, self.actor_handle, 0, gymapi.MESH_VISUAL_AND_COLLISION,
                                      gymapi.Vec3(*self.cfg.color))
