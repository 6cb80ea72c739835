"""IsaacGymEnv / TerrainGymEnv (reference shifu/gym/isaac_gym.py:16-433): the thin sim
wrapper between ShifuVecEnv and the physics backend.  Same call order as the
reference (that order is what produces quirks Q1, Q2, Q4):

    step(a):  render -> robot.step(a) -> refresh_state() [one MORE simulate] -> post_physics_step

All `self.gym.*` calls land on shifu_amd.isaacgym.gymapi.Gym, i.e. on the HIP kernels
behind include/shifu_amd.h.  Viewer / lighting code paths are accepted no-ops.
"""
from typing import List, Union

import numpy as np
import torch

from shifu_amd.isaacgym import gymapi, gymtorch, gymutil
from shifu_amd.units import Actor, Robot, Unit
from shifu_amd.utils.terrain import Terrain
from shifu_amd.utils.torch_utils import free_tensor_attrs


def quat_apply_yaw(quat, vec):
    """Rotate by the yaw part of `quat` only (reference shifu/utils/terrain.py:202-206)."""
    from shifu_amd.isaacgym.torch_utils import normalize, quat_apply
    q = quat.clone().view(-1, 4)
    q[:, :2] = 0.
    return quat_apply(normalize(q), vec)


class IsaacGymEnv:
    def __init__(self, cfg):
        self.cfg = cfg
        self.decimation = cfg.control.decimation
        self.dt = cfg.sim.dt * self.decimation          # isaac_gym.py:26 (Q1: 5 sub-steps really elapse)
        self.num_envs = cfg.num_envs
        self.device = cfg.device
        self.spacing = cfg.spacing
        self.headless = cfg.debug.headless
        self.physics_engine = cfg.physics_engine
        self.sim_params = cfg.sim_params
        self.init_done = False
        self._init_isaac_gym()
        self.viewer = None
        self.env_handles = []
        self._units: List[Unit] = []
        self._actors: List[Actor] = []

    # -- construction ---------------------------------------------------------
    def _init_isaac_gym(self):
        self.gym = gymapi.acquire_gym()
        _, dev_id = gymutil.parse_device_str(self.device)
        self.sim = self.gym.create_sim(dev_id, dev_id, self.physics_engine, self.sim_params)
        self.create_ground()

    def create_ground(self):
        self.up_axis_idx = 2
        plane = gymapi.PlaneParams()
        plane.normal = gymapi.Vec3(0.0, 0.0, 1.0)
        plane.static_friction = plane.dynamic_friction = 1.0
        plane.restitution = 0.
        self.gym.add_ground(self.sim, plane)
        self.env_origins = torch.zeros(self.num_envs, 3, device=self.device, requires_grad=False)

    def create_envs(self, robot: Robot, objects=(), sensors=()):
        self.init_done = False
        self.robot, self.objects, self.sensors = robot, list(objects), list(sensors)
        self._units += [robot, *self.objects, *self.sensors]
        self._actors += [robot, *self.objects]
        for unit in self._units:
            unit.set_env(self)
        lower = gymapi.Vec3(-self.spacing, -self.spacing, -self.spacing)
        upper = gymapi.Vec3(self.spacing, self.spacing, self.spacing)
        per_row = int(np.sqrt(self.num_envs))
        for env_id in range(self.num_envs):
            handle = self.gym.create_env(self.sim, lower, upper, per_row)
            for seg_id, unit in enumerate(self._units, 1):
                unit.load_to(env_id, handle, seg_id)
            self.env_handles.append(handle)
        self.gym.prepare_sim(self.sim)
        self._init_buffers()
        self.init_done = True

    def _init_buffers(self):
        g, s = self.gym, self.sim
        dof, root = g.acquire_dof_state_tensor(s), g.acquire_actor_root_state_tensor(s)
        body, contact = g.acquire_rigid_body_state_tensor(s), g.acquire_net_contact_force_tensor(s)
        self._refresh_all()
        self.dof_state = gymtorch.wrap_tensor(dof)          # (N*nd, 2)
        self.root_state = gymtorch.wrap_tensor(root)        # (N*A, 13) pos quat lin ang
        self.body_state = gymtorch.wrap_tensor(body)        # (N*B, 13)
        self.contact_state = gymtorch.wrap_tensor(contact)  # (N*B, 3)
        for unit in self._units:
            unit.init_buffers()

    def _refresh_all(self):
        g, s = self.gym, self.sim
        if hasattr(g, "refresh_all_state_tensors"):
            # backend extension: the six refreshes below as one call (one kinematics pass for body states + Jacobian)
            g.refresh_all_state_tensors(s)
            return
        g.refresh_actor_root_state_tensor(s)
        g.refresh_rigid_body_state_tensor(s)
        g.refresh_dof_state_tensor(s)
        g.refresh_jacobian_tensors(s)
        g.refresh_net_contact_force_tensor(s)
        g.refresh_force_sensor_tensor(s)

    # -- stepping ----------------------------------------------------------------
    def reset(self):
        self.reset_idx(torch.arange(self.num_envs, device=self.device))

    def step(self, action: torch.Tensor):
        self.render()
        self.robot.step(action)
        self.refresh_state()
        self.post_physics_step()

    def post_physics_step(self):
        pass

    def refresh_state(self):
        self.gym.simulate(self.sim)                       # the extra sub-step (isaac_gym.py:140, Q1)
        if self.device == 'cpu':
            self.gym.fetch_results(self.sim, True)
        self._refresh_all()

    def refresh_sensors(self):
        for sensor in self.sensors:
            sensor.refresh()

    def reset_idx(self, env_ids: Union[list, torch.Tensor], actors=None):
        if len(env_ids) == 0:
            return
        actors = self._actors if actors is None else actors
        rows = []
        backend = getattr(self.sim, "backend", None)
        if backend is not None and hasattr(backend, "begin_reset"):
            # this backend: the dof / position-target commits of the actors' resets go out with the root commit below as ONE
            # launch (include/shifu_amd.h: shf_sim_commit_reset); what the reference issues as three gym.set_*_tensor_indexed calls
            backend.begin_reset()
        for actor in actors:
            actor.reset_idx(env_ids)
            rows.append(actor.root_indices[env_ids])
        # (the reference de-duplicates with torch.unique -- a sort and a host sync for the output size; the actors' root
        # rows are disjoint by construction and the indexed commit copies a row named twice twice, to the same effect)
        rows = (rows[0] if len(rows) == 1 else torch.cat(rows)).to(dtype=torch.int32)
        self.gym.set_actor_root_state_tensor_indexed(self.sim, gymtorch.unwrap_tensor(self.root_state),
                                                     gymtorch.unwrap_tensor(rows), len(rows))

    # -- viewer: nothing to draw on this backend ---------------------------------
    def render(self, sync_frame_time=True):
        return

    def change_light(self, *args, **kwargs):
        return

    def destroy(self):
        for h in self.env_handles:
            self.gym.destroy_env(h)
        self.gym.destroy_viewer(self.viewer)
        self.gym.destroy_sim(self.sim)
        free_tensor_attrs(self)


class TerrainGymEnv(IsaacGymEnv):
    def create_envs(self, *args, **kwargs):
        self.spacing = 0            # env origins come from the terrain (isaac_gym.py:299-302)
        super().create_envs(*args, **kwargs)

    def _init_buffers(self):
        super()._init_buffers()
        self.height_points = self._init_height_points()
        self._glue_terrain = None
        if self.root_state.is_cuda and self.cfg.terrain.mesh_type in ('heightfield', 'trimesh'):
            # what shf_get_heights needs: the map's geometry, the samples as int16, the (P, 2) base-frame grid
            from shifu_amd import _abi
            t = _abi.ShfTerrain()
            t.rows, t.cols = self.height_samples.shape
            c = self.terrain.cfg
            t.hscale, t.vscale, t.border, t.friction = c.horizontal_scale, c.vertical_scale, c.border_size, c.static_friction
            self._glue_terrain = t
            self._glue_samples = self.height_samples.to(torch.int16).contiguous()
            self._glue_points = self.height_points[0, :, :2].contiguous()

    def _init_height_points(self):
        """(N, P, 3) base-frame sample grid, meshgrid(x, y, indexing='xy') flattened."""
        t = self.cfg.terrain
        y = torch.tensor(t.measured_points_y, device=self.device, requires_grad=False)
        x = torch.tensor(t.measured_points_x, device=self.device, requires_grad=False)
        gx, gy = torch.meshgrid(x, y, indexing='xy')
        self.num_height_points = gx.numel()
        pts = torch.zeros(self.num_envs, self.num_height_points, 3, device=self.device, requires_grad=False)
        pts[:, :, 0] = gx.flatten()
        pts[:, :, 1] = gy.flatten()
        return pts

    def post_physics_step(self):
        if self.cfg.terrain.measure_heights:
            self.measured_heights = self.get_heights()

    def create_ground(self):
        t = self.cfg.terrain
        self.up_axis_idx = 2
        if t.mesh_type not in ('heightfield', 'trimesh'):
            raise NotImplementedError("cfg.terrain.mesh_type must be one of heightfield or trimesh")
        self.terrain = Terrain(t, self.num_envs)
        (self._create_heightfield if t.mesh_type == 'heightfield' else self._create_trimesh)()
        self.height_samples = torch.tensor(self.terrain.heightsamples).view(
            self.terrain.tot_rows, self.terrain.tot_cols).to(self.device)
        self.env_origins = torch.zeros(self.num_envs, 3, device=self.device, requires_grad=False)
        max_init = t.max_init_terrain_level if t.curriculum else t.num_rows - 1
        self.terrain_levels = torch.randint(0, max_init + 1, (self.num_envs,), device=self.device)
        self.terrain_types = torch.div(torch.arange(self.num_envs, device=self.device),
                                       (self.num_envs / t.num_cols), rounding_mode='floor').to(torch.long)
        self.max_terrain_level = t.num_rows
        self.terrain_origins = torch.from_numpy(self.terrain.env_origins).to(self.device).to(torch.float)
        self.env_origins[:] = self.terrain_origins[self.terrain_levels, self.terrain_types]

    def _terrain_params(self, p):
        t = self.terrain.cfg
        p.transform.p.x = p.transform.p.y = -t.border_size
        p.transform.p.z = 0.0
        p.static_friction, p.dynamic_friction, p.restitution = t.static_friction, t.dynamic_friction, t.restitution
        return p

    def _create_heightfield(self):
        t = self.terrain.cfg
        p = self._terrain_params(gymapi.HeightFieldParams())
        p.column_scale = p.row_scale = t.horizontal_scale
        p.vertical_scale = t.vertical_scale
        p.nbRows, p.nbColumns = self.terrain.tot_cols, self.terrain.tot_rows    # (sic) isaac_gym.py:356-357
        self.gym.add_heightfield(self.sim, self.terrain.heightsamples, p)

    def _create_trimesh(self):
        t = self.terrain.cfg
        p = self._terrain_params(gymapi.TriangleMeshParams())
        p.nb_vertices, p.nb_triangles = self.terrain.vertices.shape[0], self.terrain.triangles.shape[0]
        # the backend collides the mesh as the height map plus the vertex shifts that made it
        p.height_samples = self.terrain.heightsamples
        p.horizontal_scale, p.vertical_scale = t.horizontal_scale, t.vertical_scale
        p.slope_threshold = t.slope_treshold
        self.gym.add_triangle_mesh(self.sim, self.terrain.vertices.flatten(order='C'),
                                   self.terrain.triangles.flatten(order='C'), p)

    def update_terrain_level(self, env_ids, levels):
        if not self.init_done:
            return
        self.terrain_levels = levels
        self.env_origins[env_ids] = self.terrain_origins[self.terrain_levels[env_ids], self.terrain_types[env_ids]]

    def get_heights(self, env_ids=None):
        """Terrain height under the yaw-rotated sample grid: truncate to cell indices, clip,
        take the MIN of the cell and its +x / +y neighbours (isaac_gym.py:412-433)."""
        t = self.cfg.terrain
        if t.mesh_type == 'plane':
            return torch.zeros(self.num_envs, self.num_height_points, device=self.device, requires_grad=False)
        if t.mesh_type == 'none':
            raise NameError("Can't measure height with terrain mesh type 'none'")
        if not env_ids and self.height_samples.is_cuda and getattr(self, "_glue_terrain", None) is not None:
            # the expressions below as one launch (csrc/shf_glue.hip: shf_get_heights)
            from shifu_amd import glue
            return glue.get_heights(self._glue_terrain, self._glue_samples, self.root_state, self.robot.root_indices,
                                    self._glue_points, self.num_envs)
        pose = self.robot.base_pose
        if env_ids:
            pts = quat_apply_yaw(pose[env_ids, 3:7].repeat(1, self.num_height_points),
                                 self.height_points[env_ids]) + pose[env_ids, :3].unsqueeze(1)
        else:
            pts = quat_apply_yaw(pose[:, 3:7].repeat(1, self.num_height_points),
                                 self.height_points) + pose[:, :3].unsqueeze(1)
        pts += self.terrain.cfg.border_size
        pts = (pts / self.terrain.cfg.horizontal_scale).long()
        px = torch.clip(pts[:, :, 0].view(-1), 0, self.height_samples.shape[0] - 2)
        py = torch.clip(pts[:, :, 1].view(-1), 0, self.height_samples.shape[1] - 2)
        h = torch.min(torch.min(self.height_samples[px, py], self.height_samples[px + 1, py]),
                      self.height_samples[px, py + 1])
        return h.view(self.num_envs, -1) * self.terrain.cfg.vertical_scale
