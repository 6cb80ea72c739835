"""Robot / ArmRobot / LeggedRobot (reference shifu/units/robot.py:11-236).

The tensors here are views into the sim's state tensors taken once in init_buffers
(`dof_pos` / `dof_vel` are strided views of dof_state, robot.py:51-52) and stay valid
because the backend's buffers are pointer-stable.  Every `self.gym.*` call goes to the
MI355X backend through the gymapi facade."""
import torch

from shifu_amd.isaacgym import gymapi, gymtorch
from shifu_amd.isaacgym.torch_utils import get_axis_params, quat_conjugate, quat_mul, quat_rotate_inverse, to_torch

from .units import Actor


class Robot(Actor):
    def reset_idx(self, env_ids):
        self._reset_dof_state(env_ids)
        self._reset_root_state(env_ids)

    def step(self, actions):
        self._internal_motor_step(actions)

    def _init_props(self):
        super()._init_props()
        # written even in EFFORT mode (Q11); the backend ignores drive gains for EFFORT dofs
        self.dof_props['driveMode'][:] = self.asset_options.default_dof_drive_mode
        self.dof_props['stiffness'] = self.cfg.dof_stiffness
        self.dof_props['damping'] = self.cfg.dof_damping
        self.dof_lower_limits = to_torch(self.dof_props['lower'], device=self.device)
        self.dof_upper_limits = to_torch(self.dof_props['upper'], device=self.device)
        self.dof_vel_limits = to_torch(self.dof_props['velocity'], device=self.device)
        self.torque_limits = to_torch(self.dof_props['effort'], device=self.device)

    def load_to(self, env_id, env_handle, seg_id):
        super().load_to(env_id, env_handle, seg_id)
        self.gym.set_actor_dof_properties(env_handle, self.actor_handle, self.dof_props)

    def init_buffers(self):
        super().init_buffers()
        n = self.env.num_envs
        self.default_dof_pos = to_torch(self.cfg.default_dof_pos, device=self.device)
        self.dof_pos = self.env.dof_state.view(n, self.num_dof, 2)[..., 0]
        self.dof_vel = self.env.dof_state.view(n, self.num_dof, 2)[..., 1]
        self.dof_targets = torch.zeros((n, self.num_dof), dtype=torch.float, device=self.device)

    def _internal_motor_step(self, action):
        mode = self.asset_options.default_dof_drive_mode
        if mode == gymapi.DOF_MODE_EFFORT:
            self.gym.set_dof_actuation_force_tensor(self.sim, gymtorch.unwrap_tensor(action))
        elif mode == gymapi.DOF_MODE_POS:
            self.gym.set_dof_position_target_tensor(self.sim, gymtorch.unwrap_tensor(action))
        elif mode == gymapi.DOF_MODE_VEL:
            self.gym.set_dof_velocity_target_tensor(self.sim, gymtorch.unwrap_tensor(action))
        else:
            raise NotImplementedError

    def apply_dof_targets(self, dof_targets):
        """decimation x {set targets, simulate, refresh dof state} (robot.py:66-72)."""
        for _ in range(int(self.env.decimation)):
            self.gym.set_dof_position_target_tensor(self.sim, gymtorch.unwrap_tensor(dof_targets))
            self.gym.simulate(self.sim)
            if self.device == 'cpu':
                self.gym.fetch_results(self.sim, True)
            self.gym.refresh_dof_state_tensor(self.sim)

    def _reset_dof_state(self, env_ids):
        ds = self.env.dof_state
        if ds.is_cuda and torch.is_tensor(env_ids) and env_ids.is_cuda and env_ids.dtype == torch.int64 \
                and ds.shape[0] == self.env.num_envs * self.num_dof:
            # the four statements below as one launch (csrc/shf_glue.hip: shf_reset_dof_rows); only a robot owns dofs
            # (robot.py:51-52 views dof_state as (N, num_dof, 2)), so env e's block starts at e * num_dof
            from shifu_amd import glue
            actor_ids = glue.reset_dof_rows(ds, self.dof_targets, self.default_dof_pos, env_ids, self.root_indices)
        else:
            self.dof_targets[env_ids] = self.default_dof_pos.clone()
            self.dof_pos[env_ids] = self.default_dof_pos.clone()
            self.dof_vel[env_ids] = 0.
            actor_ids = self.root_indices[env_ids].to(torch.int32)
        self.gym.set_dof_position_target_tensor_indexed(self.sim, gymtorch.unwrap_tensor(self.dof_targets),
                                                        gymtorch.unwrap_tensor(actor_ids), len(actor_ids))
        self.gym.set_dof_state_tensor_indexed(self.sim, gymtorch.unwrap_tensor(self.env.dof_state),
                                              gymtorch.unwrap_tensor(actor_ids), len(env_ids))

    def get_root_state(self):
        return self.env.root_state[self.root_indices]

    def set_root_state(self, root_state):
        self.env.root_state[self.root_indices] = root_state
        self.gym.set_actor_root_state_tensor(self.sim, gymtorch.unwrap_tensor(self.env.root_state))


class ArmRobot(Robot):
    def __init__(self, cfg):
        super().__init__(cfg)
        self.end_effector_names = cfg.end_effector_names
        self.end_effector_velocity = cfg.end_effector_velocity

    def init_buffers(self):
        super().init_buffers()
        ee = [self.rigid_body_dict[n] for n in self.cfg.end_effector_names]
        self.ee_indices = to_torch(ee, dtype=torch.long, device=self.device)
        self.num_ee = len(ee)
        self.contact_forces = self.env.contact_state.view(self.env.num_envs, -1, 3)
        self.ee_pose_targets = torch.zeros((self.env.num_envs, 7), dtype=torch.float, device=self.device)
        jac = gymtorch.wrap_tensor(self.gym.acquire_jacobian_tensor(self.sim, self.name))
        self.gym.refresh_jacobian_tensors(self.sim)
        self.j_ee = jac[:, ee[0] - 1]   # a fixed base has no Jacobian row (robot.py:128)
        self._ee0 = int(ee[0])

    def load_to(self, env_id, env_handle, seg_id):
        super().load_to(env_id, env_handle, seg_id)
        self.set_segmentation_id(env_handle, seg_id)

    def apply_target_end_positions(self, tar_pose):
        self.dof_targets[:] = self.inverse_kinematics(tar_pose)
        self.apply_dof_targets(self.dof_targets)

    @property
    def body_state(self):
        n = self.env.num_envs
        return self.env.body_state.view(n, -1, 13)[:, :self.num_bodies].view(n, self.num_bodies, -1)

    @property
    def ee_pose(self):
        return self.body_state[:, self.ee_indices, :7]

    @property
    def ee_vel(self):
        return self.body_state[:, self.ee_indices, 7:]

    @property
    def ee_forces(self):
        return self.contact_forces[:, self.ee_indices]

    @staticmethod
    def orientation_error(desired, current):
        q_r = quat_mul(desired, quat_conjugate(current))
        return q_r[:, 0:3] * torch.sign(q_r[:, 3]).unsqueeze(-1)

    def inverse_kinematics(self, goal_pose, damping=0.05):
        """Damped least squares on the EE Jacobian (robot.py:162-182)."""
        if self.j_ee.is_cuda and self.j_ee.dim() == 3 and self.j_ee.dtype == torch.float32 and goal_pose.is_cuda \
                and hasattr(self, "_ee0"):
            # the expressions below as one launch (csrc/shf_glue.hip: shf_ik_dls, LDL^T instead of torch.inverse)
            from shifu_amd import glue
            n = self.env.num_envs
            ee = self.env.body_state.view(n, -1, 13)[:, self._ee0, :7]
            return glue.ik_dls(self.j_ee, self.dof_pos, ee, goal_pose, damping)
        ee_pos, ee_quat = self.ee_pose[:, 0, :3], self.ee_pose[:, 0, 3:7]
        dpose = torch.cat([goal_pose[:, :3] - ee_pos, self.orientation_error(goal_pose[:, 3:7], ee_quat)],
                          -1).unsqueeze(-1)
        jt = torch.transpose(self.j_ee, 1, 2)
        lam = torch.eye(6, device=self.device) * (damping ** 2)
        u = (jt @ torch.inverse(self.j_ee @ jt + lam) @ dpose).view(self.env.num_envs, self.num_dof)
        return self.dof_pos + u


class LeggedRobot(ArmRobot):
    def __init__(self, cfg):
        Robot.__init__(self, cfg)        # the reference skips ArmRobot.__init__ (robot.py:190)
        self.end_effector_names = cfg.end_effector_names
        self.ee_indices = []

    def init_buffers(self):
        Robot.init_buffers(self)
        n = self.env.num_envs
        ee = [self.rigid_body_dict[k] for k in self.cfg.end_effector_names]
        self.ee_indices = to_torch(ee, dtype=torch.long, device=self.device)
        self.num_ee = len(ee)
        self.contact_forces = self.env.contact_state.view(n, -1, 3)
        jac = gymtorch.wrap_tensor(self.gym.acquire_jacobian_tensor(self.sim, self.name))
        self.gym.refresh_jacobian_tensors(self.sim)
        self.j_ee = jac[:, ee]
        self.gravity_vec = to_torch(get_axis_params(-1., self.env.up_axis_idx), device=self.device).repeat((n, 1))
        quat = self.base_pose[:, 3:7]
        self.base_lin_vel = quat_rotate_inverse(quat, self.env.root_state[self.root_indices, 7:10])
        self.base_ang_vel = quat_rotate_inverse(quat, self.env.root_state[self.root_indices, 10:13])
        self.projected_gravity = quat_rotate_inverse(quat, self.gravity_vec)

    def step(self, actions):
        self.dof_targets[:] = self.dof_pos[:, :self.num_dof] + actions
        self.apply_dof_targets(self.dof_targets)
        self.post_step()

    def post_step(self):
        """Base-frame velocities from the root_state TENSOR -- which has not been refreshed
        since the previous env step (Q2)."""
        n = self.env.num_envs
        rs = self.env.root_state
        if rs.is_cuda:
            # the four expressions below as one launch (csrc/shf_glue.hip: shf_base_frame_state)
            from shifu_amd import glue
            if glue.usable(rs, self.base_lin_vel, self.base_ang_vel, self.projected_gravity, self.gravity_vec):
                glue.base_frame_state(rs, self.root_indices, self.env.up_axis_idx, self.base_lin_vel, self.base_ang_vel,
                                      self.projected_gravity, self.gravity_vec)
                return
        self.gravity_vec[:] = to_torch(get_axis_params(-1., self.env.up_axis_idx), device=self.device).repeat((n, 1))
        quat = self.base_pose[:, 3:7]
        self.base_lin_vel[:] = quat_rotate_inverse(quat, self.env.root_state[self.root_indices, 7:10])
        self.base_ang_vel[:] = quat_rotate_inverse(quat, self.env.root_state[self.root_indices, 10:13])
        self.projected_gravity[:] = quat_rotate_inverse(quat, self.gravity_vec)

    def apply_force_on_base(self, force_tensor, pos_tensor=None):
        self.gym.apply_rigid_body_force_at_pos_tensors(
            self.sim, gymtorch.unwrap_tensor(force_tensor),
            gymtorch.unwrap_tensor(pos_tensor) if pos_tensor is not None else None)
