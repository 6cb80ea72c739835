"""AbbPushBox (BASELINE config 5) with the whole ShifuVecEnv.step in one HIP launch:
in-kernel damped-least-squares IK, 5+1 sub-steps of 20 ms, refresh of body states / Jacobian /
contacts, termination, rewards, on-device reset of arm, cube and goal, 6-dim observation.
Same buffer names as the hook-based examples/abb_pushbox_vision/a_prior_stage.AbbPushBox."""
from __future__ import annotations

import torch

from .. import _abi
from ..abb_task import ABB_BASE_POS, abb_boxes, abb_model, abb_task_params
from ..backend import AbbTask, Sim, default_sim_params

REWARD_NAMES = ["reward_reaching", "reward_success"]   # build_reward_functions order, a_prior_stage.py:115-119


class FusedAbbEnv:
    def __init__(self, num_envs: int = 4096, device="cuda:0", seed: int = 42, rank: int = 0, world_size: int = 1,
                 group: int = None, dt: float = 0.02, decimation: int = 5, episode_length_s: float = 20.0,
                 extra_boxes=(), link_contacts: bool = None, mapping: str = None, solver: str = None,
                 link_shapes: str = "box", face_manifold: bool = None):
        self.device = torch.device(device)
        self.num_envs = num_envs
        self.env_id_offset = rank * num_envs
        self.dt = dt * decimation                                         # isaac_gym.py:26
        # link_contacts: the arm's links (box stand-ins for their mesh colliders) and the rod also collide with the table,
        # the cube and the goal pad (ShfModel.link_collide) -- the reference's scene (every shape of an env
        # collides, units.py:68), what `AbbPushBox` through the gym facade does too, and the default here since round 4.
        # False: the rod against the cube only (the scene benchmarked in rounds 1-3); also what the 'chain' mapping
        # is compiled for, so asking for it without saying otherwise means the rod-only scene ('split' exists for both).
        if link_contacts is None:
            link_contacts = mapping != "chain"
        self.link_contacts = bool(link_contacts)
        # link_shapes: "box" = the links as the bounding boxes of their mesh colliders (rounds 3-5; the kernels compiled for this
        # scene); "hull" = as the convex hulls of the reference's collision meshes (abb_rod_isaac.urdf:38-113), reduced to <= 32
        # vertices, through the convex narrow phase (csrc/shf_hull.h) on the run-time-shaped kernels.  face_manifold
        # (ShfScene.flags): box pairs that touch without a vertex or an edge crossing -- an edge or a face lying flat on a face --
        # get the clipped face manifold too; on with the hulls unless said otherwise.
        if link_shapes == "hull" and not self.link_contacts:
            raise ValueError("FusedAbbEnv: link_shapes='hull' needs link_contacts=True")
        self.link_shapes = link_shapes
        self.face_manifold = bool(link_shapes == "hull" if face_manifold is None else face_manifold)
        ext = link_shapes == "hull" or self.face_manifold
        if ext and mapping not in (None, "body"):
            raise ValueError("FusedAbbEnv: hulls / the face manifold run on mapping='body' (the run-time-shaped kernels)")
        if ext:
            mapping = "body"
        self.cm = abb_model(link_contacts=link_contacts, link_shapes=link_shapes)
        # solver: "tgs" / "pgs" = the velocity-level contact solve with the reference's PhysX settings (env_config.py:50-58) -- the
        # default.  In the shipped scene it runs on k_abb_step_ws_hard (mapping 'split', 16 lanes per env: arm wave + box wave for the
        # free solve and the candidates, the solve regrouped at 32 lanes per env; round 6); otherwise (extra boxes, hulls, or
        # mapping='body') on the run-time-shaped body-per-lane kernel at 32 lanes per env (csrc/shf_hard.h).  "compliant" = rounds
        # 1-4's spring-damper law on the kernels compiled for this scene.
        # Asking, without naming a solver, for a lane mapping / width that only the compliant kernels have selects them.  (Decided
        # from what the caller passed, before the defaults for `group` and `mapping` are filled in below.)
        if solver is None:
            solver = "tgs" if (mapping in (None, "body") and group in (None, 32)) else "compliant"      # physx.solver_type = 1 (env_config.py:50); "pgs": solver_type = 0
        if group is None:
            # 16 lanes per env: sixteen envs per workgroup share one LDS copy of the model, and 4096 envs are resident at once
            # -- with link contacts too, now that only the free box owns corner slots (9.2 KB of LDS per env, was 12.6)
            group = 16
        if mapping is None:
            # 'chain': the arm's kinematic / ABA recursions on one lane (csrc/shf_arm.h), compiled for the shipped arm in
            # the shipped scene; 'split' (16 lanes per env): the same with the arm and the box actors of an env on
            # different waves of one workgroup (k_abb_step_ws, the fastest: 0.094 vs 0.104 ms at 4096 envs); 'body': the
            # level-by-level sub-step (any arm, any boxes).  Identical results.
            # With link contacts 'split' exists too (the link passes run on the box wave; 0.166 vs 0.186 ms for 'body').
            scene = not extra_boxes
            mapping = "split" if (scene and group == 16) else "chain" if (scene and not link_contacts and group == 32) else "body"
        self.solver = solver
        if solver in ("pgs", "tgs") and not (mapping == "split" and group == 16):
            group, mapping = 32, "body"
        if mapping == "chain" and self.link_contacts:
            raise ValueError("FusedAbbEnv: mapping='chain' is compiled for the rod-only scene; with link_contacts=True use "
                             "mapping='split' (16 lanes per env) or 'body'")
        self.mapping = mapping
        self.sim_params = default_sim_params(dt=dt, solver=solver)
        self.sim = Sim(self.sim_params, self.device)
        self.sim.set_plane(1.0)
        self.sim.set_articulation(self.cm.blob)
        if self.cm.hulls is not None:
            self.sim.set_hulls(self.cm.hulls)
        if self.face_manifold:
            self.sim.set_scene_flags(_abi.SCENE_FACE_MANIFOLD)
        # extra_boxes: further (fixed) box actors after table / cube / goal -- the scene then no longer matches the
        # compile-time ABB scene and the step runs on the run-time-shaped kernel instantiation (tests use this)
        self.boxes = abb_boxes() + list(extra_boxes)
        for b in self.boxes:
            self.sim.add_box(b)
        self.sim.finalize(num_envs, self.env_id_offset, group=group, mapping=mapping)
        self.task_params = abb_task_params(self.cm, dt=dt, decimation=decimation, episode_length_s=episode_length_s,
                                           seed=seed)
        for k, b in enumerate(extra_boxes):
            self.task_params.actor_default[4 + k][:] = list(b.pos) + list(b.quat)
        self.task = AbbTask(self.sim, self.task_params)
        T, S = self.task.tensors, self.sim.tensors
        self.num_obs, self.num_privileged_obs, self.num_actions = 6, None, 3
        self.max_episode_length = self.task_params.max_episode_length
        self.max_episode_length_s = episode_length_s
        self.obs_buf, self.privileged_obs_buf, self.rew_buf = T[_abi.ABB_OBS], None, T[_abi.ABB_REW]
        self.reset_buf = T[_abi.ABB_RESET].view(torch.bool)
        self.time_out_buf = T[_abi.ABB_TIMEOUT].view(torch.bool)
        self.success_buf = T[_abi.ABB_SUCCESS].view(torch.bool)
        self.episode_length_buf = T[_abi.ABB_EP_LEN]
        self.actions = T[_abi.ABB_ACTIONS]
        self.episode_rewards = {n: T[_abi.ABB_REW_SUMS][k] for k, n in enumerate(REWARD_NAMES)}
        self.dof_state, self.root_state = S[_abi.T_DOF_STATE], S[_abi.T_ROOT_STATE]
        self.body_state, self.contact_state, self.jacobian = S[_abi.T_BODY_STATE], S[_abi.T_CONTACT], S[_abi.T_JACOBIAN]
        # contacts dropped at the per-env limits since the tensor was last cleared (link contacts beyond
        # SHF_MAX_LINK_CONTACTS); the live device tensor, (N,) int32
        self.extras = {"dropped_contacts": S[_abi.T_DROPPED]}
        self.reward_names = REWARD_NAMES
        # spawn poses + the tensors Isaac Gym would show after create_actor/prepare_sim
        A = 1 + len(self.boxes)
        root = torch.zeros(num_envs * A, 13, device=self.device)
        root[:, 6] = 1.0
        root[0::A, :3] = torch.tensor(ABB_BASE_POS, device=self.device)
        for k, b in enumerate(self.boxes):
            root[1 + k::A, :3] = torch.tensor(list(b.pos), device=self.device)
        S[_abi.T_ROOT_STATE].copy_(root)
        S[_abi.T_SIM_ROOT].copy_(root)
        self.sim.refresh(_abi.REFRESH_BODY | _abi.REFRESH_JACOBIAN)   # zero-config poses, as after prepare_sim
        self.task.reset_all()

    def step(self, actions: torch.Tensor):
        self.task.step(actions)
        return self._after_step()

    def step_random(self):
        """run_policy('random') in one launch: U(-1, 1) actions drawn inside the fused step (A1 counterpart:
        FusedA1Env.step_random)."""
        self.task.step_random()
        return self._after_step()

    def _after_step(self):
        st = self.task.tensors[_abi.ABB_STATS][-1]
        self.extras["episode"] = {REWARD_NAMES[0]: st[4], REWARD_NAMES[1]: st[5], "success_rate": st[6]}
        self.extras["episode_sums"] = st[:4]
        self.extras["time_outs"] = self.time_out_buf
        return self.obs_buf, self.privileged_obs_buf, self.rew_buf, self.reset_buf, self.extras

    def reset(self):
        self.task.reset_all()
        obs, priv, _, _, _ = self.step(torch.zeros(self.num_envs, 3, device=self.device))
        return obs, priv

    def get_observations(self):
        return self.obs_buf

    def get_privileged_observations(self):
        return None

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
