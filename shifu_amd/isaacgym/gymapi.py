"""`gymapi` facade: the names shifu and its examples take from `isaacgym.gymapi`
[EXT], re-hosted on the MI355X backend.

This is a naming facade, not a CUDA shim: `acquire_gym()` returns a `Gym` whose 69
methods used by the reference (103 call sites; shifu/gym/isaac_gym.py,
shifu/units/*.py, examples/*) record the scene while envs are created and, from
`prepare_sim` on, forward to the C ABI of include/shifu_amd.h through
shifu_amd.backend.Sim.  Graphics / viewer / camera calls are accepted and ignored
(no renderer on the MI355X path: SURVEY.md section 2 rows 7, 13).

Semantics kept from Isaac Gym:
  * state tensors are sim-owned, pointer-stable, viewed once (`acquire_*`), updated
    in place by `refresh_*`; user writes reach the solver only through
    `set_*_tensor[_indexed]` (int32 actor indices in sim domain);
  * `simulate` advances one dt; `apply_rigid_body_force_at_pos_tensors` acts on the
    next `simulate` only; efforts/targets persist.
"""
from __future__ import annotations

import copy
import ctypes as C
import os
from typing import Dict, List, Optional

import numpy as np

from .. import _abi

# -- constants ---------------------------------------------------------------
SIM_PHYSX = 1
SIM_FLEX = 2
UP_AXIS_Y, UP_AXIS_Z = 0, 1
DOF_MODE_NONE, DOF_MODE_POS, DOF_MODE_VEL, DOF_MODE_EFFORT = (_abi.DOF_MODE_NONE, _abi.DOF_MODE_POS,
                                                              _abi.DOF_MODE_VEL, _abi.DOF_MODE_EFFORT)
FROM_ASSET, COMPUTE_PER_VERTEX, COMPUTE_PER_FACE = 0, 1, 2
DOMAIN_ENV, DOMAIN_SIM, DOMAIN_ACTOR = 0, 1, 2
MESH_NONE, MESH_COLLISION, MESH_VISUAL, MESH_VISUAL_AND_COLLISION = 0, 1, 2, 3
IMAGE_COLOR, IMAGE_DEPTH, IMAGE_SEGMENTATION, IMAGE_OPTICAL_FLOW = 0, 1, 2, 3
KEY_ESCAPE, KEY_V = 256, 86
ENV_SPACE, LOCAL_SPACE, GLOBAL_SPACE = 0, 1, 2


# -- plain value types -------------------------------------------------------
class Vec3:
    def __init__(self, x=0.0, y=0.0, z=0.0):
        self.x, self.y, self.z = float(x), float(y), float(z)

    def __add__(self, o):
        return Vec3(self.x + o.x, self.y + o.y, self.z + o.z)

    def __sub__(self, o):
        return Vec3(self.x - o.x, self.y - o.y, self.z - o.z)

    def __iter__(self):
        return iter((self.x, self.y, self.z))

    def __repr__(self):
        return f"Vec3({self.x}, {self.y}, {self.z})"


class Quat:
    def __init__(self, x=0.0, y=0.0, z=0.0, w=1.0):
        self.x, self.y, self.z, self.w = float(x), float(y), float(z), float(w)

    def __iter__(self):
        return iter((self.x, self.y, self.z, self.w))


class Transform:
    def __init__(self, p: Optional[Vec3] = None, r: Optional[Quat] = None):
        self.p = p if p is not None else Vec3()
        self.r = r if r is not None else Quat()


class PhysXParams:
    def __init__(self):
        self.num_threads = 0
        self.use_gpu = True
        self.solver_type = 1
        self.num_position_iterations = 4
        self.num_velocity_iterations = 1
        self.contact_offset = 0.02
        self.rest_offset = 0.0
        self.bounce_threshold_velocity = 0.2
        self.max_depenetration_velocity = 100.0
        self.max_gpu_contact_pairs = 1024 * 1024
        self.default_buffer_size_multiplier = 2.0
        self.contact_collection = 2


class SimParams:
    def __init__(self):
        self.dt = 1.0 / 60.0
        self.substeps = 2
        self.up_axis = UP_AXIS_Y
        self.gravity = Vec3(0.0, -9.8, 0.0)
        self.use_gpu_pipeline = False
        self.physx = PhysXParams()


class AssetOptions:
    def __init__(self):
        self.fix_base_link = False
        self.default_dof_drive_mode = DOF_MODE_NONE
        self.disable_gravity = False
        self.collapse_fixed_joints = False
        self.flip_visual_attachments = False
        self.replace_cylinder_with_capsule = False
        self.mesh_normal_mode = FROM_ASSET
        self.use_physx_armature = True
        self.thickness = 0.02
        self.armature = 0.0
        self.angular_damping = 0.5
        self.linear_damping = 0.0
        self.max_angular_velocity = 64.0
        self.max_linear_velocity = 1000.0
        self.density = 1000.0


class PlaneParams:
    def __init__(self):
        self.normal = Vec3(0.0, 1.0, 0.0)
        self.distance = 0.0
        self.static_friction = 1.0
        self.dynamic_friction = 1.0
        self.restitution = 0.0


class HeightFieldParams:
    def __init__(self):
        self.column_scale = 1.0
        self.row_scale = 1.0
        self.vertical_scale = 1.0
        self.nbRows = 0
        self.nbColumns = 0
        self.transform = Transform()
        self.static_friction = 1.0
        self.dynamic_friction = 1.0
        self.restitution = 0.0


class TriangleMeshParams:
    def __init__(self):
        self.nb_vertices = 0
        self.nb_triangles = 0
        self.transform = Transform()
        self.static_friction = 1.0
        self.dynamic_friction = 1.0
        self.restitution = 0.0


class CameraProperties:
    def __init__(self):
        self.enable_tensors = False
        self.use_collision_geometry = False
        self.width, self.height = 1600, 900
        self.near_plane, self.far_plane = 0.0010000000474974513, 2000000.0
        self.horizontal_fov = 90.0


class RigidShapeProperties:
    def __init__(self, friction=1.0):
        self.friction = friction
        self.rolling_friction = 0.0
        self.torsion_friction = 0.0
        self.restitution = 0.0
        self.compliance = 0.0
        self.thickness = 0.0
        self.contact_offset = 0.0
        self.rest_offset = 0.0
        self.filter = 0


class RigidBodyProperties:
    def __init__(self, mass=0.0):
        self.mass = mass
        self.com = Vec3()
        self.invMass = 0.0 if mass == 0 else 1.0 / mass


# -- scene records -----------------------------------------------------------
class Asset:
    """Result of load_asset / create_box."""

    def __init__(self, kind, name, options, model=None, box_dim=None):
        self.kind, self.name, self.options = kind, name, options
        self.model = model  # CompiledModel for articulations
        self.box_dim = box_dim
        nshapes = model.blob.np if model is not None else 1
        self.shape_props = [RigidShapeProperties() for _ in range(max(nshapes, 1))]

    @property
    def num_bodies(self):
        return self.model.blob.nb if self.model is not None else 1

    @property
    def num_dofs(self):
        return self.model.blob.nd if self.model is not None else 0

    @property
    def body_dict(self) -> Dict[str, int]:
        return self.model.rigid_body_dict if self.model is not None else {"box": 0}


class _Actor:
    def __init__(self, asset, pose, name, sim_index):
        self.asset, self.pose, self.name, self.sim_index = asset, pose, name, sim_index
        self.friction = float(np.mean([p.friction for p in asset.shape_props]))
        self.mass_override: Optional[float] = None
        self.mass_scale = None      # articulations: factor per body on the asset's mass and inertia (set_actor_rigid_body_properties)
        self.dof_props = None


class Env:
    def __init__(self, sim, index):
        self.sim, self.index = sim, index
        self.actors: List[_Actor] = []


class _TensorHandle:
    """What acquire_*_tensor returns; gymtorch.wrap_tensor unwraps it."""

    def __init__(self, tensor):
        self.tensor = tensor


class SimHandle:
    def __init__(self, device_id, params: SimParams):
        self.device_id = device_id
        self.params = params
        self.envs: List[Env] = []
        self.terrain = None          # ("plane", friction) | ("heightfield", samples, hs, vs, border, friction)
        self.backend = None          # shifu_amd.backend.Sim after prepare_sim
        self.robot_asset: Optional[Asset] = None
        self.actors_per_env = 0
        self.num_actors = 0
        self.jacobian = None


class Gym:
    """The object `gymapi.acquire_gym()` returns."""

    # ---- lifecycle ---------------------------------------------------------
    # PhysX solver settings shifu's config sets (shifu/configs/env_config.py:46-58) that have no counterpart in this
    # backend's compliant, linearly-implicit contact model.  contact_offset and
    # max_depenetration_velocity ARE used (ShfSimParams); the ones below are accepted and ignored -- said once, at
    # create_sim, instead of only in a document.
    IGNORED_PHYSX_FIELDS = {
        "max_gpu_contact_pairs": "contact candidates are fixed per articulation (SHF_MAX_POINTS), no pair buffer",
        "default_buffer_size_multiplier": "no PhysX buffers",
    }
    # honoured (the velocity-level contact solve, include/shifu_amd.h): solver_type (1 = TGS -> SHF_SOLVER_TGS, the sub-stepped
    # sweeps, round 6; 0 = PGS -> SHF_SOLVER_PGS), num_position_iterations, num_velocity_iterations, contact_offset, rest_offset,
    # bounce_threshold_velocity, max_depenetration_velocity (shifu/configs/env_config.py:50-58).  A scene beyond the solve's
    # limits (more than 32 bodies + box actors, trees deeper than 8 levels, several articulations) keeps rounds 1-4's compliant
    # law and says so once at prepare_sim, with the reason.
    SOLVER_PHYSX_FIELDS = ("solver_type", "num_position_iterations", "num_velocity_iterations", "rest_offset", "bounce_threshold_velocity")
    _warned_physx = False
    _warned_compliant = False

    def create_sim(self, compute_device=0, graphics_device=0, physics_engine=SIM_PHYSX, params: SimParams = None):
        params = params or SimParams()
        if not Gym._warned_physx:
            Gym._warned_physx = True
            import warnings
            ref = PhysXParams()
            for name, why in Gym.IGNORED_PHYSX_FIELDS.items():
                val = getattr(params.physx, name, None)
                if val is not None and val != getattr(ref, name, None):
                    warnings.warn(f"shifu_amd: sim_params.physx.{name} = {val!r} is ignored by this backend ({why})",
                                  stacklevel=2)
        return SimHandle(compute_device, params)

    def add_ground(self, sim: SimHandle, params: PlaneParams):
        sim.terrain = ("plane", 0.5 * (params.static_friction + params.dynamic_friction))

    def add_heightfield(self, sim: SimHandle, samples, params: HeightFieldParams):
        # shifu passes nbRows = tot_cols / nbColumns = tot_rows with samples laid out (tot_rows, tot_cols),
        # x <-> first index, shifted by -border (isaac_gym.py:356-367)
        hs = np.ascontiguousarray(np.asarray(samples, dtype=np.int16).reshape(params.nbColumns, params.nbRows))
        sim.terrain = ("heightfield", hs, float(params.row_scale), float(params.vertical_scale),
                       float(-params.transform.p.x), 0.5 * (params.static_friction + params.dynamic_friction))

    def add_triangle_mesh(self, sim: SimHandle, vertices, triangles, params: TriangleMeshParams):
        """Trimesh terrain (the reference's effective A1 terrain, Q5).  The mesh is expected to come from
        convert_heightfield_to_trimesh; it is collided as that warped grid (ShfTerrain.warped) when
        the caller passes the height map and slope threshold it was made from; without that hint the map, its scales
        and the vertex shifts are recovered exactly from the vertex / triangle arrays (and any other kind of mesh is
        refused)."""
        if getattr(params, "height_samples", None) is not None:
            # hint set by shifu_amd's TerrainGymEnv._create_trimesh: the map the mesh came from
            from . import terrain_utils
            hs = np.ascontiguousarray(np.asarray(params.height_samples, dtype=np.int16))
            warp = None
            if hasattr(params, "slope_threshold"):
                warp = terrain_utils.trimesh_warp_map(hs, float(params.horizontal_scale), float(params.vertical_scale),
                                                      params.slope_threshold)
            sim.terrain = ("heightfield", hs, float(params.horizontal_scale), float(params.vertical_scale),
                           float(-params.transform.p.x), 0.5 * (params.static_friction + params.dynamic_friction), warp)
            return
        # no hint (e.g. the reference's own TerrainGymEnv._create_trimesh, isaac_gym.py:369-385, on this facade): the
        # generator keeps the vertices in grid order, so samples, scales and vertex shifts are recovered exactly
        from . import terrain_utils
        if params.nb_vertices * 3 != np.asarray(vertices).size or params.nb_triangles * 3 != np.asarray(triangles).size:
            raise ValueError("add_triangle_mesh: nb_vertices / nb_triangles do not match the arrays")
        hs, hscale, vscale, warp = terrain_utils.heightfield_from_trimesh(vertices, triangles)
        sim.terrain = ("heightfield", hs, hscale, vscale, float(-params.transform.p.x),
                       0.5 * (params.static_friction + params.dynamic_friction), warp)

    def load_asset(self, sim, rootpath, filename, options: AssetOptions = None):
        import os
        from ..model import compile_urdf, asset_path
        options = options or AssetOptions()
        path = os.path.join(rootpath, filename)
        if not os.path.exists(path):
            # the reference's asset tree does not travel; the physics-only URDFs ship in shifu_amd/assets
            base = os.path.basename(filename)
            alt = {"a1.urdf": "a1.urdf", "abb_rod_isaac.urdf": "abb_rod.urdf"}.get(base, base)
            path = asset_path(alt)
        # <mesh> colliders become convex hulls where the files are present (they do not ship with this repo: the vendored
        # URDFs are the physics-only reductions); against box actors only rounded shapes are tested, so the ABB rod gets its
        # documented capsule either way
        extra, boxes = (), ()
        if os.path.basename(path) in ("abb_rod.urdf", "abb_rod_isaac.urdf"):
            from ..abb_task import ROD_CAPSULE, abb_link_boxes
            extra = ROD_CAPSULE
            # box stand-ins for the links' mesh colliders (their files do not ship): only when the URDF brought no hulls
            boxes = abb_link_boxes() if os.path.basename(path) == "abb_rod.urdf" else ()
        # every shape of an env collides with every other actor's (create_actor(..., group = env, filter = 0),
        # units.py:68): link contacts on (ShfModel.link_collide; a scene without box actors never looks at it)
        model = compile_urdf(path, extra_spheres=extra, extra_boxes=boxes, link_contacts=True,
                             fix_base_link=bool(options.fix_base_link),
                             disable_gravity=bool(options.disable_gravity),
                             collapse_fixed_joints=bool(options.collapse_fixed_joints),
                             default_dof_drive_mode=int(options.default_dof_drive_mode),
                             armature=float(getattr(options, "armature", 0.0)),
                             density=float(getattr(options, "density", 1000.0)), meshes="auto")
        return Asset("articulation", filename, options, model=model)

    def create_box(self, sim, x, y, z, options: AssetOptions = None):
        return Asset("box", "box", options or AssetOptions(), box_dim=(float(x), float(y), float(z)))

    def create_env(self, sim: SimHandle, lower, upper, num_per_row):
        env = Env(sim, len(sim.envs))
        sim.envs.append(env)
        return env

    def create_actor(self, env: Env, asset: Asset, pose: Transform, name="", group=0, filter=0, seg_id=0):
        idx = env.sim.num_actors
        env.sim.num_actors += 1
        a = _Actor(asset, copy.deepcopy(pose), name, idx)
        # [EXT] collision filter: two shapes of one actor collide unless (filter_a & filter_b) != 0; shifu passes 0
        # (units.py:68), i.e. self-collision on
        a.self_collide = (int(filter) == 0)
        env.actors.append(a)
        return len(env.actors) - 1

    def get_actor_index(self, env: Env, actor_handle, domain=DOMAIN_SIM):
        return env.actors[actor_handle].sim_index if domain == DOMAIN_SIM else actor_handle

    # ---- asset / actor queries ---------------------------------------------
    def get_asset_rigid_body_count(self, asset): return asset.num_bodies
    def get_asset_dof_count(self, asset): return asset.num_dofs
    def get_asset_rigid_body_dict(self, asset): return dict(asset.body_dict)
    def get_asset_dof_properties(self, asset):
        return asset.model.dof_properties() if asset.model is not None else np.zeros(0)
    def get_asset_rigid_shape_properties(self, asset): return copy.deepcopy(asset.shape_props)
    def set_asset_rigid_shape_properties(self, asset, props): asset.shape_props = copy.deepcopy(list(props))
    def get_actor_rigid_body_dict(self, env, actor_handle): return dict(env.actors[actor_handle].asset.body_dict)
    def find_actor_rigid_body_handle(self, env, actor_handle, name):
        return env.actors[actor_handle].asset.body_dict.get(name, -1)
    def get_actor_rigid_shape_properties(self, env, actor_handle):
        a = env.actors[actor_handle]
        return [RigidShapeProperties(a.friction) for _ in a.asset.shape_props]
    def set_actor_rigid_shape_properties(self, env, actor_handle, props):
        env.actors[actor_handle].friction = float(np.mean([p.friction for p in props]))
    def get_actor_rigid_body_properties(self, env, actor_handle):
        a = env.actors[actor_handle]
        if a.asset.model is None:
            return [RigidBodyProperties(a.mass_override or 0.0)]
        sc = a.mass_scale
        return [RigidBodyProperties(a.asset.model.blob.mass[b] * (1.0 if sc is None else float(sc[b]))) for b in range(a.asset.num_bodies)]
    def set_actor_rigid_body_properties(self, env, actor_handle, props, recomputeInertia=True):
        a = env.actors[actor_handle]
        if a.asset.model is None:
            a.mass_override = float(props[0].mass)
            return True
        # per-env link masses of an articulation (shifu/units/units.py:104-110, recomputeInertia=True): kept on the actor as a
        # factor per body on the asset's mass and inertia tensor -- uniform density, so the tensor scales with the mass and the
        # centre of mass stays -- and uploaded by prepare_sim as SHF_T_BODY_MASS_SCALE.  A body welded to its parent carries no
        # mass of its own here (the compiled model folds it into the parent): an edit of such a body is refused, not dropped.
        blob = a.asset.model.blob
        scale = np.ones(a.asset.num_bodies, np.float32)
        for b, pr in enumerate(props):
            m0, m1 = float(blob.mass[b]), float(pr.mass)
            if abs(m1 - m0) <= 1e-9 * max(1.0, abs(m0)):
                continue
            if not recomputeInertia:
                raise NotImplementedError("set_actor_rigid_body_properties: recomputeInertia=False (mass without its inertia "
                                          "tensor) is not supported")
            if m0 <= 0.0 or m1 <= 0.0:
                raise NotImplementedError(f"set_actor_rigid_body_properties: body {b} is welded to its parent (its mass is part "
                                          "of the parent's here) or the new mass is not positive")
            scale[b] = m1 / m0
        a.mass_scale = scale if bool((scale != 1.0).any()) else None
        return True
    def set_actor_dof_properties(self, env, actor_handle, props):
        env.actors[actor_handle].dof_props = props
    def set_rigid_body_segmentation_id(self, *a, **k): pass
    def set_rigid_body_color(self, *a, **k): pass

    # ---- prepare ---------------------------------------------------------------
    def prepare_sim(self, sim: SimHandle):
        import torch
        from ..backend import Sim, default_sim_params
        from .._lib import lib
        p = sim.params
        # the contact solver: the physx settings as they are (SHF_SOLVER_PGS) where the velocity-level solve is built for the
        # scene, else the compliant law (SHIFU_AMD_SOLVER=compliant forces it: A/B runs)
        env0 = sim.envs[0]
        arts = [a for a in env0.actors if a.asset.kind == "articulation"]
        probe = copy.deepcopy(arts[0].asset.model.blob) if len(arts) == 1 else None
        if probe is not None:
            probe.self_collide = int(bool(getattr(arts[0], "self_collide", False)) and probe.npair > 0 and
                                     os.environ.get("SHIFU_AMD_SELF_COLLISION", "1") != "0")
        # physx.solver_type: 1 = temporal Gauss-Seidel, the reference's value (env_config.py:50) -> SHF_SOLVER_TGS; 0 -> SHF_SOLVER_PGS;
        # SHIFU_AMD_SOLVER = tgs | pgs | compliant overrides.  The compliant law is the fallback, with the actual reason said once.
        want = os.environ.get("SHIFU_AMD_SOLVER", "tgs" if int(getattr(p.physx, "solver_type", 1)) == 1 else "pgs")
        reason = None
        if want == "compliant":
            reason = "SHIFU_AMD_SOLVER=compliant"
        elif probe is None:
            reason = f"{len(arts)} articulations per env (the solve is built for one)"
        elif int(p.physx.num_position_iterations) < 1:
            reason = f"physx.num_position_iterations = {int(p.physx.num_position_iterations)} (the solve needs at least one)"
        elif lib().shf_model_pgs_supported(C.byref(probe), len(env0.actors) - 1) != 1:
            reason = (f"{probe.nb} bodies + {len(env0.actors) - 1} box actors, {probe.nlevels} tree levels: beyond the generic solve's limits "
                      "(32 bodies + box actors, 8 levels; csrc/shf_hard.h)")
        pgs = reason is None
        # max_contacts is not a PhysX field: the 8 deepest candidates per env everywhere by default (a walking A1 offers more in 1e-6 of
        # its sub-steps, profiles/r06_play_a1_*.json); SHIFU_AMD_MAX_CONTACTS raises it to up to 16 for an A1 on its own
        solver_kw = dict(solver=("tgs" if want == "tgs" else "pgs"), pos_iters=int(p.physx.num_position_iterations),
                         vel_iters=int(p.physx.num_velocity_iterations), rest_offset=float(p.physx.rest_offset),
                         bounce_threshold=float(p.physx.bounce_threshold_velocity),
                         max_contacts=int(os.environ.get("SHIFU_AMD_MAX_CONTACTS", "8"))) if pgs else {}
        if not pgs and not Gym._warned_compliant:
            Gym._warned_compliant = True
            import warnings
            warnings.warn("shifu_amd: this scene runs the compliant contact law of rounds 1-4 instead of the velocity-level solve that "
                          "honours sim_params.physx." + " / ".join(Gym.SOLVER_PHYSX_FIELDS) + ": " + reason, stacklevel=2)
        sim.solver = solver_kw["solver"] if pgs else "compliant"
        sp = default_sim_params(dt=p.dt, gravity=tuple(p.gravity),
                                max_depen_vel=min(float(p.physx.max_depenetration_velocity), 10.0),
                                contact_offset=float(p.physx.contact_offset), **solver_kw)
        dev = torch.device("cuda", sim.device_id if isinstance(sim.device_id, int) and sim.device_id >= 0 else 0)
        be = Sim(sp, dev)
        if sim.terrain is None or sim.terrain[0] == "plane":
            be.set_plane(sim.terrain[1] if sim.terrain else 1.0)
        else:
            _, hs, hscale, vscale, border, mu = sim.terrain[:6]
            warp = sim.terrain[6] if len(sim.terrain) > 6 else None
            if any(a.asset.kind != "articulation" for a in sim.envs[0].actors):
                warp = None          # box actors collide with the height-field form only
            be.set_heightfield(hs, hscale, vscale, border, mu, warp=warp)
        env0 = sim.envs[0]
        robots = [a for a in env0.actors if a.asset.kind == "articulation"]
        if len(robots) != 1 or env0.actors[0] is not robots[0]:
            raise NotImplementedError("each env needs exactly one articulated actor, created first "
                                      "(shifu assumes only the robot has DOFs: robot.py:51-52)")
        robot = robots[0]
        sim.robot_asset = robot.asset
        m = copy.deepcopy(robot.asset.model.blob)
        m.self_collide = int(bool(getattr(robot, "self_collide", False)) and m.npair > 0 and
                             os.environ.get("SHIFU_AMD_SELF_COLLISION", "1") != "0")
        if robot.dof_props is not None:
            for d in range(m.nd):
                m.drive_mode[d] = int(robot.dof_props["driveMode"][d])
                m.kp[d] = float(robot.dof_props["stiffness"][d])
                m.kd[d] = float(robot.dof_props["damping"][d])
                # [EXT] dof_props['damping'] is the joint's passive damping (it is what <dynamics damping> of the URDF
                # is imported into); PhysX keeps applying it when the drive is off (DOF_MODE_NONE / EFFORT), where only
                # the position spring ('stiffness') has no target to act on.  shifu writes dof_stiffness / dof_damping
                # into the props in every mode (robot.py:35-37), so the torque-controlled A1 carries 0.5 N m s/rad of
                # implicit joint damping next to its explicit PD -- without it PPO on the reference schedule never
                # leaves the standing optimum in this simulator (profiles/r02_walk_ablation.json).
                if m.drive_mode[d] in (DOF_MODE_NONE, DOF_MODE_EFFORT):
                    m.damping[d] = float(robot.dof_props["damping"][d])
                if "armature" in robot.dof_props.dtype.names:
                    m.armature[d] = float(robot.dof_props["armature"][d])
        be.set_articulation(m)
        for a in env0.actors[1:]:
            if a.asset.kind != "box":
                raise NotImplementedError("extra actors must be boxes (object.py:19-39)")
            b = _abi.ShfBoxDesc()
            b.dim[:] = a.asset.box_dim
            b.mass = a.mass_override if a.mass_override is not None else 0.0
            b.friction = a.friction
            b.fixed = int(bool(a.asset.options.fix_base_link) or b.mass == 0.0)
            b.pos[:] = tuple(a.pose.p)
            b.quat[:] = tuple(a.pose.r)
            be.add_box(b)
        sim.actors_per_env = len(env0.actors)
        n = len(sim.envs)
        be.finalize(n, 0)
        # the shipped arm + scene under TGS / PGS WITHOUT self-collision has a gym.simulate kernel of its own (k_sim_step_ws_hard); with
        # self-collision on -- what collision filter 0 means for the reference's AbbPushBox -- this is a no-op and the step stays on
        # the run-time-shaped kernel
        be.use_split_step()
        A = sim.actors_per_env
        root = torch.zeros(n * A, 13)
        fr = torch.ones(n)
        for e, env in enumerate(sim.envs):
            for k, a in enumerate(env.actors):
                root[e * A + k, :3] = torch.tensor(tuple(a.pose.p))
                root[e * A + k, 3:7] = torch.tensor(tuple(a.pose.r))
            fr[e] = env.actors[0].friction
        root = root.to(dev)
        for tid in (_abi.T_SIM_ROOT, _abi.T_ROOT_STATE):
            be.tensors[tid].copy_(root)
        be.tensors[_abi.T_FRICTION].copy_(fr.to(dev))
        if any(env.actors[0].mass_scale is not None for env in sim.envs):
            one = np.ones(env0.actors[0].asset.num_bodies, np.float32)
            be.set_body_mass_scale(np.stack([one if env.actors[0].mass_scale is None else env.actors[0].mass_scale for env in sim.envs]))
        sim.backend = be
        return True

    # ---- tensors ---------------------------------------------------------------
    def _t(self, sim, tid): return _TensorHandle(sim.backend.tensors[tid])
    def acquire_dof_state_tensor(self, sim): return self._t(sim, _abi.T_DOF_STATE)
    def acquire_actor_root_state_tensor(self, sim): return self._t(sim, _abi.T_ROOT_STATE)
    def acquire_rigid_body_state_tensor(self, sim): return self._t(sim, _abi.T_BODY_STATE)
    def acquire_net_contact_force_tensor(self, sim): return self._t(sim, _abi.T_CONTACT)
    def acquire_jacobian_tensor(self, sim, name): return self._t(sim, _abi.T_JACOBIAN)
    def acquire_force_sensor_tensor(self, sim): return None

    def refresh_dof_state_tensor(self, sim): sim.backend.refresh(_abi.REFRESH_DOF)
    def refresh_actor_root_state_tensor(self, sim): sim.backend.refresh(_abi.REFRESH_ROOT)
    def refresh_rigid_body_state_tensor(self, sim): sim.backend.refresh(_abi.REFRESH_BODY)
    def refresh_net_contact_force_tensor(self, sim): sim.backend.refresh(_abi.REFRESH_CONTACT)
    def refresh_jacobian_tensors(self, sim): sim.backend.refresh(_abi.REFRESH_JACOBIAN)
    def refresh_force_sensor_tensor(self, sim): pass

    def refresh_all_state_tensors(self, sim):
        """Not an Isaac Gym call: root, rigid-body, dof, Jacobian and net-contact-force tensors refreshed by ONE backend
        call (shifu's refresh_state issues the six refresh_* calls back to back, shifu/gym/isaac_gym.py:145-154)."""
        sim.backend.refresh(_abi.REFRESH_ALL)
    def refresh_mass_matrix_tensors(self, sim): pass

    # ---- stepping --------------------------------------------------------------
    def simulate(self, sim): sim.backend.step()
    def fetch_results(self, sim, wait=True): pass

    def set_dof_actuation_force_tensor(self, sim, t): sim.backend.set_dof_command(_abi.T_EFFORT, t); return True
    def set_dof_position_target_tensor(self, sim, t): sim.backend.set_dof_command(_abi.T_POS_TARGET, t); return True
    def set_dof_velocity_target_tensor(self, sim, t): sim.backend.set_dof_command(_abi.T_VEL_TARGET, t); return True

    def apply_rigid_body_force_at_pos_tensors(self, sim, force, pos=None, space=ENV_SPACE):
        """force (N*B, 3) on every rigid body, at its centre of mass (pos None) or at the points pos (N*B, 3) -- both in env
        space, which here is the frame of the state tensors: the facade places every env at the sim's origin (create_env
        ignores the grid spacing -- envs never interact -- and TerrainGymEnv sets it to 0 anyway, isaac_gym.py:301)."""
        if space != ENV_SPACE:
            raise NotImplementedError("forces are taken in env (= world-aligned) axes only, the default the reference uses")
        sim.backend.apply_body_force(force, pos)
        return True

    def set_actor_root_state_tensor_indexed(self, sim, root, idx, n):
        sim.backend.commit_root_indexed(root, idx[:n]); return True
    def set_actor_root_state_tensor(self, sim, root): sim.backend.commit_root_all(root); return True
    def set_dof_state_tensor_indexed(self, sim, dof, idx, n):
        sim.backend.commit_dof_indexed(dof, idx[:n]); return True
    def set_dof_position_target_tensor_indexed(self, sim, tgt, idx, n):
        sim.backend.set_pos_target_indexed(tgt, idx[:n]); return True

    # ---- teardown ----------------------------------------------------------------
    def destroy_env(self, env): pass
    def destroy_viewer(self, viewer): pass
    def destroy_camera_sensor(self, *a): pass
    def destroy_sim(self, sim):
        if sim is not None and sim.backend is not None:
            sim.backend.destroy()
            sim.backend = None

    # ---- graphics: accepted and ignored (no renderer on this path) ----------------
    def create_viewer(self, *a, **k): return None
    def subscribe_viewer_keyboard_event(self, *a, **k): pass
    def viewer_camera_look_at(self, *a, **k): pass
    def query_viewer_has_closed(self, viewer): return False
    def query_viewer_action_events(self, viewer): return []
    def step_graphics(self, sim): pass
    def draw_viewer(self, *a, **k): pass
    def sync_frame_time(self, sim): pass
    def poll_viewer_events(self, viewer): pass
    def set_light_parameters(self, *a, **k): pass
    def render_all_camera_sensors(self, sim): pass
    def start_access_image_tensors(self, sim): pass
    def end_access_image_tensors(self, sim): pass
    def create_camera_sensor(self, *a, **k):
        raise NotImplementedError("rasterised camera sensors are out of scope on the MI355X path (SURVEY.md row 7)")
    get_camera_image_gpu_tensor = set_camera_location = set_camera_transform = create_camera_sensor


_GYM = None


def acquire_gym() -> Gym:
    global _GYM
    if _GYM is None:
        _GYM = Gym()
    return _GYM


Sim = SimHandle  # `gymapi.Sim` is used as a type annotation by the reference (units.py:13)
