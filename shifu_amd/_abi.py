"""ctypes mirrors of the PODs in include/shifu_amd.h (keep in lock-step)."""
import ctypes as C

SHF_ABI_VERSION = 15
MAP_BODY, MAP_CHAIN, MAP_CHAIN_SPLIT = 0, 1, 2   # shf_sim_set_mapping
MAX_BODIES = 32
MAX_DOFS = 32
MAX_POINTS = 176
MAX_BOXES = 4
MAX_SPHERES = 8
MAX_CAPSULES = 16
MAX_PAIRS = 96
MAX_SELF_CONTACTS = 8
MAX_ABOX = 16
MAX_LINK_CONTACTS = 16
MAX_HARD_CONTACTS = 16
MAX_HULLS, HULL_MAX_VERTS, HULL_MAX_FACES, HULL_MAX_EDGES, HULL_MAX_FACE_VERTS, HULL_MAX_LOOP = 8, 32, 40, 64, 8, 160
SCENE_FACE_MANIFOLD = 1
CONTACT_HIST_BINS = 26
SOLVER_COMPLIANT, SOLVER_PGS, SOLVER_TGS = 0, 1, 2

JOINT_ROOT, JOINT_REVOLUTE, JOINT_PRISMATIC, JOINT_WELD = 0, 1, 2, 3
DOF_MODE_NONE, DOF_MODE_POS, DOF_MODE_VEL, DOF_MODE_EFFORT = 0, 1, 2, 3

i32 = C.c_int32
f32 = C.c_float


class ShfModel(C.Structure):
    _fields_ = [
        ("nb", i32), ("nd", i32), ("np", i32), ("nlevels", i32),
        ("fixed_base", i32), ("gravity_on", i32), ("nsph", i32), ("nklevels", i32),
        ("parent", i32 * MAX_BODIES), ("jtype", i32 * MAX_BODIES), ("dof", i32 * MAX_BODIES),
        ("level", i32 * MAX_BODIES), ("dyn", i32 * MAX_BODIES), ("klevel", i32 * MAX_BODIES),
        ("child_start", i32 * MAX_BODIES), ("child_count", i32 * MAX_BODIES), ("child_list", i32 * MAX_BODIES),
        ("pt_start", i32 * MAX_BODIES), ("pt_count", i32 * MAX_BODIES),
        ("tpos", (f32 * 3) * MAX_BODIES), ("trot", (f32 * 9) * MAX_BODIES), ("axis", (f32 * 3) * MAX_BODIES),
        ("mass", f32 * MAX_BODIES), ("com", (f32 * 3) * MAX_BODIES), ("inertia", (f32 * 6) * MAX_BODIES),
        ("lower", f32 * MAX_DOFS), ("upper", f32 * MAX_DOFS), ("vel_limit", f32 * MAX_DOFS),
        ("effort", f32 * MAX_DOFS), ("kp", f32 * MAX_DOFS), ("kd", f32 * MAX_DOFS),
        ("armature", f32 * MAX_DOFS), ("damping", f32 * MAX_DOFS),
        ("drive_mode", i32 * MAX_DOFS), ("dof_body", i32 * MAX_DOFS),
        ("pt_body", i32 * MAX_POINTS), ("pt_pos", (f32 * 3) * MAX_POINTS), ("pt_radius", f32 * MAX_POINTS),
        ("pt_eval", i32 * MAX_POINTS), ("pt_slot", i32 * MAX_POINTS),
        ("sph_body", i32 * MAX_SPHERES), ("sph_pos", (f32 * 3) * MAX_SPHERES), ("sph_seg", (f32 * 3) * MAX_SPHERES), ("sph_radius", f32 * MAX_SPHERES), ("sph_part", i32 * MAX_SPHERES),
        ("self_collide", i32), ("ncap", i32), ("npair", i32), ("neval", i32),
        ("cap_body", i32 * MAX_CAPSULES), ("cap_a", (f32 * 3) * MAX_CAPSULES), ("cap_b", (f32 * 3) * MAX_CAPSULES),
        ("cap_radius", f32 * MAX_CAPSULES), ("pair_a", C.c_uint8 * MAX_PAIRS), ("pair_b", C.c_uint8 * MAX_PAIRS),
        ("link_collide", i32), ("nabox", i32), ("bounds_ok", i32), ("nhull", i32), ("abox_body", i32 * MAX_ABOX),
        ("abox_pos", (f32 * 3) * MAX_ABOX), ("abox_rot", (f32 * 9) * MAX_ABOX), ("abox_half", (f32 * 3) * MAX_ABOX),
        ("bbox", (f32 * 6) * MAX_BODIES),
        ("lc_range", (C.c_int16 * 4) * MAX_BODIES), ("lc_pt", C.c_int16 * MAX_POINTS), ("lc_abox", C.c_int16 * MAX_ABOX),
    ]


class ShfBoxDesc(C.Structure):
    _fields_ = [("dim", f32 * 3), ("mass", f32), ("friction", f32), ("fixed", i32),
                ("pos", f32 * 3), ("quat", f32 * 4)]


class ShfScene(C.Structure):
    _fields_ = [("nboxes", i32), ("flags", i32), ("pad", i32 * 2), ("box", ShfBoxDesc * MAX_BOXES)]


class ShfHull(C.Structure):
    _fields_ = [("body", i32), ("nv", i32), ("nf", i32), ("ne", i32), ("centroid", f32 * 3), ("pad0", f32),
                ("vert", (f32 * 3) * HULL_MAX_VERTS), ("plane", (f32 * 4) * HULL_MAX_FACES),
                ("face_start", C.c_uint8 * HULL_MAX_FACES), ("face_count", C.c_uint8 * HULL_MAX_FACES),
                ("face_loop", C.c_uint8 * HULL_MAX_LOOP), ("edge", (C.c_uint8 * 4) * HULL_MAX_EDGES)]


class ShfHullSet(C.Structure):
    _fields_ = [("nhull", i32), ("pad", i32 * 3), ("hull", ShfHull * MAX_HULLS)]


class ShfSimParams(C.Structure):
    _fields_ = [("dt", f32), ("gravity", f32 * 3), ("contact_k", f32), ("contact_d", f32),
                ("friction_vel", f32), ("limit_k", f32), ("limit_d", f32),
                ("angular_damping", f32), ("max_ang_vel", f32), ("max_depen_vel", f32), ("contact_offset", f32),
                ("solver", i32), ("pos_iters", i32), ("vel_iters", i32), ("max_contacts", i32),
                ("rest_offset", f32), ("bounce_threshold", f32), ("restitution", f32), ("erp", f32)]


class ShfTerrain(C.Structure):
    _fields_ = [("rows", i32), ("cols", i32), ("hscale", f32), ("vscale", f32), ("border", f32),
                ("friction", f32), ("warped", i32), ("nz_min", f32)]


class ShfA1TaskParams(C.Structure):
    _fields_ = [
        ("decimation", i32), ("extra_substep", i32), ("num_history", i32), ("num_height_points", i32),
        ("base_body", i32), ("curriculum", i32), ("max_terrain_level", i32), ("num_terrain_cols", i32),
        ("action_scale", f32), ("clip_actions", f32), ("clip_obs", f32), ("max_episode_length", f32),
        ("max_episode_length_s", f32), ("env_length", f32), ("max_push_force", f32), ("spawn_xy", f32),
        ("default_pos", f32 * 3), ("default_quat", f32 * 4),
        ("default_dof_pos", f32 * MAX_DOFS), ("p_gain", f32 * MAX_DOFS), ("d_gain", f32 * MAX_DOFS),
        ("num_leg_bodies", i32), ("leg_bodies", i32 * MAX_BODIES),
        ("seed", C.c_uint64),
    ]


class ShfAbbTaskParams(C.Structure):
    _fields_ = [
        ("decimation", i32), ("extra_substep", i32), ("ee_body", i32), ("cube_actor", i32), ("goal_actor", i32), ("pad0", i32),
        ("clip_actions", f32), ("clip_obs", f32), ("max_episode_length", f32), ("max_episode_length_s", f32),
        ("ee_velocity", f32), ("env_dt", f32), ("ik_damping", f32), ("pad1", f32),
        ("min_ee_pos", f32 * 3), ("max_ee_pos", f32 * 3), ("target_quat", f32 * 4),
        ("default_dof_pos", f32 * MAX_DOFS), ("actor_default", (f32 * 7) * (MAX_BOXES + 1)),
        ("cube_lo", f32 * 3), ("cube_hi", f32 * 3), ("goal_lo", f32 * 3), ("goal_hi", f32 * 3),
        ("seed", C.c_uint64),
    ]


(ABB_ACTIONS, ABB_OBS, ABB_REW, ABB_RESET, ABB_TIMEOUT, ABB_SUCCESS, ABB_EP_LEN, ABB_REW_SUMS, ABB_DOF_TARGETS,
 ABB_RESET_COUNT, ABB_DONE_SUMS, ABB_STATS, ABB_PARAMS, ABB_STATS_ACC, ABB_COUNT) = range(15)

# tensor ids (shf_sim_*)
T_DOF_STATE, T_ROOT_STATE, T_BODY_STATE, T_CONTACT, T_JACOBIAN, T_SIM_DOF, T_SIM_ROOT, T_EFFORT, \
    T_POS_TARGET, T_VEL_TARGET, T_BODY_FORCE, T_FRICTION, T_HEIGHTS, T_MODEL, T_SIM_CONTACT, T_SCENE, T_DROPPED, T_BODY_FORCE_POS, T_BODY_MASS_SCALE, T_HULLS, T_CONTACT_HIST, T_COUNT = range(22)

REFRESH_DOF, REFRESH_ROOT, REFRESH_BODY, REFRESH_CONTACT, REFRESH_JACOBIAN, REFRESH_ALL = 1, 2, 4, 8, 16, 31

# tensor ids (shf_a1_*)
(A1_ACTIONS, A1_OBS, A1_REW, A1_RESET, A1_TIMEOUT, A1_EP_LEN, A1_COMMAND, A1_HISTORY, A1_REW_SUMS, A1_TORQUES,
 A1_BASE_VEL, A1_HEIGHTS, A1_HPOINTS, A1_PUSH, A1_ORIGINS, A1_LEVELS, A1_TYPES, A1_TORIGINS, A1_RESET_COUNT,
 A1_DONE_SUMS, A1_STATS, A1_PARAMS, A1_STATS_ACC, A1_COUNT) = range(24)

DTYPE_F32, DTYPE_I32, DTYPE_I16, DTYPE_U8, DTYPE_I64 = range(5)
