"""Environment configs (reference shifu/configs/env_config.py:5-102), re-hosted on
the `gymapi` facade: the class attributes are the reference's parameter values
(cited per line); `_init_sim_params` copies `sim` / `sim.physx` onto a
gymapi.SimParams exactly like the reference (:22-36) but with getattr, not eval.
The PhysX solver settings are carried for source compatibility; the MI355X contact
model reads only dt, gravity and max_depenetration_velocity."""
from shifu_amd.isaacgym import gymapi

from .base_config import BaseConfig


class BaseEnvConfig(BaseConfig):
    num_envs = 5
    num_obs = 10
    num_privileged_obs = None   # critic obs for asymmetric training
    num_actions = 3
    num_actions_history = None
    send_timeouts = True        # extras["time_outs"] (env.py:129-130)
    episode_length_s = 20

    spacing = 1.
    device = 'cuda:0'
    physics_engine = gymapi.SIM_PHYSX

    def __init__(self):
        self._init_sim_params()
        super().__init__()

    def _init_sim_params(self):
        sim_params = gymapi.SimParams()
        for attr in dir(self.sim):
            if '__' in attr:
                continue
            if attr == 'physx':
                for pattr in dir(self.sim.physx):
                    if '__' not in pattr:
                        setattr(sim_params.physx, pattr, getattr(self.sim.physx, pattr))
            else:
                setattr(sim_params, attr, getattr(self.sim, attr))
        self.sim_params = sim_params

    class sim:
        dt = 0.005                                  # env_config.py:40
        substeps = 1
        up_axis = gymapi.UP_AXIS_Z
        gravity = gymapi.Vec3(0.0, 0.0, -9.81)
        use_gpu_pipeline = True

        class physx:                                # env_config.py:46-58
            num_threads = 10
            use_gpu = True
            solver_type = 1
            num_position_iterations = 8
            num_velocity_iterations = 1
            contact_offset = 0.01
            rest_offset = 0.0
            bounce_threshold_velocity = 0.5
            max_depenetration_velocity = 1.0
            max_gpu_contact_pairs = 2 ** 23
            default_buffer_size_multiplier = 5

    class debug:
        headless = False
        camera_pos = [1., -1., 1.]
        camera_lookat = [0, 0, 0]
        enable_viewer_sync = True
        viewer_attach_robot_env_idx = None

    class normalization:
        clip_observations = 100.
        clip_actions = 1.

    class control:
        decimation = 4


class TerrainEnvConfig(BaseEnvConfig):
    class terrain:                                  # env_config.py:78-102
        mesh_type = 'trimesh'                       # none | plane | heightfield | trimesh
        horizontal_scale = 0.1
        vertical_scale = 0.005
        border_size = 25
        static_friction = 1.0
        dynamic_friction = 1.0
        restitution = 0.
        measure_heights = True
        measured_points_x = [-0.8, -0.7, -0.6, -0.5, -0.4, -0.3, -0.2, -0.1, 0., 0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7,
                             0.8]
        measured_points_y = [-0.5, -0.4, -0.3, -0.2, -0.1, 0., 0.1, 0.2, 0.3, 0.4, 0.5]
        selected = False
        terrain_kwargs = None
        terrain_length = 8.
        terrain_width = 8.
        num_rows = 10
        num_cols = 20
        terrain_proportions = [0.1, 0.1, 0.35, 0.25, 0.2]   # smooth slope, rough slope, stairs up, stairs down, discrete
        slope_treshold = 0.75
        curriculum = True
        max_init_terrain_level = 5
