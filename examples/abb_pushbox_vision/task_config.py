"""ABB push-box, prior-information stage: configs (reference
examples/abb_pushbox_vision/task_config.py:13-117, prior stage only -- the vision stages
need a camera renderer and are out of scope)."""
from shifu_amd.configs import ArmRobotActorConfig, BaseEnvConfig, BoxActorConfig, PPOConfig

ASSET_ROOT = "./asset"


class TableConfig(BoxActorConfig):
    root_dir = ASSET_ROOT
    name = "table"
    default_pos = [0, 0, 0.05]
    default_quat = [0, 0, 0, 1]
    box_dim = [0.6, 0.6, 0.1]
    mass = 0.
    color = [0.8, 0.8, 0.8]

    class asset_options(BoxActorConfig.asset_options):
        fix_base_link = True


class PushBoxConfig(BoxActorConfig):
    root_dir = ASSET_ROOT
    name = "box"
    default_pos = [0, 0, 0.125]
    default_quat = [0, 0, 0, 1]
    box_dim = [0.05, 0.05, 0.05]
    mass = 0.1
    color = [.25, .65, .3]


class GoalBoxConfig(BoxActorConfig):
    root_dir = ASSET_ROOT
    name = "goal"
    default_pos = [0, 0, 0.1]
    default_quat = [0, 0, 0, 1]
    box_dim = [0.08, 0.08, 0.002]
    mass = 0.
    color = [0.8, 0., 0.]

    class asset_options(BoxActorConfig.asset_options):
        fix_base_link = True


class AbbRobotConfig(ArmRobotActorConfig):
    root_dir = ASSET_ROOT
    name = "AbbRobot-VacuumRod"
    urdf_filename = "urdf/abb_rod_description/urdf/abb_rod_isaac.urdf"
    end_effector_names = ['tip0']
    default_pos = [-0.48, 0, 0]
    default_quat = [0, 0, 0, 1]
    default_dof_pos = [0., 0.6437, 0.1748, 0., 0.7541, 0.]
    dof_stiffness = [800] * 6
    dof_damping = [40] * 6
    end_effector_velocity = 0.2     # m/s
    default_ee_quat = [0., 1., 0., 0]
    min_ee_pos = [-0.2, -0.2, 0.11]
    max_ee_pos = [0.2, 0.2, 0.14]


class PriorStageEnvConfig(BaseEnvConfig):
    num_envs = 3000
    num_obs = 6
    num_privileged_obs = None
    num_actions = 3
    send_timeouts = True
    episode_length_s = 20.

    class sim(BaseEnvConfig.sim):
        dt = 0.02

    class control(BaseEnvConfig.control):
        decimation = int(0.1 / 0.02)

    class debug(BaseEnvConfig.debug):
        headless = True

    class normalization(BaseEnvConfig.normalization):
        clip_observations = 10.
        clip_actions = 1.


class PriorStagePPOConfig(PPOConfig):
    seed = 42
    runner_class_name = "AbbPushBoxTask"

    class policy(PPOConfig.policy):
        init_noise_std = 1.0
        actor_hidden_dims = [512, 256, 128]
        critic_hidden_dims = [512, 256, 128]
        activation = 'elu'

    class algorithm(PPOConfig.algorithm):
        schedule = 'adaptive'

    class runner(PPOConfig.runner):
        num_steps_per_env = 24
        max_iterations = 1500
        save_interval = 50
        experiment_name = 'ppo_PushBox'
        run_name = 'AbbPushBox_PriorStage'
        load_run = -1
        checkpoint = -1
        resume_path = None
