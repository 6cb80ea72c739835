#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own Python on synthetic CPU
tensors (vectors G1-G11).  Runs only in the build container, where
/root/reference exists; the reference never travels -- only these input/output arrays do.

The reference cannot be imported as is: `isaacgym`, `rsl_rl` (and cv2, pybullet, ...) are
absent.  They are stubbed in sys.modules: `isaacgym.*` by this repo's naming facade
(shifu_amd.isaacgym: pure-Python value types and the [EXT] torch_utils / terrain_utils
helpers restated from SURVEY appendices C/D), everything else by empty modules.  What the
vectors pin is therefore the reference's *glue* (shifu/gym/env.py, shifu/gym/isaac_gym.py,
shifu/utils/*.py, examples/*), given those helper definitions.

    PYTHONDONTWRITEBYTECODE=1 python tools/make_golden.py
"""
import os
import sys
import types

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, ROOT)

import numpy as np
import torch

OUT = os.path.join(ROOT, "tests", "golden")


def install_stubs():
    import shifu_amd.isaacgym as fac
    sys.modules["isaacgym"] = fac
    for sub in ("gymapi", "gymtorch", "gymutil", "torch_utils", "terrain_utils"):
        sys.modules["isaacgym." + sub] = getattr(fac, sub)
    rsl = types.ModuleType("rsl_rl")
    rsl.env = types.ModuleType("rsl_rl.env")
    rsl.env.VecEnv = type("VecEnv", (), {})
    rsl.runners = types.ModuleType("rsl_rl.runners")
    rsl.runners.OnPolicyRunner = type("OnPolicyRunner", (), {})
    sys.modules.update({"rsl_rl": rsl, "rsl_rl.env": rsl.env, "rsl_rl.runners": rsl.runners})

    class _Any(types.ModuleType):
        def __getattr__(self, k):
            if k.startswith("__"):
                raise AttributeError(k)
            m = _Any(self.__name__ + "." + k)
            sys.modules[m.__name__] = m
            return m

        def __call__(self, *a, **k):
            return None
    for name in ("cv2", "pybullet", "pybullet_data", "torchvision", "torchvision.transforms", "torchvision.utils",
                 "torchvision.models", "tensorboard", "torch.utils.tensorboard", "matplotlib", "matplotlib.pyplot",
                 "pandas", "seaborn"):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                sys.modules[name] = _Any(name)
    if "torch.utils.tensorboard" in sys.modules and not hasattr(sys.modules["torch.utils.tensorboard"], "SummaryWriter"):
        sys.modules["torch.utils.tensorboard"].SummaryWriter = object
    sys.path.insert(0, REF)


def NS(**kw):
    return types.SimpleNamespace(**kw)


def t(x):
    return torch.as_tensor(np.asarray(x, dtype=np.float32))


def save(name, **arrays):
    os.makedirs(OUT, exist_ok=True)
    out = {}
    for k, v in arrays.items():
        out[k] = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print("wrote", name, {k: out[k].shape for k in out})


def make_a1_env(rng, n):
    """A reference A1Conditional with hand-set attributes (no simulator behind it)."""
    from examples.a1_conditional.a1_conditional import A1Conditional
    from shifu.utils.train import HistoryRecorder
    env = object.__new__(A1Conditional)
    nd, nb, P = 12, 17, 187
    quat = rng.normal(size=(n, 4)); quat /= np.linalg.norm(quat, axis=1, keepdims=True)
    robot = NS(base_pose=t(np.concatenate([rng.uniform(-3, 3, (n, 3)), quat], 1)),
               base_lin_vel=t(rng.uniform(-2, 2, (n, 3))), base_ang_vel=t(rng.uniform(-3, 3, (n, 3))),
               gravity_vec=t(np.tile([0, 0, -1.0], (n, 1))), dof_pos=t(rng.uniform(-1.5, 1.5, (n, nd))),
               default_dof_pos=t([0.1, 0.8, -1.5, 0.1, 0.8, -1.5, -0.1, 0.8, -1.5, -0.1, 0.8, -1.5]),
               dof_vel=t(rng.uniform(-150, 150, (n, nd))),   # exercises the +-100 obs clip
               contact_forces=t(rng.uniform(-1, 1, (n, nb, 3)) * (rng.random((n, nb, 1)) < 0.5)),
               leg_indices=torch.tensor([2, 3, 6, 7, 10, 11, 14, 15]), torques=t(rng.uniform(-55, 55, (n, nd))))
    env.robot = robot
    env.isg_env = NS(measured_heights=t(rng.uniform(-2, 2, (n, P))), num_envs=n, device="cpu")
    env.command_buf = t(rng.uniform(-1, 1, (n, 3)))
    env.num_envs, env.device = n, "cpu"
    env.actions_recorder = HistoryRecorder((n, nd), 3, "cpu")
    for _ in range(3):
        env.actions_recorder.add(t(rng.uniform(-1, 1, (n, nd))))
    env.contact_terminate_indices = 0
    env.max_episode_length = np.ceil(10.0 / 0.02)
    env.max_episode_length_s = 10.0
    return env


def g1_observations(rng):
    env = make_a1_env(rng, 64)
    env.compute_observations()
    obs = torch.clip(env.obs_buf, -100.0, 100.0)      # ShifuVecEnv.step's clip (env.py:90)
    r = env.robot
    save("g1_observations", command=env.command_buf, base_pose=r.base_pose, base_lin_vel=r.base_lin_vel,
         base_ang_vel=r.base_ang_vel, dof_pos=r.dof_pos, dof_vel=r.dof_vel, default_dof_pos=r.default_dof_pos,
         history=env.actions_recorder.history_buf, measured_heights=env.isg_env.measured_heights, obs=obs,
         obs_unclipped=env.obs_buf)


def g2_termination(rng):
    env = make_a1_env(rng, 64)
    cf = env.robot.contact_forces.clone()
    cf[0, 0] = t([1.0, 0, 0]); cf[1, 0] = t([0.6, 0.8, 0.0]); cf[2, 0] = t([0, 0, 1.0000001]); cf[3, 0] = 0.
    env.robot.contact_forces = cf
    ep = torch.from_numpy(rng.integers(0, 600, 64)).long()
    ep[:4] = torch.tensor([500, 501, 499, 500])
    env.episode_length_buf = ep
    env.compute_termination()
    save("g2_termination", contact_forces=cf, episode_length=ep, max_episode_length=np.float64(env.max_episode_length),
         contact_terminate=env.contact_terminate_buf, time_out=env.time_out_buf, reset=env.reset_buf)


def g3_rewards(rng):
    env = make_a1_env(rng, 64)
    env.reward_functions = env.build_reward_functions()
    env._prepare_reward_functions()
    env.rew_buf = torch.zeros(64)
    steps = []
    for k in range(3):
        r = env.robot
        r.base_lin_vel = t(rng.uniform(-2, 2, (64, 3))); r.base_ang_vel = t(rng.uniform(-3, 3, (64, 3)))
        r.contact_forces = t(rng.uniform(-1, 1, (64, 17, 3)) * (rng.random((64, 17, 1)) < 0.5))
        r.torques = t(rng.uniform(-55, 55, (64, 12)))
        env.actions_recorder.add(t(rng.uniform(-1, 1, (64, 12))))
        terms = torch.stack([f() for f in env.reward_functions])
        env.compute_reward()
        steps.append(dict(base_lin_vel=r.base_lin_vel, base_ang_vel=r.base_ang_vel, contact_forces=r.contact_forces,
                          torques=r.torques, history=env.actions_recorder.history_buf.clone(), terms=terms,
                          rew=env.rew_buf.clone(),
                          sums=torch.stack([env.episode_rewards[f.__name__] for f in env.reward_functions]).clone()))
    flat = {f"s{k}_{name}": v for k, d in enumerate(steps) for name, v in d.items()}
    save("g3_rewards", command=env.command_buf, leg_indices=env.robot.leg_indices,
         names=np.array([f.__name__ for f in env.reward_functions]), **flat)


def g4_history(rng):
    from shifu.utils.train import HistoryRecorder
    h = HistoryRecorder((10, 3), 3, "cpu")
    xs = [t(rng.uniform(-1, 1, (10, 3))) for _ in range(5)]
    snaps, flats = [], []
    for k, x in enumerate(xs):
        h.add(x)
        if k == 3:
            h.reset_idx(torch.tensor([0, 1, 2, 7, 8, 9]))
        snaps.append(h.history_buf.clone()); flats.append(h.flatten().clone())
    save("g4_history", inputs=torch.stack(xs), bufs=torch.stack(snaps), flats=torch.stack(flats),
         last1=h.get_last(1), reset_after=np.int64(3), reset_ids=np.array([0, 1, 2, 7, 8, 9]))


def g5_reset_log(rng):
    from shifu.gym.env import ShifuVecEnv
    env = make_a1_env(rng, 32)
    env.reward_functions = env.build_reward_functions()
    env._prepare_reward_functions()
    for k, name in enumerate(env.episode_rewards):
        env.episode_rewards[name] = t(rng.uniform(-50, 50, 32))
    before = torch.stack([env.episode_rewards[k] for k in env.episode_rewards]).clone()
    env.isg_env.reset_idx = lambda ids: None
    env.cfg = NS(num_actions_history=3, send_timeouts=True)
    env.episode_length_buf = torch.from_numpy(rng.integers(1, 500, 32)).long()
    env.reset_buf = torch.zeros(32, dtype=torch.long)
    env.time_out_buf = torch.from_numpy(rng.random(32) < 0.3)
    env.terrain_levels = torch.from_numpy(rng.integers(0, 10, 32)).long()
    env.extras = {}
    ids = torch.tensor([1, 4, 5, 9, 20, 31])
    hist_before = env.actions_recorder.history_buf.clone()
    ShifuVecEnv.reset_idx(env, ids)
    save("g5_reset_log", ids=ids, sums_before=before, sums_after=torch.stack([env.episode_rewards[k] for k in env.episode_rewards]),
         names=np.array(list(env.episode_rewards)), episode=torch.stack([env.extras["episode"][k] for k in env.episode_rewards]),
         terrain_levels=env.terrain_levels, terrain_levels_mean=env.extras["episode"]["terrain_levels"],
         ep_len_after=env.episode_length_buf, reset_after=env.reset_buf, history_before=hist_before,
         history_after=env.actions_recorder.history_buf, time_outs=env.extras["time_outs"])


def g6_heights(rng):
    from shifu.gym.isaac_gym import TerrainGymEnv
    env = object.__new__(TerrainGymEnv)
    n = 48
    xs = [-0.8, -0.7, -0.6, -0.5, -0.4, -0.3, -0.2, -0.1, 0., 0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8]
    ys = [-0.5, -0.4, -0.3, -0.2, -0.1, 0., 0.1, 0.2, 0.3, 0.4, 0.5]
    tc = NS(mesh_type="heightfield", measured_points_x=xs, measured_points_y=ys, border_size=2.0, horizontal_scale=0.1,
            vertical_scale=0.005)
    env.cfg = NS(terrain=tc)
    env.num_envs, env.device = n, "cpu"
    env.height_points = env._init_height_points()
    rows, cols = 70, 90
    hs = rng.integers(-300, 300, (rows, cols)).astype(np.int16)
    env.height_samples = torch.from_numpy(hs)
    env.terrain = NS(cfg=tc)
    quat = rng.normal(size=(n, 4)); quat /= np.linalg.norm(quat, axis=1, keepdims=True)
    pos = rng.uniform(0.0, 5.0, (n, 3))
    pos[:6, :2] = [[-3.0, 1.0], [-2.05, -2.3], [9.5, 1.0], [1.0, 8.9], [-1.96, 1.0], [5.0, 6.95]]  # negatives & edges
    env.robot = NS(base_pose=t(np.concatenate([pos, quat], 1)))
    h = env.get_heights()
    save("g6_heights", height_samples=hs, base_pose=env.robot.base_pose, height_points=env.height_points[0, :, :2],
         border=np.float32(2.0), hscale=np.float32(0.1), vscale=np.float32(0.005), heights=h)


def g7_curriculum(rng):
    from examples.a1_conditional.a1_conditional import A1Conditional
    from shifu.gym.isaac_gym import TerrainGymEnv
    n = 40
    env = object.__new__(A1Conditional)
    isg = object.__new__(TerrainGymEnv)
    isg.init_done = True
    isg.terrain = NS(env_length=8.0)
    isg.max_terrain_level = 10
    rows, cols = 10, 20
    isg.terrain_origins = t(rng.uniform(0, 80, (rows, cols, 3)))
    isg.terrain_types = torch.from_numpy(rng.integers(0, cols, n)).long()
    levels = torch.from_numpy(rng.integers(0, 9, n)).long()          # < max-1: the randint branch cannot fire
    isg.terrain_levels = levels.clone()
    isg.env_origins = isg.terrain_origins[levels, isg.terrain_types].clone()
    env.isg_env = isg
    env.terrain_levels = levels.clone()
    env.max_episode_length_s = 10.0
    env.command_buf = t(rng.uniform(-1, 1, (n, 3)))
    pos = isg.env_origins.numpy() + rng.uniform(-6, 6, (n, 3))
    pos[:3] = isg.env_origins.numpy()[:3] + [[4.0, 0, 0], [4.0001, 0, 0], [0, 0, 0]]
    quat = np.tile([0, 0, 0, 1.0], (n, 1))
    env.robot = NS(base_pose=t(np.concatenate([pos, quat], 1)))
    ids = torch.arange(0, n, 2)
    origins_before = isg.env_origins.clone()
    env.update_terrain_curriculum(ids)
    save("g7_curriculum", ids=ids, base_pose=env.robot.base_pose, origins_before=origins_before, command=env.command_buf,
         levels_before=levels, levels_after=env.terrain_levels, origins_after=isg.env_origins,
         terrain_origins=isg.terrain_origins, terrain_types=isg.terrain_types, env_length=np.float32(8.0),
         max_level=np.int64(10), max_episode_length_s=np.float32(10.0))


def g8_terrain(rng):
    from shifu.utils import terrain as rt
    cfg = NS(mesh_type="heightfield", horizontal_scale=0.1, vertical_scale=0.005, border_size=5, terrain_length=8.,
             terrain_width=8., num_rows=4, num_cols=10, terrain_proportions=[0.1, 0.1, 0.35, 0.25, 0.2],
             slope_treshold=0.75, curriculum=True, selected=False, terrain_kwargs=None)
    np.random.seed(7)
    ter = rt.Terrain(cfg, 64)
    sub = rt.terrain_utils.SubTerrain("t", width=80, length=80, vertical_scale=0.005, horizontal_scale=0.1)
    rt.gap_terrain(sub, gap_size=0.5, platform_size=3.)
    sub2 = rt.terrain_utils.SubTerrain("t", width=80, length=80, vertical_scale=0.005, horizontal_scale=0.1)
    rt.pit_terrain(sub2, depth=0.7, platform_size=4.)
    q = rng.normal(size=(16, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    v = rng.uniform(-1, 1, (16, 3))
    save("g8_terrain", seed=np.int64(7), height_field=ter.height_field_raw, env_origins=ter.env_origins,
         tot_rows=np.int64(ter.tot_rows), tot_cols=np.int64(ter.tot_cols), gap=sub.height_field_raw,
         pit=sub2.height_field_raw, yaw_quat=t(q), yaw_vec=t(v), yaw_out=rt.quat_apply_yaw(t(q), t(v)),
         wrap_in=t(np.linspace(-9, 9, 37)), wrap_out=rt.wrap_to_pi(t(np.linspace(-9, 9, 37)).clone()))


def g9_ik(rng):
    from shifu.utils import torch_utils as tu
    n = 24
    qa = rng.normal(size=(n, 4)); qa /= np.linalg.norm(qa, axis=1, keepdims=True)
    qb = rng.normal(size=(n, 4)); qb /= np.linalg.norm(qb, axis=1, keepdims=True)
    j = rng.uniform(-1, 1, (n, 6, 6))
    dof = rng.uniform(-1, 1, (n, 6))
    ee_pos, tar_pos = rng.uniform(-0.5, 0.5, (n, 3)), rng.uniform(-0.5, 0.5, (n, 3))
    out = tu.inverse_kinematics(t(dof), t(ee_pos), t(qa), t(tar_pos), t(qb), t(j), "cpu")
    save("g9_ik", qa=t(qa), qb=t(qb), quat_mul=tu.quat_mul(t(qa), t(qb)), quat_conjugate=tu.quat_conjugate(t(qa)),
         j_ee=t(j), dof_pos=t(dof), ee_pos=t(ee_pos), tar_pos=t(tar_pos), ik=out)


def g10_abb(rng):
    from examples.abb_pushbox_vision.a_prior_stage import AbbPushBox
    n = 64
    env = object.__new__(AbbPushBox)
    cube = rng.uniform(-0.25, 0.25, (n, 7)); goal = rng.uniform(-0.2, 0.2, (n, 7)); ee = rng.uniform(-0.25, 0.25, (n, 1, 7))
    goal[:4, :2] = cube[:4, :2] + [[0.019, 0], [0.0201, 0], [0, 0.0199], [0.0141, 0.0141]]
    ee[:8, 0, :2] = cube[:8, :2] + rng.uniform(-0.05, 0.05, (8, 2))
    env.cube = NS(base_pose=t(cube)); env.goal = NS(base_pose=t(goal))
    env.robot = NS(ee_pose=t(ee), min_ee_pos=t([-0.2, -0.2, 0.11]), max_ee_pos=t([0.2, 0.2, 0.14]))
    env.episode_length_buf = torch.from_numpy(rng.integers(0, 260, n)).long()
    env.max_episode_length = np.ceil(20.0 / 0.1)
    env.compute_observations()
    env.compute_termination()
    save("g10_abb", cube=env.cube.base_pose, goal=env.goal.base_pose, ee=env.robot.ee_pose, ep_len=env.episode_length_buf,
         max_episode_length=np.float64(env.max_episode_length), obs=env.obs_buf, time_out=env.time_out_buf,
         success=env.success_buf, reset=env.reset_buf, reward_reaching=env.reward_reaching(),
         reward_success=env.reward_success())


def g11_configs(rng):
    from examples.a1_conditional.task_config import A1ActorConfig, A1EnvConfig, A1PPOConfig
    from examples.abb_pushbox_vision.task_config import AbbRobotConfig, PriorStageEnvConfig
    from shifu.runner.utils import class_to_dict
    import json
    c = A1EnvConfig()
    eff = dict(num_envs=c.num_envs, num_obs=c.num_obs, num_actions=c.num_actions, dt=c.sim.dt,
               decimation=c.control.decimation, episode_length_s=c.episode_length_s,
               terrain_mesh_type=c.terrain.mesh_type, terrain_num_rows=c.terrain.num_rows,
               terrain_num_cols=c.terrain.num_cols, terrain_max_init_level=c.terrain.max_init_terrain_level,
               terrain_curriculum=c.terrain.curriculum, terrain_border=c.terrain.border_size,
               has_terrian_typo=hasattr(c, "terrian"), sim_params_dt=c.sim_params.dt,
               physx_max_depen=c.sim_params.physx.max_depenetration_velocity, clip_obs=c.normalization.clip_observations,
               clip_actions=c.normalization.clip_actions)
    a = A1ActorConfig()
    eff.update(a1_drive_mode=int(a.asset_options.default_dof_drive_mode), a1_collapse=bool(a.asset_options.collapse_fixed_joints),
               a1_default_pos=list(a.default_pos), a1_kp=list(a.dof_stiffness))
    p = PriorStageEnvConfig(); r = AbbRobotConfig()
    eff.update(abb_dt=p.sim.dt, abb_decimation=p.control.decimation, abb_episode_length_s=p.episode_length_s,
               abb_fix_base=bool(r.asset_options.fix_base_link), abb_disable_gravity=bool(r.asset_options.disable_gravity),
               abb_drive_mode=int(r.asset_options.default_dof_drive_mode), abb_kp=list(r.dof_stiffness))
    ppo = class_to_dict(A1PPOConfig())
    save("g11_configs", effective=np.array(json.dumps(eff, sort_keys=True)), ppo=np.array(json.dumps(ppo, sort_keys=True)))


def main():
    if not os.path.isdir(REF):
        raise SystemExit("the reference tree is not present: golden vectors can only be (re)generated in the build container")
    install_stubs()
    import inspect
    import examples.a1_conditional.a1_conditional as ref_a1
    import shifu.gym.env as ref_env
    for mod in (ref_a1, ref_env):   # the vectors must come from the REFERENCE's code, not this repo's mirror
        assert inspect.getsourcefile(mod).startswith(REF + "/"), inspect.getsourcefile(mod)
    rng = np.random.default_rng(20261001)
    for fn in (g1_observations, g2_termination, g3_rewards, g4_history, g5_reset_log, g6_heights, g7_curriculum,
               g8_terrain, g9_ik, g10_abb, g11_configs):
        torch.manual_seed(0)
        fn(rng)


if __name__ == "__main__":
    main()
