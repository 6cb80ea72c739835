"""Golden vectors produced by the reference's own Python (tools/make_golden.py, SURVEY.md
8c G1-G11) against (A) the CPU oracle's glue functions -- the same functions the fused
oracle step calls, so pinning them pins what the HIP kernel is compared with -- and
(B) this repo's host-side mirror of the reference interface.  CPU only.

Tolerances: integer / boolean / index outputs exact; float glue 1e-6 relative (torch's
exp and reduction order differ from the spec'd arithmetic by an ulp or two)."""
import json
import os
import types

import numpy as np
import torch

from shifu_amd import _abi

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
LEGS = [2, 3, 6, 7, 10, 11, 14, 15]


def load(name):
    return np.load(os.path.join(G, name + ".npz"), allow_pickle=False)


def NS(**kw):
    return types.SimpleNamespace(**kw)


def tt(x):
    return torch.from_numpy(np.asarray(x))


# ------------------------------------------------------------------ (A) oracle --
def test_oracle_observations_g1(oracle):
    g = load("g1_observations")
    dof = np.stack([g["dof_pos"], g["dof_vel"]], -1).reshape(-1, 2)
    obs = oracle.glue_obs(g["default_dof_pos"], g["command"], g["base_lin_vel"], g["base_ang_vel"], dof, g["history"],
                          g["base_pose"][:, 2], g["measured_heights"], 100.0)
    np.testing.assert_allclose(obs, g["obs"], rtol=1e-6, atol=1e-6)
    assert np.abs(g["obs_unclipped"]).max() > 100.0 and np.abs(obs).max() == 100.0   # the clip was exercised
    np.testing.assert_array_equal(obs[:, 9:12], np.tile([0, 0, -1.0], (64, 1)))      # Q3: constant gravity_vec


def test_oracle_termination_g2(oracle):
    g = load("g2_termination")
    ct, to = oracle.glue_termination(g["contact_forces"], g["episode_length"], 0, float(g["max_episode_length"]))
    np.testing.assert_array_equal(ct, g["contact_terminate"])
    np.testing.assert_array_equal(to, g["time_out"])
    np.testing.assert_array_equal(ct | to, g["reset"])
    assert list(g["time_out"][:4]) == [False, True, False, False]                  # Q6: '>' not '>='
    assert list(g["contact_terminate"][:4]) == [False, False, True, False]         # |F| = 1.0 does not terminate


def test_oracle_rewards_g3(oracle):
    g = load("g3_rewards")
    assert list(g["leg_indices"]) == LEGS
    sums = np.zeros((6, 64), np.float32)
    for k in range(3):
        terms = oracle.glue_rewards(g["command"], g[f"s{k}_base_lin_vel"], g[f"s{k}_base_ang_vel"], g[f"s{k}_history"],
                                    g[f"s{k}_contact_forces"], g[f"s{k}_torques"], LEGS)
        np.testing.assert_allclose(terms, g[f"s{k}_terms"], rtol=2e-6, atol=1e-7)
        sums += terms
        np.testing.assert_allclose(sums, g[f"s{k}_sums"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(terms.sum(0), g[f"s{k}_rew"], rtol=1e-5, atol=1e-6)
    np.testing.assert_array_equal(terms[4], g["s2_terms"][4])                      # leg_collision counts are exact


def test_oracle_heights_g6(oracle):
    g = load("g6_heights")
    hs = np.ascontiguousarray(g["height_samples"])
    terr = _abi.ShfTerrain()
    terr.rows, terr.cols = hs.shape
    terr.hscale, terr.vscale, terr.border, terr.friction = float(g["hscale"]), float(g["vscale"]), float(g["border"]), 1.0
    root = np.zeros((g["base_pose"].shape[0], 13), np.float32)
    root[:, :7] = g["base_pose"]
    h = oracle.glue_heights(terr, hs, g["height_points"], root)
    mismatch = (h != g["heights"]).mean()
    assert mismatch <= 1e-3, f"{mismatch:.4%} of the 48x187 samples differ"      # cell-edge ties only
    np.testing.assert_array_equal(h[:6], g["heights"][:6])                        # negative coords, clipping edges


def test_oracle_curriculum_g7(oracle):
    from shifu_amd.a1_task import a1_task_params
    from tests import helpers as H
    g = load("g7_curriculum")
    tp = a1_task_params(H.a1_model(), env_length=float(g["env_length"]), num_rows=int(g["max_level"]))
    ids = g["ids"]
    root = np.zeros((len(ids), 13), np.float32)
    root[:, :7] = g["base_pose"][ids]
    lv = oracle.glue_curriculum(tp, root, g["origins_before"][ids], g["command"][ids], g["levels_before"][ids])
    np.testing.assert_array_equal(lv, g["levels_after"][ids])
    assert (g["levels_after"] != g["levels_before"]).sum() > 5
    untouched = np.setdiff1d(np.arange(40), ids)
    np.testing.assert_array_equal(g["levels_after"][untouched], g["levels_before"][untouched])
    np.testing.assert_allclose(g["origins_after"][ids], g["terrain_origins"][lv, g["terrain_types"][ids]])


def test_oracle_episode_stats_g5(oracle):
    from shifu_amd.a1_task import a1_task_params
    from tests import helpers as H
    g = load("g5_reset_log")
    tp = a1_task_params(H.a1_model())
    n, ids = 32, g["ids"]
    done = np.zeros((8, n), np.float32)
    done[:6, ids] = g["sums_before"][:, ids]
    done[6] = g["terrain_levels"]
    done[7, ids] = 1.0
    out = oracle.a1_stats(tp, n, done)
    np.testing.assert_allclose(out[8:14], g["episode"], rtol=1e-6)                 # mean(sum[ids]) / 10 s
    np.testing.assert_allclose(out[14], g["terrain_levels_mean"], rtol=1e-6)
    assert out[7] == len(ids)


def test_oracle_history_recorder_g4(oracle):
    """HistoryRecorder add / reset_idx / flatten (shifu/utils/train.py:12-35) as the fused oracle steps do it."""
    g = load("g4_history")
    bufs, flats = oracle.glue_history(g["inputs"], int(g["reset_after"]), g["reset_ids"], H=3)
    np.testing.assert_array_equal(bufs, g["bufs"])
    np.testing.assert_array_equal(flats, g["flats"])
    np.testing.assert_array_equal(bufs[-1][:, :, 1], g["last1"])                     # get_last(1) = slot 1 = t-1
    assert (bufs[int(g["reset_after"])][g["reset_ids"]] == 0).all()                  # rows zeroed AFTER that step's add (Q12)


def test_oracle_quaternions_and_ik_g9(oracle):
    """quat_mul (shifu/utils/torch_utils.py:12-33) bit for bit; damped-least-squares IK (robot.py:162-182):
    the oracle solves (J J^T + 0.05^2 I) x = dpose by LDL^T where the reference calls torch.inverse, so the
    float build agrees to the conditioning of the random 6x6 systems (measured worst: 3e-5 of the row's largest entry), and the
    double build -- same code, same float32 inputs -- to float32 round-off of the reference's own result."""
    g = load("g9_ik")
    np.testing.assert_array_equal(oracle.glue_quat_mul(g["qa"], g["qb"]), g["quat_mul"])
    args = (g["j_ee"], g["dof_pos"], g["ee_pos"], g["qa"], g["tar_pos"], g["qb"])
    ik64 = oracle.glue_ik(*args, f64=True)
    ik32 = oracle.glue_ik(*args)
    scale = np.abs(g["ik"]).max(axis=1, keepdims=True)
    assert (np.abs(ik64 - g["ik"]) / scale).max() < 1e-4                              # torch.inverse in float32 is the noisy side
    assert (np.abs(ik32 - g["ik"]) / scale).max() < 1e-4
    # independent float64 restatement of the reference formula on the same inputs
    J = g["j_ee"].astype(np.float64)
    qa, qb = g["qa"].astype(np.float64), g["qb"].astype(np.float64)
    cc = np.concatenate([-qa[:, :3], qa[:, 3:]], 1)
    qr = oracle.glue_quat_mul(g["qb"], cc.astype(np.float32)).astype(np.float64)
    dpose = np.concatenate([g["tar_pos"].astype(np.float64) - g["ee_pos"], qr[:, :3] * np.sign(qr[:, 3:])], 1)[..., None]
    u = (J.transpose(0, 2, 1) @ np.linalg.inv(J @ J.transpose(0, 2, 1) + np.eye(6) * 0.05 ** 2) @ dpose)[..., 0]
    np.testing.assert_allclose(ik64, g["dof_pos"] + u, rtol=1e-5, atol=1e-5)


def test_oracle_abb_pushbox_g10(oracle):
    """AbbPushBox.compute_observations / compute_termination / reward_* (a_prior_stage.py:97-130)."""
    from shifu_amd.abb_task import abb_model, abb_task_params
    g = load("g10_abb")
    tp = abb_task_params(abb_model())
    assert tp.max_episode_length == float(g["max_episode_length"])
    obs, to, su, rs, r0, r1 = oracle.glue_abb_post(tp, g["cube"], g["goal"], g["ee"][:, 0], g["ep_len"])
    np.testing.assert_array_equal(obs, g["obs"])
    np.testing.assert_array_equal(to, g["time_out"])
    np.testing.assert_array_equal(su, g["success"])
    np.testing.assert_array_equal(rs, g["reset"])
    np.testing.assert_allclose(r0, g["reward_reaching"], rtol=2e-6, atol=1e-7)       # exp_spec vs torch.exp
    np.testing.assert_array_equal(r1, g["reward_success"])
    assert to.any() and su.any() and (r0 > 0).any() and (rs & ~to & ~su).any()       # every branch exercised


# ---------------------------------------------------------- (B) host-side mirror --
def _my_a1(g, n):
    from examples.a1_conditional.a1_conditional import A1Conditional
    from shifu_amd.utils.train import HistoryRecorder
    env = object.__new__(A1Conditional)
    env.num_envs, env.device = n, "cpu"
    env.isg_env = NS(num_envs=n, device="cpu")
    env.actions_recorder = HistoryRecorder((n, 12), 3, "cpu")
    env.contact_terminate_indices = 0
    env.max_episode_length, env.max_episode_length_s = np.ceil(10.0 / 0.02), 10.0
    return env


def test_mirror_observations_termination_rewards():
    g = load("g1_observations")
    env = _my_a1(g, 64)
    env.robot = NS(base_pose=tt(g["base_pose"]), base_lin_vel=tt(g["base_lin_vel"]), base_ang_vel=tt(g["base_ang_vel"]),
                   gravity_vec=torch.tensor([0, 0, -1.0]).repeat(64, 1), dof_pos=tt(g["dof_pos"]),
                   default_dof_pos=tt(g["default_dof_pos"]), dof_vel=tt(g["dof_vel"]))
    env.isg_env.measured_heights = tt(g["measured_heights"])
    env.command_buf = tt(g["command"])
    env.actions_recorder.history_buf = tt(g["history"]).clone()
    env.compute_observations()
    np.testing.assert_array_equal(env.obs_buf.numpy(), g["obs_unclipped"])

    g = load("g2_termination")
    env.robot.contact_forces = tt(g["contact_forces"])
    env.episode_length_buf = tt(g["episode_length"])
    env.compute_termination()
    np.testing.assert_array_equal(env.reset_buf.numpy(), g["reset"])
    np.testing.assert_array_equal(env.time_out_buf.numpy(), g["time_out"])

    g = load("g3_rewards")
    env.command_buf = tt(g["command"])
    env.robot.leg_indices = tt(g["leg_indices"])
    env.reward_functions = env.build_reward_functions()
    assert [f.__name__ for f in env.reward_functions] == list(g["names"])
    env._prepare_reward_functions()
    env.rew_buf = torch.zeros(64)
    for k in range(3):
        env.robot.base_lin_vel, env.robot.base_ang_vel = tt(g[f"s{k}_base_lin_vel"]), tt(g[f"s{k}_base_ang_vel"])
        env.robot.contact_forces, env.robot.torques = tt(g[f"s{k}_contact_forces"]), tt(g[f"s{k}_torques"])
        env.actions_recorder.history_buf = tt(g[f"s{k}_history"]).clone()
        env.compute_reward()
        np.testing.assert_array_equal(env.rew_buf.numpy(), g[f"s{k}_rew"])
        sums = torch.stack([env.episode_rewards[f.__name__] for f in env.reward_functions]).numpy()
        np.testing.assert_array_equal(sums, g[f"s{k}_sums"])


def test_mirror_history_recorder_g4():
    from shifu_amd.utils.train import HistoryRecorder
    g = load("g4_history")
    h = HistoryRecorder((10, 3), 3, "cpu")
    for k in range(5):
        h.add(tt(g["inputs"][k]))
        if k == int(g["reset_after"]):
            h.reset_idx(tt(g["reset_ids"]))
        np.testing.assert_array_equal(h.history_buf.numpy(), g["bufs"][k])
        np.testing.assert_array_equal(h.flatten().numpy(), g["flats"][k])
    np.testing.assert_array_equal(h.get_last(1).numpy(), g["last1"])


def test_mirror_reset_idx_log_info_g5():
    from shifu_amd.gym.env import ShifuVecEnv
    g = load("g5_reset_log")
    env = _my_a1(g, 32)
    env.episode_rewards = {str(n): tt(g["sums_before"][k]).clone() for k, n in enumerate(g["names"])}
    env.isg_env.reset_idx = lambda ids: None
    env.cfg = NS(num_actions_history=3, send_timeouts=True)
    env.episode_length_buf = torch.ones(32, dtype=torch.long)
    env.reset_buf = torch.zeros(32, dtype=torch.long)
    env.time_out_buf = tt(g["time_outs"])
    env.terrain_levels = tt(g["terrain_levels"])
    env.actions_recorder.history_buf = tt(g["history_before"]).clone()
    env.extras = {}
    ShifuVecEnv.reset_idx(env, tt(g["ids"]))
    ep = torch.stack([env.extras["episode"][str(n)] for n in g["names"]]).numpy()
    np.testing.assert_array_equal(ep, g["episode"])
    np.testing.assert_array_equal(env.extras["episode"]["terrain_levels"].numpy(), g["terrain_levels_mean"])
    np.testing.assert_array_equal(torch.stack(list(env.episode_rewards.values())).numpy(), g["sums_after"])
    np.testing.assert_array_equal(env.reset_buf.numpy(), g["reset_after"])
    np.testing.assert_array_equal(env.actions_recorder.history_buf.numpy(), g["history_after"])
    assert (env.episode_length_buf[tt(g["ids"])] == 0).all()


def test_mirror_get_heights_g6():
    from shifu_amd.gym.isaac_gym import TerrainGymEnv
    g = load("g6_heights")
    env = object.__new__(TerrainGymEnv)
    tc = NS(mesh_type="heightfield", measured_points_x=sorted(set(np.round(g["height_points"][:, 0], 3).tolist())),
            measured_points_y=sorted(set(np.round(g["height_points"][:, 1], 3).tolist())), border_size=float(g["border"]),
            horizontal_scale=float(g["hscale"]), vertical_scale=float(g["vscale"]))
    env.cfg, env.terrain = NS(terrain=tc), NS(cfg=tc)
    env.num_envs, env.device = g["base_pose"].shape[0], "cpu"
    env.height_points = env._init_height_points()
    np.testing.assert_allclose(env.height_points[0, :, :2].numpy(), g["height_points"], atol=1e-7)
    env.height_samples = tt(g["height_samples"])
    env.robot = NS(base_pose=tt(g["base_pose"]))
    np.testing.assert_array_equal(env.get_heights().numpy(), g["heights"])


def test_mirror_curriculum_g7():
    from examples.a1_conditional.a1_conditional import A1Conditional
    from shifu_amd.gym.isaac_gym import TerrainGymEnv
    g = load("g7_curriculum")
    env, isg = object.__new__(A1Conditional), object.__new__(TerrainGymEnv)
    isg.init_done, isg.terrain, isg.max_terrain_level = True, NS(env_length=float(g["env_length"])), int(g["max_level"])
    isg.terrain_origins, isg.terrain_types = tt(g["terrain_origins"]), tt(g["terrain_types"])
    isg.terrain_levels, isg.env_origins = tt(g["levels_before"]).clone(), tt(g["origins_before"]).clone()
    env.isg_env, env.terrain_levels = isg, tt(g["levels_before"]).clone()
    env.max_episode_length_s, env.command_buf = float(g["max_episode_length_s"]), tt(g["command"])
    env.robot = NS(base_pose=tt(g["base_pose"]))
    env.update_terrain_curriculum(tt(g["ids"]))
    np.testing.assert_array_equal(env.terrain_levels.numpy(), g["levels_after"])
    np.testing.assert_array_equal(isg.env_origins.numpy(), g["origins_after"])


def test_mirror_terrain_g8():
    from shifu_amd.gym.isaac_gym import quat_apply_yaw
    from shifu_amd.isaacgym import terrain_utils
    from shifu_amd.utils import terrain as mt
    g = load("g8_terrain")
    cfg = NS(mesh_type="heightfield", horizontal_scale=0.1, vertical_scale=0.005, border_size=5, terrain_length=8.,
             terrain_width=8., num_rows=4, num_cols=10, terrain_proportions=[0.1, 0.1, 0.35, 0.25, 0.2],
             slope_treshold=0.75, curriculum=True, selected=False, terrain_kwargs=None)
    np.random.seed(int(g["seed"]))
    ter = mt.Terrain(cfg, 64)
    assert (ter.tot_rows, ter.tot_cols) == (int(g["tot_rows"]), int(g["tot_cols"]))
    np.testing.assert_array_equal(ter.height_field_raw, g["height_field"])
    np.testing.assert_array_equal(ter.env_origins, g["env_origins"])
    sub = terrain_utils.SubTerrain("t", width=80, length=80, vertical_scale=0.005, horizontal_scale=0.1)
    mt.gap_terrain(sub, gap_size=0.5, platform_size=3.)
    np.testing.assert_array_equal(sub.height_field_raw, g["gap"])
    sub = terrain_utils.SubTerrain("t", width=80, length=80, vertical_scale=0.005, horizontal_scale=0.1)
    mt.pit_terrain(sub, depth=0.7, platform_size=4.)
    np.testing.assert_array_equal(sub.height_field_raw, g["pit"])
    np.testing.assert_array_equal(quat_apply_yaw(tt(g["yaw_quat"]), tt(g["yaw_vec"])).numpy(), g["yaw_out"])


def test_mirror_ik_and_quaternions_g9():
    from shifu_amd.utils import torch_utils as tu
    g = load("g9_ik")
    np.testing.assert_array_equal(tu.quat_mul(tt(g["qa"]), tt(g["qb"])).numpy(), g["quat_mul"])
    np.testing.assert_array_equal(tu.quat_conjugate(tt(g["qa"])).numpy(), g["quat_conjugate"])
    out = tu.inverse_kinematics(tt(g["dof_pos"]), tt(g["ee_pos"]), tt(g["qa"]), tt(g["tar_pos"]), tt(g["qb"]),
                                tt(g["j_ee"]), "cpu")
    np.testing.assert_allclose(out.numpy(), g["ik"], rtol=1e-5, atol=1e-6)


def test_mirror_configs_g11():
    from examples.a1_conditional.task_config import A1ActorConfig, A1EnvConfig, A1PPOConfig
    from shifu_amd.runner.utils import class_to_dict
    g = load("g11_configs")
    eff = json.loads(str(g["effective"]))
    c, a = A1EnvConfig(), A1ActorConfig()
    mine = dict(num_envs=c.num_envs, num_obs=c.num_obs, num_actions=c.num_actions, dt=c.sim.dt,
                decimation=c.control.decimation, episode_length_s=c.episode_length_s,
                terrain_mesh_type=c.terrain.mesh_type, terrain_num_rows=c.terrain.num_rows,
                terrain_num_cols=c.terrain.num_cols, terrain_max_init_level=c.terrain.max_init_terrain_level,
                terrain_curriculum=c.terrain.curriculum, terrain_border=c.terrain.border_size,
                has_terrian_typo=hasattr(c, "terrian"), sim_params_dt=c.sim_params.dt,
                physx_max_depen=c.sim_params.physx.max_depenetration_velocity, clip_obs=c.normalization.clip_observations,
                clip_actions=c.normalization.clip_actions, a1_drive_mode=int(a.asset_options.default_dof_drive_mode),
                a1_collapse=bool(a.asset_options.collapse_fixed_joints), a1_default_pos=list(a.default_pos),
                a1_kp=list(a.dof_stiffness))
    for k, v in mine.items():
        assert eff[k] == v, (k, eff[k], v)
    assert eff["terrain_num_rows"] == 10 and eff["has_terrian_typo"]              # Q5 documented
    assert json.loads(str(g["ppo"])) == json.loads(json.dumps(class_to_dict(A1PPOConfig())))


def test_spec_quat_rotate_inverse_matches_facade(oracle):
    """The oracle's restatement of isaacgym.torch_utils.quat_rotate_inverse ([EXT], appendix D)
    vs the facade's torch version on random inputs."""
    from shifu_amd.isaacgym.torch_utils import quat_rotate_inverse
    rng = np.random.default_rng(1)
    q = rng.normal(size=(256, 4)).astype(np.float32); q /= np.linalg.norm(q, axis=1, keepdims=True)
    v = rng.uniform(-3, 3, (256, 3)).astype(np.float32)
    np.testing.assert_allclose(oracle.quat_rotate_inverse(q, v), quat_rotate_inverse(tt(q), tt(v)).numpy(),
                               rtol=1e-6, atol=1e-6)


def test_mirror_abb_pushbox_g10():
    from examples.abb_pushbox_vision.a_prior_stage import AbbPushBox
    g = load("g10_abb")
    env = object.__new__(AbbPushBox)
    env.cube, env.goal = NS(base_pose=tt(g["cube"])), NS(base_pose=tt(g["goal"]))
    env.robot = NS(ee_pose=tt(g["ee"]), min_ee_pos=torch.tensor([-0.2, -0.2, 0.11]), max_ee_pos=torch.tensor([0.2, 0.2, 0.14]))
    env.episode_length_buf = tt(g["ep_len"])
    env.max_episode_length = float(g["max_episode_length"])
    env.compute_observations()
    env.compute_termination()
    np.testing.assert_array_equal(env.obs_buf.numpy(), g["obs"])
    np.testing.assert_array_equal(env.time_out_buf.numpy(), g["time_out"])
    np.testing.assert_array_equal(env.success_buf.numpy(), g["success"])
    np.testing.assert_array_equal(env.reset_buf.numpy(), g["reset"])
    np.testing.assert_array_equal(env.reward_reaching().numpy(), g["reward_reaching"])
    np.testing.assert_array_equal(env.reward_success().numpy(), g["reward_success"])
    assert list(g["success"][:4]) == [True, False, True, True]     # 0.02 m threshold, strict '<'
