"""The drop-in boundary end to end on the GPU: the hook-based A1Conditional (user code
in torch, physics through the `gym` facade, exactly the reference's call order) against
the fused single-launch env on the same initial state and action sequence.

Physics is the same arithmetic in both (k_sim_step vs the sub-steps inside k_a1_step),
so dof/root state must agree bit for bit; observation/reward go through torch ops in
one and the spec'd kernel arithmetic in the other (torch.exp vs exp_spec, summation
order), so they are compared to 1e-5."""
import numpy as np
import pytest

from shifu_amd import _abi

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _envs(n=64):
    from examples.a1_conditional.a1_conditional import A1Conditional
    from examples.a1_conditional.task_config import A1EnvConfig
    from shifu_amd.gym.a1_fused import FusedA1Env
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU is visible")
    cfg = A1EnvConfig()
    cfg.num_envs = n
    np.random.seed(42)
    torch.manual_seed(0)
    hook = A1Conditional(cfg)
    # A1EnvConfig builds the trimesh terrain (Q5): the fused env is given the same one
    # ... and, like every actor the reference creates (collision filter 0, units.py:68), the hook env's robot collides
    # with itself: the fused env is asked for the same
    fused = FusedA1Env(num_envs=n, terrain="trimesh", terrain_seed=42, seed=3, self_collision=True)
    return hook, fused


def _sync_state(hook, fused):
    """Give the hook env the fused env's post-reset state, commands, pushes, friction."""
    be = hook.isg_env.sim.backend
    S, T = fused.sim.tensors, fused.task.tensors
    for tid in (_abi.T_DOF_STATE, _abi.T_ROOT_STATE):
        be.tensors[tid].copy_(S[tid])
    be.tensors[_abi.T_SIM_DOF].copy_(S[_abi.T_DOF_STATE])
    be.tensors[_abi.T_SIM_ROOT].copy_(S[_abi.T_ROOT_STATE])
    be.tensors[_abi.T_FRICTION].copy_(S[_abi.T_FRICTION])
    hook.command_buf.copy_(T[_abi.A1_COMMAND])
    hook.robot.rand_force_buf.copy_(T[_abi.A1_PUSH])
    hook.isg_env.env_origins.copy_(T[_abi.A1_ORIGINS])
    hook.terrain_levels.copy_(T[_abi.A1_LEVELS])
    hook.isg_env.terrain_levels.copy_(T[_abi.A1_LEVELS])
    hook.episode_length_buf.copy_(T[_abi.A1_EP_LEN])
    hook.actions_recorder.history_buf.copy_(T[_abi.A1_HISTORY])
    for k, name in enumerate(hook.episode_rewards):
        hook.episode_rewards[name].copy_(T[_abi.A1_REW_SUMS][k])
    hook.robot.post_step()      # base-frame velocities from the (synced) root_state tensor


def test_same_terrain_and_layout():
    hook, fused = _envs(32)
    assert torch.equal(hook.isg_env.height_samples.cpu(), torch.from_numpy(fused.sim.height_samples))
    assert hook.isg_env.sim.backend.terrain.warped == 1 and fused.sim.terrain.warped == 1
    assert torch.equal(hook.isg_env.sim.backend.tensors[_abi.T_HEIGHTS], fused.sim.tensors[_abi.T_HEIGHTS])
    assert torch.equal(hook.isg_env.terrain_types, fused.task.tensors[_abi.A1_TYPES])
    assert torch.allclose(hook.isg_env.terrain_origins, fused.task.tensors[_abi.A1_TORIGINS])
    assert hook.robot.rigid_body_dict["base"] == 0 and hook.robot.num_bodies == 17 and hook.robot.num_dof == 12
    assert hook.obs_buf.shape == fused.obs_buf.shape == (32, 259)
    # Q13: the first reset_idx(all) (distance 0, zero commands) leaves every env on level 0 of its column
    assert int(fused.terrain_levels.abs().sum()) == 0
    hook.reset()
    assert int(hook.terrain_levels.abs().sum()) == 0


def _compare_step(hook, fused, it, m, o1, o2, r1, r2, obs_from=0):
    n = hook.num_envs
    assert torch.equal(hook.isg_env.dof_state.view(n, -1)[m], fused.dof_state.view(n, -1)[m]), f"dof step {it}"
    assert torch.equal(hook.isg_env.contact_state.view(n, -1)[m], fused.contact_state.view(n, -1)[m])
    assert torch.allclose(hook.isg_env.body_state.view(n, -1)[m], fused.body_state.view(n, -1)[m], atol=0, rtol=0)
    # get_heights truncates float positions to cell indices: torch's GPU kernels and the spec'd
    # kernel arithmetic may round a point lying on a cell edge to different sides -- allow 0.2 %
    hm = hook.isg_env.measured_heights[m] != fused.measured_heights[m]
    assert hm.float().mean() < 2e-3, f"heights step {it}: {hm.float().mean()}"
    assert torch.allclose(o1[m][:, obs_from:72], o2[m][:, obs_from:72], rtol=1e-5, atol=1e-5), f"obs step {it}"
    ho = ~torch.isclose(o1[m][:, 72:], o2[m][:, 72:], rtol=1e-5, atol=1e-5)
    assert ho.float().mean() < 2e-3, f"height obs step {it}"
    assert torch.allclose(r1[m], r2[m], rtol=1e-5, atol=1e-5), f"rew step {it}"


@pytest.mark.parametrize("replayed", [False, True])
def test_hook_env_matches_fused_env_through_resets(replayed):
    """The path a user runs (hooks in torch over the `gym` facade, reference call order env.py:93-130) against the
    fused kernel, THROUGH episode ends: on the step an env resets, rewards are those of the finished episode
    (computed before the reset), the observation is taken after it -- default joint state, zeroed action history,
    pre-reset measured heights against the re-spawned base height, and the base-frame velocities of the step
    before (Q2/Q15) -- and the history then receives the pre-reset action (Q12).  Only the three freshly sampled
    command entries differ that step (torch's generator vs the kernel's Philox stream); the fused env's draws
    (spawn xy, command, push) are then copied into the hook env and the comparison goes on, bit for bit in state."""
    n = 64
    hook, fused = _envs(n)
    # both run the reference's contact settings (env_config.py:50-58) through the velocity-level solve, self-collision on
    assert hook.isg_env.sim.solver == "tgs" and fused.solver == "tgs" and fused.sim_params.pos_iters == 8 and fused.sim_params.vel_iters == 1      # (physx.solver_type = 1)
    if replayed:
        # ShifuVecEnv.enable_graph_hooks: the shape-static hooks replayed from two hipGraphs, reset_idx eager in between --
        # the same step, launched differently; held to the fused kernel exactly like the eager mode
        hook.enable_graph_hooks()
    fused.task.reset_all()
    T, S = fused.task.tensors, fused.sim.tensors
    T[_abi.A1_EP_LEN][:8] = 490          # time-outs (ep_len > 500) fall inside the run, next to contact terminations
    _sync_state(hook, fused)
    be = hook.isg_env.sim.backend
    g = torch.Generator(device="cuda:0")
    g.manual_seed(5)
    resets = timeouts = after_reset = 0
    was_reset = torch.zeros(n, dtype=torch.bool, device="cuda:0")
    for it in range(80):
        a = 2 * torch.rand(n, 12, device="cuda:0", generator=g) - 1
        o1, _, r1, d1, x1 = hook.step(a)
        o2, _, r2, d2, x2 = fused.step(a)
        assert torch.equal(d1.bool(), d2), f"reset masks differ at step {it}"
        assert torch.equal(hook.time_out_buf, fused.time_out_buf), f"time-outs differ at step {it}"
        assert torch.equal(hook.episode_length_buf, fused.episode_length_buf)
        keep = ~d2
        _compare_step(hook, fused, it, keep, o1, o2, r1, r2)
        assert torch.equal(hook.isg_env.root_state[keep], fused.root_state[keep]), f"root step {it}"
        assert torch.equal(hook.actions_recorder.history_buf, T[_abi.A1_HISTORY]), f"history step {it}"
        if d2.any():
            ids = d2.nonzero().flatten()
            # the reset step itself: everything but the new command (obs[0:3]) and the random spawn xy
            _compare_step(hook, fused, it, d2, o1, o2, r1, r2, obs_from=3)
            assert torch.equal(hook.isg_env.root_state[ids][:, 2:], fused.root_state[ids][:, 2:])
            assert (o2[ids][:, 12 + 12:12 + 24] == 0).all() and (o2[ids][:, 36:72] == 0).all()   # dof_vel, history
            assert torch.equal(T[_abi.A1_HISTORY][ids][:, :, 0], fused.actions[ids])             # then a_k lands in slot 0
            assert torch.equal(hook.terrain_levels, fused.terrain_levels)
            assert torch.equal(hook.isg_env.env_origins, T[_abi.A1_ORIGINS])
            # extras["episode"] = mean(sum[env_ids]) / T_s over the envs that finished this step (env.py:149-158)
            for k, name in enumerate(hook.episode_rewards):
                torch.testing.assert_close(x1["episode"][name], x2["episode"][name], rtol=1e-5, atol=1e-6)
            assert float(x2["episode_sums"][7]) == len(ids)
            # adopt the fused env's draws and go on
            for tid in (_abi.T_ROOT_STATE, _abi.T_SIM_ROOT):
                be.tensors[tid][ids] = S[_abi.T_ROOT_STATE][ids]
            hook.command_buf[ids] = T[_abi.A1_COMMAND][ids]
            hook.robot.rand_force_buf[ids] = T[_abi.A1_PUSH][ids]
            resets += len(ids)
            timeouts += int(fused.time_out_buf.sum())
        for k, name in enumerate(hook.episode_rewards):
            torch.testing.assert_close(hook.episode_rewards[name], T[_abi.A1_REW_SUMS][k], rtol=1e-5, atol=1e-5)
        after_reset += int((was_reset & keep).sum())
        was_reset |= d2
    assert resets > 10 and timeouts >= 8 and after_reset > 100, (resets, timeouts, after_reset)


def test_hook_env_random_run_mode_is_stable():
    """run_policy(run_mode='random') shape: 200 random steps with resets, everything finite."""
    hook, _ = _envs(64)
    hook.reset()
    for _ in range(200):
        a = 2 * torch.rand(hook.num_envs, hook.num_actions, device=hook.device) - 1
        obs, _, rew, done, extras = hook.step(a)
    assert torch.isfinite(obs).all() and torch.isfinite(rew).all()
    assert "episode" in extras and "tracking_lin_vel" in extras["episode"] and "terrain_levels" in extras["episode"]
    assert extras["time_outs"].dtype == torch.bool


def _abb(n=32):
    from examples.abb_pushbox_vision.a_prior_stage import AbbPushBox
    from examples.abb_pushbox_vision.task_config import PriorStageEnvConfig
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU is visible")
    cfg = PriorStageEnvConfig()
    cfg.num_envs = n
    np.random.seed(3)
    torch.manual_seed(3)
    return AbbPushBox(cfg)


def test_abb_pushbox_random_run_mode():
    """Config 5 through the boundary: arm + table + cube + goal (4 actors / env), IK on the
    backend's Jacobian tensor, POS drives, 5+1 sub-steps of 20 ms, resets."""
    env = _abb(32)
    assert env.isg_env.root_state.shape == (32 * 4, 13) and env.isg_env.body_state.shape == (32 * 10, 13)
    assert env.robot.j_ee.shape == (32, 6, 6) and env.robot.num_dof == 6 and env.robot.num_bodies == 7
    env.reset()
    for _ in range(30):      # zero actions: the IK pulls the rod tip into the workspace slab z in [0.11, 0.14]
        env.step(torch.zeros(env.num_envs, env.num_actions, device=env.device))
    ee0 = env.robot.ee_pose[:, 0, :3].clone()
    assert ((ee0[:, 2] > 0.10) & (ee0[:, 2] < 0.15)).all(), ee0[:4]
    resets = 0
    for _ in range(120):
        a = 2 * torch.rand(env.num_envs, env.num_actions, device=env.device) - 1
        obs, _, rew, done, extras = env.step(a)
        resets += int(done.sum())
    assert torch.isfinite(obs).all() and torch.isfinite(rew).all()
    assert obs.shape == (32, 6) and "success_rate" in extras["episode"]
    cube = env.cube.base_pose
    assert ((cube[:, 2] > 0.11) & (cube[:, 2] < 0.16)).all(), "cubes must stay on the table"
    assert resets > 0


def test_graph_hooks_refuse_uncapturable_hooks_and_leave_the_env_usable():
    """A user hook that builds a device tensor from a Python list (a host-to-device copy from pageable memory, as the
    reference's AbbRobot.step does, a_prior_stage.py:72) cannot be captured: enable_graph_hooks must say so and leave the
    env on its eager path, not half-captured."""
    from examples.abb_pushbox_vision.a_prior_stage import AbbPushBox
    from examples.abb_pushbox_vision.task_config import PriorStageEnvConfig
    from shifu_amd._lib import BackendError

    class ListObs(AbbPushBox):
        def compute_observations(self):
            bias = torch.tensor([0.0, 0.0, 0.0], device=self.device)        # built from a list on every call
            super().compute_observations()
            self.obs_buf[:, :3] += bias

    cfg = PriorStageEnvConfig()
    cfg.num_envs = 32
    np.random.seed(3)
    torch.manual_seed(3)
    env = ListObs(cfg)
    env.reset()
    with pytest.raises(BackendError, match="cannot be captured"):
        env.enable_graph_hooks()
    assert getattr(env, "_hook_graphs", None) is None
    for _ in range(5):
        obs, _, rew, _, _ = env.step(torch.zeros(env.num_envs, env.num_actions, device=env.device))
    torch.cuda.synchronize()
    assert torch.isfinite(obs).all() and torch.isfinite(rew).all()


def test_abb_pushbox_replays_its_hooks_from_graphs():
    """The repo's AbbPushBox example (box re-spawn vectorised, the end-effector target quaternion built once) is capturable:
    with enable_graph_hooks it steps from two hipGraphs -- 300 steps with time-outs and re-spawns, the rod reaches the cube.
    (Replay against eager, bit for bit, is tested on A1Conditional above: a re-spawn inside a graph draws from the graph's own
    Philox offsets, so the two ABB envs part at their first reset.)"""
    env = _abb(32)
    env.reset()
    env.enable_graph_hooks()
    assert getattr(env, "_hook_graphs", None) is not None
    g = torch.Generator().manual_seed(5)
    dones, best = 0, -1e9
    for it in range(300):
        a = (2 * torch.rand(32, env.num_actions, generator=g) - 1).to(env.device)
        obs, _, rew, done, extras = env.step(a)
        dones += int(done.sum())
        best = max(best, float(rew.max()))
    torch.cuda.synchronize()
    assert torch.isfinite(obs).all() and torch.isfinite(rew).all()
    assert dones >= 32, dones            # every env timed out or succeeded at least once on average
    assert "episode" in extras


def test_abb_rod_pushes_the_cube():
    env = _abb(16)
    env.reset()
    be = env.isg_env.sim.backend
    n, A = 16, 4
    root = env.isg_env.root_state
    root[2::A, :3] = torch.tensor([0.15, 0.15, 0.125], device=root.device)     # cube and goal parked apart and out of the
    root[3::A, :3] = torch.tensor([0.18, -0.15, 0.1], device=root.device)      # rod's way: no episode ends while it settles
    be.commit_root_all(root)
    for _ in range(25):   # let the IK settle the rod tip into the workspace (the first step after a reset
        env.step(torch.zeros(env.num_envs, env.num_actions, device=env.device))   # sees a stale ee pose)
        assert not env.reset_buf.any()
    ee0 = env.robot.ee_pose[:, 0, :3].clone()
    # (reset() steps once with the cube where the task spawned it: an arm whose rod overlapped the cube there was pushed
    # by it -- the pair contact is solved consistently, the arm feels its cubes -- and holds a slightly different pose)
    calm = (ee0 - ee0.median(0).values).abs().max(1).values < 5e-3
    assert calm.float().mean() >= 0.5, "most arms settle on (nearly) the same pose"
    root[2::A, :3] = torch.tensor([0.07, 0.0, 0.125], device=root.device)      # cube in front of the rod (+x)
    root[2::A, 3:7] = torch.tensor([0, 0, 0, 1.0], device=root.device)
    root[2::A, 7:] = 0
    be.commit_root_all(root)
    a = torch.tensor([[1.0, 0.0, -1.0]], device=root.device).repeat(n, 1)      # +x, rod tip down to z = 0.11
    x0 = env.cube.base_pose[:, 0].clone()
    felt = torch.zeros(n, device=root.device)
    for _ in range(4):
        env.step(a)
        assert not env.reset_buf[calm].any()      # (an arm that the spawn pushed aside may overlap the cube placed here)
        felt += env.robot.ee_forces.abs().sum((1, 2)) if env.robot.ee_forces.dim() == 3 else env.robot.ee_forces.abs().sum(1)
        felt += env.isg_env.contact_state.view(n, 10, 3)[:, :7].abs().sum((1, 2))
    moved = env.cube.base_pose[:, 0] - x0
    ee = env.robot.ee_pose[:, 0, :3]
    assert (moved[calm] > 0.01).all(), f"cube was not pushed: {moved}"
    assert (ee[calm, 0] + 0.0194 + 0.025 <= env.cube.base_pose[calm, 0] + 0.012).all(), "rod must stay behind the cube face"
    assert (felt > 0).float().mean() > 0.8     # the arm feels the push (the tensors show the last sub-step only)
    assert torch.isfinite(root).all()


def test_fused_abb_env_tracks_hook_env():
    """FusedAbbEnv vs the hook-based AbbPushBox on the same state and actions.  Round 3: the hook path's
    ArmRobot.inverse_kinematics runs the same LDL^T kernel arithmetic as the fused step (csrc/shf_glue.hip: shf_ik_dls;
    it was torch.inverse, which left ~1e-4 between the two paths and forced cubes in contact out of the comparison), so
    until an env resets the two paths hold IDENTICAL root, dof and observation tensors, cubes in contact included."""
    from shifu_amd.gym.abb_fused import FusedAbbEnv
    n = 32
    hook = _abb(n)
    # the facade turns link contacts on (units.py:68) and the physx settings into the velocity-level solve (env_config.py:50-58)
    fused = FusedAbbEnv(num_envs=n, seed=5, link_contacts=True, solver="tgs")
    assert hook.isg_env.sim.solver == "tgs"
    be = hook.isg_env.sim.backend
    S = fused.sim.tensors
    for tid in (_abi.T_DOF_STATE, _abi.T_ROOT_STATE, _abi.T_BODY_STATE, _abi.T_JACOBIAN, _abi.T_CONTACT):
        be.tensors[tid].copy_(S[tid])
    be.tensors[_abi.T_SIM_DOF].copy_(S[_abi.T_DOF_STATE])
    be.tensors[_abi.T_SIM_ROOT].copy_(S[_abi.T_ROOT_STATE])
    hook.episode_length_buf.copy_(fused.episode_length_buf)
    g = torch.Generator(device="cuda:0")
    g.manual_seed(9)
    alive = torch.ones(n, dtype=torch.bool, device="cuda:0")
    touched = torch.zeros(n, dtype=torch.bool, device="cuda:0")
    cube0 = fused.root_state.view(n, 4, 13)[:, 2, :3].clone()
    compared, worst = 0, 0.0
    for it in range(25):
        a = 2 * torch.rand(n, 3, device="cuda:0", generator=g) - 1
        o1, _, r1, d1, _ = hook.step(a)
        o2, _, r2, d2, _ = fused.step(a)
        alive &= ~(d1.bool() | d2)
        if not alive.any():
            break
        free = alive                     # every env that has not reset yet, cubes in contact included
        if free.any():
            dr = (hook.isg_env.root_state.view(n, 4, 13)[free] - fused.root_state.view(n, 4, 13)[free]).abs()
            assert dr.max() == 0, f"root step {it}: max {dr.max()}"
            assert torch.equal(o1[free], o2[free]), f"obs step {it}"
            dq = (hook.isg_env.dof_state.view(n, 6, 2)[free] - fused.dof_state.view(n, 6, 2)[free]).abs()
            assert dq.max() == 0, f"dof_state step {it}: {dq.max()}"
            assert torch.allclose(r1[free], r2[free], atol=1e-6), f"rew step {it}"
            touched |= (fused.contact_state.view(n, 10, 3)[:, :7].abs().sum((1, 2)) > 0) & free
            worst = max(worst, float(dr.max()), float(dq.max()))
        compared += int(free.sum())
    assert compared > 3 * n and worst == 0.0
    assert int(touched.sum()) > 0, "the comparison must include envs whose rod was in contact"


@pytest.mark.gpu
def test_ppo_trainer_runs_on_the_fused_env(tmp_path):
    """OnPolicyRunner drives FusedA1Env (in-place observation buffers, time-outs, episode extras),
    checkpoints, and a reloaded policy reproduces the actions."""
    from examples.a1_conditional.task_config import A1PPOConfig
    from shifu_amd.gym.a1_fused import FusedA1Env
    from shifu_amd.rl import OnPolicyRunner
    from shifu_amd.runner.utils import class_to_dict
    torch.manual_seed(0)
    env = FusedA1Env(num_envs=256, group=32)
    runner = OnPolicyRunner(env, class_to_dict(A1PPOConfig()), log_dir=str(tmp_path), device="cuda:0")
    runner.learn(3, init_at_random_ep_len=True)
    assert len(runner.history) == 3
    for h in runner.history:
        assert np.isfinite(h["value_loss"]) and np.isfinite(h["surrogate_loss"]) and h["fps"] > 0
        assert "episode/tracking_lin_vel" in h
    assert int(env.episode_length_buf.max()) <= int(env.max_episode_length) + 1
    other = OnPolicyRunner(FusedA1Env(num_envs=64, group=32), class_to_dict(A1PPOConfig()), log_dir=None, device="cuda:0")
    other.load(str(tmp_path / "model_3.pt"))
    obs = env.get_observations()[:16].clone()
    torch.testing.assert_close(other.get_inference_policy()(obs), runner.get_inference_policy()(obs))


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["a1", "abb"])
def test_checkpoint_restore_replays_bit_exact(kind, tmp_path):
    """state_dict -> (torch.save / load) -> load_state_dict, then the same actions give the same
    run, including resets drawn after the checkpoint (the Philox counters are part of the state)."""
    from shifu_amd.checkpoint import TrajectoryRecorder
    if kind == "a1":
        from shifu_amd.gym.a1_fused import FusedA1Env
        env = FusedA1Env(num_envs=96, group=32, episode_length_s=0.6)
    else:
        from shifu_amd.gym.abb_fused import FusedAbbEnv
        env = FusedAbbEnv(num_envs=96, episode_length_s=3.0)      # 30 steps: time-outs fall inside the replayed window
    g = torch.Generator(device="cuda").manual_seed(3)
    acts = [torch.rand(env.num_envs, env.num_actions, device="cuda", generator=g) * 2 - 1 for _ in range(60)]
    env.reset()
    for a in acts[:20]:
        env.step(a)
    path = tmp_path / "ckpt.pt"
    torch.save(env.state_dict(), path)
    rec = TrajectoryRecorder(env, num_envs=8, bodies=True)

    def run():
        out = []
        for a in acts[20:]:
            obs, _, rew, done, ex = env.step(a)
            out.append((obs.clone(), rew.clone(), done.clone(), env.root_state.clone(), env.dof_state.clone()))
        return out
    first = run()
    assert sum(int(o[2].sum()) for o in first) > 0          # resets happened after the checkpoint
    env.load_state_dict(torch.load(path))
    second = []
    for a in acts[20:]:
        obs, _, rew, done, ex = env.step(a)
        rec.record()
        second.append((obs.clone(), rew.clone(), done.clone(), env.root_state.clone(), env.dof_state.clone()))
    for x, y in zip(first, second):
        for u, v in zip(x, y):
            assert torch.equal(u, v)
    z = np.load(rec.save(str(tmp_path / "traj.npz")))
    assert z["root"].shape[:2] == (40, 8) and z["dof"].shape[:2] == (40, 8) and z["body"].shape[-1] == 13
    assert len(z["body_names"]) <= z["body"].shape[2] and float(z["dt"]) > 0     # box actors follow the links
    # a checkpoint of another shape is refused
    other = type(env)(num_envs=32)
    with pytest.raises(ValueError):
        other.load_state_dict(torch.load(path))


def test_facade_uploads_per_env_link_masses():
    """gym.set_actor_rigid_body_properties(env, actor, props, recomputeInertia=True) on the robot of some envs
    (shifu/units/units.py:104-110): prepare_sim binds SHF_T_BODY_MASS_SCALE with the factors, and the heavier robots press
    harder on the ground once they have come to rest on their bellies."""
    import os
    from shifu_amd.isaacgym import gymapi
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU is visible")
    gym = gymapi.acquire_gym()
    sp = gymapi.SimParams()
    sp.dt = 0.005
    sp.up_axis = gymapi.UP_AXIS_Z
    sp.gravity = gymapi.Vec3(0.0, 0.0, -9.81)
    sim = gym.create_sim(0, 0, gymapi.SIM_PHYSX, sp)
    plane = gymapi.PlaneParams()
    plane.normal = gymapi.Vec3(0.0, 0.0, 1.0)
    gym.add_ground(sim, plane)
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "shifu_amd", "assets")
    urdf = [os.path.join(dp, f) for dp, _, fs in os.walk(root) for f in fs if f == "a1.urdf"][0]
    asset = gym.load_asset(sim, os.path.dirname(urdf), "a1.urdf", gymapi.AssetOptions())
    n, extra = 8, 3.0
    for e in range(n):
        env = gym.create_env(sim, gymapi.Vec3(-1, -1, -1), gymapi.Vec3(1, 1, 1), 3)
        pose = gymapi.Transform()
        pose.p = gymapi.Vec3(0.0, 0.0, 0.35)
        h = gym.create_actor(env, asset, pose, "a1", e, 0)
        if e % 2:
            props = gym.get_actor_rigid_body_properties(env, h)
            props[0].mass += extra
            gym.set_actor_rigid_body_properties(env, h, props, recomputeInertia=True)
    gym.prepare_sim(sim)
    be = sim.backend
    sc = be.tensors[_abi.T_BODY_MASS_SCALE].cpu().numpy()
    m0 = float(be.model.mass[0])
    assert sc.shape == (n, be.model.nb) and (sc[::2] == 1.0).all() and (sc[1::2, 1:] == 1.0).all()
    np.testing.assert_allclose(sc[1::2, 0], (m0 + extra) / m0, rtol=1e-6)
    for _ in range(600):
        gym.simulate(sim)
    gym.refresh_net_contact_force_tensor(sim)
    fz = be.tensors[_abi.T_CONTACT].reshape(n, be.model.nb, 3)[:, :, 2].sum(1).cpu().numpy()
    total = sum(float(be.model.mass[b]) for b in range(be.model.nb))
    assert np.isfinite(fz).all()
    np.testing.assert_allclose(fz[::2], total * 9.81, rtol=0.05)
    np.testing.assert_allclose(fz[1::2], (total + extra) * 9.81, rtol=0.05)


@pytest.mark.gpu
def test_contact_histogram_counts_every_substep_and_carries_the_drop_counter():
    """SHF_T_CONTACT_HIST (include/shifu_amd.h): while bound, every env adds one count per sub-step to the bin of its candidate
    count (before the max_contacts cap), the last column takes the dropped contacts -- exactly sum_k max(k - cap, 0) x bin k --
    and SHF_T_DROPPED stands still; unbound, SHF_T_DROPPED counts again.  Fused A1 step (chain kernel) and the hook path."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from shifu_amd import _abi
    from shifu_amd.gym.a1_fused import FusedA1Env
    n, steps = 256, 30
    env = FusedA1Env(num_envs=n, seed=3)
    assert env.solver == "tgs"
    env.reset()
    for _ in range(10):
        env.task.step_random()
    torch.cuda.synchronize()
    d0 = env.sim.tensors[_abi.T_DROPPED].clone()
    ht = env.sim.bind_contact_hist(True)
    for _ in range(steps):
        env.task.step_random()
    torch.cuda.synchronize()
    h = ht.cpu().numpy().astype(np.int64)
    bins = _abi.CONTACT_HIST_BINS
    assert (h[:, :bins].sum(1) == steps * 5).all()                      # 4 control sub-steps + the refresh sub-step, every env
    assert h[:, bins - 1].sum() == 0                                      # nobody beyond the last bin here
    want = (np.maximum(np.arange(bins) - 8, 0)[None, :] * h[:, :bins]).sum(1)
    np.testing.assert_array_equal(h[:, bins], want)
    assert want.sum() > 0 and torch.equal(env.sim.tensors[_abi.T_DROPPED], d0)
    env.sim.bind_contact_hist(False)
    for _ in range(5):
        env.task.step_random()
    torch.cuda.synchronize()
    assert int((env.sim.tensors[_abi.T_DROPPED] - d0).sum()) > 0
