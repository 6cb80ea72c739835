"""Boundary entry points that the fused-step parity tests do not reach: the indexed commits behind
gym.set_actor_root_state_tensor_indexed / set_dof_state_tensor_indexed / set_dof_position_target_tensor_indexed
(reference shifu/gym/isaac_gym.py:54-73, shifu/units/robot.py:74-86), the real run_policy('random') driver
(shifu/runner/policy_runner.py:33-41) and BASELINE config 4 at its full global-id range on one GPU."""
import numpy as np
import pytest

from shifu_amd import _abi

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU is visible")


def _sims():
    from shifu_amd.gym.a1_fused import FusedA1Env
    from shifu_amd.gym.abb_fused import FusedAbbEnv
    return [("a1", FusedA1Env(num_envs=37, group=32), 1), ("abb", FusedAbbEnv(num_envs=29), 4)]


def test_indexed_commits_copy_exactly_the_indexed_rows():
    """Random actor-index subsets (with repeats, sim domain, int32): indexed rows of the solver-side tensor equal the
    source, every other row keeps its value; 1-actor A1 scene and the 4-actor ABB scene (root rows of any actor,
    dof / target rows addressed by the ROBOT's actor index = env * actors_per_env, robot.py:78)."""
    _need_gpu()
    rng = np.random.default_rng(0)
    for name, env, A in _sims():
        sim, S, n, nd = env.sim, env.sim.tensors, env.num_envs, env.sim.model.nd
        dev = sim.device
        # --- root: any actor row
        before = torch.randn_like(S[_abi.T_SIM_ROOT])
        S[_abi.T_SIM_ROOT].copy_(before)
        src = torch.randn_like(before)
        rows = rng.choice(n * A, size=min(23, n * A), replace=False)
        idx = torch.tensor(np.concatenate([rows, rows[:5]]), dtype=torch.int32, device=dev)      # repeats are legal
        sim.commit_root_indexed(src, idx)
        got = S[_abi.T_SIM_ROOT]
        mask = torch.zeros(n * A, dtype=torch.bool, device=dev)
        mask[torch.from_numpy(rows).to(dev)] = True
        assert torch.equal(got[mask], src[mask]), name
        assert torch.equal(got[~mask], before[~mask]), name
        # --- dof state and position targets: robot actor indices -> env rows
        envs = rng.choice(n, size=11, replace=False)
        ridx = torch.tensor(envs * A, dtype=torch.int32, device=dev)
        emask = torch.zeros(n, dtype=torch.bool, device=dev)
        emask[torch.from_numpy(envs).to(dev)] = True
        for tid, commit, width in ((_abi.T_SIM_DOF, sim.commit_dof_indexed, 2 * nd), (_abi.T_POS_TARGET, sim.set_pos_target_indexed, nd)):
            before = torch.randn_like(S[tid])
            S[tid].copy_(before)
            src = torch.randn_like(before)
            commit(src, ridx)
            g, s0, b0 = S[tid].view(n, width), src.view(n, width), before.view(n, width)
            assert torch.equal(g[emask], s0[emask]), (name, tid)
            assert torch.equal(g[~emask], b0[~emask]), (name, tid)
        # --- an index outside the sim is skipped, not written through (nothing else changes)
        keep = S[_abi.T_SIM_ROOT].clone()
        bad = torch.tensor([n * A, -1, n * A + 7], dtype=torch.int32, device=dev)
        sim.commit_root_indexed(torch.randn_like(keep), bad)
        sim.commit_dof_indexed(torch.randn_like(S[_abi.T_SIM_DOF]), bad)
        torch.cuda.synchronize()
        assert torch.equal(S[_abi.T_SIM_ROOT], keep), name
        env.destroy()


def test_the_commits_of_one_reset_as_one_launch_equal_the_three_calls():
    """shf_sim_commit_reset (ABI v15): inside Sim.begin_reset() .. commit_root_indexed() the position-target and dof-state
    commits are held and go out with the root rows in one launch -- the solver-side tensors end up exactly as after the three
    separate calls (random subsets with repeats and out-of-range entries); a step / refresh / unrelated commit in between
    flushes what is held; outside a bracket every commit acts at once."""
    _need_gpu()
    rng = np.random.default_rng(3)
    for name, env, A in _sims():
        sim, S, n, nd = env.sim, env.sim.tensors, env.num_envs, env.sim.model.nd
        dev = sim.device
        tids = (_abi.T_SIM_ROOT, _abi.T_SIM_DOF, _abi.T_POS_TARGET)
        start = {t: torch.randn_like(S[t]) for t in tids}
        src = {t: torch.randn_like(S[t]) for t in tids}
        rows = torch.tensor(list(rng.choice(n * A, size=19, replace=False)) + [n * A + 3, -2], dtype=torch.int32, device=dev)
        ridx = torch.tensor(list(rng.choice(n, size=9, replace=False) * A) + [n * A], dtype=torch.int32, device=dev)

        def load():
            for t in tids:
                S[t].copy_(start[t])
        load()
        sim.set_pos_target_indexed(src[_abi.T_POS_TARGET], ridx)
        sim.commit_dof_indexed(src[_abi.T_SIM_DOF], ridx)
        sim.commit_root_indexed(src[_abi.T_SIM_ROOT], rows)
        want = {t: S[t].clone() for t in tids}
        load()
        sim.begin_reset()
        sim.set_pos_target_indexed(src[_abi.T_POS_TARGET], ridx)
        sim.commit_dof_indexed(src[_abi.T_SIM_DOF], ridx[:ridx.numel()])          # (the facade's idx[:n] view: same memory)
        assert torch.equal(S[_abi.T_SIM_DOF], start[_abi.T_SIM_DOF]) and torch.equal(S[_abi.T_POS_TARGET], start[_abi.T_POS_TARGET]), "held"
        sim.commit_root_indexed(src[_abi.T_SIM_ROOT], rows)
        for t in tids:
            assert torch.equal(S[t], want[t]), (name, t)
        # held commits leave with the next launch of any other kind
        load()
        sim.begin_reset()
        sim.commit_dof_indexed(src[_abi.T_SIM_DOF], ridx)
        sim.refresh(_abi.REFRESH_DOF)
        assert torch.equal(S[_abi.T_SIM_DOF], want[_abi.T_SIM_DOF]) and sim._held is None and not sim._defer, name
        # ... and outside a bracket nothing is held
        load()
        sim.commit_dof_indexed(src[_abi.T_SIM_DOF], ridx)
        assert torch.equal(S[_abi.T_SIM_DOF], want[_abi.T_SIM_DOF]), name
        env.destroy()


def test_reset_bookkeeping_in_one_launch_equals_the_five_statements():
    """shf_reset_bookkeeping (ABI v15) against ShifuVecEnv.reset_idx's buffer statements in torch (env.py:114-130, 149-158):
    episode_length[ids] = 0, reset_buf[ids] = 1 (bool and int64 buffers), history[ids] = 0, extras means and zeroed sums."""
    _need_gpu()
    from shifu_amd import glue
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(5)
    n, K = 4096, 6
    for reset_dtype in (torch.bool, torch.int64):
        sums = [(torch.rand(n, generator=g) * 40 - 5).to(dev) for _ in range(K)]
        ep = torch.randint(1, 900, (n,), generator=g).to(dev)
        rb = (torch.rand(n, generator=g) < 0.3).to(dev).to(reset_dtype)
        hist = torch.rand(n, 12, 3, generator=g).to(dev)
        ids = torch.randperm(n, generator=g)[:777].to(dev)
        want_means = [float(torch.mean(s[ids].double()) / 20.0) for s in sums]
        w_sums, w_ep, w_rb, w_hist = [s.clone() for s in sums], ep.clone(), rb.clone(), hist.clone()
        for s in w_sums:
            s[ids] = 0.0
        w_ep[ids] = 0
        w_rb[ids] = 1
        w_hist.index_fill_(0, ids, 0.0)
        log = glue.EpisodeLog(dev)
        means = log(sums, ids, 20.0, episode_length=ep, reset_buf=rb, history=hist)
        torch.cuda.synchronize()
        assert torch.equal(ep, w_ep) and torch.equal(rb, w_rb) and torch.equal(hist, w_hist)
        assert all(torch.equal(a, b) for a, b in zip(sums, w_sums))
        np.testing.assert_allclose(means.cpu().numpy(), want_means, rtol=2e-5, atol=1e-6)
        assert int(log.ws.abs().sum()) == 0, "the workspace is left zero for the next call"


def test_indexed_commit_through_the_gym_facade_resets_only_those_envs():
    """IsaacGymEnv.reset_idx + Robot._reset_dof_state on the hook env: after reset_idx(ids) the solver state of `ids`
    is the default pose at the env origin, all other envs keep stepping from where they were."""
    _need_gpu()
    from examples.a1_conditional.a1_conditional import A1Conditional
    from examples.a1_conditional.task_config import A1EnvConfig
    cfg = A1EnvConfig()
    cfg.num_envs = 24
    np.random.seed(1)
    torch.manual_seed(1)
    env = A1Conditional(cfg)
    env.reset()
    for _ in range(5):
        env.step(2 * torch.rand(24, 12, device=env.device) - 1)
    be = env.isg_env.sim.backend
    dof0, root0 = be.tensors[_abi.T_SIM_DOF].clone().view(24, -1), be.tensors[_abi.T_SIM_ROOT].clone()
    ids = torch.tensor([1, 4, 5, 17, 23], device=env.device)
    env.isg_env.reset_idx(ids)
    dof1, root1 = be.tensors[_abi.T_SIM_DOF].view(24, -1), be.tensors[_abi.T_SIM_ROOT]
    other = torch.ones(24, dtype=torch.bool, device=env.device)
    other[ids] = False
    assert torch.equal(dof1[other], dof0[other]) and torch.equal(root1[other], root0[other])
    q0 = env.robot.default_dof_pos.flatten()[:12]
    assert torch.equal(dof1[ids].view(5, 12, 2)[..., 0], q0.expand(5, 12)) and (dof1[ids].view(5, 12, 2)[..., 1] == 0).all()
    assert torch.equal(root1[ids][:, 2], 0.42 + env.isg_env.env_origins[ids][:, 2]) and (root1[ids][:, 7:] == 0).all()
    assert ((root1[ids][:, :2] - env.isg_env.env_origins[ids][:, :2]).abs() <= 1.0).all()     # spawn xy += U(-1, 1)
    env.destroy()


def test_run_policy_random_mode_on_the_a1_example(capsys):
    """The reference's own driver for this path, called as the example's __main__ calls it."""
    _need_gpu()
    from examples.a1_conditional.a1_conditional import A1Conditional
    from examples.a1_conditional.task_config import A1EnvConfig, A1PPOConfig
    from shifu_amd.runner import run_policy
    torch.manual_seed(2)
    np.random.seed(2)
    env = run_policy(run_mode="random", env_class=A1Conditional, env_cfg=A1EnvConfig(), policy_cfg=A1PPOConfig(),
                     log_root="/tmp/shifu_amd_logs", play_num_envs=50, play_iterations=150)
    assert env.num_envs == 50 and env.obs_buf.shape == (50, 259)
    assert torch.isfinite(env.obs_buf).all() and torch.isfinite(env.rew_buf).all()
    assert env.common_step_counter == 151                       # reset() steps once, then 150 random steps
    assert "episode" in env.extras and env.extras["time_outs"].dtype == torch.bool
    assert int(env.episode_length_buf.max()) <= 151 and int(env.episode_length_buf.min()) < 100   # resets happened
    env.destroy()


def test_run_policy_random_mode_on_the_abb_example():
    _need_gpu()
    from examples.abb_pushbox_vision.a_prior_stage import AbbPushBox
    from examples.abb_pushbox_vision.task_config import PriorStageEnvConfig, PriorStagePPOConfig
    from shifu_amd.runner import run_policy
    torch.manual_seed(2)
    np.random.seed(2)
    env = run_policy(run_mode="random", env_class=AbbPushBox, env_cfg=PriorStageEnvConfig(), policy_cfg=PriorStagePPOConfig(),
                     log_root="/tmp/shifu_amd_logs", play_num_envs=50, play_iterations=60)
    assert env.num_envs == 50 and env.obs_buf.shape == (50, 6) and torch.isfinite(env.obs_buf).all()
    env.destroy()


def test_config4_eight_shards_reproduce_the_32768_env_run():
    """BASELINE config 4 (32 768 envs over 8 ranks) at its full global-id range on one GPU: eight sequential
    4096-env shards (env_id_offset 0 ... 28 672, world_size 8) against one unsharded 32 768-env run -- terrain
    columns, per-env friction and every reset draw are functions of the global id, so each shard must equal its
    slice of the big run bit for bit."""
    _need_gpu()
    from shifu_amd.gym.a1_fused import FusedA1Env
    N, W, K = 4096, 8, 14
    g = torch.Generator(device="cuda:0").manual_seed(11)
    acts = [2 * torch.rand(N * W, 12, device="cuda:0", generator=g) - 1 for _ in range(K)]
    big = FusedA1Env(num_envs=N * W, seed=9, episode_length_s=0.2)           # 10-step episodes: time-out resets inside the run
    for a in acts:
        big.step(a)
    torch.cuda.synchronize()
    ids_t = [(_abi.A1_OBS, "obs"), (_abi.A1_REW, "rew"), (_abi.A1_RESET, "reset"), (_abi.A1_COMMAND, "command"),
             (_abi.A1_HISTORY, "history"), (_abi.A1_LEVELS, "levels"), (_abi.A1_TYPES, "types"), (_abi.A1_EP_LEN, "ep_len"),
             (_abi.A1_RESET_COUNT, "reset_count"), (_abi.A1_HEIGHTS, "heights"), (_abi.A1_PUSH, "push")]
    ids_s = [(_abi.T_DOF_STATE, 12), (_abi.T_ROOT_STATE, 1), (_abi.T_BODY_STATE, 17), (_abi.T_CONTACT, 17), (_abi.T_FRICTION, 1)]
    ref_t = {k: big.task.tensors[k].clone() for k, _ in ids_t}
    ref_s = {k: big.sim.tensors[k].clone() for k, _ in ids_s}
    ref_rs = big.task.tensors[_abi.A1_REW_SUMS].clone()
    assert int(ref_t[_abi.A1_RESET_COUNT].min()) >= 2 and len(torch.unique(ref_t[_abi.A1_TYPES])) == 20
    big.destroy()
    del big
    for r in range(W):
        sh = FusedA1Env(num_envs=N, seed=9, episode_length_s=0.2, rank=r, world_size=W)
        assert sh.env_id_offset == r * N
        for a in acts:
            sh.step(a[r * N:(r + 1) * N])
        torch.cuda.synchronize()
        for k, name in ids_t:
            assert torch.equal(sh.task.tensors[k], ref_t[k][r * N:(r + 1) * N]), f"rank {r}: {name}"
        for k, per in ids_s:
            assert torch.equal(sh.sim.tensors[k], ref_s[k][r * N * per:(r + 1) * N * per]), f"rank {r}: sim tensor {k}"
        assert torch.equal(sh.task.tensors[_abi.A1_REW_SUMS], ref_rs[:, r * N:(r + 1) * N]), f"rank {r}: rew_sums"
        sh.destroy()
        del sh
