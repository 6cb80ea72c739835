"""csrc/shf_glue.hip -- the library glue of the hook-compatible path as single launches --
against the oracle's exported glue functions (the same static functions the fused oracle step calls, pinned by golden
G3-G6) and against the reference's torch expressions as the mirror classes keep them for CPU tensors."""
import numpy as np
import pytest

from shifu_amd import _abi

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU is visible")


@pytest.fixture(scope="module")
def oracle():
    from oracle import pyoracle
    pyoracle.build()
    return pyoracle


def _roots(rng, rows):
    r = np.zeros((rows, 13), np.float32)
    r[:, :3] = rng.uniform(-3, 60, (rows, 3))
    q = rng.normal(0, 1, (rows, 4))
    r[:, 3:7] = q / np.linalg.norm(q, axis=1, keepdims=True)
    r[:, 7:] = rng.uniform(-3, 3, (rows, 6))
    return r


@pytest.mark.parametrize("actors", [1, 4])
def test_base_frame_state_matches_oracle_bitwise(oracle, actors):
    """K3, LeggedRobot.post_step (robot.py:222-229): root rows picked through root_indices (4 actors per env: ABB scene)."""
    _need_gpu()
    from shifu_amd import glue
    rng = np.random.default_rng(1)
    n = 777
    root = _roots(rng, n * actors)
    idx = np.arange(n, dtype=np.int64) * actors
    dev = "cuda:0"
    lin, ang, pg, gv = (torch.full((n, 3), 7.0, device=dev) for _ in range(4))
    glue.base_frame_state(torch.from_numpy(root).to(dev), torch.from_numpy(idx).to(dev), 2, lin, ang, pg, gv)
    rows = root[idx]
    g = np.tile(np.array([0, 0, -1], np.float32), (n, 1))
    np.testing.assert_array_equal(lin.cpu().numpy(), oracle.quat_rotate_inverse(rows[:, 3:7], rows[:, 7:10]))
    np.testing.assert_array_equal(ang.cpu().numpy(), oracle.quat_rotate_inverse(rows[:, 3:7], rows[:, 10:13]))
    np.testing.assert_array_equal(pg.cpu().numpy(), oracle.quat_rotate_inverse(rows[:, 3:7], g))
    np.testing.assert_array_equal(gv.cpu().numpy(), g)
    # and the reference's torch expressions (bmm-based dot product: agreement to rounding, not to the bit)
    from shifu_amd.isaacgym.torch_utils import quat_rotate_inverse
    t = torch.from_numpy(rows).to(dev)
    assert torch.allclose(lin, quat_rotate_inverse(t[:, 3:7], t[:, 7:10]), atol=2e-6, rtol=1e-6)


@pytest.mark.parametrize("plane", [False, True])
def test_get_heights_matches_oracle_bitwise(oracle, plane):
    """K4, TerrainGymEnv.get_heights (isaac_gym.py:412-433): negative coordinates (truncation toward zero), points beyond
    the map (clipping), the 3-neighbour minimum."""
    _need_gpu()
    from shifu_amd import glue
    from shifu_amd.a1_task import height_points
    rng = np.random.default_rng(2)
    n = 333
    hs = rng.integers(-200, 400, (90, 70)).astype(np.int16)
    terr = _abi.ShfTerrain()
    terr.rows, terr.cols = (0, 0) if plane else hs.shape
    terr.hscale, terr.vscale, terr.border, terr.friction = 0.1, 0.005, 2.0, 1.0
    root = _roots(rng, n)
    root[:, :2] = rng.uniform(-4, 9, (n, 2))           # well outside the 9 m x 7 m map on every side
    pts = height_points()
    dev = "cuda:0"
    got = glue.get_heights(terr, None if plane else torch.from_numpy(hs).to(dev), torch.from_numpy(root).to(dev), None,
                           torch.from_numpy(pts).to(dev), n)
    want = oracle.glue_heights(terr, hs, pts, root)
    np.testing.assert_array_equal(got.cpu().numpy(), want)
    assert plane or np.unique(want).size > 50


def test_get_heights_lands_in_the_cells_torch_picks_on_cell_boundaries(oracle):
    """`(points / horizontal_scale).long()` (isaac_gym.py:425) on a GPU tensor is x * (1 / s) in torch (a Python-float
    divisor), not a division: sample points one float either side of every cell boundary must land in the cell the torch
    expression -- evaluated here on the GPU, as the reference would -- picks, and the oracle in the same one.  The map is
    strictly increasing in both indices, so the 3-neighbour minimum is the cell's own sample and identifies it."""
    _need_gpu()
    from shifu_amd import glue
    rows, cols = 90, 70
    hs = (np.arange(rows)[:, None] * cols + np.arange(cols)[None, :]).astype(np.int16)
    terr = _abi.ShfTerrain()
    terr.rows, terr.cols = rows, cols
    terr.hscale, terr.vscale, terr.border, terr.friction = 0.1, 1.0, 2.0, 1.0
    # one sample point per env at the env's own (x, y); identity attitude, so the yaw rotation is exact
    pts = np.zeros((1, 2), np.float32)
    k = np.arange(5, 80, dtype=np.float32)
    edge = (k * np.float32(0.1)).astype(np.float32)               # the float nearest to each boundary ...
    cand = np.concatenate([edge, np.nextafter(edge, np.float32(0)), np.nextafter(edge, np.float32(99)),
                           (k / np.float32(10)).astype(np.float32)])
    xs = (cand - np.float32(2.0)).astype(np.float32)              # ... minus the border the kernel adds back
    n = xs.size
    root = np.zeros((n, 13), np.float32)
    root[:, 6] = 1.0
    root[:, 0] = xs
    root[:, 1] = xs[::-1] * np.float32(0.7)
    dev = "cuda:0"
    rt = torch.from_numpy(root).to(dev)
    got = glue.get_heights(terr, torch.from_numpy(hs).to(dev), rt, None, torch.from_numpy(pts).to(dev), n)
    # the reference's statements on the GPU (isaac_gym.py:418-433)
    points = torch.zeros(n, 1, 3, device=dev) + rt[:, :3].unsqueeze(1)
    points += 2.0
    points = (points / 0.1).long()
    px = torch.clip(points[:, :, 0].view(-1), 0, rows - 2)
    py = torch.clip(points[:, :, 1].view(-1), 0, cols - 2)
    hst = torch.from_numpy(hs).to(dev)
    want = torch.min(torch.min(hst[px, py], hst[px + 1, py]), hst[px, py + 1]).view(n, -1) * 1.0
    assert torch.equal(got, want.to(torch.float32))
    np.testing.assert_array_equal(got.cpu().numpy(), oracle.glue_heights(terr, hs, pts, root))
    # the division the CPU pipeline performs lands elsewhere for some of these samples: the test has teeth
    cpu = np.clip(np.trunc((root[:, 0] + np.float32(2.0)) / np.float32(0.1)), 0, rows - 2)
    assert (cpu != px.cpu().numpy()).any()


def test_history_add_and_reset_match_oracle_and_torch(oracle):
    """K9, HistoryRecorder (train.py:12-35) through the mirror class: GPU kernels vs the oracle's glue_history and vs the
    class's own torch statements on CPU."""
    _need_gpu()
    from shifu_amd.utils.train import HistoryRecorder
    rng = np.random.default_rng(3)
    n, A, H, K = 130, 12, 3, 7
    x = rng.normal(0, 1, (K, n, A)).astype(np.float32)
    ids = np.array([0, 5, 77, 129], np.int64)
    bufs, flats = oracle.glue_history(x, 3, ids, H=H)
    g = HistoryRecorder((n, A), H, device="cuda:0")
    c = HistoryRecorder((n, A), H, device="cpu")
    for k in range(K):
        g.add(torch.from_numpy(x[k]).cuda())
        c.add(torch.from_numpy(x[k]))
        if k == 3:
            g.reset_idx(torch.from_numpy(ids).cuda())
            c.reset_idx(torch.from_numpy(ids))
        np.testing.assert_array_equal(g.history_buf.cpu().numpy(), bufs[k], err_msg=f"step {k}")
        np.testing.assert_array_equal(g.flatten().cpu().numpy(), flats[k])
        np.testing.assert_array_equal(c.history_buf.numpy(), bufs[k])


def test_episode_log_is_exact_and_zeroes_the_sums():
    """K7, ShifuVecEnv.log_info (env.py:149-158): per key mean(sums[ids]) / T in 2^-20 fixed point (the arithmetic of the
    fused kernels' statistics / oracle a1_stats), sums[ids] = 0, other rows untouched; repeated calls reuse the workspace."""
    _need_gpu()
    from shifu_amd import glue
    rng = np.random.default_rng(4)
    n, K, T = 4096, 6, 20.0
    dev = "cuda:0"
    log = glue.EpisodeLog(torch.device(dev))
    for rep, nids in enumerate((1, 300, 4096)):
        sums = [torch.from_numpy(rng.normal(0, 30, n).astype(np.float32)).to(dev) for _ in range(K)]
        before = [s.cpu().numpy().copy() for s in sums]
        ids = np.sort(rng.choice(n, nids, replace=False)).astype(np.int64)
        out = log(sums, torch.from_numpy(ids).to(dev), T).cpu().numpy()
        for k in range(K):
            acc = int(np.rint(before[k][ids].astype(np.float32) * np.float32(1048576.0)).astype(np.int64).sum())
            want = np.float32(acc) * np.float32(1.0 / 1048576.0) / np.float32(nids) / np.float32(T)
            assert out[k] == want, (rep, k, out[k], want)
            assert abs(out[k] - before[k][ids].mean() / T) < 1e-4 * max(1.0, abs(out[k]))      # = torch.mean to rounding
            after = sums[k].cpu().numpy()
            assert (after[ids] == 0).all()
            keep = np.ones(n, bool); keep[ids] = False
            np.testing.assert_array_equal(after[keep], before[k][keep])
    assert int(log.ws.abs().sum()) == 0, "the kernel leaves its workspace zero"


def test_reward_accumulate_equals_the_torch_loop_bitwise():
    """compute_reward (env.py:180-185): rew = 0 + r_0 + r_1 ... and sums_k += r_k -- float32 additions in the same order."""
    _need_gpu()
    from shifu_amd import glue
    g = torch.Generator(device="cuda:0").manual_seed(5)
    n, K = 5000, 6
    terms = [torch.randn(n, device="cuda:0", generator=g) * (10.0 ** (k - 3)) for k in range(K)]
    sums = [torch.randn(n, device="cuda:0", generator=g) for _ in range(K)]
    ref_sums = [s.clone() for s in sums]
    rew = torch.full((n,), 3.0, device="cuda:0")
    ref = torch.zeros(n, device="cuda:0")
    for k in range(K):
        ref_sums[k] += terms[k]
        ref[:] += terms[k]
    glue.reward_accumulate(terms, sums, rew)
    assert torch.equal(rew, ref)
    for a, b in zip(sums, ref_sums):
        assert torch.equal(a, b)


def test_hook_env_uses_the_glue_kernels_and_still_matches_the_torch_expressions():
    """The mirror classes on the GPU take the kernels (LeggedRobot.post_step, get_heights, HistoryRecorder, log_info,
    compute_reward); their outputs agree with the reference's torch expressions evaluated on the same state."""
    _need_gpu()
    from examples.a1_conditional.a1_conditional import A1Conditional
    from examples.a1_conditional.task_config import A1EnvConfig
    from shifu_amd.isaacgym.torch_utils import quat_rotate_inverse
    cfg = A1EnvConfig()
    cfg.num_envs = 64
    env = A1Conditional(cfg)
    env.reset()
    isg = env.isg_env
    assert isg._glue_terrain is not None
    g = torch.Generator(device=env.device).manual_seed(0)
    for it in range(30):
        env.step(2 * torch.rand(64, 12, device=env.device, generator=g) - 1)
    # get_heights: kernel vs the torch expressions (force the torch branch by hiding the glue terrain)
    got = isg.get_heights()
    keep, isg._glue_terrain = isg._glue_terrain, None
    want = isg.get_heights()
    isg._glue_terrain = keep
    assert torch.equal(got, want), "same truncation, clipping and 3-neighbour minimum"
    # base-frame state of the robot: post_step's kernel vs quat_rotate_inverse on the same root rows
    rob = isg.robot
    rob.post_step()
    rs = isg.root_state[rob.root_indices]
    assert torch.allclose(rob.base_lin_vel, quat_rotate_inverse(rs[:, 3:7], rs[:, 7:10]), atol=2e-6, rtol=1e-6)
    assert torch.allclose(rob.projected_gravity, quat_rotate_inverse(rs[:, 3:7], rob.gravity_vec), atol=2e-6, rtol=1e-6)
    assert "episode" in env.extras and all(torch.isfinite(v).all() for v in env.extras["episode"].values())
    assert torch.isfinite(env.obs_buf).all() and torch.isfinite(env.rew_buf).all()


def test_reset_dof_rows_equals_the_torch_statements():
    """Robot._reset_dof_state (robot.py:74-86): targets / positions to the defaults, velocities to zero, int32 actor ids."""
    _need_gpu()
    from shifu_amd import glue
    g = torch.Generator(device="cuda:0").manual_seed(6)
    n, nd, A = 500, 12, 4
    ds = torch.randn(n * nd, 2, device="cuda:0", generator=g)
    tg = torch.randn(n, nd, device="cuda:0", generator=g)
    q0 = torch.randn(nd, device="cuda:0", generator=g)
    ri = torch.arange(n, device="cuda:0") * A
    ids = torch.tensor([0, 7, 8, 499, 250], device="cuda:0")
    ds2, tg2 = ds.clone(), tg.clone()
    tg2[ids] = q0.clone()
    ds2.view(n, nd, 2)[..., 0][ids] = q0.clone()
    ds2.view(n, nd, 2)[..., 1][ids] = 0.
    aid = glue.reset_dof_rows(ds, tg, q0, ids, ri)
    assert torch.equal(ds, ds2) and torch.equal(tg, tg2)
    assert aid.dtype == torch.int32 and torch.equal(aid, ri[ids].to(torch.int32))


def test_ik_dls_matches_oracle_bitwise_and_torch_inverse_closely(oracle):
    """K10, ArmRobot.inverse_kinematics (robot.py:162-182) on strided views, as the class hands them over: bit-equal to
    the oracle's glue_ik (the fused ABB step's arithmetic, golden G9), and within G9's 3e-5 of the row scale of the
    reference's torch.inverse formulation."""
    _need_gpu()
    from shifu_amd import glue
    from shifu_amd.utils.torch_utils import quat_conjugate, quat_mul
    g = torch.Generator(device="cuda:0").manual_seed(8)
    n, nd, links = 600, 6, 6
    jac = torch.randn(n, links, 6, nd, device="cuda:0", generator=g)
    body = torch.randn(n, 10, 13, device="cuda:0", generator=g)
    body[:, :, 3:7] /= body[:, :, 3:7].norm(dim=-1, keepdim=True)
    dofs = torch.randn(n * nd, 2, device="cuda:0", generator=g)
    goal = torch.randn(n, 7, device="cuda:0", generator=g)
    goal[:, 3:7] /= goal[:, 3:7].norm(dim=-1, keepdim=True)
    j_ee, ee, q = jac[:, 4], body[:, 6, :7], dofs.view(n, nd, 2)[..., 0]
    got = glue.ik_dls(j_ee, q, ee, goal, 0.05)
    want = oracle.glue_ik(j_ee.cpu().numpy(), q.cpu().numpy(), ee[:, :3].cpu().numpy(), ee[:, 3:7].cpu().numpy(),
                          goal[:, :3].cpu().numpy(), goal[:, 3:7].cpu().numpy(), damping=0.05)
    np.testing.assert_array_equal(got.cpu().numpy(), want)          # the oracle returns dof_pos + u as well
    # against the float64 evaluation of the same formula, next to the reference's torch.inverse formulation (float32):
    # on these random (often ill-conditioned) Jacobians the LDL^T solve is at least as accurate
    exact = oracle.glue_ik(j_ee.cpu().numpy(), q.cpu().numpy(), ee[:, :3].cpu().numpy(), ee[:, 3:7].cpu().numpy(),
                           goal[:, :3].cpu().numpy(), goal[:, 3:7].cpu().numpy(), damping=0.05, f64=True)
    qr = quat_mul(goal[:, 3:7], quat_conjugate(ee[:, 3:7]))
    dpose = torch.cat([goal[:, :3] - ee[:, :3], qr[:, :3] * torch.sign(qr[:, 3]).unsqueeze(-1)], -1).unsqueeze(-1)
    jt = j_ee.transpose(1, 2)
    ref = q + (jt @ torch.inverse(j_ee @ jt + torch.eye(6, device="cuda:0") * 0.05 ** 2) @ dpose).view(n, nd)
    scale = np.abs(exact - q.cpu().numpy()).max(axis=1, keepdims=True)
    err_ours = (np.abs(got.cpu().numpy() - exact) / scale).max()
    err_torch = (np.abs(ref.cpu().numpy() - exact) / scale).max()
    assert err_ours < 2e-3 and err_ours <= 2.0 * err_torch + 1e-6, (err_ours, err_torch)   # 4e-4 each on this draw
