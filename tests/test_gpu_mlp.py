"""The trainer's MFMA layers (shifu_amd/csrc/shf_mlp.hip) against the plain fp32 torch reference of
the same op -- forward, input gradient, weight / bias gradients -- on the shapes of the A1 ActorCritic
(259 -> 512 -> 256 -> 128 -> 12 / 1, 24 576-row mini-batches and 4096-row rollout batches) and on ragged ones.

Tolerance, by operand precision (shf_mlp_set_precision): "bf16x3" (default: bf16 head + tail, three MFMAs, products good
to 2^-16) -- every output within 2e-4 of the tensor's largest entry and 2e-5 of it on average; "bf16" (operands rounded
once, 2^-9) -- 2e-2 / 3e-3.  The torch reference itself runs fp32 library GEMMs whose summation order differs, hence not
tighter.  Identity / asymmetric-operand checks pin the fragment layouts bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU is visible")


TOL = {"bf16x3": (2e-4, 2e-5), "bf16": (2e-2, 3e-3)}


def _close(got, ref, what, mode="bf16x3"):
    scale = float(ref.abs().max()) + 1e-12
    err = (got - ref).abs()
    tmax, tmean = TOL[mode]
    assert float(err.max()) <= tmax * scale, f"{what} [{mode}]: max err {float(err.max()):.3g} vs scale {scale:.3g}"
    assert float(err.mean()) <= tmean * scale, f"{what} [{mode}]: mean err {float(err.mean()):.3g} vs scale {scale:.3g}"


@pytest.fixture(params=["bf16x3", "bf16"])
def precision(request):
    from shifu_amd.rl import mfma_linear
    _need_gpu()
    before = mfma_linear.get_precision()
    mfma_linear.set_precision(request.param)
    yield request.param
    mfma_linear.set_precision(before)


def test_identity_and_asymmetric_operands_pin_the_fragment_layout(precision):
    _need_gpu()
    from shifu_amd.rl.mfma_linear import MfmaLinear
    dev = "cuda:0"
    lin = MfmaLinear(160, 160, elu=False).to(dev)
    with torch.no_grad():
        lin.weight.copy_(torch.eye(160)); lin.bias.zero_()
    x = (torch.arange(200 * 160, device=dev, dtype=torch.float32).reshape(200, 160) % 251) - 125.0      # exact in bf16
    assert torch.equal(lin(x), x)                                     # W = I: output = input, element for element
    with torch.no_grad():
        w = torch.zeros(160, 160)
        w[3, 7] = 2.0; w[150, 1] = -1.0; w[31, 159] = 0.5            # asymmetric: a row/column swap cannot pass
        lin.weight.copy_(w)
    y = lin(x)
    ref = x @ w.to(dev).t()
    assert torch.equal(y, ref)


@pytest.mark.parametrize("M,K,N,elu", [(24576, 259, 512, True), (24576, 512, 256, True), (4096, 256, 128, True),
                                       (24576, 128, 12, False), (24576, 128, 1, False), (50, 259, 512, True), (1000, 37, 5, True)])
def test_forward_and_backward_match_the_fp32_reference(M, K, N, elu, precision):
    _need_gpu()
    from shifu_amd.rl.mfma_linear import MfmaLinear
    torch.manual_seed(0)
    dev = "cuda:0"
    lin = MfmaLinear(K, N, elu=elu).to(dev)
    ref = torch.nn.Linear(K, N).to(dev)
    ref.load_state_dict({k: v for k, v in lin.state_dict().items()})
    x = torch.randn(M, K, device=dev) * 1.5
    x1, x2 = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    y = lin(x1)
    yr = ref(x2)
    if elu:
        yr = torch.nn.functional.elu(yr)
    _close(y, yr, "forward", precision)
    g = torch.randn(M, N, device=dev)
    y.backward(g)
    yr.backward(g)
    _close(x1.grad, x2.grad, "input gradient", precision)
    _close(lin.weight.grad, ref.weight.grad, "weight gradient", precision)
    _close(lin.bias.grad, ref.bias.grad, "bias gradient", precision)


def test_actor_critic_on_the_mfma_backend_matches_the_torch_backend_and_trains():
    _need_gpu()
    from shifu_amd.rl.actor_critic import ActorCritic
    torch.manual_seed(1)
    a = ActorCritic(259, 259, 12, actor_hidden_dims=[512, 256, 128], critic_hidden_dims=[512, 256, 128], mlp_backend="mfma").to("cuda:0")
    b = ActorCritic(259, 259, 12, actor_hidden_dims=[512, 256, 128], critic_hidden_dims=[512, 256, 128], mlp_backend="torch").to("cuda:0")
    assert list(a.state_dict().keys()) == list(b.state_dict().keys())       # rsl_rl's parameter names either way
    b.load_state_dict(a.state_dict())
    obs = torch.randn(4096, 259, device="cuda:0")
    _close(a.act_inference(obs), b.act_inference(obs), "actor")         # four layers deep, default precision (bf16x3)
    _close(a.evaluate(obs), b.evaluate(obs), "critic")
    # a few Adam steps on a regression target: the loss falls on the MFMA backend as on the fp32 one
    tgt = torch.tanh(obs[:, :12])
    for net in (a, b):
        opt = torch.optim.Adam(net.actor.parameters(), lr=1e-3)
        first = None
        for it in range(60):
            opt.zero_grad()
            loss = (net.act_inference(obs) - tgt).square().mean()
            loss.backward()
            opt.step()
            first = first if first is not None else float(loss)
        net.final = float(loss)
        assert net.final < 0.35 * first
    assert abs(a.final - b.final) < 0.25 * b.final


def test_two_rank_training_keeps_parameters_in_sync(tmp_path):
    """SURVEY 8f f1 + 8e: tools/train_a1.py on two ranks (gloo, sharing this box's one GPU -- RCCL refuses two ranks on one
    device): env shards per rank, gradients all-reduced per mini-batch, the KL averaged so that every rank takes the same
    learning-rate decision; after the updates both ranks hold bit-identical parameters."""
    _need_gpu()
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, SHIFU_AMD_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "tools", "train_a1.py"), "--iters", "3", "--envs", "256", "--quiet",
           "--out", str(tmp_path / "log")]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=root, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == 2 and rec["iterations"] == 3 and rec["ranks_in_sync"] is True
    assert rec["samples_per_s"] > 0 and all(np.isfinite(rec["param_checksum"]))


def _ppo_with_filled_storage(seed, graph_update, backend="mfma"):
    """A PPO trainer whose rollout buffer holds a synthetic rollout (256 envs x 24 steps)."""
    from shifu_amd.rl.actor_critic import ActorCritic
    from shifu_amd.rl.ppo import PPO
    torch.manual_seed(seed)
    dev = "cuda:0"
    ac = ActorCritic(259, 259, 12, actor_hidden_dims=[512, 256, 128], critic_hidden_dims=[512, 256, 128], mlp_backend=backend)
    alg = PPO(ac, num_learning_epochs=2, num_mini_batches=4, schedule="adaptive", desired_kl=0.01, learning_rate=1e-3,
              entropy_coef=0.01, device=dev, graph_update=graph_update)
    alg.init_storage(256, 24, [259], [259], [12])

    def fill(gen_seed):
        g = torch.Generator(device=dev).manual_seed(gen_seed)
        obs = torch.randn(256, 259, device=dev, generator=g)
        for _ in range(24):
            with torch.no_grad():
                alg.act(obs, obs)
            rew = torch.randn(256, device=dev, generator=g)
            done = torch.rand(256, device=dev, generator=g) < 0.05
            alg.process_env_step(rew, done, {})
            obs = torch.randn(256, 259, device=dev, generator=g)
        alg.compute_returns(obs)
    return alg, fill


@pytest.mark.parametrize("backend", ["mfma", "torch"])
def test_captured_update_equals_the_eager_update(backend):
    """PPO.graph_update replays one captured hipGraph per mini-batch step; parameters, optimizer state and learning rate
    after three updates must be bit-identical to the eagerly launched ones -- with the MFMA layers and with the stock ones."""
    _need_gpu()
    res = []
    for graph in (False, True):
        alg, fill = _ppo_with_filled_storage(3, graph, backend)
        for it in range(3):
            fill(100 + it)
            torch.manual_seed(7 + it)               # the mini-batch permutation
            alg.update()
        assert (alg._upd_graph is not None) == graph
        res.append(([p.detach().clone() for p in alg.actor_critic.parameters()], float(alg.lr),
                    [s["exp_avg"].clone() for s in alg.optimizer.state.values()]))
    (pa, lra, ma), (pb, lrb, mb) = res
    assert lra == lrb
    for x, y in zip(pa, pb):
        assert torch.equal(x, y)
    for x, y in zip(ma, mb):
        assert torch.equal(x, y)
