"""bench.py end to end as the driver launches it: the N > 1 launcher path in a child process --
two ranks over gloo sharing the one GPU of the test box (RCCL refuses two ranks on one device), and the RCCL path itself
with one rank (SHIFU_AMD_FORCE_DIST=1: process group, barriers, all-gather of the episode statistics, MAX all-reduce
of the elapsed time).  The JSON lines are kept under gpurun_out/ (copied to profiles/ by the builder)."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_bench(args, extra_env, tag):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    env = dict(os.environ)
    env.update(extra_env)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, f"rank 0 prints exactly one JSON line, got {len(lines)}"
    out = json.loads(lines[0])
    log_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(log_dir):
        with open(os.path.join(log_dir, f"bench_{tag}.json"), "w") as f:
            f.write(lines[0] + "\n")
    return out


def test_bench_two_ranks_in_child_processes_over_gloo():
    """`python bench.py --gpus 2 ...` started bare becomes the launcher (torch.distributed.run, one child per rank)."""
    out = _run_bench(["--gpus", "2", "--steps", "30", "--warmup", "5", "--no-cpu-baseline"],
                     {"SHIFU_AMD_DIST_BACKEND": "gloo"}, "2rank_gloo")
    assert out["n_gpus"] == 2 and out["config"]["total_envs"] == 2 * 4096 and out["config"]["envs_per_gpu"] == 4096
    assert out["config"]["obs_finite"] is True
    assert out["config"]["gathers_in_timed_region"] >= 1, "steps 6..35 of the run contain the 24th: one all-gather is timed"
    assert out["scaling"] == "weak" and out["value"] > 0 and out["steps"] == 30 and out["warmup"] == 5
    assert out["cpu_baseline"] is None, "the CPU leg runs at N = 1 only"
    # the process group's own count and the spread of the per-rank clocks
    assert out["dist"]["backend"] == "gloo" and out["dist"]["world_size"] == 2 and out["rccl_world_size"] is None
    assert out["dist"]["ms_per_step_rank_min"] <= out["dist"]["ms_per_step_rank_max"] == out["ms_per_step"]
    # every rank's own report: which device it ran on, its clock, what its timed all-gathers cost
    ranks = out["dist"]["ranks"]
    assert [r["rank"] for r in ranks] == [0, 1] and len({r["pid"] for r in ranks}) == 2
    assert all(r["device"] and r["ms_per_step"] > 0 and len(r["all_gather_ms"]) >= 1 for r in ranks)
    assert out["dist"]["distinct_devices"] == 1, "two gloo ranks share the test box's one GPU"
    assert max(r["ms_per_step"] for r in ranks) == pytest.approx(out["ms_per_step"])
    assert out["dist"]["all_gather_ms_max"] > 0


def test_bench_two_ranks_over_rccl_on_two_gpus():
    """The real thing where the box has two GPUs (skipped on the one-GPU test box): two ranks over RCCL, each on its own
    device -- distinct PCI ids in the line, rccl_world_size 2."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    out = _run_bench(["--gpus", "2", "--steps", "30", "--warmup", "5", "--no-cpu-baseline"], {}, "2rank_rccl")
    assert out["n_gpus"] == 2 and out["rccl_world_size"] == 2 and out["dist"]["backend"] == "nccl"
    assert out["dist"]["distinct_devices"] == 2 and out["config"]["total_envs"] == 2 * 4096
    assert all(len(r["all_gather_ms"]) >= 1 for r in out["dist"]["ranks"])


def test_a_failing_rank_fails_the_job():
    """A rank that dies must take the job's exit status with it (no hang, no silent success): rank 1 is made to exit with
    status 3 before the process group forms; the launcher returns non-zero within the rendezvous timeout."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    env = dict(os.environ)
    env.update({"SHIFU_AMD_DIST_BACKEND": "gloo", "SHIFU_AMD_DIST_TIMEOUT_S": "60"})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1", "--no-cpu-baseline",
                        "--test-fail-rank", "1"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")], "no result line from a job that lost a rank"


def test_bench_one_rank_over_rccl():
    """The same code path on RCCL (backend nccl) with a single rank: what every rank of the driver's 8-GPU run executes."""
    out = _run_bench(["--gpus", "1", "--steps", "30", "--warmup", "5", "--no-cpu-baseline"],
                     {"SHIFU_AMD_FORCE_DIST": "1", "RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1",
                      "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(_free_port())}, "1rank_rccl")
    assert out["n_gpus"] == 1 and out["config"]["total_envs"] == 4096
    assert out["config"]["obs_finite"] is True and out["config"]["gathers_in_timed_region"] >= 1
    assert out["rccl_world_size"] == 1 and out["dist"]["backend"] == "nccl"
    rf = out["roofline"]
    assert abs(rf["frac"] * rf["peak"] - rf["frac_of_achievable"] * rf["peak_achievable"]) < 1e-6 * rf["achieved"] + 1e-9
    assert ("kernel_ms_note" in rf) == (rf["kernel_ms"] > out["ms_per_step"])
