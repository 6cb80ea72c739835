#!/usr/bin/env python3
"""Build / run the kernel experiments kept under shifu_amd/csrc/experiments/ (tried, measured, not shipped).

    python tools/experiment.py build shuffle_handoff        # here: shifu_amd/libshifu_amd_exp_shuffle_handoff.so
    python tools/experiment.py run shuffle_handoff          # on the MI355X: bit-exactness vs the oracle + bench line

`run` executes the fused-A1 parity tests and bench.py with SHIFU_AMD_LIB pointing at the experiment's library, and the
same bench with the product library, and prints both kernel times."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
FLAGS = {"shuffle_handoff": ["-DSHF_EXP_SHUFFLE_HANDOFF"], "no_elink": ["-DSHF_EXP_NO_ELINK"], "no_ebox": ["-DSHF_EXP_NO_EBOX"],
         "no_edge": ["-DSHF_EXP_NO_ELINK", "-DSHF_EXP_NO_EBOX"]}


def lib(name):
    return os.path.join(ROOT, "shifu_amd", f"libshifu_amd_exp_{name}.so")


def build(name):
    from shifu_amd import build as b
    b.compile_all(b.FLAGS, lib(name), extra=FLAGS[name])
    print(lib(name))


def bench_kernel_ms(env):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "300", "--warmup", "30", "--no-cpu-baseline"],
                         capture_output=True, text=True, env=env, cwd=ROOT)
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    return d["roofline"]["kernel_ms"], d["value"]


def run(name):
    env = dict(os.environ, SHIFU_AMD_LIB=lib(name))
    t = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-m", "gpu", "-q", "-k",
                        "fused_a1_step_matches or simulate_matches or shard"], env=env, cwd=ROOT, capture_output=True, text=True)
    print(t.stdout[-400:])
    exp = bench_kernel_ms(env)
    ref = bench_kernel_ms(dict(os.environ))
    print(json.dumps({"experiment": name, "parity_tests_rc": t.returncode, "kernel_ms": exp[0], "env_steps_per_s": exp[1],
                      "product_kernel_ms": ref[0], "product_env_steps_per_s": ref[1]}))


if __name__ == "__main__":
    {"build": build, "run": run}[sys.argv[1]](sys.argv[2])
