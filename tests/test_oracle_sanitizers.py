"""The CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer (sanitizers run on the CPU
restatement; the pool has no GPU sanitizer).  `make -C oracle asan` builds oracle/libshf_oracle_asan.so; the oracle
suites then run in a child interpreter with libasan preloaded and SHF_ORACLE_LIB pointing at that build."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SUITES = ["tests/test_golden.py", "tests/test_oracle_physics.py", "tests/test_contact_kats.py",
          "tests/test_self_collision.py", "tests/test_capsule_box.py", "tests/test_trimesh.py"]


def test_oracle_suites_are_clean_under_asan_and_ubsan():
    if os.environ.get("SHF_ORACLE_LIB"):
        pytest.skip("already inside the sanitizer run")
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    libubsan = subprocess.run(["gcc", "-print-file-name=libubsan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("gcc has no libasan here")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"])
    env = dict(os.environ)
    env.update({"LD_PRELOAD": libasan + (":" + libubsan if os.path.exists(libubsan) else ""),
                "SHF_ORACLE_LIB": os.path.join(ROOT, "oracle", "libshf_oracle_asan.so"),
                # CPython itself leaks at exit and maps memory ASan does not know: only the oracle's own accesses matter
                "ASAN_OPTIONS": "detect_leaks=0:abort_on_error=0:halt_on_error=1",
                "UBSAN_OPTIONS": "print_stacktrace=1:halt_on_error=1", "OMP_NUM_THREADS": "4"})
    suites = [s for s in SUITES if os.path.exists(os.path.join(ROOT, s))]
    p = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider"] + suites,
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    text = p.stdout + p.stderr
    assert "ERROR: AddressSanitizer" not in text and "runtime error:" not in text, text[-4000:]
    assert p.returncode == 0, text[-4000:]
    assert " passed" in p.stdout
