"""Builds the HIP library in-tree: shifu_amd/libshifu_amd.so (gfx950 only).

hipcc cross-compiles without a GPU.  -ffp-contract=off / no fast-math are part of
the arithmetic contract documented in csrc/shf_device.h.
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libshifu_amd.so")
SOURCES = ["shf_api.hip"]
DEPS = ["shf_api.hip", "shf_device.h", "shf_boxes.h", os.path.join("..", "..", "include", "shifu_amd.h")]
# -fno-slp-vectorize: the SLP vectoriser packs neighbouring scalar f32 ops into v_pk_* pairs plus the
# v_mov shuffles that feed them -- slower for this kernel (measured -6 % at 2 envs/wave, -25 % at one
# wavefront per env; cf. MI355X_MICROARCH.md "packed f32 VALU ... an anti-lever").
# -amdgpu-sched-strategy=max-ilp: the default scheduler minimises registers for occupancy and ends up waiting
# on every LDS read individually (s_waitcnt lgkmcnt(0) after each ds_read); these kernels run at <= 2 waves per
# SIMD and are bound by single-wave latency, so batching the reads matters more (measured 0.087 -> 0.079 ms).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize",
         "-mllvm", "-amdgpu-sched-strategy=max-ilp", "-fPIC", "-shared"]


def hipcc() -> str:
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the MI355X backend cannot be built")


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS)


def build_native(force: bool = False, verbose: bool = False) -> str:
    if force or needs_build():
        cmd = [hipcc()] + FLAGS + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_native(force=True, verbose=True))
