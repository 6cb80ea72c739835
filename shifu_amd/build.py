"""Builds the HIP library in-tree: shifu_amd/libshifu_amd.so (gfx950 only).

hipcc cross-compiles without a GPU.  -ffp-contract=off / no fast-math are part of
the arithmetic contract documented in csrc/shf_device.h.
"""
import json
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libshifu_amd.so")
SOURCES = ["shf_api.hip", "shf_a1_chain.hip", "shf_glue.hip", "shf_mlp.hip"]
DEPS = SOURCES + ["shf_device.h", "shf_boxes.h", "shf_task.h", "shf_chain.h", "shf_chain_hard.h", "shf_hard.h", "shf_link.h", "shf_arm.h", os.path.join("..", "..", "include", "shifu_amd.h")]
# -fno-slp-vectorize: the SLP vectoriser packs neighbouring scalar f32 ops into v_pk_* pairs plus the
# v_mov shuffles that feed them -- slower for this kernel (measured -6 % at 2 envs/wave, -25 % at one
# wavefront per env; cf. MI355X_MICROARCH.md "packed f32 VALU ... an anti-lever").
# -amdgpu-sched-strategy=max-ilp: the default scheduler minimises registers for occupancy and ends up waiting
# on every LDS read individually (s_waitcnt lgkmcnt(0) after each ds_read); these kernels run at <= 2 waves per
# SIMD and are bound by single-wave latency, so batching the reads matters more (measured 0.087 -> 0.079 ms).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize",
         "-mllvm", "-amdgpu-sched-strategy=max-ilp", "-fPIC"]


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


RESOURCES = os.path.join(HERE, "libshifu_amd.resources.json")
# (kernel-name prefix, max VGPRs, max scratch bytes): the fused A1 step at two envs per wavefront runs two
# waves per SIMD (4096 envs resident at once) only while it stays within 256 VGPRs, and any scratch use there
# means an array stopped living in registers -- both have cost >30 % when they slipped in unnoticed.
BUDGETS = [("_Z16k_a1_step_a1_g32", 256, 0),          # default: A1, two envs per wavefront, height field
           ("_Z9k_a1_stepILi32E9FixedDims", 256, 0),  # the same with the trimesh terrain query compiled in
           ("_Z21k_a1_step_self_a1_g32", 256, 96),    # with self-collision: a few spilled registers are tolerated (68 B in round 3)
           # the kernels the fused envs launch by default since round 3 -- the chain-mapped A1 step (two waves per SIMD at 32
           # lanes per env, four envs per wave at 16) and the two-wave ABB step: a spill or a 257th register there has cost
           # more than 30 % before it was noticed
           ("_Z10k_a1_chainILi32ELb0E", 256, 0), ("_Z10k_a1_chainILi32ELb1E", 256, 0),   # (all four: height field / trimesh, without / with self-collision)
           ("_Z10k_a1_chainILi16ELb0E", 512, 0),      # four envs per wave = one wave per SIMD: no occupancy step to lose below 512

           ("_Z13k_abb_step_wsILi256ELb0EE", 256, 0),
           # FusedAbbEnv's default since round 4: the arm with link contacts at 16 lanes per env, one wave per SIMD (all 4096 envs
           # resident: 16 envs per CU share the 160 KB of LDS) -- 512 registers, and the 20 B of scratch every body-mapped
           # ABB instantiation has had since round 3
           ("_Z10k_abb_stepILi16E9FixedDimsILi7ELi6ELi59ELi6ELi6EE10FixedSceneILi3ELi1ELi2EELb1ELi0ELb0EE", 512, 32),
           # ... and its arm-wave / box-wave form, the default at 16 lanes: two waves per SIMD, so 256 registers, of which the
           # link passes spill some (120 B of scratch at the end of round 4; 496 B cost 19 %)
           ("_Z13k_abb_step_wsILi512ELb1EE", 256, 160)]


def parse_resources(remarks: str) -> dict:
    """kernel -> {vgprs, agprs, sgprs, scratch, spill, occupancy} from -Rpass-analysis=kernel-resource-usage."""
    out, cur = {}, None
    keys = {"VGPRs:": "vgprs", "AGPRs:": "agprs", "TotalSGPRs:": "sgprs", "ScratchSize [bytes/lane]:": "scratch",
            "VGPRs Spill:": "spill", "Occupancy [waves/SIMD]:": "occupancy"}
    for line in remarks.splitlines():
        if "remark:" not in line:
            continue
        body = line.split("remark:", 1)[1].replace("[-Rpass-analysis=kernel-resource-usage]", "").strip()
        if body.startswith("Function Name:"):
            cur = out.setdefault(body.split(":", 1)[1].strip(), {})
        elif cur is not None:
            for k, name in keys.items():
                if body.startswith(k):
                    cur[name] = int(body[len(k):].strip())
    return out


def check_budgets(res: dict):
    for prefix, max_vgpr, max_scratch in BUDGETS:
        hits = [(k, v) for k, v in res.items() if k.startswith(prefix)]
        if not hits:
            raise RuntimeError(f"kernel {prefix}* missing from the build")
        for k, v in hits:
            if v.get("vgprs", 0) + v.get("agprs", 0) > max_vgpr or v.get("scratch", 0) > max_scratch:
                raise RuntimeError(f"{k}: {v} exceeds the register budget ({max_vgpr} VGPRs, {max_scratch} B scratch)")


def compile_all(flags, lib, extra=(), verbose=False) -> str:
    """One object per translation unit (compiled side by side), then the link; returns the resource-usage remarks."""
    from concurrent.futures import ThreadPoolExecutor
    objdir = os.path.join(HERE, "build", os.path.basename(lib) + ".obj")
    os.makedirs(objdir, exist_ok=True)

    def one(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = [hipcc()] + list(flags) + list(extra) + ["-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        if p.returncode != 0:
            raise RuntimeError("hipcc failed on " + src + ":\n" + p.stderr[-4000:])
        return obj, p.stderr

    with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        done = list(ex.map(one, SOURCES))
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + [o for o, _ in done] + ["-o", lib]
    if verbose:
        print(" ".join(cmd))
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    if p.returncode != 0:
        raise RuntimeError("link failed:\n" + p.stderr[-4000:])
    return "\n".join(r for _, r in done)


def build_native(force: bool = False, verbose: bool = False) -> str:
    if force or needs_build() or not os.path.exists(RESOURCES):
        remarks = compile_all(FLAGS, LIB, verbose=verbose)
        res = parse_resources(remarks)
        try:
            check_budgets(res)        # before the report is cached: a failed budget must fail the next build_native() too
        except RuntimeError:
            if os.path.exists(RESOURCES):
                os.remove(RESOURCES)
            raise
        with open(RESOURCES, "w") as f:
            json.dump(res, f, indent=1, sort_keys=True)
    return LIB


if __name__ == "__main__":
    print(build_native(force=True, verbose=True))
