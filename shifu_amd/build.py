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
# shf_api.hip holds the C ABI and the launches; the kernel families of shf_kernels.h are instantiated by the shf_k_*.hip units
# (csrc/shf_kernel_list.h) so that they compile side by side -- as one unit the library took 4.5 minutes to build.
KERNEL_UNITS = ["shf_k_sim.hip", "shf_k_sim_link.hip", "shf_k_sim_hard.hip", "shf_k_sim_hard_wide.hip", "shf_k_a1.hip", "shf_k_abb.hip",
                "shf_k_abb_link.hip", "shf_k_abb_hard.hip", "shf_k_abb_ws.hip", "shf_k_abb_ws_hard.hip", "shf_k_sim_ext.hip", "shf_k_abb_ext.hip", "shf_k_hull_test.hip"]
UNITY_SOURCES = ["shf_api.hip", "shf_a1_chain.hip", "shf_glue.hip", "shf_mlp.hip", "shf_k_hull_test.hip"]      # with -DSHF_UNITY: shf_api.hip instantiates every kernel
SOURCES = UNITY_SOURCES + [u for u in KERNEL_UNITS if u not in UNITY_SOURCES]
HEADERS = ["shf_device.h", "shf_boxes.h", "shf_task.h", "shf_chain.h", "shf_chain_hard.h", "shf_hard.h", "shf_link.h", "shf_arm.h", "shf_kernels.h",
           "shf_kernel_list.h", "shf_hull.h", os.path.join("..", "..", "include", "shifu_amd.h")]
DEPS = SOURCES + HEADERS
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
           ("_Z10k_abb_stepILi16E9FixedDimsILi7ELi6ELi59ELi6ELi6EE10FixedSceneILi3ELi1ELi2EELb1ELi0ELb0ELb0EE", 512, 32),
           # ... and its arm-wave / box-wave form, the default at 16 lanes: two waves per SIMD, so 256 registers, of which the
           # link passes spill some (120 B of scratch at the end of round 4; 496 B cost 19 %)
           ("_Z13k_abb_step_wsILi512ELb1EE", 256, 160),
           # the velocity-level solve (ShfSimParams.solver = SHF_SOLVER_PGS): the default of both fused envs and of the gym facade since
           # round 5.  The chain-mapped A1 step must stay at two waves per SIMD without scratch (all four forms: height field / trimesh,
           # without / with self-collision); the hook path's sub-step at three waves per SIMD
           ("_Z14k_a1_chain_pgsILb0ELb0EE", 256, 0), ("_Z14k_a1_chain_pgsILb0ELb1EE", 256, 0),
           ("_Z14k_a1_chain_pgsILb1ELb0EE", 256, 0), ("_Z14k_a1_chain_pgsILb1ELb1EE", 256, 0),
           ("_Z20k_sim_step_chain_pgs", 168, 0),
           # ... and their TGS instantiations, the default since round 6 (physx.solver_type = 1)
           ("_Z14k_a1_chain_tgsILb0ELb0EE", 256, 0), ("_Z14k_a1_chain_tgsILb0ELb1EE", 256, 0),
           ("_Z14k_a1_chain_tgsILb1ELb0EE", 256, 0), ("_Z14k_a1_chain_tgsILb1ELb1EE", 256, 0),
           ("_Z20k_sim_step_chain_tgs", 168, 0),
           # config 5 under that solve: the wave-specialised step on compile-time shapes (round 6) without spills; the run-time-shaped
           # generic kernels (any articulation / scene) as they stand: sixteen envs per 512-thread workgroup at 256 registers
           ("_Z19k_abb_step_pgs_wide", 256, 272), ("_Z19k_sim_step_pgs_wide", 256, 48),
           # config 5's default under that solve since round 6 (arm wave + box wave, the solve regrouped at 32 lanes per env): two
           # waves per SIMD; the link passes and the solve share 256 registers (1184 B of scratch when a struct copy and a pointer
           # select had put five structs on the stack: 0.50 ms instead of 0.37)
           ("_Z18k_abb_step_ws_hardILb1EE", 256, 64), ("_Z18k_abb_step_ws_hardILb0EE", 256, 0),
           ("_Z10k_sim_stepILi32ELb0ELb0ELb0ELb1ELb0EE", 168, 0), ("_Z10k_sim_stepILi32ELb1ELb0ELb1ELb1ELb0EE", 256, 32)]


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


def _llvm_tool(name: str) -> str:
    for c in (os.path.join("/opt/rocm/lib/llvm/bin", name), shutil.which(name)):
        if c and os.path.exists(c):
            return c
    raise RuntimeError(name + " not found")


def kernel_code_hashes(obj: str) -> dict:
    """kernel symbol -> sha256 (first 16 hex digits) of its machine code in the gfx950 code object embedded in `obj`.
    profiles/traffic.json entries carry the hash of the kernel the counters were collected from; bench.py compares it with the
    loaded library's (libshifu_amd.resources.json) and says `counters_stale` when they differ."""
    import hashlib
    import struct
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        fat, co = os.path.join(td, "fat.bin"), os.path.join(td, "k.co")
        p = subprocess.run([_llvm_tool("llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", obj, fat], capture_output=True)
        if p.returncode != 0 or not os.path.exists(fat) or os.path.getsize(fat) == 0:
            return {}
        p = subprocess.run([_llvm_tool("clang-offload-bundler"), "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--unbundle",
                            "--input=" + fat, "--output=" + co], capture_output=True)
        if p.returncode != 0 or not os.path.exists(co):
            return {}
        data = open(co, "rb").read()
    if data[:4] != b"\x7fELF" or data[4] != 2:
        return {}
    shoff, = struct.unpack_from("<Q", data, 0x28)
    shentsize, shnum, _ = struct.unpack_from("<HHH", data, 0x3A)
    secs = [struct.unpack_from("<IIQQQQIIQQ", data, shoff + i * shentsize) for i in range(shnum)]   # name type flags addr off size link info align entsize
    out = {}
    for (_, typ, _, _, off, size, link, _, _, entsize) in secs:
        if typ != 2:                       # SHT_SYMTAB
            continue
        stroff = secs[link][4]
        for i in range(size // entsize):
            name_i, info, _, shndx, value, sz = struct.unpack_from("<IBBHQQ", data, off + i * entsize)
            if (info & 0xF) != 2 or sz == 0 or shndx == 0 or shndx >= shnum:      # STT_FUNC
                continue
            end = data.index(b"\0", stroff + name_i)
            name = data[stroff + name_i:end].decode()
            sec = secs[shndx]
            start = sec[4] + (value - sec[3])
            out[name] = hashlib.sha256(data[start:start + sz]).hexdigest()[:16]
    return out


def check_budgets(res: dict):
    for prefix, max_vgpr, max_scratch in BUDGETS:
        hits = [(k, v) for k, v in res.items() if k.startswith(prefix)]
        if not hits:
            raise RuntimeError(f"kernel {prefix}* missing from the build")
        for k, v in hits:
            if v.get("vgprs", 0) + v.get("agprs", 0) > max_vgpr or v.get("scratch", 0) > max_scratch:
                raise RuntimeError(f"{k}: {v} exceeds the register budget ({max_vgpr} VGPRs, {max_scratch} B scratch)")


def _unit_fresh(obj: str, dep: str, stamp: str, cmdline: str) -> bool:
    """True when `obj` was built by `cmdline` and is newer than every file its dependency list (-MD) names."""
    try:
        if open(stamp).read() != cmdline:
            return False
        t = os.path.getmtime(obj)
        words = open(dep).read().replace("\\\n", " ").split()[1:]
        return all(os.path.getmtime(w) <= t for w in words if not w.endswith(":"))
    except OSError:
        return False


def compile_all(flags, lib, extra=(), verbose=False, force=True) -> str:
    """One object per translation unit (compiled side by side), then the link; returns the resource-usage remarks.
    force=False reuses the objects whose sources, headers and command line have not changed (the remarks are kept beside them)."""
    from concurrent.futures import ThreadPoolExecutor
    objdir = os.path.join(HERE, "build", os.path.basename(lib) + ".obj")
    os.makedirs(objdir, exist_ok=True)
    # debug builds that read one set of device globals (the phase clock) compile the kernels in one unit
    unity = any("SHF_PHASE_CLOCK" in x or "SHF_UNITY" in x for x in extra)
    sources = UNITY_SOURCES if unity else SOURCES
    extra = list(extra) + (["-DSHF_UNITY"] if unity and "-DSHF_UNITY" not in extra else [])

    def one(src):
        base = os.path.join(objdir, src.replace(".hip", ""))
        obj, dep, stamp, rem = base + ".o", base + ".d", base + ".cmd", base + ".remarks"
        cmd = [hipcc()] + list(flags) + list(extra) + ["-Rpass-analysis=kernel-resource-usage", "-MD", "-MF", dep, "-c", os.path.join(CSRC, src), "-o", obj]
        line = " ".join(cmd)
        if not force and _unit_fresh(obj, dep, stamp, line) and os.path.exists(rem):
            return obj, open(rem).read()
        if verbose:
            print(line)
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        if p.returncode != 0:
            raise RuntimeError("hipcc failed on " + src + ":\n" + p.stderr[-4000:])
        with open(rem, "w") as f:
            f.write(p.stderr)
        with open(stamp, "w") as f:
            f.write(line)
        return obj, p.stderr

    compile_all.objects = []
    with ThreadPoolExecutor(max_workers=min(len(sources), os.cpu_count() or 4)) as ex:
        done = list(ex.map(one, sources))
    compile_all.objects = [o for o, _ in done]
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + [o for o, _ in done] + ["-o", lib]
    if verbose:
        print(" ".join(cmd))
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    if p.returncode != 0:
        raise RuntimeError("link failed:\n" + p.stderr[-4000:])
    return "\n".join(r for _, r in done)


def build_native(force: bool = False, verbose: bool = False) -> str:
    if force or needs_build() or not os.path.exists(RESOURCES):
        remarks = compile_all(FLAGS, LIB, verbose=verbose, force=force)
        res = parse_resources(remarks)
        for obj in getattr(compile_all, "objects", []):
            for name, h in kernel_code_hashes(obj).items():
                if name in res:
                    res[name]["code_sha"] = h
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
