#!/usr/bin/env python3
"""Per-phase latency breakdown of k_a1_step from in-kernel s_memtime marks.

    python tools/phase_clock.py build         # here (no GPU): debug library with -DSHF_PHASE_CLOCK
    python tools/phase_clock.py [G] [steps] [--abb]   # on the MI355X box

Thread 0 of block 0 accumulates the cycle count between PHASE_MARKs (csrc/shf_device.h); the marks
serialise the wave a little (s_memtime + waitcnt), so the total is a few % above the production kernel.
"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "shifu_amd", "libshifu_amd_phase.so")
NAMES = ["kin: local rotation", "kin: level loop", "inertia + ext force", "contact eval (3 rounds)", "contact accumulate",
         "dof efforts", "ABA inward", "root solve", "ABA outward", "contact force out", "integrate",
         "prologue (stage, load, actions)", "copy-out + history load", "body_states tail", "get_heights", "post_step (lane 0)",
         "obs + stores"]


def build():
    from shifu_amd import build as b
    b.compile_all(b.FLAGS, LIB, extra=["-DSHF_PHASE_CLOCK"])
    print(LIB)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        return build()
    abb = "--abb" in sys.argv
    chain = "--chain" in sys.argv     # the chain-per-lane A1 step (csrc/shf_chain.h); G = 16 or 32
    levels = "--levels" in sys.argv   # --abb: the level-by-level sub-step instead of the arm's recursions on one lane
    split = "--split" in sys.argv     # --abb: arm and boxes on different waves (k_abb_step_ws); marks 24-29 are the arm wave's
    link = "--link" in sys.argv       # --abb: with link contacts (the run-time-shaped kernel)
    pgs = "--pgs" in sys.argv         # --chain: the velocity-level contact solve (k_a1_chain_pgs, csrc/shf_chain_hard.h)
    selfc = "--self" in sys.argv      # --chain: with self-collision
    terrain = "trimesh" if "--trimesh" in sys.argv else "heightfield"
    hull = "--hull" in sys.argv       # --abb --link: the links as convex hulls (the EXT instantiation of the run-time-shaped kernel)
    argv = [a for a in sys.argv if a not in ("--abb", "--chain", "--levels", "--split", "--link", "--pgs", "--self", "--trimesh", "--hull")]
    G = int(argv[1]) if len(argv) > 1 else 32
    steps = int(argv[2]) if len(argv) > 2 else 100
    from shifu_amd import build as b
    deps = [os.path.join(b.CSRC, d) for d in b.DEPS]
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(d) for d in deps):
        build()                                  # a stale debug library would miss symbols the product library has
    os.environ["SHIFU_AMD_LIB"] = LIB
    import torch
    from shifu_amd import _lib
    from shifu_amd.gym.a1_fused import FusedA1Env
    from shifu_amd.gym.abb_fused import FusedAbbEnv
    # --abb: the sub-step phases (0-10) of the push-box env; its kernel has no marks outside the sub-steps
    env = (FusedAbbEnv(num_envs=4096, group=G, link_contacts=link, mapping="split" if split else ("body" if (levels or link) else "chain"),
                       solver="pgs" if pgs else "compliant", **({"link_shapes": "hull"} if hull else {})) if abb else
           FusedA1Env(num_envs=4096, group=G, mapping="chain" if chain else "body", solver="pgs" if pgs else "compliant",
                      self_collision=selfc, terrain=terrain))
    if chain:
        NAMES[0] = "dof lanes: drive efforts + local joint rotations"; NAMES[1] = "chain lanes: poses, velocities, motion subspaces"
        NAMES[2] = "body lanes: rigid inertias + external forces"; NAMES[5] = "(unused on the chain mapping)"
        NAMES[6] = "chain: dof efforts + inward"; NAMES[7] = "root: add hips + 6x6 solve"; NAMES[8] = "chain: outward + integrate"
        NAMES[10] = "root: integrate"
    if pgs and abb:
        NAMES[3] = "contact eval: candidate gaps of the terrain points"; NAMES[4] = "candidates of the box / link families + the solve (marks 17-20, 32-37 are its parts)"
        NAMES[9] = "integration + contact force out"
    if pgs and not abb:
        NAMES[3] = "points -> candidate constraints, selection"; NAMES[4] = "contact solve: H1-H4 (marks 5, 10, 17, 18 are its parts)"
        NAMES[5] = "  H1 response matrix W (column lanes)"; NAMES[10] = "  H2 owner setup (u0, targets, block inverses)"
        NAMES[6] = "free inward pass (row lanes)"; NAMES[7] = "root: sum, LDL^T, free acceleration"; NAMES[8] = "free outward pass + velocity rates"
        NAMES[9] = "integration + contact force out"
    lib = _lib.lib()
    fn = lib.shf_debug_phase_cycles
    fn.argtypes = [ctypes.POINTER(ctypes.c_uint64), ctypes.c_int, ctypes.c_int]
    buf = (ctypes.c_uint64 * 48)()
    env.reset()
    act = torch.empty(env.num_envs, env.num_actions, device=env.device)
    for _ in range(20):
        env.step(act.uniform_(-1, 1))
    torch.cuda.synchronize()
    assert fn(buf, 48, 1) == 0
    for _ in range(steps):
        env.step(act.uniform_(-1, 1))
    torch.cuda.synchronize()
    assert fn(buf, 48, 0) == 0
    tot = sum(buf[:17]) if not split else sum(buf[11:17]) + sum(buf[24:30])  # (marks 17-23: inside phase 4 for the box scene)
    if pgs and not abb:
        tot = sum(buf[:19])                       # chain kernel: the solve's parts are marks 5, 10, 17, 18
    if pgs and abb:
        tot = sum(buf[:24]) + sum(buf[32:38]) + (sum(buf[24:29]) if link else 0)    # generic kernel: every mark is its own interval
    if pgs and abb and split:
        tot = sum(buf[11:17]) + sum(buf[24:30])     # k_abb_step_ws_hard: the arm wave's intervals (6 and 32-37 lie inside 26 and 28)
    print(f"G={G}: {tot / steps:.0f} cycles per env-step in block 0 / wave 0 (s_memtime ticks)")
    extra_names = ['box: corner slots', 'box: sphere slots', 'box: fold', 'fold: box lane corners', 'fold: pair law', 'fold: sync', 'fold: arm lanes', 'split: arm wave before S1', 'split: wait at S1', 'split: arm wave S1 -> S4', 'split: wait at S4', 'split: arm wave after S4', 'split: final barrier']
    if link and split:
        # k_abb_step_ws<512, true>: marks 17-22 are the BOX wave's own clock (its first lane), 24-28 the arm wave's
        extra_names[0:6] = ['box wave: finish of the sub-step before (+ prologue)', 'box wave: poses + corner / edge slots', 'box wave: parking inertia and ballots',
                            'box wave: wait at S0\'', 'box wave: link passes', 'box wave: idle from its link passes to S4']
        extra_names[12] = '(debug counter, not a time)'
    if pgs and abb and split:
        # k_abb_step_ws_hard: marks 24-29 are wave 0's clock (arm wave of envs 0-3, then the solve of envs 0-1)
        extra_names[0:6] = ['(unused)'] * 6
        extra_names[7:13] = ['arm wave: joints, drives, chain composition', "arm wave: wait at S0' (box wave: poses, corner candidates)",
                             'arm wave: inertias, point candidates, free ABA (mark 6 inside)', 'arm wave: wait at S1 (box wave: rod + link candidates)',
                             'the regrouped solve, integration, forces (marks 32-37 inside)', 'wait at S2 (the partner wave\'s solve)']
    if link and not split:
        # the link passes' own clock (csrc/shf_boxes.h: link_contacts): marks 24-28; mark 20 then spans all of them once more
        extra_names[7:12] = ['link: broad phase (body, box) pairs', 'link: pair set-up', 'link: sample points vs the box (family A)',
                             'link: capsule / rounded shapes', 'link: box volumes vs the box (corners, edges: family B / E)']
        extra_names[3] = 'all link passes (marks 24-28 once more) + sync'
        print(f"  link contacts: {buf[29] / max(buf[31], 1):.2f} live (body, box) pairs per wavefront and sub-step, "
              f"{buf[30] / max(buf[31], 1):.2f} in its first env; stage 0 (broad phase) = mark 24")
    if pgs and not abb:
        extra_names[0:2] = ['  H3 sweeps (position + velocity iterations)', '  H4 impulse passes (inward / root / outward, x2)']
        hist = [buf[38 + k] for k in range(9)]
        print("  wavefronts by constraint count Kw = 0..8 (all blocks, per sub-step): " + " ".join(f"{100.0 * h / max(sum(hist), 1):.1f}%" for h in hist))
    hard_names = []
    if pgs and abb:
        # the generic solve on the body-per-lane sub-step (csrc/shf_hard.h): its parts are marks 32-37
        hard_names = ['hard: body records + velocity rates', 'hard: gather candidates', 'hard: response matrix columns',
                      'hard: owner setup', 'hard: sweeps', 'hard: impulse passes']
    for k, nme in enumerate(NAMES + extra_names):
        print(f"  {k:2d} {nme:52s} {buf[k] / steps:9.0f}  {100.0 * buf[k] / tot:5.1f} %")
    for k, nme in enumerate(hard_names):
        print(f"  {32 + k:2d} {nme:52s} {buf[32 + k] / steps:9.0f}  {100.0 * buf[32 + k] / tot:5.1f} %")


if __name__ == "__main__":
    main()
