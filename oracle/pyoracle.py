"""ctypes loader for the CPU oracle (oracle/libshf_oracle.so).

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package (shifu_amd/) never imports it.
"""
import contextlib
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libshf_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("shf_oracle.c", "Makefile")] + \
           [os.path.join(_HERE, "..", "include", "shifu_amd.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.environ.get("SHF_ORACLE_LIB")      # e.g. the sanitizer build (make -C oracle asan)
        if not so:
            so = os.path.join(_HERE, "libshf_oracle.so")
            if not os.path.exists(so):
                build()
        _LIB = C.CDLL(so)
    return _LIB


_FLOP = None


def flop_lib():
    """The oracle compiled with a flop-counting real type (oracle/flopcount.cpp): entry points *_cnt plus
    shf_flopcount_read(reset).  Built on first use; single-threaded counting (the counter is thread-local)."""
    global _FLOP
    if _FLOP is None:
        so = os.path.join(_HERE, "libshf_flopcount.so")
        srcs = [os.path.join(_HERE, f) for f in ("shf_oracle.c", "flopcount.cpp")]
        if not os.path.exists(so) or any(os.path.getmtime(x) > os.path.getmtime(so) for x in srcs):
            subprocess.check_call(["g++", "-O1", "-fPIC", "-shared", "-fpermissive", "-w", "-std=c++17", "-fopenmp",
                                   os.path.join(_HERE, "flopcount.cpp"), "-o", so])
        _FLOP = C.CDLL(so)
        _FLOP.shf_flopcount_read.restype = C.c_ulonglong
    return _FLOP


def box_primitives(bR, bpos, h, pt, rad=0.0):
    """point_in_box and sphere_vs_box (float64 build) for one box and one point: (inside, phi, n), (phi_s, n_s, rc)."""
    a = np.concatenate([np.asarray(bR, np.float64).reshape(9), np.asarray(bpos, np.float64), np.asarray(h, np.float64),
                        np.asarray(pt, np.float64), [float(rad)]]).reshape(1, 19)
    out = np.zeros((1, 12))
    lib().shf_oracle_box_primitives_f64(C.c_int(1), _p(a, C.c_double), _p(out, C.c_double))
    o = out[0]
    return (bool(o[0]), float(o[1]), o[2:5].copy()), (float(o[5]), o[6:9].copy(), o[9:12].copy())


def box_box_edge(RA, cA, hA, RB, cB, hB, offset=0.01, f64=True):
    """box_box_edge (edge-edge contact of two oriented boxes by the separating-axis test): (hit, phi, n, r)."""
    dt, ct, suf = (np.float64, C.c_double, "f64") if f64 else (np.float32, C.c_float, "f32")
    a = np.concatenate([np.asarray(RA, dt).reshape(9), np.asarray(cA, dt), np.asarray(hA, dt), np.asarray(RB, dt).reshape(9),
                        np.asarray(cB, dt), np.asarray(hB, dt), [offset]]).astype(dt).reshape(1, 31)
    out = np.zeros((1, 8), dt)
    getattr(lib(), "shf_oracle_box_box_edge_" + suf)(C.c_int(1), _p(a, ct), _p(out, ct))
    o = out[0]
    return bool(o[0]), float(o[1]), o[2:5].copy(), o[5:8].copy()


def convex_manifold(Ra, pa, Rb, pb, ha=None, hb=None, hull_a=None, hull_b=None, offset=0.01, f64=True, seps=False):
    """convex_manifold of the oracle: polytope A (receives +f) = ShfHull `hull_a` on pose (Ra, pa) or the box of half extents
    `ha`; B likewise.  Returns (n (3,), [(r (3,), phi), ...]) with at most four contacts; n points from B towards A."""
    from shifu_amd import _abi
    dt, ct, suf = (np.float64, C.c_double, "f64") if f64 else (np.float32, C.c_float, "f32")
    z = np.zeros(3)
    a = np.concatenate([np.asarray(Ra, dt).reshape(9), np.asarray(pa, dt), np.asarray(z if ha is None else ha, dt),
                        np.asarray(Rb, dt).reshape(9), np.asarray(pb, dt), np.asarray(z if hb is None else hb, dt)]).astype(dt).reshape(1, 30)
    out = np.zeros((1, 24), dt)
    fn = getattr(lib(), "shf_oracle_convex_manifold_" + suf)
    fn.restype = None
    fn(C.c_int(1), None if hull_a is None else C.byref(hull_a), None if hull_b is None else C.byref(hull_b), _p(a, ct), ct(offset), _p(out, ct))
    o = out[0]
    if seps:
        return o[1:4].copy(), [(o[4 + 4 * q:7 + 4 * q].copy(), float(o[7 + 4 * q])) for q in range(int(o[0]))], o[20:23].copy()
    return o[1:4].copy(), [(o[4 + 4 * q:7 + 4 * q].copy(), float(o[7 + 4 * q])) for q in range(int(o[0]))]


@contextlib.contextmanager
def scene_extras(hulls=None, flags: int = 0):
    """The articulation's convex hulls (ShfHullSet, ShfModel.nhull > 0) and ShfScene.flags for the scene_step / abb_step calls
    inside the `with` (process-wide, like body_mass_scale)."""
    L = lib()
    for sfx in ("_f32", "_f64"):
        f, g = getattr(L, "shf_oracle_set_hulls" + sfx), getattr(L, "shf_oracle_set_scene_flags" + sfx)
        f.restype = g.restype = None
        f(None if hulls is None else C.byref(hulls))
        g(C.c_int(flags))
    try:
        yield
    finally:
        for sfx in ("_f32", "_f64"):
            getattr(L, "shf_oracle_set_hulls" + sfx)(None)
            getattr(L, "shf_oracle_set_scene_flags" + sfx)(C.c_int(0))


def point_in_box(bR, bpos, h, pt):
    return box_primitives(bR, bpos, h, pt)[0]


def dropped(reset: bool = True) -> int:
    """Contacts the float build dropped at the per-env limits (self / link contacts) since the last reset."""
    f = lib().shf_oracle_dropped
    f.restype = C.c_long
    return int(f(C.c_int(1 if reset else 0)))


def random_actions(seed: int, n: int, env_off: int, step: int, nd: int) -> np.ndarray:
    """run_policy('random') raw actions as shf_a1_step_random / shf_abb_step_random draw them in-kernel: (n, nd) float32,
    U(-1, 1), Philox4x32-10 keyed by `seed`, counter (global env id, vec-step index, dof)."""
    out = np.zeros((n, nd), np.float32)
    f = lib().shf_oracle_random_actions
    f.restype = None
    f(C.c_uint64(seed), C.c_int(n), C.c_int64(env_off), C.c_uint64(step), C.c_int(nd), out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def _p(a, ct):
    return None if a is None else a.ctypes.data_as(C.POINTER(ct))


def step(model, params, n, dof_state, root_state, *, nsteps=1, terrain=None, heights=None, effort=None,
         pos_target=None, vel_target=None, body_force=None, friction=None, want_contact=False,
         want_body_state=False, f64=False, body_force_pos=None):
    """Advance `n` envs by `nsteps` simulate() calls, in place.  Arrays are
    float32 (f64=False) or float64 (f64=True); friction is always float32."""
    dt = np.float64 if f64 else np.float32
    ct = C.c_double if f64 else C.c_float
    for a in (dof_state, root_state, effort, pos_target, vel_target, body_force, body_force_pos):
        assert a is None or (a.dtype == dt and a.flags.c_contiguous)
    contact = np.zeros((n * model.nb, 3), dt) if want_contact else None
    bstate = np.zeros((n * model.nb, 13), dt) if want_body_state else None
    # body_force_pos: world-space points of application (gym.apply_rigid_body_force_at_pos_tensors(force, pos)); None = CoM
    fn = lib().shf_oracle_step_at_pos_f64 if f64 else lib().shf_oracle_step_at_pos_f32
    fn.restype = None
    fn(C.byref(model), C.byref(params), C.byref(terrain) if terrain is not None else None,
       _p(heights, C.c_int16), C.c_int(n), C.c_int(nsteps), _p(dof_state, ct), _p(root_state, ct), _p(effort, ct),
       _p(pos_target, ct), _p(vel_target, ct), _p(body_force, ct), _p(body_force_pos, ct), _p(friction, C.c_float),
       _p(contact, ct), _p(bstate, ct))
    return contact, bstate


@contextlib.contextmanager
def body_mass_scale(scale):
    """Per-env factors on the bodies' mass and inertia (SHF_T_BODY_MASS_SCALE; gym.set_actor_rigid_body_properties with
    recomputeInertia=True, shifu/units/units.py:104-110) for the step / a1_step / abb_step calls inside the `with`:
    scale (n, nb) float32, row = the call's env index."""
    a = np.ascontiguousarray(scale, np.float32)
    fns = [getattr(lib(), "shf_oracle_set_body_mass_scale" + sfx) for sfx in ("_f32", "_f64")]
    for f in fns:
        f.restype = None
        f(_p(a, C.c_float))
    try:
        yield a
    finally:
        for f in fns:
            f(None)


def accel(model, params, dof_state, root_state, effort, f64=True):
    dt = np.float64 if f64 else np.float32
    ct = C.c_double if f64 else C.c_float
    qdd = np.zeros(model.nd, dt)
    racc = np.zeros(6, dt)
    fn = lib().shf_oracle_accel_f64 if f64 else lib().shf_oracle_accel_f32
    fn.restype = None
    fn(C.byref(model), C.byref(params), _p(np.ascontiguousarray(dof_state, dt), ct),
       _p(np.ascontiguousarray(root_state, dt), ct), _p(np.ascontiguousarray(effort, dt), ct), _p(qdd, ct),
       _p(racc, ct))
    return qdd, racc


def sincos(x):
    x = np.ascontiguousarray(x, np.float32)
    s = np.zeros_like(x); c = np.zeros_like(x)
    lib().shf_oracle_sincos_f32(C.c_int(x.size), _p(x, C.c_float), _p(s, C.c_float), _p(c, C.c_float))
    return s, c


def rcp_rsqrt(x):
    """rcp_spec / rsqrt_spec of the float build (the Newton sequences the HIP kernels share)."""
    x = np.ascontiguousarray(x, np.float32)
    r = np.zeros_like(x); q = np.zeros_like(x)
    lib().shf_oracle_rcp_rsqrt_f32(C.c_int(x.size), _p(x, C.c_float), _p(r, C.c_float), _p(q, C.c_float))
    return r, q


def exp(x):
    x = np.ascontiguousarray(x, np.float32)
    y = np.zeros_like(x)
    lib().shf_oracle_exp_f32(C.c_int(x.size), _p(x, C.c_float), _p(y, C.c_float))
    return y


def terrain_query(terrain, heights, xy, f64=False):
    dt = np.float64 if f64 else np.float32
    ct = C.c_double if f64 else C.c_float
    xy = np.ascontiguousarray(xy, dt)
    n = xy.shape[0]
    h = np.zeros(n, dt); nrm = np.zeros((n, 3), dt)
    fn = lib().shf_oracle_terrain_query_f64 if f64 else lib().shf_oracle_terrain_query_f32
    fn(C.byref(terrain), _p(heights, C.c_int16), C.c_int(n), _p(xy, ct), _p(h, ct), _p(nrm, ct))
    return h, nrm


class A1Buffers(C.Structure):
    """Mirror of ShfOracleA1Buffers (oracle/shf_oracle.c)."""
    _F = C.POINTER(C.c_float)
    _fields_ = [("dof_state", _F), ("root_state", _F), ("body_state", _F), ("contact", _F), ("friction", _F),
                ("actions", _F), ("obs", _F), ("rew", _F),
                ("reset", C.POINTER(C.c_uint8)), ("timeout", C.POINTER(C.c_uint8)), ("ep_len", C.POINTER(C.c_int64)),
                ("command", _F), ("history", _F), ("rew_sums", _F), ("torques", _F), ("base_vel", _F), ("heights", _F),
                ("hpoints", _F), ("push", _F), ("origins", _F),
                ("levels", C.POINTER(C.c_int64)), ("types", C.POINTER(C.c_int64)), ("torigins", _F),
                ("reset_count", C.POINTER(C.c_int32)), ("done_sums", _F)]


A1_FIELDS = [f[0] for f in A1Buffers._fields_]


def a1_step(model, params, task_params, n, env_id_offset, bufs: dict, raw_actions, terrain=None, heights=None,
            nthreads=1, count_flops=False):
    """One fused A1Conditional env step on NumPy buffers (dict keyed by A1_FIELDS), in place."""
    B = A1Buffers()
    for name, ctype in A1Buffers._fields_:
        a = bufs[name]
        assert a.flags.c_contiguous, name
        setattr(B, name, a.ctypes.data_as(ctype))
    raw = np.ascontiguousarray(raw_actions, np.float32)
    fn = flop_lib().shf_oracle_a1_step_cnt if count_flops else lib().shf_oracle_a1_step_f32
    fn.restype = None
    fn(C.byref(model), C.byref(params), C.byref(terrain) if terrain is not None else None, _p(heights, C.c_int16),
       C.byref(task_params), C.c_int(n), C.c_int64(env_id_offset), C.byref(B), _p(raw, C.c_float), C.c_int(nthreads))


def a1_stats(task_params, n, done_sums):
    out = np.zeros(16, np.float32)
    fn = lib().shf_oracle_a1_stats_f32
    fn.restype = None
    fn(C.byref(task_params), C.c_int(n), _p(done_sums, C.c_float), _p(out, C.c_float))
    return out


# ---- glue pieces, one entry point each (pinned against tests/golden) -------------------
def _f(a):
    return np.ascontiguousarray(a, np.float32)


def glue_heights(terrain, heights, hpoints, root_state):
    root_state, hpoints = _f(root_state), _f(hpoints)
    n, P = root_state.shape[0], hpoints.shape[0]
    out = np.zeros((n, P), np.float32)
    lib().shf_oracle_glue_heights_f32(C.byref(terrain), _p(heights, C.c_int16), C.c_int(n), C.c_int(P),
                                      _p(hpoints, C.c_float), _p(root_state, C.c_float), _p(out, C.c_float))
    return out


def glue_rewards(cmd, blv, bav, hist, contact, tau, leg_bodies):
    cmd, blv, bav, hist, contact, tau = map(_f, (cmd, blv, bav, hist, contact, tau))
    n, nd, H = hist.shape
    nb = contact.shape[1]
    legs = np.ascontiguousarray(leg_bodies, np.int32)
    out = np.zeros((6, n), np.float32)
    lib().shf_oracle_glue_rewards_f32(C.c_int(n), C.c_int(nd), C.c_int(H), C.c_int(nb), C.c_int(len(legs)),
                                      _p(legs, C.c_int32), _p(cmd, C.c_float), _p(blv, C.c_float), _p(bav, C.c_float),
                                      _p(hist, C.c_float), _p(contact, C.c_float), _p(tau, C.c_float), _p(out, C.c_float))
    return out


def glue_termination(contact, ep_len, base_body, max_len):
    contact = _f(contact)
    ep = np.ascontiguousarray(ep_len, np.int64)
    n, nb = contact.shape[0], contact.shape[1]
    ct, to = np.zeros(n, np.uint8), np.zeros(n, np.uint8)
    lib().shf_oracle_glue_termination_f32(C.c_int(n), C.c_int(nb), C.c_int(base_body), C.c_float(max_len),
                                          _p(contact, C.c_float), _p(ep, C.c_int64), _p(ct, C.c_uint8), _p(to, C.c_uint8))
    return ct.astype(bool), to.astype(bool)


def glue_obs(q0, cmd, blv, bav, dof_state, hist, base_z, mh, clip_obs=100.0):
    q0, cmd, blv, bav, dof_state, hist, base_z, mh = map(_f, (q0, cmd, blv, bav, dof_state, hist, base_z, mh))
    n, nd, H = hist.shape
    P = mh.shape[1]
    out = np.zeros((n, 12 + 2 * nd + nd * H + P), np.float32)
    lib().shf_oracle_glue_obs_f32(C.c_int(n), C.c_int(nd), C.c_int(H), C.c_int(P), C.c_float(clip_obs),
                                  _p(q0, C.c_float), _p(cmd, C.c_float), _p(blv, C.c_float), _p(bav, C.c_float),
                                  _p(dof_state, C.c_float), _p(hist, C.c_float), _p(base_z, C.c_float), _p(mh, C.c_float),
                                  _p(out, C.c_float))
    return out


def glue_curriculum(task_params, root_state, origins, cmd, levels, rnd=None):
    root_state, origins, cmd = _f(root_state), _f(origins), _f(cmd)
    n = root_state.shape[0]
    lv = np.ascontiguousarray(levels, np.int64).copy()
    rnd = np.zeros(n, np.uint32) if rnd is None else np.ascontiguousarray(rnd, np.uint32)
    lib().shf_oracle_glue_curriculum_f32(C.byref(task_params), C.c_int(n), _p(root_state, C.c_float),
                                         _p(origins, C.c_float), _p(cmd, C.c_float), _p(rnd, C.c_uint32), _p(lv, C.c_int64))
    return lv


def quat_rotate_inverse(q, v):
    q, v = _f(q), _f(v)
    out = np.zeros_like(v)
    lib().shf_oracle_quat_rotate_inverse_f32(C.c_int(q.shape[0]), _p(q, C.c_float), _p(v, C.c_float), _p(out, C.c_float))
    return out


def scene_step(model, params, boxes, n, dof_state, root_state, *, nsteps=1, terrain=None, heights=None, effort=None,
               pos_target=None, vel_target=None, friction=None, want_contact=True, want_body_state=True,
               want_jacobian=False, f64=False):
    """simulate() for an articulation + box actors (boxes: list of ShfBoxDesc)."""
    from shifu_amd import _abi
    dt = np.float64 if f64 else np.float32
    ct = C.c_double if f64 else C.c_float
    nbx = len(boxes)
    arr = (_abi.ShfBoxDesc * max(nbx, 1))(*boxes)
    B = model.nb + nbx
    contact = np.zeros((n * B, 3), dt) if want_contact else None
    bstate = np.zeros((n * B, 13), dt) if want_body_state else None
    jac = np.zeros((n, model.nb - 1, 6, model.nd), dt) if want_jacobian else None
    fn = lib().shf_oracle_scene_step_f64 if f64 else lib().shf_oracle_scene_step_f32
    fn.restype = None
    fn(C.byref(model), C.byref(params), C.byref(terrain) if terrain is not None else None, _p(heights, C.c_int16),
       C.c_int(nbx), arr, C.c_int(n), C.c_int(nsteps), _p(dof_state, ct), _p(root_state, ct), _p(effort, ct),
       _p(pos_target, ct), _p(vel_target, ct), _p(friction, C.c_float), _p(contact, ct), _p(bstate, ct), _p(jac, ct))
    return contact, bstate, jac



class AbbBuffers(C.Structure):
    """Mirror of ShfOracleAbbBuffers (oracle/shf_oracle.c)."""
    _F = C.POINTER(C.c_float)
    _fields_ = [("dof_state", _F), ("root_state", _F), ("body_state", _F), ("contact", _F), ("jacobian", _F),
                ("friction", _F), ("actions", _F), ("obs", _F), ("rew", _F),
                ("reset", C.POINTER(C.c_uint8)), ("timeout", C.POINTER(C.c_uint8)), ("success", C.POINTER(C.c_uint8)),
                ("ep_len", C.POINTER(C.c_int64)), ("rew_sums", _F), ("dof_targets", _F),
                ("reset_count", C.POINTER(C.c_int32)), ("done_sums", _F)]


ABB_FIELDS = [f[0] for f in AbbBuffers._fields_]


def abb_step(model, params, boxes, task_params, n, env_id_offset, bufs: dict, raw_actions, nthreads=1, count_flops=False):
    """One fused AbbPushBox env step on NumPy buffers (dict keyed by ABB_FIELDS), in place."""
    from shifu_amd import _abi
    B = AbbBuffers()
    for name, ctype in AbbBuffers._fields_:
        a = bufs[name]
        assert a.flags.c_contiguous, name
        setattr(B, name, a.ctypes.data_as(ctype))
    arr = (_abi.ShfBoxDesc * max(len(boxes), 1))(*boxes)
    raw = np.ascontiguousarray(raw_actions, np.float32)
    fn = flop_lib().shf_oracle_abb_step_cnt if count_flops else lib().shf_oracle_abb_step_f32
    fn.restype = None
    fn(C.byref(model), C.byref(params), None, None, C.c_int(len(boxes)), arr, C.byref(task_params), C.c_int(n),
       C.c_int64(env_id_offset), C.byref(B), _p(raw, C.c_float), C.c_int(nthreads))


def abb_stats(task_params, n, done_sums):
    out = np.zeros(8, np.float32)
    fn = lib().shf_oracle_abb_stats_f32
    fn.restype = None
    fn(C.byref(task_params), C.c_int(n), _p(done_sums, C.c_float), _p(out, C.c_float))
    return out


# ---- golden G4 / G9 / G10 pieces (the static functions the fused oracle steps call) ----
def glue_history(inputs, reset_after, reset_ids, H=3):
    """HistoryRecorder.add per step (+ reset_idx after step `reset_after`): returns (bufs (K,n,nd,H), flats (K,n,nd*H))."""
    x = _f(inputs)
    K, n, nd = x.shape
    ids = np.ascontiguousarray(reset_ids, np.int64)
    bufs, flats = np.zeros((K, n, nd, H), np.float32), np.zeros((K, n, nd * H), np.float32)
    lib().shf_oracle_glue_history_f32(C.c_int(n), C.c_int(nd), C.c_int(H), C.c_int(K), _p(x, C.c_float),
                                      C.c_int(int(reset_after)), C.c_int(len(ids)), _p(ids, C.c_int64),
                                      _p(bufs, C.c_float), _p(flats, C.c_float))
    return bufs, flats


def glue_quat_mul(a, b):
    a, b = _f(a), _f(b)
    out = np.zeros_like(a)
    lib().shf_oracle_glue_quat_mul_f32(C.c_int(a.shape[0]), _p(a, C.c_float), _p(b, C.c_float), _p(out, C.c_float))
    return out


def glue_ik(j_ee, dof_pos, ee_pos, ee_quat, tar_pos, tar_quat, damping=0.05, f64=False):
    """ArmRobot.inverse_kinematics: inputs float32 (as the tensors are), arithmetic in the build's real type."""
    j, q, ep, eq, tp, tq = map(_f, (j_ee, dof_pos, ee_pos, ee_quat, tar_pos, tar_quat))
    n, _, nd = j.shape
    out = np.zeros((n, nd), np.float64 if f64 else np.float32)
    fn = lib().shf_oracle_glue_ik_f64 if f64 else lib().shf_oracle_glue_ik_f32
    fn(C.c_int(n), C.c_int(nd), _p(j, C.c_float), _p(q, C.c_float), _p(ep, C.c_float), _p(eq, C.c_float),
       _p(tp, C.c_float), _p(tq, C.c_float), C.c_float(damping), _p(out, C.c_double if f64 else C.c_float))
    return out


def glue_abb_post(task_params, cube, goal, ee, ep_len):
    """AbbPushBox obs / termination / rewards for given cube, goal, ee poses (n,7) and episode lengths."""
    cube, goal, ee = _f(cube), _f(goal), _f(ee)
    n = cube.shape[0]
    ep = np.ascontiguousarray(ep_len, np.int64)
    obs = np.zeros((n, 6), np.float32)
    to, su, rs = np.zeros(n, np.uint8), np.zeros(n, np.uint8), np.zeros(n, np.uint8)
    r0, r1 = np.zeros(n, np.float32), np.zeros(n, np.float32)
    lib().shf_oracle_glue_abb_post_f32(C.byref(task_params), C.c_int(n), _p(cube, C.c_float), _p(goal, C.c_float),
                                       _p(ee, C.c_float), _p(ep, C.c_int64), _p(obs, C.c_float), _p(to, C.c_uint8),
                                       _p(su, C.c_uint8), _p(rs, C.c_uint8), _p(r0, C.c_float), _p(r1, C.c_float))
    return obs, to.astype(bool), su.astype(bool), rs.astype(bool), r0, r1


def segment_closest(segs, f64=True):
    """Closest points of segment pairs: segs (n,4,3) = p1, q1, p2, q2 -> (n,2,3) c1, c2 (the self-collision geometry)."""
    dt, ct = (np.float64, C.c_double) if f64 else (np.float32, C.c_float)
    s = np.ascontiguousarray(segs, dt)
    out = np.zeros((s.shape[0], 2, 3), dt)
    fn = lib().shf_oracle_segment_closest_f64 if f64 else lib().shf_oracle_segment_closest_f32
    fn(C.c_int(s.shape[0]), _p(s, ct), _p(out, ct))
    return out


def segment_box_contact(bR, bpos, h, c0, s, part, f64=True):
    """Capsule vs box contact point `part` (0 / 1, ShfModel.sph_part): (exists (n,) bool, t (n,)).  A line contact (the
    segment runs along a face) has two: the ends of the stretch; otherwise part 0 is the closest point."""
    dt, ct = (np.float64, C.c_double) if f64 else (np.float32, C.c_float)
    n = np.reshape(bR, (-1, 9)).shape[0]
    pack = np.ascontiguousarray(np.concatenate([np.reshape(bR, (-1, 9)), bpos, h, c0, s, np.full((n, 1), float(part))], axis=1), dt)
    out = np.zeros((n, 3), dt)
    fn = lib().shf_oracle_segment_box_contact_f64 if f64 else lib().shf_oracle_segment_box_contact_f32
    fn.restype = None
    fn(C.c_int(n), _p(pack, ct), _p(out, ct))
    return out[:, 0] > 0, out[:, 1]


def segment_box_param(bR, bpos, h, c0, s, f64=True):
    """Capsule vs box: parameter t of the point c0 + t s closest to the box (bR (n,9) row-major world<-box, bpos, half
    extents h, all (n,3)) -> (n,)."""
    dt, ct = (np.float64, C.c_double) if f64 else (np.float32, C.c_float)
    pack = np.ascontiguousarray(np.concatenate([np.reshape(bR, (-1, 9)), bpos, h, c0, s], axis=1), dt)
    out = np.zeros(pack.shape[0], dt)
    fn = lib().shf_oracle_segment_box_param_f64 if f64 else lib().shf_oracle_segment_box_param_f32
    fn.restype = None
    fn(C.c_int(pack.shape[0]), _p(pack, ct), _p(out, ct))
    return out
