"""Host-side owner of one simulation: allocates every device buffer with torch,
binds it to the C-ABI library and exposes the launches.  This is the layer the
`gym` facade (shifu_amd/isaacgym/gymapi.py) and the fused A1 env sit on.

PyTorch is plumbing here (device memory, current stream); all arithmetic happens
in the HIP kernels behind include/shifu_amd.h.
"""
from __future__ import annotations

import ctypes as C
import functools
from typing import Dict, Optional

import numpy as np
import torch

from . import _abi
from ._lib import check, lib

_TORCH_DTYPE = {_abi.DTYPE_F32: torch.float32, _abi.DTYPE_I32: torch.int32, _abi.DTYPE_I16: torch.int16,
                _abi.DTYPE_U8: torch.uint8, _abi.DTYPE_I64: torch.int64}


def default_sim_params(dt: float = 0.005, gravity=(0.0, 0.0, -9.81), **kw) -> _abi.ShfSimParams:
    p = _abi.ShfSimParams()
    p.dt = dt
    p.gravity[:] = gravity
    p.contact_k = kw.get("contact_k", 5e4)
    p.contact_d = kw.get("contact_d", 300.0)
    p.friction_vel = kw.get("friction_vel", 0.002)
    p.limit_k = kw.get("limit_k", 2000.0)
    p.limit_d = kw.get("limit_d", 20.0)
    p.angular_damping = kw.get("angular_damping", 0.5)  # AssetOptions default [EXT]
    p.max_ang_vel = kw.get("max_ang_vel", 64.0)          # AssetOptions default [EXT]
    p.max_depen_vel = kw.get("max_depen_vel", 1.0)       # env_config.py:57
    p.contact_offset = kw.get("contact_offset", 0.01)    # env_config.py:54
    set_solver(p, kw.get("solver", "compliant"), **{k: v for k, v in kw.items() if k in _SOLVER_KEYS})
    return p


_SOLVER_KEYS = ("pos_iters", "vel_iters", "max_contacts", "rest_offset", "bounce_threshold", "restitution", "erp")


def set_solver(p: _abi.ShfSimParams, solver: str = "pgs", **kw) -> _abi.ShfSimParams:
    """Contact solver of a ShfSimParams.  "pgs": the velocity-level projected Gauss-Seidel solve with the PhysX settings
    the reference configures (shifu/configs/env_config.py:50-58: num_position_iterations 8, num_velocity_iterations 1,
    rest_offset 0, bounce_threshold_velocity 0.5; max_depenetration_velocity and contact_offset are fields of their own);
    "compliant": rounds 1-4's linearly-implicit spring-damper law (contact_k, contact_d, friction_vel)."""
    if solver in ("pgs", "tgs"):
        # "tgs": physx.solver_type = 1, the reference's value (env_config.py:50): the same sweeps as sub-iterations of dt / pos_iters
        # whose constraint errors follow the advanced motion (include/shifu_amd.h: SHF_SOLVER_TGS); "pgs": solver_type = 0
        p.solver = _abi.SOLVER_TGS if solver == "tgs" else _abi.SOLVER_PGS
        p.pos_iters, p.vel_iters = kw.get("pos_iters", 8), kw.get("vel_iters", 1)
        p.max_contacts = kw.get("max_contacts", 8)
        # (fail here, not at the first step: the chain-mapped A1 kernels hold up to MAX_HARD_CONTACTS = 16 constraints per env, the
        # run-time-shaped kernels of other scenes 8 -- those refuse more at launch, shf_sim_step / shf_abb_step)
        if not (1 <= p.pos_iters) or p.vel_iters < 0 or not (0 <= p.max_contacts <= _abi.MAX_HARD_CONTACTS):
            raise ValueError(f"set_solver: pos_iters >= 1, vel_iters >= 0 and 0 <= max_contacts <= {_abi.MAX_HARD_CONTACTS} (0 reads as 8)")
        p.rest_offset = kw.get("rest_offset", 0.0)
        p.bounce_threshold = kw.get("bounce_threshold", 0.5)
        p.restitution = kw.get("restitution", 0.0)
        p.erp = kw.get("erp", 0.2)
    elif solver == "compliant":
        p.solver = _abi.SOLVER_COMPLIANT
    else:
        raise ValueError("solver must be 'pgs', 'tgs' or 'compliant'")
    return p


def _on_device(method):
    """Run a launching method with the sim's GPU as the current HIP device: the library launches on, and records LDS
    opt-ins for, whatever device is current, and the stream handed in belongs to this one."""
    @functools.wraps(method)
    def wrapped(self, *a, **k):
        dev = self.device if hasattr(self, "device") else self.sim.device
        if torch.cuda.current_device() == dev.index:       # the common case: no device switch, no context object
            return method(self, *a, **k)
        with torch.cuda.device(dev):
            return method(self, *a, **k)
    return wrapped


def _stream_ptr(device) -> C.c_void_p:
    idx = torch.device(device).index
    return C.c_void_p(torch._C._cuda_getCurrentRawStream(torch.cuda.current_device() if idx is None else idx))


def _struct_to_device(s, device) -> torch.Tensor:
    return torch.frombuffer(bytearray(bytes(s)), dtype=torch.uint8).to(device)


def heightfield_nz_min(samples: np.ndarray, hscale: float, vscale: float) -> float:
    """ShfTerrain.nz_min: a lower bound on n_z of every normal a height-field query can return.  The query's gradients
    are differences of neighbouring samples along x and along y (either triangle of a cell), so the steepest normal is
    bounded by the largest such differences; 1 % is taken off for the float32 arithmetic of the kernel."""
    s = samples.astype(np.int64)
    gx = float(np.abs(np.diff(s, axis=0)).max()) * vscale / hscale if s.shape[0] > 1 else 0.0
    gy = float(np.abs(np.diff(s, axis=1)).max()) * vscale / hscale if s.shape[1] > 1 else 0.0
    return float(0.99 / np.sqrt(1.0 + gx * gx + gy * gy))


class Sim:
    """One `gym.create_sim` worth of state on one GPU."""

    def __init__(self, params: _abi.ShfSimParams, device="cuda:0"):
        self.device = torch.device(device)
        if self.device.type == "cuda" and self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        if self.device.type != "cuda":
            raise RuntimeError("shifu_amd.backend.Sim runs on an MI355X only (device must be cuda:N); "
                               "there is no CPU fallback")
        self.params = params
        self._h = C.c_void_p()
        check(lib().shf_sim_create(C.byref(params), C.byref(self._h)))
        self.tensors: Dict[int, torch.Tensor] = {}
        self.model: Optional[_abi.ShfModel] = None
        self.terrain = _abi.ShfTerrain()
        self.terrain.friction = 1.0
        self.num_envs = 0
        self.nboxes = 0
        self.group = 64
        self.boxes = []
        self._heights: Optional[torch.Tensor] = None

    # -- construction -------------------------------------------------------
    def set_plane(self, friction: float = 1.0):
        t = _abi.ShfTerrain()
        t.rows = t.cols = 0
        t.hscale, t.vscale, t.border, t.friction = 1.0, 1.0, 0.0, friction
        t.nz_min = 0.99          # the plane's normal is (0, 0, 1)
        self.terrain = t
        check(lib().shf_sim_set_terrain(self._h, C.byref(t)))

    def set_heightfield(self, samples: np.ndarray, hscale: float, vscale: float, border: float, friction: float,
                        warp: Optional[np.ndarray] = None):
        """Height-field terrain; with `warp` (isaacgym.terrain_utils.trimesh_warp_map) the trimesh made of the same
        samples (ShfTerrain.warped: vertical risers at steep steps)."""
        assert samples.dtype == np.int16 and samples.ndim == 2
        t = _abi.ShfTerrain()
        t.rows, t.cols = samples.shape
        t.hscale, t.vscale, t.border, t.friction = hscale, vscale, border, friction
        t.warped = 0 if warp is None else 1
        t.nz_min = 0.0 if warp is not None else heightfield_nz_min(samples, hscale, vscale)
        self.terrain = t
        self.height_samples = np.ascontiguousarray(samples)
        if warp is None:
            payload = self.height_samples
        else:
            from .isaacgym.terrain_utils import pack_trimesh_samples
            assert warp.shape == samples.shape and warp.dtype == np.uint8
            payload = pack_trimesh_samples(samples, warp)
        self.terrain_payload = payload                      # what the oracle takes as `heights`
        self._heights = torch.from_numpy(payload).to(self.device)
        check(lib().shf_sim_set_terrain(self._h, C.byref(t)))

    def set_articulation(self, model: _abi.ShfModel):
        check(lib().shf_model_bounds(C.byref(model)))      # derived broad-phase data: the blob uploaded as SHF_T_MODEL carries it
        self.model = model
        check(lib().shf_sim_set_articulation(self._h, C.byref(model)))

    def set_hulls(self, hulls: "_abi.ShfHullSet"):
        """The articulation's convex hulls (mesh colliders; CompiledModel.hulls) -- after set_articulation, before finalize."""
        check(lib().shf_sim_set_hulls(self._h, C.byref(hulls)))
        self.hulls = hulls

    def set_scene_flags(self, flags: int):
        """ShfScene.flags (SCENE_FACE_MANIFOLD: the clipped face manifold for box pairs that touch without a vertex or an
        edge crossing) -- before finalize."""
        check(lib().shf_sim_set_scene_flags(self._h, int(flags)))
        self.scene_flags = int(flags)

    def add_box(self, box: _abi.ShfBoxDesc):
        check(lib().shf_sim_add_box(self._h, C.byref(box)))
        self.boxes.append(box)
        self.nboxes += 1

    def layout(self, tid: int):
        shape = (C.c_int64 * 4)()
        nd, dt = C.c_int32(), C.c_int32()
        check(lib().shf_sim_layout(self._h, tid, shape, C.byref(nd), C.byref(dt)))
        return tuple(shape[:nd.value]), _TORCH_DTYPE[dt.value]

    def bind(self, tid: int, t: torch.Tensor):
        assert t.is_contiguous() and t.device == self.device
        self.tensors[tid] = t
        check(lib().shf_sim_bind(self._h, tid, C.c_void_p(t.data_ptr())))

    def finalize(self, num_envs: int, env_id_offset: int = 0, group: int = 64, mapping: str = "body"):
        """group: lanes per env; mapping: 'body' (lane = rigid body, any articulation) or 'chain' (lane = kinematic chain:
        the A1's tree, csrc/shf_chain.h, or the ABB's serial arm, csrc/shf_arm.h; 16 or 32 lanes) -- kernel selection,
        identical results.  For the A1's chain mapping `group` is the width of the fused step only: gym.simulate, refresh_*
        and the reset kernels keep the library's body-mapped width (64 lanes), whatever self.group says."""
        self.num_envs = num_envs
        self.group = group
        self.mapping = mapping
        check(lib().shf_sim_finalize(self._h, num_envs, env_id_offset))
        if mapping in ("chain", "split"):
            # 'split': the chain mapping with arm and boxes on different waves of a workgroup (the ABB step at 16 lanes)
            check(lib().shf_sim_set_mapping(self._h, _abi.MAP_CHAIN if mapping == "chain" else _abi.MAP_CHAIN_SPLIT))
            check(lib().shf_sim_set_group(self._h, group))
        elif mapping != "body":
            raise ValueError("mapping must be 'body', 'chain' or 'split'")
        elif group != 64:
            check(lib().shf_sim_set_group(self._h, group))
        for tid in range(_abi.T_COUNT):
            if tid in (_abi.T_BODY_MASS_SCALE, _abi.T_CONTACT_HIST):        # optional: bound by set_body_mass_scale() / bind_contact_hist()
                continue
            if tid == _abi.T_HULLS:                  # only for an articulation with convex hulls (set_hulls)
                if getattr(self, "hulls", None) is not None and self.model.nhull > 0:
                    self.bind(tid, _struct_to_device(self.hulls, self.device))
                continue
            if tid == _abi.T_HEIGHTS:
                if self._heights is not None:
                    self.bind(tid, self._heights)
                continue
            if tid == _abi.T_MODEL:
                self.bind(tid, _struct_to_device(self.model, self.device))
                continue
            if tid == _abi.T_SCENE:
                sc = _abi.ShfScene()
                sc.nboxes = self.nboxes
                sc.flags = getattr(self, "scene_flags", 0)
                for k, b in enumerate(self.boxes):
                    sc.box[k] = b
                self.bind(tid, _struct_to_device(sc, self.device))
                continue
            shape, dt = self.layout(tid)
            t = torch.zeros(shape, dtype=dt, device=self.device)
            if tid == _abi.T_FRICTION:
                t.fill_(1.0)
            self.bind(tid, t)

    def use_split_step(self) -> bool:
        """gym.simulate on the kernel compiled for this sim's arm + scene, when there is one (shf_sim_step_split_supported: the
        shipped 6-link arm with table / cube / pad under a velocity-level solver -> k_sim_step_ws_hard; identical results).  The gym
        facade asks after prepare_sim; False: shf_sim_step stays on the run-time-shaped kernels."""
        if self.params.solver == _abi.SOLVER_COMPLIANT or not lib().shf_sim_step_split_supported(self._h):
            return False
        check(lib().shf_sim_set_mapping(self._h, _abi.MAP_CHAIN_SPLIT))
        self.mapping = "split"
        return True

    def bind_contact_hist(self, on: bool = True):
        """SHF_T_CONTACT_HIST (velocity-level solve): per env, how many sub-steps offered k candidate constraints before the
        max_contacts cap -- (num_envs, CONTACT_HIST_BINS + 1) int32 (last column: the env's drop counter, which
        T_DROPPED does not receive meanwhile), accumulated by the step kernels while bound.  Returns the tensor
        (None after unbinding).  Bind before capturing a step into a hipGraph."""
        if not on:
            check(lib().shf_sim_bind(self._h, _abi.T_CONTACT_HIST, None))
            self.tensors.pop(_abi.T_CONTACT_HIST, None)
            return None
        t = torch.zeros(self.num_envs, _abi.CONTACT_HIST_BINS + 1, dtype=torch.int32, device=self.device)   # last column: the drop counters
        self.bind(_abi.T_CONTACT_HIST, t)
        return t

    def set_body_mass_scale(self, scale) -> torch.Tensor:
        """Per-env factors on the mass and inertia tensor of each body of the articulation (centre of mass unchanged):
        gym.set_actor_rigid_body_properties(env, actor, props, recomputeInertia=True) with props[b].mass = factor x the
        asset's (shifu/units/units.py:104-110).  scale: (num_envs, nb), > 0; returns the bound tensor -- later edits in place
        are seen by the next step.  None unbinds (factors 1).  Bind before capturing a step into a hipGraph: the kernels take the
        tensor's address as a launch argument, and a graph captured earlier keeps replaying with the address it saw (none)."""
        if scale is None:
            check(lib().shf_sim_bind(self._h, _abi.T_BODY_MASS_SCALE, None))
            self.tensors.pop(_abi.T_BODY_MASS_SCALE, None)
            return None
        t = torch.as_tensor(scale, dtype=torch.float32).to(self.device).reshape(self.num_envs, self.model.nb).contiguous().clone()
        if not bool((t > 0).all()):
            raise ValueError("set_body_mass_scale: factors must be positive")
        self.bind(_abi.T_BODY_MASS_SCALE, t)
        return t

    # -- launches -----------------------------------------------------------
    # step / refresh / set_dof_command are what a user's control loop calls decimation x 3 times per vec-step
    # (a1_conditional.py:64-75): they skip the generic wrapper when the sim's GPU is already current (bound library
    # functions, the raw stream handle as an int) -- ~3 us less host time per call, same calls into the library.
    def _fast(self):
        f = getattr(self, "_fast_fns", None)
        if f is None:
            L = lib()
            f = self._fast_fns = (L.shf_sim_step, L.shf_sim_refresh, L.shf_sim_set_dof_command, self.device.index,
                                  torch._C._cuda_getCurrentRawStream)
        return f

    # -- the commits of one reset as one launch (shf_sim_commit_reset) --------------------------------------------------------
    # IsaacGymEnv.reset_idx brackets its actors' resets with begin_reset() .. commit_root_indexed(): inside, the dof-state and
    # position-target commits of Robot._reset_dof_state are held back and go out together with the root rows.  Outside such a
    # bracket every commit acts at once, as gym.set_*_tensor_indexed does; any other launch flushes what is held first.
    _defer = False
    _held = None           # (dof tensor or None, position-target tensor or None, int32 actor ids)

    def begin_reset(self):
        self._flush_held()
        self._defer = True

    @_on_device
    def _flush_held(self):
        held, self._held, self._defer = self._held, None, False
        if held is not None:
            dof, tgt, idx = held
            check(lib().shf_sim_commit_reset(self._h, None, None, 0, C.c_void_p(dof.data_ptr()) if dof is not None else None,
                                             C.c_void_p(tgt.data_ptr()) if tgt is not None else None, C.c_void_p(idx.data_ptr()),
                                             idx.numel(), _stream_ptr(self.device)))

    def _hold(self, dof, tgt, idx) -> bool:
        """Inside a reset bracket: keep this commit for the root commit's launch.  False: act now."""
        if not self._defer:
            return False
        if self._held is not None:
            d0, t0, i0 = self._held
            if i0.data_ptr() != idx.data_ptr() or (dof is not None and d0 is not None) or (tgt is not None and t0 is not None):
                self._flush_held()            # a second articulation's commits: the first set goes out now
                self._defer = True
                d0 = t0 = None
            dof, tgt = (dof if dof is not None else d0), (tgt if tgt is not None else t0)
        self._held = (dof, tgt, idx)
        return True

    def step(self):
        if self._held is not None or self._defer:
            self._flush_held()
        f_step, _, _, idx, raw_stream = self._fast()
        if torch.cuda.current_device() == idx:
            if f_step(self._h, raw_stream(idx)):
                check(1)
            return
        with torch.cuda.device(self.device):
            check(f_step(self._h, _stream_ptr(self.device)))

    def refresh(self, mask: int = _abi.REFRESH_ALL):
        if self._held is not None or self._defer:
            self._flush_held()
        _, f_refresh, _, idx, raw_stream = self._fast()
        if torch.cuda.current_device() == idx:
            if f_refresh(self._h, mask, raw_stream(idx)):
                check(1)
            return
        with torch.cuda.device(self.device):
            check(f_refresh(self._h, mask, _stream_ptr(self.device)))

    def set_dof_command(self, tid: int, values: torch.Tensor):
        v = values if values.is_contiguous() else values.contiguous()
        assert v.dtype == torch.float32 and v.numel() == self.num_envs * self.model.nd
        _, _, f_cmd, idx, raw_stream = self._fast()
        if torch.cuda.current_device() == idx:
            if f_cmd(self._h, tid, v.data_ptr(), raw_stream(idx)):
                check(1)
            return
        with torch.cuda.device(self.device):
            check(f_cmd(self._h, tid, C.c_void_p(v.data_ptr()), _stream_ptr(self.device)))

    @_on_device
    def set_pos_target_indexed(self, values: torch.Tensor, idx: torch.Tensor):
        assert idx.dtype == torch.int32
        if values.is_contiguous() and self._hold(None, values, idx):
            return
        check(lib().shf_sim_set_pos_target_indexed(self._h, C.c_void_p(values.data_ptr()), C.c_void_p(idx.data_ptr()),
                                                   idx.numel(), _stream_ptr(self.device)))

    @_on_device
    def apply_body_force(self, force: torch.Tensor, pos: Optional[torch.Tensor] = None):
        """gym.apply_rigid_body_force_at_pos_tensors(force, pos): pos None = at the centres of mass."""
        f = force.contiguous()
        assert f.dtype == torch.float32 and f.numel() == self.num_envs * (self.model.nb + self.nboxes) * 3
        if pos is None:
            check(lib().shf_sim_apply_body_force(self._h, C.c_void_p(f.data_ptr()), _stream_ptr(self.device)))
            return
        p = pos.contiguous()
        assert p.dtype == torch.float32 and p.numel() == f.numel()
        check(lib().shf_sim_apply_body_force_at_pos(self._h, C.c_void_p(f.data_ptr()), C.c_void_p(p.data_ptr()), _stream_ptr(self.device)))

    @_on_device
    def commit_root_indexed(self, root: torch.Tensor, idx: torch.Tensor):
        assert idx.dtype == torch.int32 and root.is_contiguous()
        held, self._held, self._defer = self._held, None, False
        if held is not None:                  # the reset's three commits in one launch
            dof, tgt, didx = held
            check(lib().shf_sim_commit_reset(self._h, C.c_void_p(root.data_ptr()), C.c_void_p(idx.data_ptr()), idx.numel(),
                                             C.c_void_p(dof.data_ptr()) if dof is not None else None,
                                             C.c_void_p(tgt.data_ptr()) if tgt is not None else None, C.c_void_p(didx.data_ptr()),
                                             didx.numel(), _stream_ptr(self.device)))
            return
        check(lib().shf_sim_commit_root_indexed(self._h, C.c_void_p(root.data_ptr()), C.c_void_p(idx.data_ptr()),
                                                idx.numel(), _stream_ptr(self.device)))

    @_on_device
    def commit_root_all(self, root: torch.Tensor):
        self._flush_held()
        check(lib().shf_sim_commit_root_all(self._h, C.c_void_p(root.data_ptr()), _stream_ptr(self.device)))

    @_on_device
    def commit_dof_indexed(self, dof: torch.Tensor, idx: torch.Tensor):
        assert idx.dtype == torch.int32 and dof.is_contiguous()
        if self._hold(dof, None, idx):
            return
        check(lib().shf_sim_commit_dof_indexed(self._h, C.c_void_p(dof.data_ptr()), C.c_void_p(idx.data_ptr()),
                                               idx.numel(), _stream_ptr(self.device)))

    @_on_device
    def reset_all(self, default_root: torch.Tensor, default_dof: torch.Tensor, origins: Optional[torch.Tensor]):
        check(lib().shf_sim_reset_all(self._h, C.c_void_p(default_root.data_ptr()), C.c_void_p(default_dof.data_ptr()),
                                      C.c_void_p(origins.data_ptr()) if origins is not None else None,
                                      _stream_ptr(self.device)))

    def destroy(self):
        if self._h:
            lib().shf_sim_destroy(self._h)
            self._h = C.c_void_p()
        self.tensors.clear()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class A1Task:
    """The fused A1Conditional env step (shf_a1_*)."""

    def __init__(self, sim: Sim, params: _abi.ShfA1TaskParams):
        self.sim = sim
        self.params = params
        self._h = C.c_void_p()
        check(lib().shf_a1_create(sim._h, C.byref(params), C.byref(self._h)))
        self.tensors: Dict[int, torch.Tensor] = {}
        self.step_index = 0
        for tid in range(_abi.A1_COUNT):
            if tid == _abi.A1_PARAMS:
                self.bind(tid, _struct_to_device(params, sim.device))
                continue
            shape, dt = self.layout(tid)
            self.bind(tid, torch.zeros(shape, dtype=dt, device=sim.device))

    def layout(self, tid: int):
        shape = (C.c_int64 * 4)()
        nd, dt = C.c_int32(), C.c_int32()
        check(lib().shf_a1_layout(self._h, tid, shape, C.byref(nd), C.byref(dt)))
        return tuple(shape[:nd.value]), _TORCH_DTYPE[dt.value]

    def bind(self, tid: int, t: torch.Tensor):
        assert t.is_contiguous()
        self.tensors[tid] = t
        check(lib().shf_a1_bind(self._h, tid, C.c_void_p(t.data_ptr())))

    num_sums = 8            # leading (sum, count) entries of a statistics row: what the multi-GPU all-gather carries

    @_on_device
    def step(self, raw_actions: torch.Tensor, stats: bool = True) -> int:
        """One fused vec-step (episode statistics included); returns the statistics ring row it writes."""
        a = raw_actions.contiguous()
        assert a.dtype == torch.float32 and a.shape == (self.sim.num_envs, self.sim.model.nd)
        check(lib().shf_a1_step(self._h, C.c_void_p(a.data_ptr()), _stream_ptr(self.sim.device)))
        return self.advance_slot()

    @_on_device
    def step_random(self) -> int:
        """One fused vec-step of run_policy('random') (policy_runner.py:38-41): the U(-1, 1) actions are drawn inside the
        launch from the task's counter-based generator -- no per-step RNG launch; returns the statistics ring row."""
        check(lib().shf_a1_step_random(self._h, _stream_ptr(self.sim.device)))
        return self.advance_slot()

    @_on_device
    def launch_step(self, raw_actions: torch.Tensor):
        """Only the launch (no host-side bookkeeping): what a hipGraph capture records.  The kernel picks the ring row
        from its device-side step counter; pair every replay with advance_slot()."""
        check(lib().shf_a1_step(self._h, C.c_void_p(raw_actions.data_ptr()), _stream_ptr(self.sim.device)))

    def advance_slot(self) -> int:
        """Host mirror of the device-side step counter: the ring row of the step just launched / replayed."""
        idx = self.step_index
        self.step_index += 1
        return idx % (self.tensors[_abi.A1_STATS].shape[0] - 1)

    def kernel_symbol(self) -> str:
        """Mangled-name prefix of the instantiation shf_a1_step launches for this sim (build resource table)."""
        g, warped = self.sim.group, bool(self.sim.terrain.warped)
        mdl = self.sim.model
        if self.sim.params.solver != _abi.SOLVER_COMPLIANT:
            k16 = int(self.sim.params.max_contacts) > 8      # up to 16 constraints per env: the packed-response-matrix kernel
            name = ("k_a1_chain_tgs" if self.sim.params.solver == _abi.SOLVER_TGS else "k_a1_chain_pgs") + ("16" if k16 else "")
            return f"_Z{len(name)}{name}ILb{int(warped)}ELb{int(bool(mdl.self_collide and mdl.npair > 0))}EE"
        if getattr(self.sim, "mapping", "body") == "chain":
            return f"_Z10k_a1_chainILi{g}ELb{int(warped)}ELb{int(bool(mdl.self_collide and mdl.npair > 0))}EE"
        a1 = mdl.nb == 17 and mdl.nd == 12 and mdl.np == 76
        if mdl.self_collide and mdl.npair > 0:
            if a1:
                return "_Z21k_a1_step_self_a1_g32" if not warped else "_Z14k_a1_step_selfILi32E9FixedDims"
            return f"_Z14k_a1_step_selfILi{g}E7DynDims"
        if a1:
            return "_Z16k_a1_step_a1_g32" if (g == 32 and not warped) else f"_Z9k_a1_stepILi{g}E9FixedDims"
        return f"_Z9k_a1_stepILi{g}E7DynDims"

    @_on_device
    def reset_all(self):
        check(lib().shf_a1_reset_all(self._h, _stream_ptr(self.sim.device)))

    def destroy(self):
        if self._h:
            lib().shf_a1_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class AbbTask:
    """The fused AbbPushBox env step (shf_abb_*)."""

    def __init__(self, sim: Sim, params: _abi.ShfAbbTaskParams):
        self.sim = sim
        self.params = params
        self._h = C.c_void_p()
        check(lib().shf_abb_create(sim._h, C.byref(params), C.byref(self._h)))
        self.tensors: Dict[int, torch.Tensor] = {}
        self.step_index = 0
        for tid in range(_abi.ABB_COUNT):
            if tid == _abi.ABB_PARAMS:
                self.bind(tid, _struct_to_device(params, sim.device))
                continue
            shape = (C.c_int64 * 4)()
            nd, dt = C.c_int32(), C.c_int32()
            check(lib().shf_abb_layout(self._h, tid, shape, C.byref(nd), C.byref(dt)))
            self.bind(tid, torch.zeros(tuple(shape[:nd.value]), dtype=_TORCH_DTYPE[dt.value], device=sim.device))

    def bind(self, tid: int, t: torch.Tensor):
        assert t.is_contiguous()
        self.tensors[tid] = t
        check(lib().shf_abb_bind(self._h, tid, C.c_void_p(t.data_ptr())))

    num_sums = 4

    @_on_device
    def step(self, raw_actions: torch.Tensor, stats: bool = True) -> int:
        a = raw_actions.contiguous()
        assert a.dtype == torch.float32 and a.shape == (self.sim.num_envs, 3)
        check(lib().shf_abb_step(self._h, C.c_void_p(a.data_ptr()), _stream_ptr(self.sim.device)))
        return self.advance_slot()

    @_on_device
    def step_random(self) -> int:
        """As A1Task.step_random: run_policy('random') with the actions drawn inside the launch."""
        check(lib().shf_abb_step_random(self._h, _stream_ptr(self.sim.device)))
        return self.advance_slot()

    @_on_device
    def launch_step(self, raw_actions: torch.Tensor):
        check(lib().shf_abb_step(self._h, C.c_void_p(raw_actions.data_ptr()), _stream_ptr(self.sim.device)))

    def advance_slot(self) -> int:
        idx = self.step_index
        self.step_index += 1
        return idx % (self.tensors[_abi.ABB_STATS].shape[0] - 1)

    def kernel_symbol(self) -> str:
        """Mangled-name prefix of the instantiation shf_abb_step launches for this sim (build resource table)."""
        mdl = self.sim.model
        link = bool(mdl.link_collide and self.sim.nboxes > 0)
        fixed = mdl.nb == 7 and mdl.np == (59 if link else 3) and self.sim.nboxes == 3
        pre = f"_Z10k_abb_stepILi{self.sim.group}E"
        if mdl.nhull > 0 or getattr(self.sim, "scene_flags", 0):     # the convex narrow phase compiled in (csrc/shf_hull.h): run-time shapes
            hard = int(self.sim.params.solver != _abi.SOLVER_COMPLIANT)
            return f"_Z10k_abb_stepILi{32 if hard else self.sim.group}E7DynDims8DynSceneLb1ELi0ELb{hard}ELb1EE"
        if self.sim.params.solver != _abi.SOLVER_COMPLIANT and getattr(self.sim, "mapping", "body") == "split":
            return f"_Z18k_abb_step_ws_hardILb{int(link)}EE"     # arm wave + box wave, the solve regrouped at 32 lanes per env
        if self.sim.params.solver != _abi.SOLVER_COMPLIANT:    # the generic velocity-level solve: run-time shapes, 32 lanes per env
            if lib().shf_abb_step_pgs_is_wide(self._h):   # sixteen envs per workgroup of 512 threads
                return f"_Z19k_abb_step_pgs_wideILb{int(link)}EE"
            return f"_Z10k_abb_stepILi32E7DynDims8DynSceneLb{int(link)}ELi0ELb1ELb0EE"
        if not fixed:
            return pre + "7DynDims"
        if link:    # the shipped arm with its link volumes in the shipped scene (AbbLinkDims, AbbScene)
            if getattr(self.sim, "mapping", "body") == "split":
                return "_Z13k_abb_step_wsILi512ELb1EE"
            return pre + "9FixedDimsILi7ELi6ELi59ELi6ELi6EE10FixedSceneILi3ELi1ELi2EELb1ELi0ELb0ELb0EE"
        mp = getattr(self.sim, "mapping", "body")
        if mp == "split":
            return "_Z13k_abb_step_wsILi256ELb0EE"
        arm = 6 if mp == "chain" else 0
        return pre + f"9FixedDimsILi7ELi6ELi3ELi6ELi6EE10FixedSceneILi3ELi1ELi2EELb0ELi{arm}ELb0ELb0EE"

    @_on_device
    def reset_all(self):
        check(lib().shf_abb_reset_all(self._h, _stream_ptr(self.sim.device)))

    def destroy(self):
        if self._h:
            lib().shf_abb_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass
