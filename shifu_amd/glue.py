"""Host side of csrc/shf_glue.hip: the LIBRARY glue of the hook-compatible path as single HIP launches.

ShifuVecEnv with torch hooks (the source-compatible path) spends most of its 2.4 ms per vec-step in small torch launches,
and a good part of those are not user hooks but library code: LeggedRobot.post_step (reference
shifu/units/robot.py:222-229), TerrainGymEnv.get_heights (shifu/gym/isaac_gym.py:393-433), HistoryRecorder
(shifu/utils/train.py:12-35), ShifuVecEnv.log_info / compute_reward (shifu/gym/env.py:149-185).  Each function below is
that piece as one kernel, used by the mirror classes whenever their tensors live on the GPU.  Results equal the oracle's
glue_* functions bit for bit (tests/test_gpu_glue.py); there is no CPU path here -- CPU tensors (golden-vector tests of
the Python mirror) keep the reference's torch expressions in the classes themselves."""
import ctypes as C

import torch

from ._lib import check, lib

_vp = C.c_void_p


def _stream(dev):
    return _vp(torch._C._cuda_getCurrentRawStream(dev.index if dev.index is not None else torch.cuda.current_device()))


def _f32c(t):
    assert t.dtype == torch.float32 and t.is_contiguous() and t.is_cuda, "float32 contiguous CUDA tensor expected"
    return _vp(t.data_ptr())


def usable(*tensors) -> bool:
    return all(t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() for t in tensors)


def base_frame_state(root_state, root_indices, up_axis, lin, ang, pg, gvec):
    """In place: lin / ang / pg / gvec (n, 3) from root_state rows root_indices (int64) -- LeggedRobot.post_step."""
    n = lin.shape[0]
    idx = None if root_indices is None else root_indices.contiguous()
    assert idx is None or (idx.dtype == torch.int64 and idx.is_cuda and idx.numel() == n)
    with torch.cuda.device(root_state.device):
        check(lib().shf_base_frame_state(_f32c(root_state), None if idx is None else _vp(idx.data_ptr()), root_state.shape[0], n,
                                         int(up_axis), _f32c(lin), _f32c(ang), _f32c(pg), _f32c(gvec), _stream(root_state.device)))


def get_heights(terrain_struct, height_samples, root_state, root_indices, points_xy, n):
    """(n, P) measured heights -- TerrainGymEnv.get_heights.  height_samples: int16 (rows, cols) device tensor."""
    P = points_xy.shape[0]
    out = torch.empty(n, P, device=root_state.device, dtype=torch.float32)
    idx = None if root_indices is None else root_indices.contiguous()
    assert height_samples is None or (height_samples.dtype == torch.int16 and height_samples.is_contiguous())
    with torch.cuda.device(root_state.device):
        check(lib().shf_get_heights(C.byref(terrain_struct), None if height_samples is None else _vp(height_samples.data_ptr()),
                                    _f32c(root_state), None if idx is None else _vp(idx.data_ptr()), root_state.shape[0],
                                    _f32c(points_xy), n, P, _f32c(out), _stream(root_state.device)))
    return out


def history_add(history_buf, x):
    """HistoryRecorder.add on a contiguous (..., H) buffer."""
    H = history_buf.shape[-1]
    rows = history_buf.numel() // H
    x = x.contiguous()
    assert x.numel() == rows
    with torch.cuda.device(history_buf.device):
        check(lib().shf_history_add(_f32c(history_buf), _f32c(x), rows, H, _stream(history_buf.device)))


def rows_fill_indexed(buf, idx, value=0.0):
    """buf[idx] = value along dim 0 (HistoryRecorder.reset_idx)."""
    idx = idx.contiguous()
    assert idx.dtype == torch.int64 and idx.is_cuda
    row_words = buf.numel() // buf.shape[0]
    with torch.cuda.device(buf.device):
        check(lib().shf_rows_fill_indexed(_f32c(buf), _vp(idx.data_ptr()), idx.numel(), buf.shape[0], row_words, float(value),
                                          _stream(buf.device)))


class EpisodeLog:
    """extras["episode"] means for one reset set in one launch (ShifuVecEnv.log_info): keeps the 17-word workspace and
    hands out a fresh (K,) result tensor per call (the caller stores 0-d views of it in `extras`)."""

    def __init__(self, device):
        self.ws = torch.zeros(17, dtype=torch.int64, device=device)

    def __call__(self, sums_list, env_ids, episode_length_s, episode_length=None, reset_buf=None, history=None):
        """episode_length / reset_buf / history given: the rest of ShifuVecEnv.reset_idx's buffer writes for the same ids in the
        same launch (shf_reset_bookkeeping): episode_length[ids] = 0, reset_buf[ids] = 1, history[ids] = 0."""
        K, dev = len(sums_list), self.ws.device
        if env_ids.numel() == 0:          # torch.mean over an empty selection (the kernel returns before writing)
            return torch.full((K,), float("nan"), dtype=torch.float32, device=dev)
        out = torch.empty(K, dtype=torch.float32, device=dev)
        ptrs = (_vp * K)(*[t.data_ptr() for t in sums_list])
        ids = env_ids.contiguous()
        assert ids.dtype == torch.int64
        with torch.cuda.device(dev):
            if episode_length is None and reset_buf is None and history is None:
                check(lib().shf_episode_log(ptrs, K, _vp(ids.data_ptr()), ids.numel(), sums_list[0].numel(), float(episode_length_s),
                                            _vp(self.ws.data_ptr()), _vp(out.data_ptr()), _stream(dev)))
            else:
                n = sums_list[0].numel()
                check(lib().shf_reset_bookkeeping(
                    ptrs, K, _vp(ids.data_ptr()), ids.numel(), n, float(episode_length_s), _vp(self.ws.data_ptr()), _vp(out.data_ptr()),
                    _vp(episode_length.data_ptr()) if episode_length is not None else None,
                    _vp(reset_buf.data_ptr()) if reset_buf is not None else None, reset_buf.element_size() if reset_buf is not None else 0,
                    _vp(history.data_ptr()) if history is not None else None, history.numel() // n if history is not None else 0,
                    _stream(dev)))
        return out


def reward_accumulate(terms, sums_list, rew_buf):
    """rew_buf = terms[0] + terms[1] + ...; sums_list[k] += terms[k] (ShifuVecEnv.compute_reward)."""
    K = len(terms)
    tp = (_vp * K)(*[t.data_ptr() for t in terms])
    sp = (_vp * K)(*[t.data_ptr() for t in sums_list])
    with torch.cuda.device(rew_buf.device):
        check(lib().shf_reward_accumulate(tp, sp, K, rew_buf.numel(), _f32c(rew_buf), _stream(rew_buf.device)))


def reset_dof_rows(dof_state, dof_targets, default_dof_pos, env_ids, root_indices):
    """Robot._reset_dof_state's tensor writes in one launch; returns the int32 actor ids for the indexed commits."""
    ids = env_ids.contiguous()
    assert ids.dtype == torch.int64 and ids.is_cuda
    n, nd = dof_targets.shape
    actor_ids = torch.empty(ids.numel(), dtype=torch.int32, device=ids.device)
    ri = root_indices.contiguous()
    with torch.cuda.device(ids.device):
        check(lib().shf_reset_dof_rows(_f32c(dof_state), _f32c(dof_targets), _f32c(default_dof_pos), _vp(ids.data_ptr()), ids.numel(),
                                       n, nd, _vp(ri.data_ptr()), _vp(actor_ids.data_ptr()), _stream(ids.device)))
    return actor_ids


def ik_dls(j_ee, dof_pos, ee_pose, goal_pose, damping):
    """ArmRobot.inverse_kinematics in one launch.  j_ee (n, 6, nd), ee_pose (n, 7) and dof_pos (n, nd) may be strided
    views into the Jacobian / rigid-body-state / dof-state tensors (unit stride inside a block, any stride between envs)."""
    n, six, nd = j_ee.shape
    assert six == 6 and j_ee.stride(2) == 1 and j_ee.stride(1) == nd and ee_pose.stride(1) == 1
    assert dof_pos.shape == (n, nd) and dof_pos.stride(0) == nd * dof_pos.stride(1)
    goal = goal_pose.contiguous()
    out = torch.empty(n, nd, device=j_ee.device, dtype=torch.float32)
    with torch.cuda.device(j_ee.device):
        check(lib().shf_ik_dls(_vp(j_ee.data_ptr()), j_ee.stride(0), _vp(dof_pos.data_ptr()), dof_pos.stride(1),
                               _vp(ee_pose.data_ptr()), ee_pose.stride(0), _f32c(goal), n, nd, float(damping), _f32c(out),
                               _stream(j_ee.device)))
    return out
