"""ctypes binding of include/shifu_amd.h.  There is no CPU fallback: if the HIP
library is missing this module raises, and so does everything built on it."""
import ctypes as C
import os

from . import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
# SHIFU_AMD_LIB overrides the in-tree library (kernel A/B experiments only)
_PATH = os.environ.get("SHIFU_AMD_LIB") or os.path.join(_HERE, "libshifu_amd.so")
_lib = None

vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
_SIGS = {
    "shf_abi_version": ([], i32),
    "shf_sim_create": ([C.POINTER(_abi.ShfSimParams), C.POINTER(vp)], i32),
    "shf_sim_destroy": ([vp], i32),
    "shf_sim_set_terrain": ([vp, C.POINTER(_abi.ShfTerrain)], i32),
    "shf_sim_set_articulation": ([vp, C.POINTER(_abi.ShfModel)], i32),
    "shf_model_bounds": ([C.POINTER(_abi.ShfModel)], i32),
    "shf_model_pgs_supported": ([C.POINTER(_abi.ShfModel), i32], i32),
    "shf_abb_step_pgs_is_wide": ([vp], i32),
    "shf_sim_add_box": ([vp, C.POINTER(_abi.ShfBoxDesc)], i32),
    "shf_sim_set_hulls": ([vp, C.POINTER(_abi.ShfHullSet)], i32),
    "shf_sim_set_scene_flags": ([vp, i32], i32),
    "shf_convex_manifold": ([i32, vp, vp, C.c_float, i32, vp, vp], i32),
    "shf_sim_finalize": ([vp, i32, i64], i32),
    "shf_sim_set_group": ([vp, i32], i32),
    "shf_sim_set_mapping": ([vp, i32], i32),
    "shf_sim_layout": ([vp, i32, C.POINTER(i64), C.POINTER(i32), C.POINTER(i32)], i32),
    "shf_sim_bind": ([vp, i32, vp], i32),
    "shf_sim_reset_all": ([vp, vp, vp, vp, vp], i32),
    "shf_sim_step": ([vp, vp], i32),
    "shf_sim_refresh": ([vp, i32, vp], i32),
    "shf_sim_set_dof_command": ([vp, i32, vp, vp], i32),
    "shf_sim_set_pos_target_indexed": ([vp, vp, vp, i32, vp], i32),
    "shf_sim_apply_body_force": ([vp, vp, vp], i32),
    "shf_sim_apply_body_force_at_pos": ([vp, vp, vp, vp], i32),
    "shf_sim_commit_root_indexed": ([vp, vp, vp, i32, vp], i32),
    "shf_sim_commit_root_all": ([vp, vp, vp], i32),
    "shf_sim_commit_dof_indexed": ([vp, vp, vp, i32, vp], i32),
    "shf_sim_commit_reset": ([vp, vp, vp, i32, vp, vp, vp, i32, vp], i32),
    "shf_sim_step_split_supported": ([vp], i32),
    "shf_a1_create": ([vp, C.POINTER(_abi.ShfA1TaskParams), C.POINTER(vp)], i32),
    "shf_a1_destroy": ([vp], i32),
    "shf_a1_layout": ([vp, i32, C.POINTER(i64), C.POINTER(i32), C.POINTER(i32)], i32),
    "shf_a1_bind": ([vp, i32, vp], i32),
    "shf_a1_step": ([vp, vp, vp], i32),
    "shf_a1_step_random": ([vp, vp], i32),
    "shf_a1_reset_all": ([vp, vp], i32),
    "shf_abb_create": ([vp, C.POINTER(_abi.ShfAbbTaskParams), C.POINTER(vp)], i32),
    "shf_abb_destroy": ([vp], i32),
    "shf_abb_layout": ([vp, i32, C.POINTER(i64), C.POINTER(i32), C.POINTER(i32)], i32),
    "shf_abb_bind": ([vp, i32, vp], i32),
    "shf_abb_step": ([vp, vp, vp], i32),
    "shf_abb_step_random": ([vp, vp], i32),
    "shf_abb_reset_all": ([vp, vp], i32),
    "shf_mlp_linear_forward": ([vp, vp, vp, vp, i32, i32, i32, i32, vp], i32),
    "shf_mlp_linear_backward_input": ([vp, vp, vp, vp, i32, i32, i32, vp], i32),
    "shf_mlp_backward_weight_workspace": ([i32, i32, i32, C.POINTER(i64)], i32),
    "shf_mlp_pack_bytes": ([i32, i32, C.POINTER(i64)], i32),
    "shf_mlp_pack_weights": ([vp, vp, i32, i32, vp], i32),
    "shf_mlp_panel_forward": ([vp, vp, vp, vp, i32, i32, i32, i32, vp], i32),
    "shf_mlp_panel_backward_input": ([vp, vp, vp, vp, i32, i32, i32, vp], i32),
    "shf_mlp_chain_forward": ([vp, i32, vp, vp], i32),
    "shf_mlp_chain_fits": ([vp], i32),
    "shf_mlp_set_precision": ([i32], i32),
    "shf_mlp_get_precision": ([], i32),
    "shf_mlp_linear_backward_weight": ([vp, vp, vp, vp, vp, vp, i32, i32, i32, vp], i32),
    "shf_copy_many": ([C.POINTER(vp), C.POINTER(vp), C.POINTER(i64), i32, vp], i32),
    "shf_episode_bookkeeping": ([vp, vp, i32, i64, vp, vp, vp, vp], i32),
    "shf_gather_rows": ([C.POINTER(vp), C.POINTER(vp), C.POINTER(i32), i32, vp, i64, vp], i32),
    "shf_adapt_lr": ([vp, vp] + [C.c_float] * 6 + [vp], i32),
    # library glue of the hook-compatible path (csrc/shf_glue.hip)
    "shf_base_frame_state": ([vp, vp, i64, i32, i32, vp, vp, vp, vp, vp], i32),
    "shf_get_heights": ([C.POINTER(_abi.ShfTerrain), vp, vp, vp, i64, vp, i32, i32, vp, vp], i32),
    "shf_history_add": ([vp, vp, i64, i32, vp], i32),
    "shf_rows_fill_indexed": ([vp, vp, i32, i64, i32, C.c_float, vp], i32),
    "shf_episode_log": ([C.POINTER(vp), i32, vp, i32, i64, C.c_float, vp, vp, vp], i32),
    "shf_reset_bookkeeping": ([C.POINTER(vp), i32, vp, i32, i64, C.c_float, vp, vp, vp, vp, i32, vp, i32, vp], i32),
    "shf_reset_dof_rows": ([vp, vp, vp, vp, i32, i64, i32, vp, vp, vp], i32),
    "shf_ik_dls": ([vp, i64, vp, i32, vp, i64, vp, i32, i32, C.c_float, vp, vp], i32),
    "shf_reward_accumulate": ([C.POINTER(vp), C.POINTER(vp), i32, i64, vp, vp], i32),
    "shf_gae": ([vp, vp, vp, vp, i32, i64, C.c_float, C.c_float, vp, vp], i32),
    "shf_ppo_loss_workspace": ([i64, i32, C.POINTER(i64)], i32),
    "shf_ppo_loss": ([vp] * 10 + [i64, i32, C.c_float, C.c_float, C.c_float, i32] + [vp] * 6, i32),
}
EXPORTS = sorted(list(_SIGS) + ["shf_last_error", "shf_mlp_last_error"])


class BackendError(RuntimeError):
    pass


def lib():
    """The loaded HIP library.  torch is imported first so that the HIP runtime
    already in the process (torch's bundled libamdhip64.so.7) is the one used."""
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            raise BackendError(f"{_PATH} is missing: build it with `python -m shifu_amd.build` "
                               "(there is no CPU fallback for the MI355X backend)")
        import torch  # noqa: F401  (loads the process-wide HIP runtime)
        l = C.CDLL(_PATH)
        for name, (args, res) in _SIGS.items():
            f = getattr(l, name)
            f.argtypes, f.restype = args, res
        l.shf_last_error.restype = C.c_char_p
        l.shf_mlp_last_error.restype = C.c_char_p
        if l.shf_abi_version() != _abi.SHF_ABI_VERSION:
            raise BackendError("libshifu_amd.so ABI version mismatch: rebuild")
        _lib = l
    return _lib


def check(rc: int):
    if rc != 0:
        raise BackendError(lib().shf_last_error().decode())
