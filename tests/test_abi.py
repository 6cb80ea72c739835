"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950,
loads, and exports every symbol include/shifu_amd.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def libpath():
    from shifu_amd.build import build_native
    return build_native()


def _declared():
    hdr = open(os.path.join(ROOT, "include", "shifu_amd.h")).read()
    return sorted(set(re.findall(r"\b(shf_[a-z0-9_]+)\s*\(", hdr)))


def test_header_symbols_are_exported(libpath):
    lib = ctypes.CDLL(libpath)
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/shifu_amd.h but not exported"


def test_binding_table_covers_header(libpath):
    from shifu_amd import _lib
    assert set(_declared()) == set(_lib.EXPORTS)


def test_struct_sizes_match_header(libpath, tmp_path):
    """ctypes mirrors in shifu_amd/_abi.py must have the C layout."""
    import subprocess
    from shifu_amd import _abi
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "shifu_amd.h"\nint main(){printf("%zu %zu %zu %zu %zu\\n",'
                   'sizeof(ShfModel),sizeof(ShfSimParams),sizeof(ShfTerrain),sizeof(ShfA1TaskParams),sizeof(ShfBoxDesc));'
                   'printf("%zu %zu\\n",sizeof(ShfScene),sizeof(ShfAbbTaskParams));}')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    want = [ctypes.sizeof(t) for t in (_abi.ShfModel, _abi.ShfSimParams, _abi.ShfTerrain, _abi.ShfA1TaskParams,
                                       _abi.ShfBoxDesc, _abi.ShfScene, _abi.ShfAbbTaskParams)]
    assert got == want


def test_error_path_without_gpu(libpath):
    """Argument validation happens on the host and reports through shf_last_error."""
    from shifu_amd import _abi, _lib
    l = _lib.lib()
    h = ctypes.c_void_p()
    p = _abi.ShfSimParams()
    p.dt = 0.0
    assert l.shf_sim_create(ctypes.byref(p), ctypes.byref(h)) != 0
    assert b"dt" in l.shf_last_error()
    p.dt = 0.005
    assert l.shf_sim_create(ctypes.byref(p), ctypes.byref(h)) == 0
    assert l.shf_sim_finalize(h, 4, 0) != 0 and b"articulation" in l.shf_last_error()
    with pytest.raises(_lib.BackendError):
        _lib.check(l.shf_sim_step(h, None))
    l.shf_sim_destroy(h)


def test_register_budgets_of_the_fused_step():
    """The build records every kernel's register use; the default instantiation of the fused A1 step must stay
    within two waves per SIMD and out of scratch (shifu_amd/build.py: BUDGETS)."""
    import json
    from shifu_amd import build
    build.build_native()
    res = json.load(open(build.RESOURCES))
    build.check_budgets(res)
    k = [v for n, v in res.items() if n.startswith("_Z16k_a1_step_a1_g32")][0]
    assert k["spill"] == 0 and k["vgprs"] <= 256
    # the kernels that are the defaults since round 5 (velocity-level solve) are under the same guard
    guarded = [p for p, _, _ in build.BUDGETS]
    for fam in ("_Z14k_a1_chain_pgs", "_Z20k_sim_step_chain_pgs", "_Z19k_abb_step_pgs_wide", "_Z19k_sim_step_pgs_wide",
                "_Z14k_a1_chain_tgs", "_Z20k_sim_step_chain_tgs", "_Z18k_abb_step_ws_hard"):      # (round 6: the TGS forms, config 5's own kernel)
        assert any(p.startswith(fam) for p in guarded), fam
    for sym in ("_Z14k_a1_chain_pgsILb0ELb0EE", "_Z14k_a1_chain_tgsILb0ELb0EE"):       # the default A1 kernel of `solver_type` 0 / 1
        k = [v for n, v in res.items() if n.startswith(sym)][0]
        assert k["spill"] == 0 and k["scratch"] == 0 and k["vgprs"] + k["agprs"] <= 256, sym
    ws = [v for n, v in res.items() if n.startswith("_Z18k_abb_step_ws_hardILb0EE")][0]
    assert ws["spill"] == 0 and ws["scratch"] == 0


def test_a_bloated_default_kernel_fails_the_budget_check():
    """A 257th register or new scratch in the headline kernel must not build green (shifu_amd/build.py: check_budgets)."""
    import copy
    import json
    from shifu_amd import build
    build.build_native()
    res = json.load(open(build.RESOURCES))
    name = [n for n in res if n.startswith("_Z14k_a1_chain_tgsILb0ELb0EE")][0]      # the headline kernel since round 6
    for field, value in (("vgprs", 257), ("scratch", 16), ("agprs", 64)):
        bad = copy.deepcopy(res)
        bad[name][field] = value
        with pytest.raises(RuntimeError, match="register budget"):
            build.check_budgets(bad)
    missing = {k: v for k, v in res.items() if not k.startswith("_Z14k_a1_chain_tgs")}
    with pytest.raises(RuntimeError, match="missing"):
        build.check_budgets(missing)


def test_empty_and_oversize_inputs_are_refused_on_the_host(libpath):
    """Edge cases of the boundary that need no GPU: zero envs, a model beyond SHF_MAX_*, a model wider than the
    lane group, a fifth box actor, a bad lane-group size, layouts before finalize."""
    import copy
    from shifu_amd import _abi, _lib
    from shifu_amd.backend import default_sim_params
    from shifu_amd.model import asset_path, compile_urdf
    l = _lib.lib()
    cm = compile_urdf(asset_path("a1.urdf"), default_dof_drive_mode=_abi.DOF_MODE_EFFORT)
    h = ctypes.c_void_p()
    sp = default_sim_params()
    assert l.shf_sim_create(ctypes.byref(sp), ctypes.byref(h)) == 0
    shape, nd, dt = (ctypes.c_int64 * 4)(), ctypes.c_int32(), ctypes.c_int32()
    assert l.shf_sim_layout(h, _abi.T_DOF_STATE, shape, ctypes.byref(nd), ctypes.byref(dt)) != 0
    assert b"finalized" in l.shf_last_error()
    big = copy.deepcopy(cm.blob)
    big.nb = 33
    assert l.shf_sim_set_articulation(h, ctypes.byref(big)) != 0 and b"SHF_MAX" in l.shf_last_error()
    big.nb, big.np = cm.blob.nb, _abi.MAX_POINTS + 1
    assert l.shf_sim_set_articulation(h, ctypes.byref(big)) != 0
    assert l.shf_sim_set_articulation(h, ctypes.byref(cm.blob)) == 0
    assert l.shf_sim_finalize(h, 0, 0) != 0 and b"num_envs" in l.shf_last_error()
    assert l.shf_sim_finalize(h, -3, 0) != 0
    assert l.shf_sim_set_group(h, 48) != 0 and b"16, 32 or 64" in l.shf_last_error()
    assert l.shf_sim_set_group(h, 16) != 0 and b"exceed the lane group" in l.shf_last_error()     # 17 bodies > 16 lanes
    assert l.shf_sim_set_group(h, 32) == 0
    assert l.shf_sim_set_group(None, 32) != 0 and b"null" in l.shf_last_error()
    box = _abi.ShfBoxDesc()
    box.dim[0] = box.dim[1] = box.dim[2] = 0.1
    for _ in range(4):
        assert l.shf_sim_add_box(h, ctypes.byref(box)) == 0
    assert l.shf_sim_add_box(h, ctypes.byref(box)) != 0 and b"too many boxes" in l.shf_last_error()
    assert l.shf_sim_finalize(h, 1, 0) == 0                                            # the smallest legal env count
    assert l.shf_sim_layout(h, _abi.T_DOF_STATE, shape, ctypes.byref(nd), ctypes.byref(dt)) == 0
    assert tuple(shape[:nd.value]) == (12, 2)
    l.shf_sim_destroy(h)
