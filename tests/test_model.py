"""Model compiler robustness: every URDF the reference ships (asset/urdf/**, present only in the build container)
flattens into a ShfModel -- missing <inertial> children, capsules, fixed-joint chains, mesh-only links -- or is
refused with a capacity message; the two vendored physics-only URDFs are checked for their known shape."""
import glob
import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shifu_amd import _abi
from shifu_amd.model import asset_path, compile_urdf

REF = "/root/reference/asset/urdf"


def test_vendored_robots():
    a1 = compile_urdf(asset_path("a1.urdf"), default_dof_drive_mode=_abi.DOF_MODE_EFFORT)
    assert (a1.blob.nb, a1.blob.nd, a1.blob.np, a1.blob.nlevels, a1.blob.nklevels) == (17, 12, 76, 3, 4)
    assert abs(a1.total_mass - 12.454) < 1e-3 and a1.body_names[0] == "base" and "FL_foot" in a1.body_names
    from shifu_amd.abb_task import abb_model
    abb = abb_model()
    assert abb.blob.nd == 6 and abb.blob.fixed_base == 1 and abb.blob.nsph == 8


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is only present in the build container")
def test_every_reference_urdf_compiles_or_is_refused_with_a_reason():
    paths = sorted(glob.glob(os.path.join(REF, "**", "*.urdf"), recursive=True))
    assert len(paths) > 20
    refused = []
    for p in paths:
        try:
            cm = compile_urdf(p)
        except AssertionError as e:
            assert "SHF_MAX" in str(e), (p, e)
            refused.append(os.path.basename(p))
            continue
        m = cm.blob
        assert 1 <= m.nb <= _abi.MAX_BODIES and 0 <= m.nd <= 32 and len(cm.body_names) == m.nb and len(cm.dof_names) == m.nd
        assert all(m.parent[b] < b for b in range(1, m.nb)), "bodies are numbered parents-first"
        assert sum(m.pt_count[b] for b in range(m.nb)) == m.np
    assert refused == ["anymal.urdf"]          # 143 collision sample points > SHF_MAX_POINTS (96)
