"""The boundary as the reference's own code sees it (build container only: needs /root/reference, never shipped).

1. `shifu_amd.compat.install()` + the reference's UNMODIFIED example files
   (examples/a1_conditional/a1_conditional.py, examples/abb_pushbox_vision/a_prior_stage.py) import, their classes
   bind to this repo's ShifuVecEnv / LeggedRobot / ArmRobot / Box, and their configs instantiate.
2. Every `gym.X(...)`, `gymapi.X`, `gymtorch.X`, `gymutil.X` and isaacgym.torch_utils name used anywhere on the hot
   path of the reference (shifu/gym, shifu/units except the camera sensor, shifu/utils/terrain.py, the two examples)
   exists on the facade -- an AST scan, so a call site added upstream fails here by name.
3. gym.add_triangle_mesh without this repo's hint attributes (what the reference's own TerrainGymEnv passes,
   isaac_gym.py:369-385) recovers the height map, scales and vertex shifts exactly, and refuses other meshes."""
import ast
import os
import subprocess
import sys

import numpy as np
import pytest

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
needs_ref = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is only present in the build container")

_CHILD = r'''
import sys
sys.dont_write_bytecode = True
sys.path.insert(0, %(root)r)
import shifu_amd.compat
shifu_amd.compat.install(force=True)                  # isaacgym.* and shifu.* -> this repo (before the reference tree is visible)
sys.path.insert(0, %(ref)r)                           # `examples` now resolves to the reference's package
import examples.a1_conditional.a1_conditional as a1
import examples.abb_pushbox_vision.a_prior_stage as abb
import examples.a1_conditional.task_config as a1cfg
import examples.abb_pushbox_vision.task_config as abbcfg
for m in (a1, abb, a1cfg, abbcfg):
    assert m.__file__.startswith(%(ref)r), m.__file__
import shifu_amd
from shifu_amd.gym import ShifuVecEnv
from shifu_amd.units import ArmRobot, Box, LeggedRobot
assert issubclass(a1.A1Conditional, ShifuVecEnv) and issubclass(a1.A1Robot, LeggedRobot)
assert issubclass(abb.AbbPushBox, ShifuVecEnv) and issubclass(abb.AbbRobot, ArmRobot) and issubclass(abb.RandPosBox, Box)
c = a1cfg.A1EnvConfig()
assert c.num_envs == 4000 and c.terrain.mesh_type == "trimesh" and c.terrain.num_rows == 10 and hasattr(c, "terrian")   # Q5
assert c.sim_params.dt == 0.005 and c.sim_params.physx.contact_offset == 0.01 and c.control.decimation == 4
a = a1cfg.A1ActorConfig()
assert int(a.asset_options.default_dof_drive_mode) == 3 and a.dof_damping == [.5] * 12
p = abbcfg.PriorStageEnvConfig()
assert p.num_envs == 3000 and p.num_obs == 6 and p.num_actions == 3
r = abbcfg.AbbRobotConfig()
assert r.asset_options.fix_base_link and r.asset_options.disable_gravity
from shifu_amd.runner.utils import class_to_dict
d = class_to_dict(a1cfg.A1PPOConfig())
assert d["runner"]["max_iterations"] == 3000 and d["algorithm"]["entropy_coef"] == 0.01
# the user hooks are the reference's own functions
import inspect
assert inspect.getsourcefile(a1.A1Conditional.compute_observations).startswith(%(ref)r)
assert [f.__name__ for f in a1.A1Conditional.build_reward_functions(a1.A1Conditional.__new__(a1.A1Conditional))] == \
    ["tracking_lin_vel", "tracking_ang_vel", "stabilizing_base", "smoothing_action", "leg_collision", "torques_penalize"]
import os
assert not any(f.endswith(".pyc") for _, _, fs in os.walk(%(ref)r + "/examples") for f in fs), "bytecode written into the reference tree"
print("COMPAT-OK")
'''


@needs_ref
def test_unmodified_reference_examples_bind_to_this_backend():
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    out = subprocess.run([sys.executable, "-c", _CHILD % {"root": ROOT, "ref": REF}], capture_output=True, text=True, env=env,
                         cwd="/tmp")
    assert out.returncode == 0 and "COMPAT-OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


HOT_PATH_FILES = ["shifu/gym/isaac_gym.py", "shifu/gym/env.py", "shifu/units/units.py", "shifu/units/robot.py",
                  "shifu/units/object.py", "shifu/utils/terrain.py", "shifu/configs/env_config.py",
                  "shifu/configs/asset_config.py", "examples/a1_conditional/a1_conditional.py",
                  "examples/a1_conditional/task_config.py", "examples/abb_pushbox_vision/a_prior_stage.py",
                  "examples/abb_pushbox_vision/task_config.py"]
# graphics / camera entry points: accepted no-ops or NotImplementedError by design (out of scope)
TORCH_UTILS = ["to_torch", "quat_rotate_inverse", "quat_apply", "quat_mul", "quat_conjugate", "normalize",
               "get_axis_params", "torch_rand_float", "quat_from_euler_xyz"]


def _scan(path):
    tree = ast.parse(open(path).read(), filename=path)
    gym_calls, mod_attrs, names = set(), set(), set()
    for node in ast.walk(tree):
        if isinstance(node, ast.Attribute):
            v = node.value
            if isinstance(v, ast.Attribute) and v.attr == "gym" or isinstance(v, ast.Name) and v.id == "gym":
                gym_calls.add(node.attr)
            if isinstance(v, ast.Name) and v.id in ("gymapi", "gymtorch", "gymutil", "terrain_utils"):
                mod_attrs.add((v.id, node.attr))
        if isinstance(node, ast.Name):
            names.add(node.id)
    return gym_calls, mod_attrs, names


@needs_ref
def test_every_gym_name_the_reference_hot_path_uses_exists_on_the_facade():
    from shifu_amd.isaacgym import gymapi, gymtorch, gymutil, terrain_utils, torch_utils
    mods = {"gymapi": gymapi, "gymtorch": gymtorch, "gymutil": gymutil, "terrain_utils": terrain_utils}
    gym = gymapi.acquire_gym()
    calls, attrs, used_tu = set(), set(), set()
    for f in HOT_PATH_FILES:
        c, a, n = _scan(os.path.join(REF, f))
        calls |= c; attrs |= a
        used_tu |= n & set(TORCH_UTILS)
    assert len(calls) >= 55, sorted(calls)                 # SURVEY 8b counts 69 with the camera sensor's calls
    missing = sorted(c for c in calls if not hasattr(gym, c))
    assert not missing, f"gym.* methods the reference calls but the facade lacks: {missing}"
    missing = sorted(f"{m}.{a}" for m, a in attrs if not hasattr(mods[m], a))
    assert not missing, f"isaacgym names the reference uses but the facade lacks: {missing}"
    missing = sorted(n for n in used_tu if not hasattr(torch_utils, n))
    assert not missing and len(used_tu) >= 5, (missing, used_tu)
    for name in ("simulate", "refresh_dof_state_tensor", "set_dof_actuation_force_tensor", "set_actor_root_state_tensor_indexed",
                 "set_dof_state_tensor_indexed", "acquire_jacobian_tensor", "apply_rigid_body_force_at_pos_tensors",
                 "add_triangle_mesh", "create_box", "find_actor_rigid_body_handle"):
        assert name in calls, name                          # the scan really saw the hot path


def test_add_triangle_mesh_without_hints_recovers_the_grid_exactly():
    from types import SimpleNamespace as NS
    from shifu_amd.gym.a1_fused import default_terrain_cfg
    from shifu_amd.isaacgym import gymapi, terrain_utils
    from shifu_amd.utils.terrain import Terrain
    cfg = default_terrain_cfg(mesh_type="trimesh", num_rows=3, num_cols=6, border_size=5)
    np.random.seed(5)
    ter = Terrain(cfg, 32)
    assert ter.vertices.shape[0] == ter.tot_rows * ter.tot_cols
    gym = gymapi.acquire_gym()

    def params(hint):
        p = gymapi.TriangleMeshParams()
        p.nb_vertices, p.nb_triangles = ter.vertices.shape[0], ter.triangles.shape[0]
        p.transform.p.x = p.transform.p.y = -cfg.border_size
        p.static_friction = p.dynamic_friction = 1.0
        if hint:
            p.height_samples, p.horizontal_scale, p.vertical_scale = ter.heightsamples, cfg.horizontal_scale, cfg.vertical_scale
            p.slope_threshold = cfg.slope_treshold
        return p
    a, b = NS(terrain=None), NS(terrain=None)
    gym.add_triangle_mesh(a, ter.vertices.flatten(order="C"), ter.triangles.flatten(order="C"), params(True))
    gym.add_triangle_mesh(b, ter.vertices.flatten(order="C"), ter.triangles.flatten(order="C"), params(False))
    assert a.terrain[0] == b.terrain[0] == "heightfield"
    np.testing.assert_array_equal(a.terrain[1], b.terrain[1])                      # int16 samples
    assert np.float32(a.terrain[2]) == np.float32(b.terrain[2]) and np.float32(a.terrain[3]) == np.float32(b.terrain[3])
    assert a.terrain[4:6] == b.terrain[4:6]
    np.testing.assert_array_equal(a.terrain[6], b.terrain[6])                      # vertex shifts + plain bits
    assert (a.terrain[6] & 0x0f != 5).sum() > 50, "the terrain must contain shifted vertices"
    # anything that is not such a grid is refused loudly
    v = ter.vertices.copy(); v[7, 0] += 0.033
    with pytest.raises(NotImplementedError, match="convert_heightfield_to_trimesh"):
        terrain_utils.heightfield_from_trimesh(v, ter.triangles)
    t = ter.triangles.copy(); t[3] = t[3][::-1]
    with pytest.raises(NotImplementedError):
        terrain_utils.heightfield_from_trimesh(ter.vertices, t)


def test_create_sim_honours_the_physx_solver_settings_and_names_the_ignored_ones_once():
    """shifu's config sets PhysX solver parameters (shifu/configs/env_config.py:46-58).  Since round 5 the solver fields
    (solver_type, num_position_iterations, num_velocity_iterations, rest_offset, bounce_threshold_velocity) configure the
    velocity-level contact solve and are not reported any more; the two buffer-size fields have no counterpart and the
    facade says so at create_sim -- once per process -- rather than staying silent."""
    import warnings
    from shifu_amd.isaacgym import gymapi
    gymapi.Gym._warned_physx = False
    gym = gymapi.acquire_gym()
    sp = gymapi.SimParams()
    sp.physx.solver_type = 1                      # env_config.py:50
    sp.physx.num_position_iterations = 8          # env_config.py:51
    sp.physx.bounce_threshold_velocity = 0.5      # env_config.py:56
    sp.physx.contact_offset = 0.01                # env_config.py:54
    sp.physx.max_gpu_contact_pairs = 2 ** 23      # env_config.py:58
    sp.physx.default_buffer_size_multiplier = 5   # env_config.py:59
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        gym.create_sim(0, 0, gymapi.SIM_PHYSX, sp)
        gym.create_sim(0, 0, gymapi.SIM_PHYSX, sp)
    msgs = [str(x.message) for x in w]
    assert sum("max_gpu_contact_pairs" in m for m in msgs) == 1
    assert sum("default_buffer_size_multiplier" in m for m in msgs) == 1
    for honoured in ("solver_type", "num_position_iterations", "bounce_threshold_velocity", "contact_offset"):
        assert not any(honoured in m for m in msgs), honoured


def test_pgs_support_query():
    """shf_model_pgs_supported: the A1 on its own (chain-mapped kernels), the ABB arm with its three box actors and any other
    scene of at most 32 bodies + boxes (the generic solve, csrc/shf_hard.h) -- yes; more box actors than the scene holds -- no
    (the facade would then keep the compliant law and say so)."""
    import ctypes as C
    from shifu_amd._lib import lib
    from shifu_amd.abb_task import abb_model
    from tests.helpers import a1_model
    a1 = a1_model().blob
    assert lib().shf_model_pgs_supported(C.byref(a1), 0) == 1
    assert lib().shf_model_pgs_supported(C.byref(a1), 1) == 1
    assert lib().shf_model_pgs_supported(C.byref(a1_model(self_collision=True).blob), 0) == 1
    abb = abb_model(link_contacts=True).blob
    assert lib().shf_model_pgs_supported(C.byref(abb), 3) == 1
    assert lib().shf_model_pgs_supported(C.byref(abb), 5) == 0 and lib().shf_model_pgs_supported(None, 0) == 0


def test_trimesh_recovery_with_sparse_height_levels_and_shifted_border():
    """a valid convert_heightfield_to_trimesh mesh whose height levels are never one vertical step apart
    ({0, 2, 5} x vs) and whose outermost row has shifted vertices is still recovered exactly (vs from the gap divided by
    1..4, hs from the median vertex spacing)."""
    from shifu_amd.isaacgym import terrain_utils as tu
    hs, vs = 0.1, 0.005
    hf = np.zeros((12, 10), np.int16)
    hf[4:8, 3:7] = 2
    hf[9:, :] = 5              # a tall step reaching the last row: its riser shifts vertices of the border rows
    hf[0, :] = 5
    v, t = tu.convert_heightfield_to_trimesh(hf, hs, vs, 0.01)
    got, ghs, gvs, warp = tu.heightfield_from_trimesh(v, t)
    np.testing.assert_array_equal(got, hf)
    assert abs(ghs - hs) < 1e-6 and abs(gvs - vs) < 1e-7
    np.testing.assert_array_equal(warp, tu.trimesh_warp_map(hf, hs, vs, 0.01))
    # and with the caller's scales
    got2, _, gvs2, _ = tu.heightfield_from_trimesh(v, t, horizontal_scale=hs, vertical_scale=vs)
    np.testing.assert_array_equal(got2, hf)
    assert gvs2 == vs
