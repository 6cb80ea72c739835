"""Scene constants of the ABB push-box task (reference examples/abb_pushbox_vision/
task_config.py:13-84, a_prior_stage.py:24-73): arm + table + cube + goal pad."""
from __future__ import annotations

from . import _abi
from .model import asset_path, compile_urdf

ABB_DEFAULT_DOF_POS = [0., 0.6437, 0.1748, 0., 0.7541, 0.]   # task_config.py:57
ABB_BASE_POS = [-0.48, 0.0, 0.0]                              # task_config.py:55

# The reference collides the rod as the convex hull of rod.stl (a cylinder r = 0.0194 m,
# z in [-0.0025, 0.2145] m in the tool0 frame, measured from the STL).  Mesh colliders
# are not supported here; the rod is one native capsule of that radius over the rod's whole length (its flat end
# becomes a half sphere: the tip reaches the same z, the rim is rounded off by < r (1 - 1/sqrt 2) = 5.7 mm).
ROD_RADIUS = 0.0194
ROD_CAPSULE = [("tool0", (0.0, 0.0, 0.2145 - ROD_RADIUS), (0.0, 0.0, ROD_RADIUS - 0.0025), ROD_RADIUS)]


def abb_link_boxes():
    """Box stand-ins for the reference's mesh colliders of the arm's links (asset/urdf/abb_rod_description/meshes/
    irb1200_5_90/collision/*.stl): the bounding boxes of their convex hulls, tools/make_link_boxes.py."""
    import json
    with open(asset_path("abb_link_boxes.json")) as f:
        return [tuple(b) for b in json.load(f)["boxes"]]


def abb_link_hulls():
    """The reference's mesh colliders of the arm's links as reduced convex hulls (<= 32 vertices each, link frame):
    tools/make_link_hulls.py."""
    import json
    with open(asset_path("abb_link_hulls.json")) as f:
        return [(name, verts) for name, verts in json.load(f)["hulls"]]


def abb_model(kp=800.0, kd=40.0, link_contacts=False, link_shapes="box"):
    """link_contacts: the arm's links collide with the table, the cube and the goal pad (SURVEY 8f f3;
    ShfModel.link_collide) through box stand-ins for their mesh colliders -- off for the fused / benchmarked scene, whose
    only arm collider is the rod (BASELINE config 5), on for `AbbPushBox` through the gym facade (every shape of an env
    collides there: create_actor(..., group, 0), units.py:68)."""
    if link_shapes not in ("box", "hull"):
        raise ValueError("link_shapes must be 'box' or 'hull'")
    hull = link_contacts and link_shapes == "hull"
    # link_shapes="hull": the links collide as the convex hulls of the reference's collision meshes (reduced to <= 32 vertices,
    # shifu_amd/assets/abb_link_hulls.json) through the convex narrow phase; "box": their bounding boxes (rounds 3-5)
    cm = compile_urdf(asset_path("abb_rod.urdf"), fix_base_link=True, disable_gravity=True,
                      default_dof_drive_mode=_abi.DOF_MODE_POS, extra_spheres=ROD_CAPSULE, link_contacts=link_contacts,
                      extra_boxes=abb_link_boxes() if (link_contacts and not hull) else (),
                      extra_hulls=abb_link_hulls() if hull else (), hull_contacts=hull)
    for d in range(cm.blob.nd):
        cm.blob.kp[d], cm.blob.kd[d] = kp, kd
    return cm


def box_desc(dim, mass, friction, fixed, pos, quat=(0, 0, 0, 1)) -> _abi.ShfBoxDesc:
    b = _abi.ShfBoxDesc()
    b.dim[:] = dim
    b.mass, b.friction, b.fixed = mass, friction, int(fixed)
    b.pos[:] = pos
    b.quat[:] = quat
    return b


def abb_boxes():
    """table (fixed), cube (free, 0.1 kg, mu 0.5), goal pad (fixed) -- task_config.py:13-46."""
    return [box_desc([0.6, 0.6, 0.1], 0.0, 0.5, True, [0, 0, 0.05]),
            box_desc([0.05, 0.05, 0.05], 0.1, 0.5, False, [0, 0, 0.125]),
            box_desc([0.08, 0.08, 0.002], 0.0, 0.5, True, [0, 0, 0.1])]


def abb_task_params(cm, *, dt=0.02, decimation=5, episode_length_s=20.0, extra_substep=True, seed=42,
                    clip_obs=10.0, clip_actions=1.0) -> _abi.ShfAbbTaskParams:
    """ShfAbbTaskParams from the reference's numbers (task_config.py:49-93, a_prior_stage.py:24-73)."""
    import numpy as np
    tp = _abi.ShfAbbTaskParams()
    tp.decimation, tp.extra_substep = decimation, int(extra_substep)
    tp.ee_body = cm.rigid_body_dict["tip0"]
    tp.cube_actor, tp.goal_actor = 2, 3
    tp.clip_actions, tp.clip_obs = clip_actions, clip_obs
    tp.env_dt = dt * decimation
    tp.max_episode_length = float(np.ceil(episode_length_s / (dt * decimation)))
    tp.max_episode_length_s = episode_length_s
    tp.ee_velocity = 0.2
    tp.ik_damping = 0.05
    tp.min_ee_pos[:] = [-0.2, -0.2, 0.11]
    tp.max_ee_pos[:] = [0.2, 0.2, 0.14]
    tp.target_quat[:] = [0.0, 1.0, 0.0, 0.0]
    for d in range(cm.blob.nd):
        tp.default_dof_pos[d] = ABB_DEFAULT_DOF_POS[d]
    defaults = [ABB_BASE_POS + [0, 0, 0, 1], [0, 0, 0.05, 0, 0, 0, 1], [0, 0, 0.125, 0, 0, 0, 1], [0, 0, 0.1, 0, 0, 0, 1]]
    for a, row in enumerate(defaults):
        tp.actor_default[a][:] = row
    tp.cube_lo[:], tp.cube_hi[:] = [-0.1, -0.1, 0.125], [0.1, 0.1, 0.125]
    tp.goal_lo[:], tp.goal_hi[:] = [-0.1, -0.1, 0.1], [0.1, 0.1, 0.1]
    tp.seed = seed
    return tp
