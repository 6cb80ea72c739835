#!/usr/bin/env python3
"""Box stand-ins for mesh colliders whose files do not ship with this repo.

The reference's ABB arm collides through STL meshes (asset/urdf/abb_rod_description/meshes/irb1200_5_90/collision/
link_1.stl ... link_6.stl, base_link.stl; [EXT] PhysX: their convex hulls).  The vendored physics-only URDF
(shifu_amd/assets/abb_rod.urdf, tools/strip_urdf.py) has no meshes, so for link contacts each mesh collider
is reduced HERE -- where /root/reference exists -- to the bounding box of its convex hull in the collision frame, and only
those numbers travel: shifu_amd/assets/abb_link_boxes.json, a list of [link, xyz, rpy, size] that
compile_urdf(extra_boxes=...) adds as <box> shapes.  A derived data reduction like the stripped URDF itself.

    python tools/make_link_boxes.py /root/reference/asset/urdf/abb_rod_description/urdf/abb_rod_isaac.urdf \
        shifu_amd/assets/abb_link_boxes.json
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def rpy_of(R):
    """Inverse of model._rpy (R = Rz(y) Ry(p) Rx(r))."""
    p = -np.arcsin(np.clip(R[2, 0], -1.0, 1.0))
    r = np.arctan2(R[2, 1], R[2, 2])
    y = np.arctan2(R[1, 0], R[0, 0])
    return [float(r), float(p), float(y)]


def main(src, dst, skip=("tool0",)):
    from shifu_amd.model import parse_urdf
    links, _ = parse_urdf(src, meshes="error")
    out = []
    for name, l in links.items():
        if name in skip:            # the rod: a native capsule (abb_task.ROD_CAPSULE), not a box
            continue
        for s in l.shapes:
            if s.kind != "hull":
                continue
            lo, hi = s.verts.min(0), s.verts.max(0)
            c = s.pos + s.rot @ (0.5 * (lo + hi))
            out.append([name, [round(float(v), 6) for v in c], [round(v, 6) for v in rpy_of(s.rot)],
                        [round(float(v), 6) for v in (hi - lo)]])
    with open(dst, "w") as f:
        json.dump({"source": os.path.relpath(src, "/root/reference") if src.startswith("/root/reference") else os.path.basename(src),
                   "what": "bounding boxes of the convex hulls of the reference's <mesh> colliders, collision frame of each link: "
                           "[link, xyz, rpy, size]", "boxes": out}, f, indent=1)
    print(f"{dst}: {len(out)} boxes")
    for b in out:
        print(" ", b)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
