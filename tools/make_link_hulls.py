#!/usr/bin/env python3
"""Reduced convex hulls of mesh colliders whose files do not ship with this repo.

The reference's ABB arm collides through STL meshes (asset/urdf/abb_rod_description/meshes/irb1200_5_90/collision/*.stl,
asset/urdf/abb_rod_description/urdf/abb_rod_isaac.urdf:38-113; [EXT] PhysX: their convex hulls, cooked to <= 64 vertices).
The vendored physics-only URDF (shifu_amd/assets/abb_rod.urdf, tools/strip_urdf.py) has no meshes, so each mesh collider is
reduced HERE -- where /root/reference exists -- to the vertices of a convex polytope within the narrow phase's limits
(shifu_amd/model.py: reduce_hull, <= 32 vertices: support points of the true hull along a fixed set of directions), in the
link's frame, and only those numbers travel: shifu_amd/assets/abb_link_hulls.json, a list of [link, [[x, y, z], ...]] that
compile_urdf(extra_hulls=..., hull_contacts=True) turns into ShfHull records.  A derived data reduction like
abb_link_boxes.json (tools/make_link_boxes.py).

    python tools/make_link_hulls.py /root/reference/asset/urdf/abb_rod_description/urdf/abb_rod_isaac.urdf \
        shifu_amd/assets/abb_link_hulls.json
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(src, dst, skip=("tool0",)):
    from shifu_amd.model import parse_urdf, reduce_hull
    links, _ = parse_urdf(src, meshes="error")
    out = []
    for name, l in links.items():
        if name in skip:            # the rod: a native capsule (abb_task.ROD_CAPSULE)
            continue
        for s in l.shapes:
            if s.kind != "hull":
                continue
            h = reduce_hull(s.verts)
            v = s.pos + h["verts"] @ s.rot.T          # link frame
            out.append([name, [[round(float(x), 6) for x in q] for q in v]])
    with open(dst, "w") as f:
        json.dump({"source": os.path.relpath(src, "/root/reference") if src.startswith("/root/reference") else os.path.basename(src),
                   "what": "vertices of the reduced convex hulls (<= 32 vertices: support points of the mesh's hull along a fixed direction "
                           "set) of the reference's <mesh> colliders, link frame: [link, [[x, y, z], ...]]", "hulls": out}, f)
    print(f"{dst}: {len(out)} hulls, {sum(len(h[1]) for h in out)} vertices")


if __name__ == "__main__":
    main(*sys.argv[1:3])
