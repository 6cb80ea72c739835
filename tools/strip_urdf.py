#!/usr/bin/env python3
"""Reduce a URDF to the physics-only data the model compiler needs.

The reference ships robot descriptions as URDF *data* files
(asset/urdf/a1/urdf/a1.urdf, asset/urdf/abb_rod_description/urdf/abb_rod_isaac.urdf;
SURVEY.md row 20).  The GPU box has no /root/reference, so the model data has to
travel with the repo.  This tool keeps only what the dynamics needs --
links (inertial + primitive collision shapes), joints (origin, axis, limits,
dont_collapse) -- and drops visuals, materials, meshes, gazebo/transmission tags.
Mesh collision shapes are dropped as well (the backend has no mesh collider,
DESIGN.md "out of scope"); shapes the backend substitutes for them are added by
shifu_amd/model.py, not here.

Usage: tools/strip_urdf.py <in.urdf> <out.urdf>
"""
import sys
import xml.etree.ElementTree as ET


def fmt(x):
    return " ".join(repr(float(t)) for t in x.split())


def main(src, dst):
    root = ET.parse(src).getroot()
    out = ET.Element("robot", {"name": root.get("name", "robot")})
    for link in root.findall("link"):
        l = ET.SubElement(out, "link", {"name": link.get("name")})
        ine = link.find("inertial")
        if ine is not None:
            i2 = ET.SubElement(l, "inertial")
            o = ine.find("origin")
            if o is not None:
                ET.SubElement(i2, "origin", {"xyz": fmt(o.get("xyz", "0 0 0")), "rpy": fmt(o.get("rpy", "0 0 0"))})
            ET.SubElement(i2, "mass", {"value": repr(float(ine.find("mass").get("value")))})
            I = ine.find("inertia")
            ET.SubElement(i2, "inertia", {k: repr(float(I.get(k))) for k in ("ixx", "ixy", "ixz", "iyy", "iyz", "izz")})
        for col in link.findall("collision"):
            g = col.find("geometry")
            prim = None
            for tag in ("box", "sphere", "cylinder", "capsule"):
                if g is not None and g.find(tag) is not None:
                    prim = g.find(tag)
            if prim is None:
                continue  # mesh collider: not representable, see module docstring
            c2 = ET.SubElement(l, "collision")
            o = col.find("origin")
            if o is not None:
                ET.SubElement(c2, "origin", {"xyz": fmt(o.get("xyz", "0 0 0")), "rpy": fmt(o.get("rpy", "0 0 0"))})
            g2 = ET.SubElement(c2, "geometry")
            ET.SubElement(g2, prim.tag, dict(prim.attrib))
    for j in root.findall("joint"):
        attrs = {"name": j.get("name"), "type": j.get("type")}
        if j.get("dont_collapse"):
            attrs["dont_collapse"] = j.get("dont_collapse")
        j2 = ET.SubElement(out, "joint", attrs)
        o = j.find("origin")
        if o is not None:
            ET.SubElement(j2, "origin", {"xyz": fmt(o.get("xyz", "0 0 0")), "rpy": fmt(o.get("rpy", "0 0 0"))})
        ET.SubElement(j2, "parent", {"link": j.find("parent").get("link")})
        ET.SubElement(j2, "child", {"link": j.find("child").get("link")})
        a = j.find("axis")
        if a is not None:
            ET.SubElement(j2, "axis", {"xyz": fmt(a.get("xyz"))})
        lim = j.find("limit")
        if lim is not None:
            ET.SubElement(j2, "limit", {k: repr(float(v)) for k, v in lim.attrib.items()})
        d = j.find("dynamics")
        if d is not None:
            ET.SubElement(j2, "dynamics", {k: repr(float(v)) for k, v in d.attrib.items()})
    ET.indent(out, space=" ")
    ET.ElementTree(out).write(dst, encoding="unicode", xml_declaration=False)
    print(f"{src} -> {dst}: {len(out.findall('link'))} links, {len(out.findall('joint'))} joints")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
