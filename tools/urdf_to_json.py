#!/usr/bin/env python3
"""Extract the numeric model parameters of the reference's URDF robots into small JSON
fixtures (dojo.jl_amd/host/dojo_amd/data/*.json).

The GPU box has no /root/reference, so the parameters the mechanism builders need
(link masses / inertias / inertial poses, joint types / origins / axes / damping, first
visual capsule radius) are extracted once, here, and committed.  Only numbers and names
are kept -- no XML, no meshes.  Semantics (COM frames, fixed-joint merging, limits,
contacts) live in dojo_amd/mechanisms.py, following src/mechanism/urdf.jl (SURVEY.md
Appendix A).

usage: python tools/urdf_to_json.py            (run in the build container)
"""
import json, os, sys
import xml.etree.ElementTree as ET

REF = "/root/reference/DojoEnvironments/src/mechanisms"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dojo.jl_amd", "host", "dojo_amd", "data")
ROBOTS = {
    "ant": "ant/dependencies/ant.urdf",
    "quadruped": "quadruped/dependencies/gazebo_a1.urdf",
    "atlas": "atlas/dependencies/atlas_simple.urdf",
}

def fvec(s, default):
    return [float(x) for x in (s if s is not None else default).split()]

def pose(el):
    if el is None:
        return [0.0, 0.0, 0.0], [0.0, 0.0, 0.0]
    return fvec(el.get("xyz"), "0 0 0"), fvec(el.get("rpy"), "0 0 0")

def convert(path):
    root = ET.parse(path).getroot()          # ElementTree drops XML comments
    links, joints = [], []
    for l in root.findall("link"):
        inert = l.find("inertial")
        if inert is None:
            xyz, rpy, m, J = [0.0] * 3, [0.0] * 3, 0.0, [0.0] * 6
        else:
            xyz, rpy = pose(inert.find("origin"))
            mel = inert.find("mass")
            m = float(mel.get("value", "0")) if mel is not None else 0.0
            I = inert.find("inertia")
            J = [float(I.get(k, "0")) for k in ("ixx", "ixy", "ixz", "iyy", "iyz", "izz")] if I is not None else [0.0] * 6
        # radius of the first visual geometry if it is a capsule/sphere/cylinder (ant contact radii,
        # DojoEnvironments/src/mechanisms/ant/mechanism.jl:56,77: body.shape.shapes[1].rh[1])
        radius = None
        vis = l.find("visual")
        if vis is not None and vis.find("geometry") is not None:
            for g in vis.find("geometry"):
                if g.tag in ("capsule", "sphere", "cylinder"):
                    radius = float(g.get("radius", "0.5"))
                break
        links.append(dict(name=l.get("name"), mass=m, inertia=J, xyz=xyz, rpy=rpy, radius=radius))
    for j in root.findall("joint"):
        xyz, rpy = pose(j.find("origin"))
        ax = j.find("axis")
        axis = fvec(ax.get("xyz") if ax is not None else None, "1 0 0")
        dyn = j.find("dynamics")
        damping = float(dyn.get("damping", "0")) if dyn is not None else 0.0
        joints.append(dict(name=j.get("name"), type=j.get("type"), parent=j.find("parent").get("link"),
                           child=j.find("child").get("link"), xyz=xyz, rpy=rpy, axis=axis, damping=damping))
    return dict(links=links, joints=joints)

if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    for name, rel in ROBOTS.items():
        d = convert(os.path.join(REF, rel))
        d["source"] = "DojoEnvironments/src/mechanisms/" + rel
        with open(os.path.join(OUT, name + ".json"), "w") as f:
            json.dump(d, f, indent=1)
        print(name, len(d["links"]), "links", len(d["joints"]), "joints")
