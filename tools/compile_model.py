#!/usr/bin/env python3
"""URDF -> flat `wbc_model` table (JSON fixture) for the batched whole-body-QP path.

Run HERE (build container) only: it reads the reference's URDF assets
(`/root/reference/models/...`, cited in SURVEY.md section 8 rows a1/a1').  The generated
JSON files under `quadruped_drake_amd/models/` are *data* (masses, offsets, inertias) and
are what travels to the GPU box; the URDF text itself is never copied.

What the table holds (canonical leg order [LF, RF, LH, RH], link order [abduct/HAA,
thigh/HFE, shank/KFE]):

  base   : mass, com[3], I_origin[6]            (link frame, about the link origin)
  link   : 4 x 3 x { off[3] (joint origin in the parent link frame),
                     axis[3] (unit, in the link frame; must be +-x/+-y/+-z),
                     mass, com[3], I_origin[6] }
  foot   : 4 x off[3]  (foot frame origin in the shank frame)

`fixed` joints are welded: the child's inertia is lumped into the nearest moving
ancestor (ANYmal `base_inertia`, `*_ADAPTER`), exactly what Drake's parser + MultibodyPlant
do for welded bodies.  I_origin uses the URDF ixy/ixz/iyz as inertia-matrix entries as
written (Drake convention), shifted with the parallel-axis theorem.
Inertia order everywhere: [xx, yy, zz, xy, xz, yz].
"""
import json
import os
import sys
import xml.etree.ElementTree as ET

import numpy as np

FOOT_NAMES = ["LF_FOOT", "RF_FOOT", "LH_FOOT", "RH_FOOT"]


def _vec(s, n=3):
    v = [float(x) for x in s.split()]
    assert len(v) == n, s
    return np.array(v)


def _rpy_to_R(rpy):
    r, p, y = rpy
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def _origin(elem):
    o = elem.find("origin") if elem is not None else None
    xyz = np.zeros(3)
    rpy = np.zeros(3)
    if o is not None:
        if o.get("xyz"):
            xyz = _vec(o.get("xyz"))
        if o.get("rpy"):
            rpy = _vec(o.get("rpy"))
    return xyz, rpy


def _sym(ixx, iyy, izz, ixy, ixz, iyz):
    return np.array([[ixx, ixy, ixz], [ixy, iyy, iyz], [ixz, iyz, izz]])


def _pack_I(I):
    return [I[0, 0], I[1, 1], I[2, 2], I[0, 1], I[0, 2], I[1, 2]]


class Inertia:
    """mass, first moment (m*c) and rotational inertia about the frame origin."""

    def __init__(self, m=0.0, mc=None, I=None):
        self.m = m
        self.mc = np.zeros(3) if mc is None else mc
        self.I = np.zeros((3, 3)) if I is None else I

    @staticmethod
    def from_urdf(link):
        inert = link.find("inertial")
        if inert is None:
            return Inertia()
        m = float(inert.find("mass").get("value"))
        c, rpy = _origin(inert)
        it = inert.find("inertia")
        Ic = _sym(*[float(it.get(k)) for k in ("ixx", "iyy", "izz", "ixy", "ixz", "iyz")])
        R = _rpy_to_R(rpy)
        Ic = R @ Ic @ R.T
        Io = Ic + m * (np.dot(c, c) * np.eye(3) - np.outer(c, c))
        return Inertia(m, m * c, Io)

    def moved(self, p, R):
        """Express in a parent frame: child frame origin at p, orientation R."""
        mc_rot = R @ self.mc
        I_rot = R @ self.I @ R.T
        # shift reference point from child origin to parent origin (offset p)
        c = mc_rot / self.m if self.m > 0 else np.zeros(3)
        Ic = I_rot - self.m * (np.dot(c, c) * np.eye(3) - np.outer(c, c))
        cp = c + p
        Io = Ic + self.m * (np.dot(cp, cp) * np.eye(3) - np.outer(cp, cp))
        return Inertia(self.m, self.m * cp, Io)

    def __add__(self, o):
        return Inertia(self.m + o.m, self.mc + o.mc, self.I + o.I)


def compile_urdf(path, name, body_frame):
    root = ET.parse(path).getroot()
    links = {l.get("name"): l for l in root.findall("link")}
    joints = {}
    children = {}
    parent_of = {}
    for j in root.findall("joint"):
        jn = j.get("name")
        p = j.find("parent").get("link")
        c = j.find("child").get("link")
        xyz, rpy = _origin(j)
        ax = j.find("axis")
        axis = _vec(ax.get("xyz")) if ax is not None else np.array([1.0, 0, 0])
        lim = j.find("limit")
        effort = float(lim.get("effort")) if lim is not None and lim.get("effort") else float("inf")
        joints[jn] = dict(name=jn, type=j.get("type"), parent=p, child=c, xyz=xyz, rpy=rpy,
                          axis=axis, effort=effort)
        children.setdefault(p, []).append(jn)
        parent_of[c] = jn
    roots = [l for l in links if l not in parent_of]
    assert roots == [body_frame], (roots, body_frame)

    def lump(link):
        """Inertia of `link` plus everything welded below it, in `link`'s frame."""
        tot = Inertia.from_urdf(links[link])
        for jn in children.get(link, []):
            j = joints[jn]
            if j["type"] == "fixed":
                tot = tot + lump(j["child"]).moved(j["xyz"], _rpy_to_R(j["rpy"]))
        return tot

    def moving_chain(foot):
        """joints from the base down to the foot frame + foot offset in the last moving link."""
        chain = []
        off = np.zeros(3)
        R_acc = np.eye(3)
        link = foot
        while link in parent_of:
            j = joints[parent_of[link]]
            if j["type"] == "fixed":
                if not chain:
                    # still below the last moving joint: accumulate the welded offset
                    R = _rpy_to_R(j["rpy"])
                    off = j["xyz"] + R @ off
                    R_acc = R @ R_acc
                else:
                    raise ValueError("fixed joint between moving joints is not supported: " + j["name"])
            else:
                assert j["type"] in ("revolute", "continuous"), j
                chain.append(j)
            link = j["parent"]
        assert np.allclose(R_acc, np.eye(3)), "rotated foot frames are not supported"
        return chain[::-1], off

    table = {"name": name, "body_frame": body_frame, "gravity": 9.81,
             "source": os.path.relpath(path, "/root/reference")}
    b = lump(body_frame)
    table["base"] = {"mass": b.m, "com": (b.mc / b.m).tolist(), "I": _pack_I(b.I)}
    legs = []
    act_names = []
    for foot in FOOT_NAMES:
        chain, foot_off = moving_chain(foot)
        assert len(chain) == 3, (foot, [j["name"] for j in chain])
        assert chain[0]["parent"] == body_frame
        leg = {"foot": foot, "foot_off": foot_off.tolist(), "links": []}
        for j in chain:
            assert np.allclose(j["rpy"], 0.0), "rotated joint frames are not supported: " + j["name"]
            a = j["axis"] / np.linalg.norm(j["axis"])
            assert np.isclose(np.abs(a).max(), 1.0), "axis must be axis-aligned: " + j["name"]
            li = lump(j["child"])
            leg["links"].append({"joint": j["name"], "link": j["child"], "off": j["xyz"].tolist(),
                                 "axis": a.tolist(), "mass": li.m, "com": (li.mc / li.m).tolist(),
                                 "I": _pack_I(li.I), "effort": j["effort"]})
            act_names.append(j["name"])
        legs.append(leg)
    table["legs"] = legs
    # actuator order = URDF transmission order (Drake's MakeActuationMatrix column order)
    trans = []
    for t in root.findall("transmission"):
        trans.append(t.find("joint").get("name"))
    table["transmission_order"] = trans
    # act_perm[k] = canonical (leg-major) joint index of actuator k
    table["act_perm"] = [act_names.index(n) for n in trans] if trans else list(range(12))
    table["total_mass"] = b.m + sum(l["mass"] for leg in legs for l in leg["links"])
    return table


def flatten(table):
    """178 doubles in the order the C ABI's `wbc_model` expects (include/wbc.h)."""
    out = [table["base"]["mass"]] + table["base"]["com"] + table["base"]["I"]
    for leg in table["legs"]:
        for l in leg["links"]:
            out += l["off"] + l["axis"] + [l["mass"]] + l["com"] + l["I"]
    for leg in table["legs"]:
        out += leg["foot_off"]
    out.append(table["gravity"])
    return out


def main():
    ref = "/root/reference/models"
    outdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "quadruped_drake_amd", "models")
    os.makedirs(outdir, exist_ok=True)
    jobs = [
        (ref + "/mini_cheetah/mini_cheetah_mesh.urdf", "mini_cheetah", "body"),
        (ref + "/anymal_b_simple_description/urdf/anymal_drake.urdf", "anymal_b", "base"),
    ]
    for path, name, body in jobs:
        t = compile_urdf(path, name, body)
        t["flat"] = flatten(t)
        with open(os.path.join(outdir, name + ".json"), "w") as f:
            json.dump(t, f, indent=1)
        print(name, "total mass", t["total_mass"], "flat len", len(t["flat"]), "act_perm", t["act_perm"])


if __name__ == "__main__":
    sys.exit(main())
