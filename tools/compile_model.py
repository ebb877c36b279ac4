#!/usr/bin/env python3
"""Model compiler: URDF + collision meshes -> constant tables for the oracle and the HIP engine.

Reads the robot description the reference loads with ``p.loadURDF(model_urdf, flags=0)``
(reference solo.py:69-73; assets solo_description/robots/{solo,solo12}.urdf and the collision
meshes named therein) and emits

  * ``solorl_amd/models/<robot>.json``  -- human-readable table (committed, small)
  * ``include/solorl_model_data.h``     -- the same numbers as C initialisers, consumed by
                                           ``oracle/solo_oracle.c`` and ``solorl_amd/csrc/*.hip``

Only DATA leaves the reference (masses, origins, extents); no reference code is copied.
The GPU box has no /root/reference, so the outputs are committed and this script is only re-run
when the assets change:  ``python tools/compile_model.py /root/reference/solo_description``.

Modelling rules (see DESIGN.md "Physics model"):
  K2  inertia: Bullet's default for URDF import without URDF_USE_INERTIA_FROM_FILE -- keep URDF
      mass and inertial origin, replace the tensor by that of a solid box with the extents of the
      collision shape's AABB (+2*0.001 collision margin), axes = link axes (all rpy are 0).
      The URDF tensor is emitted too (``inertia_urdf``) for the ``use_urdf_inertia`` switch.
  K6' contact margin: Bullet's *relative* contact breaking threshold
      0.02 * (|aabb_centre - com| + |aabb_half_extent|) per link.
  collision primitives: each link's convex hull is reduced to a few analytic primitives whose
      support point towards the ground is a smooth function of pose:
        disc  (centre c, axis a, radius r) -- support point c + r*proj_perp_a(d)/|.|
        point (r = 0)
      foot  -> one disc (axis y) of the hull's x-z radius;
      knee  -> one disc (axis y, on the upper leg at the KFE axis) of the hull's radius there,
               centred over the y-span of the upper+lower leg knee housings;
      base  -> the 8 hull vertices extremal along the box-diagonal directions + the 4 corners of
               the bottom plate (lowest hull face).
      Shoulder housings (Solo12) lie inside the base AABB and get no primitive.
"""
import json
import os
import struct
import sys
import xml.etree.ElementTree as ET

import numpy as np
from scipy.spatial import ConvexHull

MARGIN = 0.001  # URDF importer collision margin on convex hulls [K6]
LEGS = ["FL", "FR", "HL", "HR"]


def load_mesh(path):
    if path.endswith(".stl"):
        b = open(path, "rb").read()
        n = struct.unpack("<I", b[80:84])[0]
        a = np.frombuffer(b[84:84 + 50 * n], dtype=np.uint8).reshape(n, 50)
        v = a[:, 12:48].copy().view("<f4").reshape(n * 3, 3).astype(np.float64)
    else:
        v = np.array([[float(x) for x in l.split()[1:4]] for l in open(path) if l.startswith("v ")])
    return np.unique(v, axis=0)


def vec(s):
    return [float(x) for x in s.split()]


MAX_RINGS = 4


def ring_profile(y, rho, tol=5e-5):
    """Profile rings (y_i >= 0, r_i) of a body of revolution given its vertices' |axis coordinate| and distance from the axis: the
    support function towards a unit direction with components (s, a) = (|perpendicular|, |axial|) is max_i (r_i s + y_i a).  Ring 0 is
    the outermost one (largest radius: the tread); further rings are added where the support error is largest until it is < tol."""
    P = np.unique(np.round(np.stack([y, rho], axis=1), 6), axis=0)
    th = np.linspace(0.0, 0.5 * np.pi, 721)
    S, A = np.cos(th), np.sin(th)
    full = (P[:, 1:2] * S + P[:, 0:1] * A).max(axis=0)
    r0 = P[:, 1].max()
    first = P[np.abs(P[:, 1] - r0) < 1e-6]
    sel = [first[np.argmax(first[:, 0])]]                 # the tread's edge: largest |y| at the largest radius
    while len(sel) < MAX_RINGS:
        cur = np.max([p[1] * S + p[0] * A for p in sel], axis=0)
        k = int(np.argmax(full - cur))
        if full[k] - cur[k] < tol:
            break
        sel.append(P[np.argmax(P[:, 1] * S[k] + P[:, 0] * A[k])])
    sel = [sel[0]] + sorted(sel[1:], key=lambda p: p[0])
    return [round(float(p[0]), 6) for p in sel], [round(float(p[1]), 6) for p in sel]


def compile_robot(desc_dir, urdf_name):
    root = ET.parse(os.path.join(desc_dir, "robots", urdf_name)).getroot()
    links_xml = {l.get("name"): l for l in root.findall("link")}
    joints_xml = root.findall("joint")
    child_joint = {j.find("child").get("link"): j for j in joints_xml}

    # Bullet link order = URDF joint order (reference solo.py:95-106 walks getJointInfo in order)
    order = ["base_link"] + [j.find("child").get("link") for j in joints_xml]
    index = {n: i for i, n in enumerate(order)}
    links = []
    hulls = {}
    for name in order:
        lx = links_xml[name]
        inert = lx.find("inertial")
        com = vec(inert.find("origin").get("xyz"))
        assert vec(inert.find("origin").get("rpy")) == [0, 0, 0]
        mass = float(inert.find("mass").get("value"))
        it = inert.find("inertia").attrib
        col = lx.find("collision")
        col_org = np.array(vec(col.find("origin").get("xyz")))
        assert vec(col.find("origin").get("rpy")) == [0, 0, 0]
        mesh_rel = col.find("geometry/mesh").get("filename")
        mesh = os.path.normpath(os.path.join(desc_dir, "robots", mesh_rel))
        v = load_mesh(mesh) + col_org            # link frame
        hv = v[ConvexHull(v).vertices]
        hulls[name] = hv
        lo, hi = v.min(0), v.max(0)
        ext = (hi - lo) + 2 * MARGIN
        box = [mass / 12.0 * (ext[1] ** 2 + ext[2] ** 2),
               mass / 12.0 * (ext[0] ** 2 + ext[2] ** 2),
               mass / 12.0 * (ext[0] ** 2 + ext[1] ** 2)]
        centre = 0.5 * (lo + hi) - np.array(com)
        disc = float(np.linalg.norm(centre) + np.linalg.norm(0.5 * ext))
        ct = lx.find("contact")
        fric = float(ct.find("lateral_friction").get("value")) if ct is not None else 0.5
        if name == "base_link":
            parent, jtype, axis, jorg, jname = -1, -1, [0, 0, 0], [0, 0, 0], ""
        else:
            j = child_joint[name]
            parent = index[j.find("parent").get("link")]
            jtype = 0 if j.get("type") == "revolute" else 1
            axis = vec(j.find("axis").get("xyz")) if j.find("axis") is not None else [0, 0, 0]
            jorg = vec(j.find("origin").get("xyz"))
            assert vec(j.find("origin").get("rpy")) == [0, 0, 0]
            jname = j.get("name")
            if jtype == 0:
                lim = j.find("limit")
                assert float(lim.get("lower")) == -10 and float(lim.get("upper")) == 10
        links.append(dict(name=name, joint=jname, parent=parent, jtype=jtype, axis=axis, jorigin=jorg,
                          com=com, mass=mass, inertia_box=box,
                          inertia_urdf=[float(it[k]) for k in ("ixx", "iyy", "izz", "ixy", "ixz", "iyz")],
                          friction=fric, margin=0.02 * disc,
                          aabb_min=lo.tolist(), aabb_max=hi.tolist()))

    # ---- collision primitives -------------------------------------------------------------
    prims = []
    hv = hulls["base_link"]
    ext = 0.5 * (hv.max(0) - hv.min(0))
    for sx in (1, -1):
        for sy in (1, -1):
            for sz in (-1, 1):
                d = np.array([sx, sy, sz]) / ext
                p = hv[np.argmax(hv @ d)]
                prims.append(dict(link=0, kind="point", center=p.tolist(), axis=-1, radius=0.0))
    belly = hv[hv[:, 2] < hv[:, 2].min() + 5e-4]           # bottom plate: its 4 corners
    for sx in (1, -1):
        for sy in (1, -1):
            p = belly[np.argmax(belly[:, :2] @ (np.array([sx, sy]) / ext[:2]))]
            prims.append(dict(link=0, kind="point", center=p.tolist(), axis=-1, radius=0.0))
    for leg in LEGS:
        up, low, foot = index[leg + "_UPPER_LEG"], index[leg + "_LOWER_LEG"], index[leg + "_FOOT"]
        kfe = np.array(links[low]["jorigin"])              # KFE origin in the upper-leg frame
        hu = hulls[leg + "_UPPER_LEG"]
        m = hu[:, 2] < kfe[2] + 0.03
        r_knee = float(np.hypot(hu[m, 0], hu[m, 2] - kfe[2]).max())
        hl = hulls[leg + "_LOWER_LEG"] + kfe               # lower-leg hull at q=0, upper-leg frame
        ml = hl[:, 2] > kfe[2] - 0.03
        ys = np.concatenate([hu[m, 1], hl[ml, 1]])
        yc = 0.5 * (ys.min() + ys.max())
        # halfw: half-thickness of the disc along its axis (round 3, K6): a leg lying on its SIDE rests on the housing's face, half a
        # thickness below the disc's plane -- the zero-thickness discs of rounds 1-2 let lying legs sink 8-19 mm deeper than their hulls
        prims.append(dict(link=up, kind="disc", center=[0.0, float(yc), float(kfe[2])], axis=1, radius=r_knee,
                          halfw=float(0.5 * (ys.max() - ys.min()))))
        # Foot (round 4, K6): the foot hull is a body of revolution about the link's y axis -- a wheel with a 4 mm wide tread of radius
        # 16 mm chamfered down to 11.7 mm at its faces (+-8.25 mm) -- and its support function is that of its PROFILE, the rings
        # (|y - yc|, rho).  Rings are picked greedily from the profile's convex hull until the support error is < 0.05 mm (<= 4).  radius /
        # halfw describe ring 0, the tread.  (Round 3 took the full 16 mm radius out to the faces: a foot tilted by 10-45 degrees reached
        # 1-3 mm too far and its contact point sat 8 mm off centre from a tilt of 6 degrees on, where the hull's stays on the tread up to
        # 32 degrees -- measured with a trained walking policy: tests/test_parity_gpu3.py)
        hf = hulls[leg + "_FOOT"]
        yc = float(0.5 * (hf[:, 1].min() + hf[:, 1].max()))
        ring_y, ring_r = ring_profile(np.abs(hf[:, 1] - yc), np.hypot(hf[:, 0], hf[:, 2]))
        prims.append(dict(link=foot, kind="disc", center=[0.0, yc, 0.0], axis=1, radius=ring_r[0], halfw=ring_y[0], ring_y=ring_y, ring_r=ring_r))
    # Solo12 shoulder housings (K6, measured in tests/test_oracle_k6.py: -3 % terminations without them): one disc about the
    # link's x axis (= the HAA axis direction).  The hull's support function in the directions perpendicular to x,
    # h(theta) = max_v (v_y cos theta + v_z sin theta), is fitted by a circle yc cos + zc sin + r in the least-squares
    # sense; the disc sits at the x of the widest section.  Appended AFTER the 20 primitives both robots share, so the
    # knee / foot indices (12 + 2 leg, 13 + 2 leg) stay what every consumer assumes.
    for leg in LEGS:
        name = leg + "_SHOULDER"
        if name not in hulls:
            continue
        hs = hulls[name]
        th = np.linspace(0, 2 * np.pi, 360, endpoint=False)
        h = (hs[:, 1:2] * np.cos(th) + hs[:, 2:3] * np.sin(th)).max(axis=0)
        A = np.stack([np.cos(th), np.sin(th), np.ones_like(th)], axis=1)
        (yc, zc, r), *_ = np.linalg.lstsq(A, h, rcond=None)
        rad = np.hypot(hs[:, 1] - yc, hs[:, 2] - zc)
        xw = float(hs[np.argmax(rad), 0])
        clean = lambda v: 0.0 if abs(v) < 1e-9 else round(float(v), 9)     # (the four legs' fits then mirror exactly)
        prims.append(dict(link=index[name], kind="disc", center=[clean(xw), clean(yc), clean(zc)], axis=0, radius=clean(r)))
    for p in prims:
        p.setdefault("halfw", 0.0)
        p.setdefault("ring_y", []); p.setdefault("ring_r", [])
        p["friction"] = links[p["link"]]["friction"] * 1.0   # x plane.urdf lateral friction 1.0 [K6]
        p["margin"] = links[p["link"]]["margin"]

    dof_links = [i for i, l in enumerate(links) if l["jtype"] == 0]
    foot_links = [i for i, l in enumerate(links) if l["jtype"] == 1]
    return dict(name=urdf_name.replace(".urdf", "").replace("solo", "solo8") if urdf_name == "solo.urdf"
                else urdf_name.replace(".urdf", ""),
                urdf=urdf_name, nlinks=len(links), ndof=len(dof_links), dof_links=dof_links,
                foot_links=foot_links, total_mass=sum(l["mass"] for l in links),
                links=links, prims=prims)


def c_array(vals, fmt="%.17g"):
    return "{" + ", ".join(fmt % v for v in vals) + "}"


def emit_header(models, path):
    o = []
    o.append("/* GENERATED by tools/compile_model.py from the reference's solo_description assets")
    o.append(" * (robots/solo.urdf, robots/solo12.urdf + collision meshes). DATA ONLY. Do not edit. */")
    o.append("#ifndef SOLORL_MODEL_DATA_H")
    o.append("#define SOLORL_MODEL_DATA_H")
    o.append("#ifdef __cplusplus")
    o.append("#define SOLORL_MODEL_CONST static constexpr")
    o.append("#else")
    o.append("#define SOLORL_MODEL_CONST static const")
    o.append("#endif")
    o.append("#define SOLORL_MAX_LINKS 17")
    o.append("#define SOLORL_MAX_DOF 12")
    o.append("#define SOLORL_MAX_PRIMS 24")
    o.append("typedef struct solorl_link_data {")
    o.append("  int parent; int jtype; /* -1 base, 0 revolute, 1 fixed */ int dof; /* index into q or -1 */")
    o.append("  double axis[3]; double jorigin[3]; double com[3]; double mass;")
    o.append("  double inertia_box[3]; double inertia_urdf[6]; /* ixx iyy izz ixy ixz iyz */")
    o.append("} solorl_link_data;")
    o.append("typedef struct solorl_prim_data {")
    o.append("  int link; int axis; /* -1 point, 0/1/2 disc axis */ double center[3]; double radius;")
    o.append("  double friction; double margin; double halfw; /* half-thickness of a disc along its axis */")
    o.append("  /* nring > 1: a body of revolution about the disc axis given by its profile rings (ring_y[i] >= 0 from the centre plane, ring_r[i]);")
    o.append("   * support point towards a direction with |perpendicular| s and |axial| a components: the ring maximising ring_r s + ring_y a.  Ring 0")
    o.append("   * (radius, halfw) is the tread. */")
    o.append("  int nring; double ring_y[4]; double ring_r[4];")
    o.append("} solorl_prim_data;")
    o.append("typedef struct solorl_model_data {")
    o.append("  const char* name; int nlinks; int ndof; int nprims; int foot_prim[4];")
    o.append("  solorl_link_data links[SOLORL_MAX_LINKS]; solorl_prim_data prims[SOLORL_MAX_PRIMS];")
    o.append("} solorl_model_data;")
    for m in models:
        dof_of = {l: i for i, l in enumerate(m["dof_links"])}
        foot_prims = [i for i, p in enumerate(m["prims"]) if p["link"] in m["foot_links"]]
        assert len(foot_prims) == 4 and len(m["prims"]) <= 24
        o.append("SOLORL_MODEL_CONST solorl_model_data SOLORL_MODEL_%s = {" % m["name"].upper())
        o.append('  "%s", %d, %d, %d, %s,' % (m["name"], m["nlinks"], m["ndof"], len(m["prims"]),
                                            c_array(foot_prims, "%d")))
        o.append("  {")
        for i, l in enumerate(m["links"]):
            o.append("    {%d, %d, %d, %s, %s, %s, %.17g, %s, %s}, /* %s */" % (
                l["parent"], l["jtype"], dof_of.get(i, -1), c_array(l["axis"]), c_array(l["jorigin"]),
                c_array(l["com"]), l["mass"], c_array(l["inertia_box"]), c_array(l["inertia_urdf"]), l["name"]))
        o.append("  },")
        o.append("  {")
        for p in m["prims"]:
            ry = list(p["ring_y"]) + [0.0] * (4 - len(p["ring_y"])); rr = list(p["ring_r"]) + [0.0] * (4 - len(p["ring_r"]))
            o.append("    {%d, %d, %s, %.17g, %.17g, %.17g, %.17g, %d, %s, %s}," % (
                p["link"], p["axis"], c_array(p["center"]), p["radius"], p["friction"], p["margin"], p["halfw"], len(p["ring_y"]),
                c_array(ry), c_array(rr)))
        o.append("  }")
        o.append("};")
    o.append("#endif")
    open(path, "w").write("\n".join(o) + "\n")


def main():
    desc = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/solo_description"
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    models = [compile_robot(desc, "solo.urdf"), compile_robot(desc, "solo12.urdf")]
    for m in models:
        with open(os.path.join(repo, "solorl_amd", "models", m["name"] + ".json"), "w") as f:
            json.dump(m, f, indent=1)
        print(m["name"], "links", m["nlinks"], "dof", m["ndof"], "prims", len(m["prims"]),
              "mass %.5f" % m["total_mass"])
    emit_header(models, os.path.join(repo, "include", "solorl_model_data.h"))


if __name__ == "__main__":
    main()
