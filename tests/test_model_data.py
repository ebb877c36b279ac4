"""The compiled robot tables (include/solorl_model_data.h, shared by oracle and engine; solorl_amd/models/*.json from the
same generator, tools/compile_model.py) against values typed in by hand from SURVEY.md Appendix A (the reference's
solo_description/robots/{solo,solo12}.urdf) -- a wrong COM sign, axis or joint origin in the generator would otherwise
be invisible to every oracle-vs-engine parity test, since both consume the same table."""
import json
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LEGS = [("FL", 1, 1), ("FR", 1, -1), ("HL", -1, 1), ("HR", -1, -1)]          # name, sx (front +), sy (left +)


def load_json(robot):
    return json.load(open(os.path.join(ROOT, "solorl_amd", "models", robot + ".json")))


def parse_header(symbol):
    """The C initialiser of `symbol` as nested python lists."""
    txt = open(os.path.join(ROOT, "include", "solorl_model_data.h")).read()
    body = txt[txt.index(symbol + " = ") + len(symbol) + 3:]
    body = body[:body.index("};") + 1]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S).replace("{", "[").replace("}", "]")
    return eval(body)


def test_solo12_matches_survey_appendix_a():
    m = load_json("solo12")
    L = {l["name"]: l for l in m["links"]}
    assert m["nlinks"] == 17 and m["ndof"] == 12 and abs(m["total_mass"] - 2.5) < 1e-5
    b = L["base_link"]
    assert b["mass"] == pytest.approx(1.16115091) and b["com"] == [0, 0, 0] and b["friction"] == 1.0
    assert b["inertia_urdf"][:3] == pytest.approx([0.00578574, 0.01938108, 0.02476124])
    dof = 0
    for name, sx, sy in LEGS:
        sh, ul, ll, ft = (L["%s_%s" % (name, k)] for k in ("SHOULDER", "UPPER_LEG", "LOWER_LEG", "FOOT"))
        # HAA: revolute about x at (sx 0.1946, sy 0.0875, 0); shoulder m 0.14853845, COM (-sx 0.078707, sy 0.01, 0), no <contact> tag -> mu 0.5
        assert sh["jtype"] == 0 and sh["axis"] == [1, 0, 0] and sh["jorigin"] == pytest.approx([sx * 0.1946, sy * 0.0875, 0])
        assert sh["mass"] == pytest.approx(0.14853845) and sh["com"] == pytest.approx([-sx * 0.078707, sy * 0.01, 0]) and sh["friction"] == 0.5
        assert sh["inertia_urdf"] == pytest.approx([3.024e-5, 4.1193e-4, 4.1107e-4, sy * 4.671e-5, 0, 0])
        # HFE: revolute about y at (0, sy 0.014, 0); upper leg COM (sy 1.377e-5, sy 0.01935853, -0.078707)
        assert ul["jtype"] == 0 and ul["axis"] == [0, 1, 0] and ul["jorigin"] == pytest.approx([0, sy * 0.014, 0])
        assert ul["mass"] == pytest.approx(0.14853845) and ul["com"] == pytest.approx([sy * 1.377e-5, sy * 0.01935853, -0.078707])
        assert ul["inertia_urdf"] == pytest.approx([4.1107e-4, 4.1193e-4, 3.024e-5, 0, 0, sy * 4.671e-5])
        # KFE: revolute about y at (0, sy 0.03745, -0.16); lower leg COM y = +0.00787644 on ALL four legs (not mirrored)
        assert ll["jtype"] == 0 and ll["axis"] == [0, 1, 0] and ll["jorigin"] == pytest.approx([0, sy * 0.03745, -0.16])
        assert ll["mass"] == pytest.approx(0.03070001) and ll["com"] == pytest.approx([0, 0.00787644, -0.08928215])
        # ANKLE: fixed at (0, sy 0.008, -0.16); foot m 0.00693606, COM (0, 0, 0.00035767)
        assert ft["jtype"] == 1 and ft["jorigin"] == pytest.approx([0, sy * 0.008, -0.16])
        assert ft["mass"] == pytest.approx(0.00693606) and ft["com"] == pytest.approx([0, 0, 0.00035767]) and ft["friction"] == 1.0
        # Bullet joint order = URDF order: per leg [HAA, HFE, KFE] for FL, FR, HL, HR
        assert [m["links"][i]["name"] for i in m["dof_links"][dof:dof + 3]] == [sh["name"], ul["name"], ll["name"]]
        assert ul["parent"] == m["links"].index(sh) and ll["parent"] == m["links"].index(ul) and ft["parent"] == m["links"].index(ll)
        dof += 3


def test_solo8_matches_survey_appendix_a():
    m = load_json("solo8")
    L = {l["name"]: l for l in m["links"]}
    assert m["nlinks"] == 13 and m["ndof"] == 8 and abs(m["total_mass"] - 2.17785) < 1e-5
    assert L["base_link"]["mass"] == pytest.approx(1.43315091)
    for name, sx, sy in LEGS:
        ul, ll, ft = (L["%s_%s" % (name, k)] for k in ("UPPER_LEG", "LOWER_LEG", "FOOT"))
        assert ul["axis"] == [0, 1, 0] and ul["jorigin"] == pytest.approx([sx * 0.19, sy * 0.1046, 0]) and ul["parent"] == 0
        assert ul["com"] == pytest.approx([sy * 1.377e-5, sy * 0.01935853, -0.078707]) and ul["mass"] == pytest.approx(0.14853845)
        assert ll["jorigin"] == pytest.approx([0, sy * 0.03745, -0.16]) and ll["com"] == pytest.approx([0, 0.00787644, -0.08928215])
        assert ft["jtype"] == 1 and ft["jorigin"] == pytest.approx([0, sy * 0.008, -0.16])


@pytest.mark.parametrize("robot,symbol", [("solo8", "SOLORL_MODEL_SOLO8"), ("solo12", "SOLORL_MODEL_SOLO12")])
def test_c_header_equals_the_json_table(robot, symbol):
    m = load_json(robot)
    name, nlinks, ndof, nprims, foot_prim, links, prims = parse_header(symbol)
    assert (nlinks, ndof, nprims) == (m["nlinks"], m["ndof"], len(m["prims"])) and len(links) == nlinks
    for h, j in zip(links, m["links"]):
        parent, jtype, dof, axis, jorigin, com, mass, box, urdf = h
        assert (parent, jtype) == (j["parent"], j["jtype"])
        assert np.allclose(axis, j["axis"]) and np.allclose(jorigin, j["jorigin"], atol=1e-12) and np.allclose(com, j["com"], atol=1e-12)
        assert mass == pytest.approx(j["mass"]) and np.allclose(box, j["inertia_box"], rtol=1e-9) and np.allclose(urdf, j["inertia_urdf"])
    for h, j in zip(prims, m["prims"]):
        link, axis, center, radius, friction, margin, halfw, nring, ring_y, ring_r = h
        assert halfw == pytest.approx(j["halfw"], rel=1e-9) and (halfw > 0) == (axis == 1)      # thick discs: knees and feet (K6, round 3)
        assert nring == len(j["ring_y"]) == len(j["ring_r"]) and np.allclose(ring_y[:nring], j["ring_y"]) and np.allclose(ring_r[:nring], j["ring_r"])
        if nring:           # the feet (round 4): ring 0 is the tread (radius, halfw), the others follow the chamfer inwards
            assert link in m["foot_links"] and nring == 4 and (ring_y[0], ring_r[0]) == (halfw, radius)
            assert all(ring_y[i] < ring_y[i + 1] and ring_r[i] > ring_r[i + 1] for i in range(3))
        assert link == j["link"] and np.allclose(center, j["center"], atol=1e-9) and radius == pytest.approx(j["radius"], abs=1e-7)
        assert friction == j["friction"] and margin == pytest.approx(j["margin"], rel=1e-6)
    assert [prims[p][0] for p in foot_prim] == m["foot_links"]        # the feet sensor reads the foot primitives


def test_foot_profile_is_the_hulls(tmp_path):
    """Hand-typed from the foot mesh (solo_foot: a wheel about the link's y axis, AABB 32 x 16.5 x 32 mm): tread of
    radius 16 mm and half-width 2 mm, chamfered to a face of radius 11.7 mm at +-8.25 mm.  The profile rings reproduce the hull's support
    function; round 3's full-radius thick disc overshot it by 1.1 / 2.1 / 2.8 mm at tilts of 10 / 20 / 45 degrees."""
    for robot in ("solo8", "solo12"):
        p = load_json(robot)["prims"][13]
        ry, rr = np.array(p["ring_y"]), np.array(p["ring_r"])
        assert (ry[0], rr[0]) == pytest.approx((0.002, 0.016), abs=1e-6) and (ry[-1], rr[-1]) == pytest.approx((0.00825, 0.01167), abs=2e-5)
        for tilt, want in ((0, 16.0), (10, 16.10), (20, 15.77), (45, 14.37), (80, 10.15)):      # hull support heights (mm) measured from the mesh
            t = np.radians(tilt)
            assert 1e3 * (rr * np.cos(t) + ry * np.sin(t)).max() == pytest.approx(want, abs=0.06)


def test_box_inertia_rule_k2():
    """URDF mass and inertial origin are kept, the tensor is that of a solid box with the extents
    of the collision AABB (+ 2 x 0.001 collision margin); the survey's margin-free example for the Solo12 base is
    diag(0.00491, 0.01971, 0.02408)."""
    m = load_json("solo12")
    for l in m["links"]:
        ext = np.array(l["aabb_max"]) - np.array(l["aabb_min"]) + 2e-3
        want = l["mass"] / 12.0 * np.array([ext[1] ** 2 + ext[2] ** 2, ext[0] ** 2 + ext[2] ** 2, ext[0] ** 2 + ext[1] ** 2])
        assert np.allclose(l["inertia_box"], want, rtol=1e-6), l["name"]
    assert np.allclose(m["links"][0]["inertia_box"], [0.00491, 0.01971, 0.02408], rtol=0.03)
