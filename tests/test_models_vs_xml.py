"""The compiled MuJoCo MODELS against the MJCF files the reference loads.

The reference holds one MuJoCo input in its tree: `third_party/mujoco_gym_xml_patches/*_envpool.xml`
(`envpool/mujoco/gym/mujoco_env.h:50-58` prefers them; e.g. `half_cheetah_envpool.xml:52-110`,
`ant_envpool.xml:19-37`).  This repository transcribes those files by hand twice -- `oracle/mjcpu/models.c` (the
oracle) and `envpool_amd/csrc/mj_*_model.h` (the product's build-time model compilers -> `gen_mj_consts` tables) --
and both agreeing proves only that the author copied consistently.  Here a THIRD reading, `tools/mjcf_subset.py`
(an `xml.etree` MJCF-subset parser: defaults inheritance, angle units, `fromto` -> pos / half length / axis,
`settotalmass`, `density`, inertiafromgeom), is the judge:

* `tests/golden/mjcf_models.json` = its output for all 13 files (`tests/golden/make_mjcf_golden.py`), replayed on
  any box; where `/root/reference` exists the JSON is re-derived from the XML and must equal the committed file;
* `test_oracle_models_match_the_xml`: EVERY raw attribute of `mjc_model` (option, body tree, joint type / axis /
  range / limited / armature / damping / stiffness / ref / margin / solref / solimp, geom type / size / pos / axis /
  friction / solref / solimp / margin / contype / conaffinity / condim / density, actuator joint / gear / ctrlrange)
  and the compiled masses, centres of mass, inertias and qpos0, for the 13 models (`mjcpu_model_dump`);
* `test_product_tables_match_the_xml`: the same for the 13 `constexpr` tables the kernels are compiled with,
  through the layout each kernel family uses (planar: x-z numbers per body and capsule end spheres; Ant / Pusher:
  welded bodies merged into their parents; chains: link-frame numbers in the plane of motion; Humanoid: generic tree).
"""
import ctypes
import json
import math
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
GOLDEN = os.path.join(ROOT, "tests", "golden", "mjcf_models.json")
XML_DIR = "/root/reference/third_party/mujoco_gym_xml_patches"
STEMS = ["half_cheetah", "ant", "hopper", "walker2d", "walker2d_v5", "swimmer", "reacher", "inverted_pendulum",
         "inverted_double_pendulum", "pusher", "pusher_v5", "humanoid", "humanoidstandup"]
INTEGRATOR = {"Euler": 0, "RK4": 1}
SOLVER = {"Newton": 0, "PGS": 1}
PLANE, SPHERE, CAPSULE, CYLINDER = 0, 2, 3, 5
FREE, SLIDE, HINGE = 0, 2, 3


@pytest.fixture(scope="module")
def golden():
    with open(GOLDEN) as f:
        return json.load(f)


class Diff:
    """collects mismatches so that one run lists every transcription slip of a model"""

    def __init__(self, what):
        self.what, self.bad, self.count = what, [], 0

    def eq(self, name, want, got, rtol=1e-12, atol=1e-13):
        self.count += 1
        a, b = np.asarray(want, float), np.asarray(got, float)
        if a.shape != b.shape or not np.allclose(a, b, rtol=rtol, atol=atol):
            self.bad.append(f"{name}: xml {np.asarray(want).tolist()} != {np.asarray(got).tolist()}")

    def done(self, at_least):
        assert not self.bad, f"{self.what}: {len(self.bad)} mismatches vs the XML\n  " + "\n  ".join(self.bad[:30])
        assert self.count >= at_least, (self.what, self.count)  # the comparison really covered the model


def test_golden_is_what_the_xml_says():
    """the committed JSON == a fresh reading of the reference's XML (only where the reference tree exists)"""
    if not os.path.isdir(XML_DIR):
        pytest.skip("no /root/reference on this box: the committed fixture is replayed as is")
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_mjcf_golden

    fresh = json.loads(json.dumps(make_mjcf_golden.build(), sort_keys=True))
    with open(GOLDEN) as f:
        assert fresh == json.load(f)
    # and the file names are the ones the reference's loader resolves (mujoco_env.h:50-58: "<name>_envpool.xml")
    have = sorted(f[:-len("_envpool.xml")] for f in os.listdir(XML_DIR) if f.endswith("_envpool.xml"))
    assert have == sorted(STEMS)


# --------------------------------------------------------------------------------------------------------------------
# oracle/mjcpu/models.c
# --------------------------------------------------------------------------------------------------------------------
def _oracle_dump(stem):
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "port"], check=True)
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "_build", "liboracle.so"))
    lib.mjcpu_model_dump.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int]
    buf = ctypes.create_string_buffer(1 << 19)
    n = lib.mjcpu_model_dump(stem.encode(), buf, 1 << 19)
    assert 0 < n < (1 << 19), (stem, n)
    return json.loads(buf.value)


def _zaxis(q):
    w, x, y, z = q
    return [2 * (x * z + w * y), 2 * (y * z - w * x), 1 - 2 * (x * x + y * y)]


@pytest.mark.parametrize("stem", STEMS)
def test_oracle_models_match_the_xml(golden, stem):
    x, o = golden[stem], _oracle_dump(stem)
    d = Diff(f"oracle/mjcpu/models.c [{stem}]")
    d.eq("nq", x["nq"], o["nq"])
    d.eq("nv", x["nv"], o["nv"])
    d.eq("nbody", len(x["bodies"]), o["nbody"])
    d.eq("njnt", len(x["joints"]), o["njnt"])
    d.eq("nu", len(x["motors"]), o["nu"])
    d.eq("timestep", x["timestep"], o["timestep"])
    d.eq("gravity", x["gravity"], o["gravity"])
    d.eq("integrator", INTEGRATOR[x["integrator"]], o["integrator"])
    d.eq("solver", SOLVER[x["solver"]], o["solver"])
    d.eq("iterations", x["iterations"], o["iterations"])
    d.eq("option density", x["opt_density"], o["opt_density"])
    d.eq("option viscosity", x["opt_viscosity"], o["opt_viscosity"])
    d.eq("settotalmass", max(x["settotalmass"], 0), max(o["settotalmass"], 0))
    for b, xb in enumerate(x["bodies"]):
        d.eq(f"body {b} parent", xb["parent"], o["body_parent"][b])
        d.eq(f"body {b} pos", xb["pos"], o["body_pos"][b])
        d.eq(f"body {b} quat", xb["quat"], o["body_quat"][b])
        if b == 0:
            continue  # the world body's mass is not used (and the oracle drops the world's non-colliding geoms)
        d.eq(f"body {b} mass", xb["mass"], o["body_mass"][b])
        d.eq(f"body {b} ipos", xb["ipos"], o["body_ipos"][b])
        d.eq(f"body {b} inertia", np.ravel(xb["inertia"]), o["body_inertia"][b], rtol=1e-11, atol=1e-15)
    for j, xj in enumerate(x["joints"]):
        for k, ok in (("type", "jnt_type"), ("body", "jnt_body"), ("limited", "jnt_limited"), ("pos", "jnt_pos"),
                      ("stiffness", "jnt_stiffness"), ("ref", "jnt_ref")):
            d.eq(f"joint {j} {k}", xj[k], o[ok][j])
        assert xj["springref"] == 0 and xj["frictionloss"] == 0  # not modelled by the oracle: must not be needed
        if xj["type"] != FREE:
            d.eq(f"joint {j} axis", xj["axis"], o["jnt_axis"][j])
        if xj["limited"]:
            d.eq(f"joint {j} range", xj["range"], o["jnt_range"][j])
            d.eq(f"joint {j} margin", xj["margin"], o["jnt_margin"][j])
            d.eq(f"joint {j} solreflimit", xj["solref"], o["jnt_solref"][j])
            d.eq(f"joint {j} solimplimit", xj["solimp"], o["jnt_solimp"][j])
        adr = o["jnt_dofadr"][j]
        for i in range(6 if xj["type"] == FREE else 1):
            d.eq(f"joint {j} armature", xj["armature"], o["dof_armature"][adr + i])
            d.eq(f"joint {j} damping", xj["damping"], o["dof_damping"][adr + i])
    xg = x["geoms"]
    if len(xg) != o["ngeom"]:  # Reacher: the arena's visual-only world geoms (contype = conaffinity = 0) are left out
        xg = [g for g in xg if not (g["body"] == 0 and g["contype"] == 0 and g["conaffinity"] == 0)]
    d.eq("ngeom", len(xg), o["ngeom"])
    for g, gx in enumerate(xg):
        for k, ok in (("type", "geom_type"), ("body", "geom_body"), ("contype", "geom_contype"),
                      ("conaffinity", "geom_conaffinity"), ("condim", "geom_condim"), ("pos", "geom_pos"),
                      ("friction", "geom_friction"), ("margin", "geom_margin"), ("density", "geom_density"),
                      ("solref", "geom_solref"), ("solimp", "geom_solimp")):
            d.eq(f"geom {g} {k}", gx[k], o[ok][g])
        assert gx["gap"] == 0
        if gx["type"] != PLANE:
            d.eq(f"geom {g} size", gx["size"], o["geom_size"][g])
        if gx["type"] in (CAPSULE, CYLINDER):  # the axis INCLUDING its sign: the +axis end sphere collides first
            d.eq(f"geom {g} axis", gx["zaxis"], _zaxis(o["geom_quat"][g]))
    for u, xu in enumerate(x["motors"]):
        d.eq(f"motor {u} joint", xu["joint"], o["act_jnt"][u])
        d.eq(f"motor {u} gear", xu["gear"], o["act_gear"][u])
        d.eq(f"motor {u} ctrlrange", xu["ctrlrange"], o["act_ctrlrange"][u])
        assert xu["ctrllimited"] == 1  # the oracle clamps every ctrl
    d.eq("qpos0", x["qpos0"], o["qpos0"])
    d.done(at_least=90)


# --------------------------------------------------------------------------------------------------------------------
# the product's tables (envpool_amd/csrc/gen_mj_consts.cpp over mj_*_model.h)
# --------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def tables(tmp_path_factory):
    """runs the build-time generator exactly as the Makefile does and parses its `/*field*/ value,` output"""
    csrc = os.path.join(ROOT, "envpool_amd", "csrc")
    exe = str(tmp_path_factory.mktemp("gen") / "gen_mj_consts")
    subprocess.run(["g++", "-O1", "-std=c++17", os.path.join(csrc, "gen_mj_consts.cpp"), "-o", exe], check=True)
    out = {}
    for which in ("cheetah", "walker", "ant", "humanoid", "pusher", "chain"):
        txt = subprocess.run([exe, which], check=True, capture_output=True, text=True).stdout
        for m in re.finditer(r"constexpr\s+[^=]*?\s(\w+)\s*=\s*\{(.*?)\n\};", txt, re.S):
            fields = {}
            for f in re.finditer(r"/\*(\w+)\*/\s*(.*?),\s*\n", m.group(2) + "\n"):
                v = f.group(2).strip().rstrip(",").replace("{", "[").replace("}", "]")
                fields[f.group(1)] = json.loads(re.sub(r"(\d)u\b", r"\1", v))
            out[m.group(1)] = fields
    assert len(out) == 13, sorted(out)
    return out


def _solref_kb(solref, solimp, timestep):
    """MuJoCo's reference acceleration constants for positive solref (timeconst, dampratio): the time constant is
    floored at 2 x timestep (refsafe), K = 1 / (dmax^2 tc^2 dr^2), B = 2 / (dmax tc)"""
    tc, dr = max(solref[0], 2 * timestep), solref[1]
    dmax = solimp[1]
    return 1.0 / (dmax * dmax * tc * tc * dr * dr), 2.0 / (dmax * tc)


def _clamp_imp(v):
    return min(max(v, 0.0001), 0.9999)  # mjMINIMP / mjMAXIMP


def _ends(g):
    """capsule end-sphere centres in the body frame, +axis end first (the order mjc_PlaneCapsule emits contacts in)"""
    p, z, h = np.array(g["pos"]), np.array(g["zaxis"]), g["size"][1]
    return p + h * z, p - h * z


def _merge_welded(x, keep):
    """bodies without joints are welded to their parents: MuJoCo keeps them as bodies, the specialised kernels fold
    their mass into the moving ancestor.  `keep` = the XML body ids that stay; returns {id: (mass, com, inertia 3x3
    about com, [(geom, offset of its body in the kept body's frame)])}.  All welded bodies here have identity quat."""
    out = {b: {"mass": 0.0, "mcom": np.zeros(3), "parts": [], "geoms": []} for b in keep}
    for b, xb in enumerate(x["bodies"]):
        if b == 0:
            continue
        off, a = np.zeros(3), b
        while a not in keep:
            assert np.allclose(x["bodies"][a]["quat"], [1, 0, 0, 0])
            off += np.array(x["bodies"][a]["pos"])
            a = x["bodies"][a]["parent"]
        out[a]["parts"].append((xb["mass"], off + np.array(xb["ipos"]), np.array(xb["inertia"])))
        out[a]["geoms"] += [(g, off) for g in x["geoms"] if g["body"] == b]
    for v in out.values():
        v["mass"] = sum(p[0] for p in v["parts"])
        com = sum(p[0] * p[1] for p in v["parts"]) / v["mass"]
        inertia = np.zeros((3, 3))
        for m, c, i in v["parts"]:
            r = c - com
            inertia += i + m * (r @ r * np.eye(3) - np.outer(r, r))
        v["com"], v["inertia"] = com, inertia
    return out


def _sym6(i):  # xx yy zz xy xz yz
    return [i[0][0], i[1][1], i[2][2], i[0][1], i[0][2], i[1][2]]


PLANAR = {"half_cheetah": ("kCheetahModelConst", 7, 6), "walker2d": ("kWalkerModelConst", 7, 6),
          "walker2d_v5": ("kWalkerV5ModelConst", 7, 6), "hopper": ("kHopperModelConst", 4, 3)}


@pytest.mark.parametrize("stem", sorted(PLANAR))
def test_product_planar_tables_match_the_xml(golden, tables, stem):
    """CheetahModel<double> (mj_cheetah.hip.h:223): planar robots in the x-z plane, 2 slides + 1 hinge on the torso"""
    name, nb, nu = PLANAR[stem]
    x, t = golden[stem], tables[name]
    d = Diff(f"{name} (mj_cheetah_model.h)")
    assert [j["type"] for j in x["joints"][:3]] == [SLIDE, SLIDE, HINGE]
    d.eq("root axes", [x["joints"][0]["axis"], x["joints"][1]["axis"], x["joints"][2]["axis"]],
         [[1, 0, 0], [0, 0, 1], [0, 1, 0]])
    floor = x["geoms"][0]
    assert floor["type"] == PLANE
    # Two exact changes of variables of the model compiler (mj_cheetah_model.h:170-182), applied to the XML numbers
    # here: (1) a hinge at `pos` != 0 in its body frame (Walker2d / Hopper leg and foot joints): every body frame is
    # re-centred on its hinge anchor; (2) a hinge about -y: the kernel integrates q' = -q (range mirrored, gear
    # negated; the step kernel flips those qpos / qvel entries on load and store).
    anchor = [np.zeros(3)] + [np.array(x["joints"][3 + u]["pos"]) for u in range(nu)]  # per planar body
    parent = [x["bodies"][b + 1]["parent"] - 1 for b in range(nb)]
    assert all(x["joints"][r]["body"] == 1 for r in range(3))
    assert x["joints"][2]["pos"] == [0, 0, 0]  # rooty at the torso origin (the `pos` of a slide has no meaning)
    for b in range(nb):
        xb = x["bodies"][b + 1]
        assert xb["pos"][1] == 0 and np.allclose(xb["quat"], [1, 0, 0, 0])
        pos = np.array(xb["pos"]) + anchor[b] - (anchor[parent[b]] if b else 0)
        if b == 0:  # qpos[0], qpos[1] ARE the world x, z of the torso origin: body_pos - ref of the two slides
            pos = pos - np.array([x["joints"][0]["ref"], 0, x["joints"][1]["ref"]])
        d.eq(f"body {b} lx lz", [pos[0], pos[2]], [t["lx"][b], t["lz"][b]], atol=1e-15)
        d.eq(f"body {b} mass", xb["mass"], t["mass"][b])
        d.eq(f"body {b} iyy", xb["inertia"][1][1], t["iyy"][b], rtol=1e-11)
        com = np.array(xb["ipos"]) - anchor[b]
        d.eq(f"body {b} cx cz", [com[0], com[2]], [t["cx"][b], t["cz"][b]], atol=1e-15)
        gs = [g for g in x["geoms"] if g["body"] == b + 1]
        assert all(g["type"] == CAPSULE for g in gs) and len(gs) == (2 if (b == 0 and stem == "half_cheetah") else 1)
        for k, g in enumerate(gs):
            slot = 2 * k if b == 0 else 2 * b + 2  # BodyEnd0 (mj_cheetah.hip.h:249); the torso's head: slots 2, 3
            for w, e in enumerate(_ends(g)):
                e = e - anchor[b]
                assert abs(e[1]) < 1e-15
                d.eq(f"body {b} capsule {k} end {w}", [e[0], e[2]], [t["ex"][slot + w], t["ez"][slot + w]], atol=1e-15)
                d.eq(f"body {b} capsule {k} radius", g["size"][0], t["er"][slot + w])
            # floor contact: friction = max of the two geoms', margin = max, solref / solimp identical on both geoms
            d.eq(f"body {b} friction", max(floor["friction"][0], g["friction"][0]), t["bmu"][b])
            d.eq(f"body {b} margin", max(floor["margin"], g["margin"]), t["con_margin"])
            d.eq(f"body {b} solref", g["solref"], floor["solref"])
            d.eq(f"body {b} solimp", g["solimp"], floor["solimp"])
            assert max(g["condim"], floor["condim"]) == 3  # a pair takes the larger condim: pyramidal friction rows
    K, B = _solref_kb(floor["solref"], floor["solimp"], x["timestep"])
    d.eq("contact K B", [K, B], [t["con_K"], t["con_B"]])
    d.eq("contact solimp", [_clamp_imp(floor["solimp"][0]), _clamp_imp(floor["solimp"][1]), floor["solimp"][2]],
         [t["con_d0"], t["con_dmax"], t["con_width"]])
    assert floor["solimp"][3:] == [0.5, 2]  # the kernels' Impedance() is the midpoint 0.5 / power 2 form
    for u in range(nu):
        xj = x["joints"][3 + u]
        sgn = xj["axis"][1]
        assert xj["type"] == HINGE and xj["limited"] == 1 and xj["axis"] == [0, sgn, 0] and abs(sgn) == 1
        assert xj["body"] == u + 2 and xj["ref"] == 0 and xj["margin"] == 0
        d.eq(f"hinge {u} stiffness damping armature", [xj["stiffness"], xj["damping"], xj["armature"]],
             [t["stiff"][u], t["damp"][u], t["arm"][u]])
        d.eq(f"hinge {u} range", sorted(sgn * r for r in xj["range"]), [t["lo"][u], t["hi"][u]])
        K, B = _solref_kb(xj["solref"], xj["solimp"], x["timestep"])
        d.eq(f"hinge {u} limit K B", [K, B], [t["lim_K"], t["lim_B"]])
        d.eq(f"hinge {u} limit solimp", [_clamp_imp(xj["solimp"][0]), _clamp_imp(xj["solimp"][1]), xj["solimp"][2]],
             [t["lim_d0"], t["lim_dmax"], t["lim_width"]])
    for r in range(3):
        xj = x["joints"][r]
        assert (xj["stiffness"], xj["damping"], xj["armature"], xj["limited"]) == (0, 0, 0, 0)
    assert len(x["motors"]) == nu
    for u, mo in enumerate(x["motors"]):
        d.eq(f"motor {u} gear", mo["gear"] * x["joints"][mo["joint"]]["axis"][1], t["gear"][mo["joint"] - 3])
        assert mo["ctrlrange"] == [-1, 1] and mo["ctrllimited"] == 1  # the kernels clamp ctrl to [-1, 1]
    d.eq("total mass", x["total_mass"], t["total_mass"])
    d.eq("timestep", x["timestep"], t["timestep"])
    d.eq("gravity", x["gravity"], [0, 0, -t["gravity"]])
    # Hopper / Walker2d geoms collide with each other (contype = conaffinity = 1): the Hopper kernels carry its three
    # non-adjacent body pairs; HalfCheetah's body geoms have conaffinity 0 (floor only)
    body_geoms = [g for g in x["geoms"][1:]]
    selfcollide = all(g["contype"] & g["conaffinity"] for g in body_geoms)
    if stem == "hopper":
        assert selfcollide and t["n_pairs"] == 3
        # MuJoCo filters parent-child pairs: what is left are torso-leg, torso-foot, thigh-foot, the order of
        # PairBody1 / PairBody2 (mj_cheetah.hip.h:246-247)
        pairs = [(a, b) for a in range(nb) for b in range(a + 1, nb) if parent[b] != a]
        assert pairs == [(0, 2), (0, 3), (1, 3)]
        assert all(g["condim"] == 1 for g in body_geoms)  # body-body contacts: one frictionless row each
    if stem == "half_cheetah":
        assert not any(g["conaffinity"] for g in body_geoms) and t["n_pairs"] == 0
    d.done(at_least=120 if nb == 7 else 60)


def test_product_ant_table_matches_the_xml(golden, tables):
    """AntModel<double> (mj_ant.hip.h:78): torso (+ the four jointless leg stubs) and 4 x (hip body, ankle body)"""
    x, t = golden["ant"], tables["kAntModelConst"]
    d = Diff("kAntModelConst (mj_ant_model.h)")
    keep = [1] + [b for leg in range(4) for b in (3 + 3 * leg, 4 + 3 * leg)]
    merged = _merge_welded(x, keep)
    for i, b in enumerate(keep):
        d.eq(f"body {i} mass", merged[b]["mass"], t["mass"][i])
        d.eq(f"body {i} com", merged[b]["com"], t["com"][i], atol=1e-15)
        d.eq(f"body {i} inertia", _sym6(merged[b]["inertia"]), t["inertia"][i], rtol=1e-11, atol=1e-15)
    floor = x["geoms"][0]
    torso_sphere = x["geoms"][1]
    assert floor["type"] == PLANE and torso_sphere["type"] == SPHERE
    d.eq("torso sphere", torso_sphere["pos"] + [torso_sphere["size"][0]], t["sph"][0] + [t["sph_r"][0]])
    assert x["joints"][0]["type"] == FREE
    for leg in range(4):
        stub, hip, ankle = x["bodies"][2 + 3 * leg], x["bodies"][3 + 3 * leg], x["bodies"][4 + 3 * leg]
        d.eq(f"leg {leg} aux_pos", np.add(stub["pos"], hip["pos"]), t["aux_pos"][leg])
        d.eq(f"leg {leg} foot_pos", ankle["pos"], t["foot_pos"][leg])
        jh, ja = x["joints"][1 + 2 * leg], x["joints"][2 + 2 * leg]
        assert (jh["body"], ja["body"]) == (3 + 3 * leg, 4 + 3 * leg) and jh["pos"] == ja["pos"] == [0, 0, 0]
        d.eq(f"leg {leg} hip axis", jh["axis"], [0, 0, 1])
        d.eq(f"leg {leg} ankle axis", ja["axis"], t["ankle_axis"][leg])
        for k, xj in enumerate((jh, ja)):
            u = 2 * leg + k
            assert xj["limited"] == 1 and xj["stiffness"] == 0 and xj["margin"] == 0 and xj["ref"] == 0
            d.eq(f"joint {u} range", xj["range"], [t["lo"][u], t["hi"][u]])
            d.eq(f"joint {u} damping armature", [xj["damping"], xj["armature"]], [t["damp"][u], t["arm"][u]])
            K, B = _solref_kb(xj["solref"], xj["solimp"], x["timestep"])
            d.eq(f"joint {u} limit K B (the contact's constants are used for limits too)", [K, B],
                 [t["con_K"], t["con_B"]])
            d.eq(f"joint {u} limit solimp", xj["solimp"][:3], [t["imp_d0"], t["imp_dmax"], t["imp_width"]])
        for k in range(3):  # aux capsule (on the stub, torso frame), leg capsule (hip frame), ankle capsule
            g = x["geoms"][2 + 3 * leg + k]
            assert g["type"] == CAPSULE and g["body"] == 2 + 3 * leg + k
            off = np.array(stub["pos"]) if k == 0 else np.zeros(3)
            for w, e in enumerate(_ends(g)):
                s = 1 + 6 * leg + 2 * k + w
                d.eq(f"leg {leg} capsule {k} end {w}", e + off, t["sph"][s], atol=1e-15)
                d.eq(f"leg {leg} capsule {k} radius", g["size"][0], t["sph_r"][s])
    for g in x["geoms"][1:]:
        d.eq("friction", max(floor["friction"][0], g["friction"][0]), t["mu"])
        d.eq("margin", max(floor["margin"], g["margin"]), t["margin"])
        assert g["condim"] == 3 and g["conaffinity"] == 0 and g["solref"] == floor["solref"]
        K, B = _solref_kb(g["solref"], g["solimp"], x["timestep"])
        d.eq("contact K B", [K, B], [t["con_K"], t["con_B"]])
        d.eq("contact solimp", g["solimp"][:3], [t["imp_d0"], t["imp_dmax"], t["imp_width"]])
    for u, mo in enumerate(x["motors"]):
        # ant_envpool.xml lists the motors hip_4, ankle_4, hip_1, ankle_1, ...: the kernel's `gear` is one number
        d.eq(f"motor {u} gear", mo["gear"], t["gear"])
        assert mo["ctrlrange"] == [-1, 1]
    d.eq("total mass", x["total_mass"], t["total_mass"])
    d.eq("timestep", x["timestep"], t["timestep"])
    d.eq("gravity", x["gravity"], [0, 0, -t["gravity"]])
    assert x["integrator"] == "RK4"
    d.done(at_least=150)


@pytest.mark.parametrize("stem,name", [("humanoid", "kHumanoidModelConst"),
                                       ("humanoidstandup", "kHumanoidStandupModelConst")])
def test_product_humanoid_tables_match_the_xml(golden, tables, stem, name):
    """tree::TreeModel (mj_tree.hip.h): the generic tree form, raw attributes one to one"""
    x, t = golden[stem], tables[name]
    d = Diff(f"{name} (mj_tree_model.h)")
    nb, nj, ng, nu = len(x["bodies"]), len(x["joints"]), len(x["geoms"]), len(x["motors"])
    d.eq("counts", [nb, nj, x["nq"], x["nv"], ng, nu], [t[k] for k in ("nbody", "njnt", "nq", "nv", "ngeom", "nu")])
    d.eq("iterations", x["iterations"], t["iterations"])
    assert x["solver"] == "PGS" and x["integrator"] == "RK4"
    d.eq("timestep", x["timestep"], t["timestep"])
    d.eq("gravity", x["gravity"], [0, 0, -t["gravity"]])
    d.eq("total mass", x["total_mass"], t["total_mass"])
    for b, xb in enumerate(x["bodies"]):
        d.eq(f"body {b} parent", xb["parent"], t["body_parent"][b])
        d.eq(f"body {b} pos", xb["pos"], t["body_pos"][b])
        d.eq(f"body {b} quat", xb["quat"], t["body_quat"][b])
        if b:
            d.eq(f"body {b} mass", xb["mass"], t["body_mass"][b])
            d.eq(f"body {b} ipos", xb["ipos"], t["body_ipos"][b], atol=1e-15)
            d.eq(f"body {b} inertia", _sym6(xb["inertia"]), t["body_inertia"][b], rtol=1e-11, atol=1e-15)
    dof = 0
    for j, xj in enumerate(x["joints"]):
        d.eq(f"joint {j} type body limited", [xj["type"], xj["body"], xj["limited"]],
             [t["jnt_type"][j], t["jnt_body"][j], t["jnt_limited"][j]])
        d.eq(f"joint {j} dof address", dof, t["jnt_dadr"][j])
        d.eq(f"joint {j} pos", xj["pos"], t["jnt_pos"][j])
        d.eq(f"joint {j} stiffness", xj["stiffness"], t["jnt_stiff"][j])
        assert xj["ref"] == 0 and xj["margin"] == 0
        if xj["type"] != FREE:
            d.eq(f"joint {j} axis", xj["axis"], t["jnt_axis"][j])
        if xj["limited"]:
            d.eq(f"joint {j} range", xj["range"], [t["jnt_lo"][j], t["jnt_hi"][j]])
            K, B = _solref_kb(xj["solref"], xj["solimp"], x["timestep"])
            d.eq(f"joint {j} limit K B", [K, B], [t["sol_K"], t["sol_B"]])
            d.eq(f"joint {j} limit solimp", xj["solimp"][:3], [t["sol_d0"], t["sol_dmax"], t["sol_width"]])
        for i in range(6 if xj["type"] == FREE else 1):
            d.eq(f"joint {j} armature damping", [xj["armature"], xj["damping"]],
                 [t["dof_arm"][dof + i], t["dof_damp"][dof + i]])
            d.eq(f"dof {dof + i} body", xj["body"], t["dof_body"][dof + i])
        dof += 6 if xj["type"] == FREE else 1
    d.eq("limited joints", [j for j, xj in enumerate(x["joints"]) if xj["limited"]], t["limit_jnt"][:t["nlimit"]])
    floor = x["geoms"][0]
    assert floor["type"] == PLANE and floor["condim"] == 3
    for g, gx in enumerate(x["geoms"]):
        d.eq(f"geom {g} type body", [gx["type"], gx["body"]], [t["geom_type"][g], t["geom_body"][g]])
        if g == 0:
            continue
        d.eq(f"geom {g} pos", gx["pos"], t["geom_pos"][g], atol=1e-15)
        d.eq(f"geom {g} radius", gx["size"][0], t["geom_rad"][g])
        if gx["type"] == CAPSULE:
            d.eq(f"geom {g} axis", gx["zaxis"], t["geom_axis"][g], atol=1e-15)
            d.eq(f"geom {g} half length", gx["size"][1], t["geom_hl"][g])
        # floor pair: condim = max (3), friction = max, margin = max; geom-geom pairs: condim 1 (frictionless)
        d.eq(f"geom {g} floor friction", max(floor["friction"][0], gx["friction"][0]), t["floor_mu"])
        d.eq(f"geom {g} margin", max(floor["margin"], gx["margin"]), t["margin"])
        assert gx["condim"] == 1 and gx["contype"] == 1 and gx["conaffinity"] == 1
        K, B = _solref_kb(gx["solref"], gx["solimp"], x["timestep"])
        d.eq(f"geom {g} contact K B", [K, B], [t["sol_K"], t["sol_B"]])
        d.eq(f"geom {g} solimp", gx["solimp"][:3], [t["sol_d0"], t["sol_dmax"], t["sol_width"]])
    jd = np.cumsum([0] + [6 if xj["type"] == FREE else 1 for xj in x["joints"]])
    for u, mo in enumerate(x["motors"]):
        d.eq(f"motor {u} dof", jd[mo["joint"]], t["act_dof"][u])
        d.eq(f"motor {u} gear", mo["gear"], t["act_gear"][u])
        d.eq(f"motor {u} ctrlrange", mo["ctrlrange"], [t["ctrl_lo"], t["ctrl_hi"]])
    d.eq("qpos0", x["qpos0"], t["qpos0"][:x["nq"]])
    d.done(at_least=400)


# chains in a plane (mj_pendulum.hip.h): name, links, base (2 = cart slider, 0 = fixed, 3 = planar free base),
# XML body of each link, in-plane coordinates (u, w) of a body-frame vector
CHAINS = {
    "inverted_pendulum": ("kInvertedPendulumModelConst", [2], "cart", lambda v: (v[0], v[2])),
    "inverted_double_pendulum": ("kInvertedDoublePendulumModelConst", [2, 3], "cart", lambda v: (v[0], v[2])),
    "reacher": ("kReacherModelConst", [1, 2], "fixed", lambda v: (v[0], -v[1])),   # z := -y
    "swimmer": ("kSwimmerModelConst", [1, 2, 3], "free", lambda v: (v[0], -v[1])),
}


@pytest.mark.parametrize("stem", sorted(CHAINS))
def test_product_chain_tables_match_the_xml(golden, tables, stem):
    name, links, base, plane = CHAINS[stem]
    x, t = golden[stem], tables[name]
    d = Diff(f"{name} (mj_pendulum_model.h)")
    merged = _merge_welded(x, [b for b in range(1, len(x["bodies"])) if any(j["body"] == b for j in x["joints"])])
    if base == "cart":
        d.eq("cart mass", x["bodies"][1]["mass"], t["cart_mass"])
    normal = 1 if base == "cart" else 2  # inertia about the plane's normal: y for the x-z plane, z for x-y
    for i, b in enumerate(links):
        mb = merged[b]
        d.eq(f"link {i} mass", mb["mass"], t["mass"][i])
        d.eq(f"link {i} inertia about the plane normal", mb["inertia"][normal][normal], t["iyy"][i], rtol=1e-11)
        d.eq(f"link {i} com", plane(mb["com"]), [t["cx"][i], t["cz"][i]], atol=1e-15)
        if i + 1 < len(links):  # where the next link hangs
            d.eq(f"link {i} -> next", plane(x["bodies"][links[i + 1]]["pos"]), [t["lx"][i], t["lz"][i]])
        elif stem == "reacher":  # the fingertip body's origin
            d.eq("fingertip", plane(x["bodies"][3]["pos"]), [t["lx"][i], t["lz"][i]])
        elif base == "cart":  # far end of the last pole
            g = [g for g in x["geoms"] if g["body"] == b][0]
            d.eq("pole tip", plane(_ends(g)[0]), [t["lx"][i], t["lz"][i]], atol=1e-15)
    # cart: slider + one hinge per pole; fixed: one hinge per link; free: x, y, rotation of link 0 + a hinge per further link
    nv = {"cart": 1 + len(links), "fixed": len(links), "free": 3 + len(links) - 1}[base]
    for j in range(nv):
        xj = x["joints"][j]
        d.eq(f"dof {j} damping armature", [xj["damping"], xj["armature"]], [t["damp"][j], t["arm"][j]])
        d.eq(f"dof {j} limited", xj["limited"], t["limited"][j])
        assert xj["stiffness"] == 0 and xj["ref"] == 0
        if xj["limited"]:
            d.eq(f"dof {j} range", xj["range"], [t["lo"][j], t["hi"][j]])
            d.eq(f"dof {j} margin", xj["margin"], t["margin"][j])
            K, B = _solref_kb(xj["solref"], xj["solimp"], x["timestep"])
            d.eq(f"dof {j} limit K B", [K, B], [t["lim_K"], t["lim_B"]])
            d.eq(f"dof {j} limit solimp", xj["solimp"][:3], [t["lim_d0"], t["lim_dmax"], t["lim_width"]])
    gear = [0.0] * nv
    for mo in x["motors"]:
        gear[mo["joint"]] = mo["gear"]
        d.eq("ctrlrange", mo["ctrlrange"], [t["ctrl_lo"], t["ctrl_hi"]])
    d.eq("gear", gear, t["gear"][:nv])
    gu, gw = plane(x["gravity"])
    d.eq("in-plane gravity", [gu, gw], [t["grav_x"], t["grav_z"]])
    d.eq("medium", [x["opt_density"], x["opt_viscosity"]], [t["fluid_density"], t["fluid_viscosity"]])
    d.eq("timestep", x["timestep"], t["timestep"])
    assert x["integrator"] == "RK4"
    d.done(at_least=20)


@pytest.mark.parametrize("stem,name", [("pusher", "kPusherModelConst"), ("pusher_v5", "kPusherV5ModelConst")])
def test_product_pusher_tables_match_the_xml(golden, tables, stem, name):
    """PusherModel<double> (mj_pusher.hip.h): 7 hinge links (13 bodies, the jointless ones folded in) + the object"""
    x, t = golden[stem], tables[name]
    d = Diff(f"{name} (mj_pusher_model.h)")
    hinge_bodies = [j["body"] for j in x["joints"][:7]]
    assert hinge_bodies == [1, 2, 3, 5, 6, 8, 9] and all(j["type"] == HINGE for j in x["joints"][:7])
    merged = _merge_welded(x, hinge_bodies + [11, 12])
    for i, b in enumerate(hinge_bodies):
        off, a = np.zeros(3), b
        stop = hinge_bodies[i - 1] if i else 0
        while a != stop:  # offset of this link's frame in the previous link's
            off += np.array(x["bodies"][a]["pos"])
            a = x["bodies"][a]["parent"]
        d.eq(f"link {i} offset", off, t["off"][i])
        d.eq(f"link {i} mass", merged[b]["mass"], t["mass"][i])
        d.eq(f"link {i} com", merged[b]["com"], t["com"][i], atol=1e-15)
        d.eq(f"link {i} inertia", _sym6(merged[b]["inertia"]), t["inertia"][i], rtol=1e-11, atol=1e-15)
        xj = x["joints"][i]
        assert xj["limited"] == 1 and xj["pos"] == [0, 0, 0] and xj["stiffness"] == 0 and xj["margin"] == 0
        d.eq(f"joint {i} range", xj["range"], [t["lo"][i], t["hi"][i]])
        d.eq(f"joint {i} damping armature", [xj["damping"], xj["armature"]], [t["damp"][i], t["arm"][i]])
        K, B = _solref_kb(xj["solref"], xj["solimp"], x["timestep"])
        d.eq(f"joint {i} limit K B", [K, B], [t["sol_K"], t["sol_B"]])
        d.eq(f"joint {i} limit solimp", xj["solimp"][:3], [t["imp_d0"], t["imp_dmax"], t["imp_width"]])
    for k, j in enumerate((7, 8)):  # the object's slides (y, then x)
        xj = x["joints"][j]
        assert xj["type"] == SLIDE and xj["body"] == 11
        d.eq(f"object slide {k} damping armature", [xj["damping"], xj["armature"]], [t["damp"][7 + k], t["arm"][7 + k]])
    table = x["geoms"][0]
    assert table["type"] == PLANE
    d.eq("table height", table["pos"][2], t["table_z"])
    caps = [g for g in x["geoms"] if g["body"] == 9 and g["type"] == CAPSULE and g["contype"] == 1]
    assert len(caps) == 3
    for k, g in enumerate(caps):  # fromto: p0 = from, p1 = to
        hi, lo = _ends(g)
        d.eq(f"wrist capsule {k} from", lo, t["cap_p0"][k], atol=1e-15)
        d.eq(f"wrist capsule {k} to", hi, t["cap_p1"][k], atol=1e-15)
        d.eq(f"wrist capsule {k} radius", g["size"][0], t["cap_r"])
        d.eq(f"wrist capsule {k} margin", max(g["margin"], table["margin"]), t["margin"])
        assert g["condim"] == 1
    d.eq("object pos", x["bodies"][11]["pos"], t["obj_pos"])
    d.eq("object mass", merged[11]["mass"], t["obj_mass"])
    cyl = [g for g in x["geoms"] if g["body"] == 11 and g["type"] == CYLINDER][0]
    d.eq("object cylinder", cyl["size"][:2], [t["cyl_r"], t["cyl_h"]])
    d.eq("goal pos", x["bodies"][12]["pos"], t["goal_pos"])
    for mo in x["motors"]:
        d.eq("ctrlrange", mo["ctrlrange"], [t["ctrl_lo"], t["ctrl_hi"]])
        assert mo["gear"] == 1
    d.eq("timestep", x["timestep"], t["timestep"])
    assert x["gravity"] == [0, 0, 0] and x["integrator"] == "Euler" and x["iterations"] == 20
    d.done(at_least=80)
