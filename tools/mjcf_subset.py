"""A minimal MJCF reader + model compiler for the gym-MuJoCo XML files the reference loads
(`/root/reference/third_party/mujoco_gym_xml_patches/*_envpool.xml`, preferred by
`envpool/mujoco/gym/mujoco_env.h:50-58`).  TEST INFRASTRUCTURE: it exists so that the two hand transcriptions of
those files in this repository (`oracle/mjcpu/models.c`, `envpool_amd/csrc/mj_*_model.h`) can be held against the
XML itself by a third, independent reading (`tests/golden/make_mjcf_golden.py` -> `tests/golden/mjcf_models.json`,
`tests/test_models_vs_xml.py`).

Subset (everything the 13 files use; anything else raises): `<compiler angle settotalmass inertiafromgeom>`,
`<option timestep gravity integrator solver iterations density viscosity>`, one global `<default>` with
`<joint> <geom> <motor>`, nested `<body>` with `pos`, `quat`; `<joint>` free / slide / hinge; `<geom>` plane /
sphere / capsule / cylinder with `size`, `pos`, `quat` | `axisangle` | `fromto`; `<motor>`.  Conventions follow the
MuJoCo XML reference (defaults per attribute; `angle="degree"` is the compiler default and applies to hinge
`range` / `ref` and to `axisangle`; `autolimits`: a joint with a `range` is limited unless `limited="false"`;
body / joint / geom ids in depth-first order of appearance, geoms grouped by body).
"""
from __future__ import annotations

import math
import xml.etree.ElementTree as ET

JNT_TYPE = {"free": 0, "ball": 1, "slide": 2, "hinge": 3}
GEOM_TYPE = {"plane": 0, "hfield": 1, "sphere": 2, "capsule": 3, "ellipsoid": 4, "cylinder": 5, "box": 6}

JOINT_DEFAULTS = {
    "type": "hinge", "pos": "0 0 0", "axis": "0 0 1", "range": "0 0", "armature": "0", "damping": "0",
    "stiffness": "0", "ref": "0", "springref": "0", "margin": "0", "solreflimit": "0.02 1",
    "solimplimit": "0.9 0.95 0.001 0.5 2", "frictionloss": "0",
}
GEOM_DEFAULTS = {
    "type": "sphere", "pos": "0 0 0", "size": "0 0 0", "friction": "1 0.005 0.0001", "density": "1000",
    "margin": "0", "gap": "0", "solref": "0.02 1", "solimp": "0.9 0.95 0.001 0.5 2", "condim": "3",
    "contype": "1", "conaffinity": "1",
}
MOTOR_DEFAULTS = {"gear": "1 0 0 0 0 0", "ctrlrange": "0 0", "ctrllimited": "auto"}
IGNORED_ATTRS = {"name", "rgba", "material", "user", "class"}


def _floats(s: str) -> list[float]:
    return [float(x) for x in s.split()]


def _merge_vec(default: str, *given: str | None) -> list[float]:
    """MuJoCo reads as many numbers as the attribute holds; the rest keep what they had: the built-in default,
    then the <default> element's value, then the element's own (`given`, in that order)."""
    out = _floats(default)
    for s in given:
        if s is not None:
            g = _floats(s)
            out[:len(g)] = g[:len(out)]
    return out


def _quat_mul(a, b):
    return [a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3],
            a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
            a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1],
            a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]]


def _normalize(v):
    n = math.sqrt(sum(x * x for x in v))
    return [x / n for x in v]


def quat_to_mat(q):
    w, x, y, z = q
    return [[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
            [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
            [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]]


def _z_to_quat(vec):
    """rotation taking +z onto `vec` about z x vec (MuJoCo's fromto convention)"""
    v = _normalize(vec)
    axis = [-v[1], v[0], 0.0]  # z x v
    s = math.sqrt(axis[0] ** 2 + axis[1] ** 2)
    if s < 1e-12:
        return [1.0, 0.0, 0.0, 0.0] if v[2] > 0 else [0.0, 1.0, 0.0, 0.0]
    ang = math.atan2(s, v[2])
    axis = [a / s for a in axis]
    return [math.cos(ang / 2)] + [a * math.sin(ang / 2) for a in axis]


class Model(dict):
    pass


def parse(path: str) -> dict:
    root = ET.parse(path).getroot()
    assert root.tag == "mujoco"
    comp = root.find("compiler")
    comp = dict(comp.attrib) if comp is not None else {}
    for k in comp:
        if k not in ("angle", "coordinate", "inertiafromgeom", "settotalmass"):
            raise ValueError(f"compiler attribute {k} outside the subset")
    assert comp.get("coordinate", "local") == "local"
    assert comp.get("inertiafromgeom", "auto") in ("true", "auto")
    degree = comp.get("angle", "degree") == "degree"
    ang = math.pi / 180.0 if degree else 1.0
    opt = root.find("option")
    opt = dict(opt.attrib) if opt is not None else {}
    for k in opt:
        if k not in ("timestep", "gravity", "integrator", "solver", "iterations", "density", "viscosity"):
            raise ValueError(f"option attribute {k} outside the subset")
    dflt = {"joint": {}, "geom": {}, "motor": {}}
    for d in root.findall("default"):
        for child in d:
            if child.tag == "default":
                raise ValueError("default classes are outside the subset")
            if child.tag in dflt:
                dflt[child.tag].update(child.attrib)
    m = {
        "source": path.split("/")[-1],
        "angle": "degree" if degree else "radian",
        "settotalmass": float(comp.get("settotalmass", -1)),
        "timestep": float(opt.get("timestep", 0.002)),
        "gravity": _floats(opt.get("gravity", "0 0 -9.81")),
        "integrator": opt.get("integrator", "Euler"),
        "solver": opt.get("solver", "Newton"),
        "iterations": int(opt.get("iterations", 100)),
        "opt_density": float(opt.get("density", 0)),
        "opt_viscosity": float(opt.get("viscosity", 0)),
        "bodies": [{"name": "world", "parent": 0, "pos": [0.0, 0.0, 0.0], "quat": [1.0, 0.0, 0.0, 0.0]}],
        "joints": [], "geoms": [], "motors": [],
    }
    geoms_by_body: dict[int, list] = {}

    def attr(el, kind, key, defaults):
        if key in el.attrib:
            return el.attrib[key]
        if key in dflt[kind]:
            return dflt[kind][key]
        return defaults[key]

    def read_joint(el, body):
        known = set(JOINT_DEFAULTS) | {"limited"} | IGNORED_ATTRS
        for k in list(el.attrib) + list(dflt["joint"]):
            if k not in known:
                raise ValueError(f"joint attribute {k} outside the subset")
        typ = attr(el, "joint", "type", JOINT_DEFAULTS)
        j = {"name": el.attrib.get("name", ""), "body": body, "type": JNT_TYPE[typ]}
        for k in ("armature", "damping", "stiffness", "margin", "frictionloss"):
            j[k] = float(attr(el, "joint", k, JOINT_DEFAULTS))
        j["pos"] = _floats(attr(el, "joint", "pos", JOINT_DEFAULTS))
        j["axis"] = _normalize(_floats(attr(el, "joint", "axis", JOINT_DEFAULTS)))
        has_range = "range" in el.attrib or "range" in dflt["joint"]
        rng = _floats(attr(el, "joint", "range", JOINT_DEFAULTS))
        ref = float(attr(el, "joint", "ref", JOINT_DEFAULTS))
        sref = float(attr(el, "joint", "springref", JOINT_DEFAULTS))
        if typ == "hinge":  # angles in the compiler's unit
            rng = [r * ang for r in rng]
            ref *= ang
            sref *= ang
        j["range"], j["ref"], j["springref"] = rng, ref, sref
        lim = el.attrib.get("limited", dflt["joint"].get("limited", "auto"))
        j["limited"] = {"true": 1, "false": 0, "auto": int(has_range)}[lim]
        j["solref"] = _merge_vec(JOINT_DEFAULTS["solreflimit"], dflt["joint"].get("solreflimit"),
                                 el.attrib.get("solreflimit"))
        j["solimp"] = _merge_vec(JOINT_DEFAULTS["solimplimit"], dflt["joint"].get("solimplimit"),
                                 el.attrib.get("solimplimit"))
        if typ == "free":  # MuJoCo: a free joint has no limits, armature / damping stay as given
            j["limited"] = 0
        m["joints"].append(j)

    def read_geom(el, body):
        known = set(GEOM_DEFAULTS) | {"quat", "axisangle", "fromto"} | IGNORED_ATTRS
        for k in list(el.attrib) + list(dflt["geom"]):
            if k not in known:
                raise ValueError(f"geom attribute {k} outside the subset")
        typ = attr(el, "geom", "type", GEOM_DEFAULTS)
        g = {"name": el.attrib.get("name", ""), "body": body, "type": GEOM_TYPE[typ]}
        size = _merge_vec("0 0 0", dflt["geom"].get("size"), el.attrib.get("size"))
        pos = _floats(attr(el, "geom", "pos", GEOM_DEFAULTS))
        quat = [1.0, 0.0, 0.0, 0.0]
        if "fromto" in el.attrib:
            ft = _floats(el.attrib["fromto"])
            a, b = ft[:3], ft[3:]
            pos = [(x + y) / 2 for x, y in zip(a, b)]
            d = [y - x for x, y in zip(a, b)]
            size = [size[0], math.sqrt(sum(x * x for x in d)) / 2, 0.0]
            quat = _z_to_quat(d)
        elif "axisangle" in el.attrib:
            aa = _floats(el.attrib["axisangle"])
            ax, th = _normalize(aa[:3]), aa[3] * ang
            quat = [math.cos(th / 2)] + [x * math.sin(th / 2) for x in ax]
        elif "quat" in el.attrib:
            quat = _normalize(_floats(el.attrib["quat"]))
        if typ == "sphere":
            size = [size[0], 0.0, 0.0]
        elif typ in ("capsule", "cylinder"):
            size = [size[0], size[1], 0.0]
        g["size"], g["pos"], g["quat"] = size, pos, quat
        g["zaxis"] = [row[2] for row in quat_to_mat(quat)]
        g["friction"] = _merge_vec(GEOM_DEFAULTS["friction"], dflt["geom"].get("friction"), el.attrib.get("friction"))
        g["solref"] = _merge_vec(GEOM_DEFAULTS["solref"], dflt["geom"].get("solref"), el.attrib.get("solref"))
        g["solimp"] = _merge_vec(GEOM_DEFAULTS["solimp"], dflt["geom"].get("solimp"), el.attrib.get("solimp"))
        for k in ("density", "margin", "gap"):
            g[k] = float(attr(el, "geom", k, GEOM_DEFAULTS))
        for k in ("condim", "contype", "conaffinity"):
            g[k] = int(attr(el, "geom", k, GEOM_DEFAULTS))
        geoms_by_body.setdefault(body, []).append(g)

    def read_body(el, parent):
        for k in el.attrib:
            if k not in ("name", "pos", "quat"):
                raise ValueError(f"body attribute {k} outside the subset")
        bid = len(m["bodies"])
        m["bodies"].append({"name": el.attrib.get("name", ""), "parent": parent,
                            "pos": _floats(el.attrib.get("pos", "0 0 0")),
                            "quat": _normalize(_floats(el.attrib.get("quat", "1 0 0 0")))})
        children(el, bid)

    def children(el, bid):
        for c in el:
            if c.tag == "joint":
                read_joint(c, bid)
            elif c.tag == "freejoint":
                raise ValueError("freejoint outside the subset")
            elif c.tag == "geom":
                read_geom(c, bid)
            elif c.tag == "body":
                read_body(c, bid)
            elif c.tag in ("camera", "light", "site"):
                pass  # no dynamics
            else:
                raise ValueError(f"<{c.tag}> outside the subset")

    children(root.find("worldbody"), 0)
    for b in range(len(m["bodies"])):
        m["geoms"].extend(geoms_by_body.get(b, []))
    # joints are read in document order == body order for a depth-first walk as long as a body's joints precede
    # its child bodies' (true for these files; asserted)
    assert [j["body"] for j in m["joints"]] == sorted(j["body"] for j in m["joints"])
    names = [j["name"] for j in m["joints"]]
    act = root.find("actuator")
    for el in (act if act is not None else []):
        if el.tag != "motor":
            raise ValueError(f"actuator <{el.tag}> outside the subset")
        for k in list(el.attrib) + list(dflt["motor"]):
            if k not in ("name", "joint", "gear", "ctrlrange", "ctrllimited"):
                raise ValueError(f"motor attribute {k} outside the subset")
        cr = _floats(el.attrib.get("ctrlrange", dflt["motor"].get("ctrlrange", MOTOR_DEFAULTS["ctrlrange"])))
        cl = el.attrib.get("ctrllimited", dflt["motor"].get("ctrllimited", "auto"))
        m["motors"].append({
            "joint": names.index(el.attrib["joint"]),
            "gear": _merge_vec(MOTOR_DEFAULTS["gear"], dflt["motor"].get("gear"), el.attrib.get("gear"))[0],
            "ctrlrange": cr,
            "ctrllimited": {"true": 1, "false": 0, "auto": int("ctrlrange" in el.attrib or
                                                              "ctrlrange" in dflt["motor"])}[cl],
        })
    # tendons of the humanoid files: fixed, no limits / springs / actuators on them => no dynamics; anything else raises
    for t in root.findall("tendon"):
        for f in t:
            if f.tag != "fixed" or any(k not in ("name",) for k in f.attrib):
                raise ValueError("tendon with dynamics outside the subset")
    for tag in ("equality", "contact", "sensor", "keyframe"):
        if root.find(tag) is not None:
            raise ValueError(f"<{tag}> outside the subset")
    compile_model(m)
    return m


# ---- model compiler: what MuJoCo's compiler derives from the geoms (inertiafromgeom) -------------------------------
def _geom_mass_inertia(g):
    """mass and the diagonal inertia about the geom's own centre / axes"""
    t, s, rho = g["type"], g["size"], g["density"]
    if t == GEOM_TYPE["plane"]:
        return 0.0, [0.0, 0.0, 0.0]
    if t == GEOM_TYPE["sphere"]:
        mass = rho * 4.0 / 3.0 * math.pi * s[0] ** 3
        i = 0.4 * mass * s[0] ** 2
        return mass, [i, i, i]
    r, h = s[0], 2 * s[1]
    if t == GEOM_TYPE["cylinder"]:
        mass = rho * math.pi * r * r * h
        ix = mass * (3 * r * r + h * h) / 12
        return mass, [ix, ix, mass * r * r / 2]
    if t == GEOM_TYPE["capsule"]:
        mc = rho * math.pi * r * r * h            # cylinder part
        ms = rho * 4.0 / 3.0 * math.pi * r ** 3   # the two hemispheres together
        iz = mc * r * r / 2 + ms * 2 * r * r / 5
        # hemisphere: own-COM inertia 83/320 m r^2, COM 3r/8 beyond the cylinder's end
        half = ms / 2
        ix = (mc * (3 * r * r + h * h) / 12 +
              2 * (half * (83.0 / 320.0) * r * r + half * (h / 2 + 3 * r / 8) ** 2))
        return mc + ms, [ix, ix, iz]
    raise ValueError(f"geom type {t}")


def compile_model(m: dict) -> None:
    nb = len(m["bodies"])
    total = 0.0
    for b in range(nb):
        gs = [g for g in m["geoms"] if g["body"] == b]
        mass, com = 0.0, [0.0, 0.0, 0.0]
        mi = [_geom_mass_inertia(g) for g in gs]
        for g, (gm, _) in zip(gs, mi):
            mass += gm
            com = [c + gm * p for c, p in zip(com, g["pos"])]
        if mass > 0:
            com = [c / mass for c in com]
        inertia = [[0.0] * 3 for _ in range(3)]
        for g, (gm, gi) in zip(gs, mi):
            if gm <= 0:
                continue
            rot = quat_to_mat(g["quat"])
            d = [p - c for p, c in zip(g["pos"], com)]
            d2 = sum(x * x for x in d)
            for r in range(3):
                for c in range(3):
                    inertia[r][c] += (sum(rot[r][k] * gi[k] * rot[c][k] for k in range(3)) +
                                      gm * ((d2 if r == c else 0.0) - d[r] * d[c]))
        m["bodies"][b].update({"mass": mass, "ipos": com, "inertia": inertia})
        total += mass
    if m["settotalmass"] > 0:
        s = m["settotalmass"] / total
        for b in m["bodies"]:
            b["mass"] *= s
            b["inertia"] = [[x * s for x in row] for row in b["inertia"]]
    m["total_mass"] = sum(b["mass"] for b in m["bodies"])
    # qpos0: free joint = the body's pos + quat; slide / hinge = ref
    q0 = []
    for j in m["joints"]:
        if j["type"] == JNT_TYPE["free"]:
            b = m["bodies"][j["body"]]
            q0 += list(b["pos"]) + list(b["quat"])
        else:
            q0.append(j["ref"])
    m["qpos0"] = q0
    m["nq"], m["nv"] = len(q0), sum(6 if j["type"] == 0 else 1 for j in m["joints"])


if __name__ == "__main__":
    import json
    import sys

    print(json.dumps(parse(sys.argv[1]), indent=1))
