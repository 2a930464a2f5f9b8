"""Which source files a step kernel is compiled from, and a hash over them + the Makefile.
`tools/make_pmc_json.py` stamps every `profiles/pmc.json` entry with the hash of the build it profiled;
`bench.py` recomputes it and refuses to price a changed kernel with old PMC counts ("stale": true).
The hash is over the CODE: comments and whitespace are stripped first, so that a comment-only edit does not
turn the headline `roofline` into the HBM fallback (round 4: commit ebe05a1 did exactly that)."""
import hashlib
import os
import re

_TOKEN = re.compile(r"//[^\n]*|/\*.*?\*/|\"(?:\\.|[^\"\\])*\"|'(?:\\.|[^'\\])*'", re.S)


def strip_comments(text: str) -> str:
    """C / C++ / Makefile-rule text without comments, every whitespace run collapsed to one blank (string and
    character literals are kept as they are)."""
    def keep(m):
        t = m.group(0)
        return " " if t.startswith("/") else t
    return " ".join(_TOKEN.sub(keep, text).split())

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "envpool_amd", "csrc")
COMMON = ["device_common.hip.h", "gen_mj_consts.cpp"]  # + the Makefile lines that decide how the TU is compiled
SOURCES = {
    "PlanarLgStepKernel": ["mujoco_planar_lg.hip", "mj_planar_lg.hip.h", "mj_cheetah.hip.h", "mj_cheetah_model.h",
                           "mujoco_planar_common.h"],
    "CheetahStepKernel": ["mujoco_gym.hip", "mj_cheetah.hip.h", "mj_cheetah_model.h", "mujoco_planar_common.h"],
    "AntStepKernel": ["mujoco_ant.hip", "mj_ant4.hip.h", "mj_ant.hip.h", "mj_ant_model.h", "mj_quad.hip.h",
                      "mj_cheetah.hip.h"],
    "Humanoid4StepKernel": ["mujoco_humanoid4.hip", "mj_hum4.hip.h", "mj_tree.hip.h", "mj_tree_model.h",
                            "mj_quad.hip.h", "mujoco_humanoid_common.h", "mj_cheetah.hip.h"],
    "HumanoidStepKernel": ["mujoco_humanoid.hip", "mj_tree.hip.h", "mj_tree_model.h", "mujoco_humanoid_common.h",
                           "mj_cheetah.hip.h"],
    "PusherStepKernel": ["mujoco_pusher.hip", "mj_pusher.hip.h", "mj_pusher_model.h", "mj_cheetah.hip.h"],
}


def base_name(kernel: str) -> str:
    """'Humanoid4StepKernel<double>[Standup]' / 'void epa::(...)::PlanarLgStepKernel<2, 0, 1>' -> family key"""
    for key in sorted(SOURCES, key=len, reverse=True):
        if key in kernel:
            return key
    raise KeyError(kernel)


def _makefile_lines(main_source: str) -> bytes:
    """the flag variables and the rule(s) that compile `main_source`"""
    keep, take_cmd = [], False
    for line in open(os.path.join(CSRC, "Makefile")):
        if line.startswith(("COMMON =", "EXACT =", "MJFLAGS =", "ANTFLAGS =", "ARCH ", "HIPCC ")):
            keep.append(line)
        elif line.startswith("build/") and f" {main_source} " in line:
            keep.append(line)
            take_cmd = True
        elif take_cmd and line.startswith("\t"):
            keep.append(line)
        else:
            take_cmd = False
    # `#` starts a Makefile comment (none of the kept lines holds a literal '#')
    return " ".join(" ".join(line.split("#", 1)[0] for line in keep).split()).encode()


def source_hash(kernel: str) -> str:
    h = hashlib.sha256()
    h.update(_makefile_lines(SOURCES[base_name(kernel)][0]))
    for name in COMMON + SOURCES[base_name(kernel)]:
        with open(os.path.join(CSRC, name), encoding="utf-8") as f:
            h.update(name.encode() + b"\0" + strip_comments(f.read()).encode() + b"\0")
    return h.hexdigest()[:16]
