"""Reads the 13 gym-MuJoCo MJCF files the reference loads
(/root/reference/third_party/mujoco_gym_xml_patches/*_envpool.xml; envpool/mujoco/gym/mujoco_env.h:50-58 prefers
them over the stock gym XML) with tools/mjcf_subset.py and writes tests/golden/mjcf_models.json: every raw
attribute (defaults resolved, angles in radians, fromto -> pos / half length / axis) plus the masses, centres of mass
and inertias an inertiafromgeom compile derives from them.  tests/test_models_vs_xml.py holds BOTH hand
transcriptions of these files (oracle/mjcpu/models.c and the product's mj_*_model.h -> gen_mj_consts tables) against
this file on any box, and re-derives it from the XML where /root/reference exists.
usage: python tests/golden/make_mjcf_golden.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import mjcf_subset  # noqa: E402

XML_DIR = "/root/reference/third_party/mujoco_gym_xml_patches"
STEMS = ["half_cheetah", "ant", "hopper", "walker2d", "walker2d_v5", "swimmer", "reacher", "inverted_pendulum",
         "inverted_double_pendulum", "pusher", "pusher_v5", "humanoid", "humanoidstandup"]


def build():
    return {s: mjcf_subset.parse(os.path.join(XML_DIR, s + "_envpool.xml")) for s in STEMS}


if __name__ == "__main__":
    dst = os.path.join(ROOT, "tests", "golden", "mjcf_models.json")
    with open(dst, "w") as f:
        json.dump(build(), f, indent=0, sort_keys=True)
    print(dst, os.path.getsize(dst), "bytes")
