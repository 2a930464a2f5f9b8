"""Extracts the `DefaultConfig()` tables of the reference env families (C++ headers
under /root/reference/envpool) into tests/golden/spec_defaults.json, so that the Python
spec tables of envpool_amd are pinned to the reference's own defaults (key order and
values).  Run in the build container (the reference is not on the GPU boxes):

    python tests/golden/make_spec_golden.py
"""
import json
import os
import re

REF = "/root/reference/envpool"
HERE = os.path.dirname(os.path.abspath(__file__))

# class name in the header -> family name used by envpool_amd's FamilyDef
FAMILIES = {
    "classic_control/cartpole.h": {"CartPoleEnvFns": "CartPole"},
    "classic_control/pendulum.h": {"PendulumEnvFns": "Pendulum"},
    "classic_control/mountain_car.h": {"MountainCarEnvFns": "MountainCar"},
    "classic_control/mountain_car_continuous.h": {"MountainCarContinuousEnvFns": "MountainCarContinuous"},
    "classic_control/acrobot.h": {"AcrobotEnvFns": "Acrobot"},
    "toy_text/catch.h": {"CatchEnvFns": "Catch"},
    "toy_text/frozen_lake.h": {"FrozenLakeEnvFns": "FrozenLake"},
    "toy_text/taxi.h": {"TaxiEnvFns": "Taxi"},
    "toy_text/nchain.h": {"NChainEnvFns": "NChain"},
    "toy_text/cliffwalking.h": {"CliffWalkingEnvFns": "CliffWalking"},
    "toy_text/blackjack.h": {"BlackjackEnvFns": "Blackjack"},
    "mujoco/gym/half_cheetah.h": {"HalfCheetahEnvFns": "GymHalfCheetah"},
    "mujoco/gym/ant.h": {"AntEnvFns": "GymAnt"},
    "mujoco/gym/walker2d.h": {"Walker2dEnvFns": "GymWalker2d"},
    "mujoco/gym/hopper.h": {"HopperEnvFns": "GymHopper"},
    "mujoco/gym/swimmer.h": {"SwimmerEnvFns": "GymSwimmer"},
    "mujoco/gym/reacher.h": {"ReacherEnvFns": "GymReacher"},
    "mujoco/gym/pusher.h": {"PusherEnvFns": "GymPusher"},
    "mujoco/gym/inverted_pendulum.h": {"InvertedPendulumEnvFns": "GymInvertedPendulum"},
    "mujoco/gym/inverted_double_pendulum.h": {"InvertedDoublePendulumEnvFns": "GymInvertedDoublePendulum"},
    "mujoco/gym/humanoid.h": {"HumanoidEnvFns": "GymHumanoid"},
    "mujoco/gym/humanoid_standup.h": {"HumanoidStandupEnvFns": "GymHumanoidStandup"},
}


def parse_value(tok: str):
    tok = tok.strip()
    m = re.fullmatch(r'std::string\("(.*)"\)', tok)
    if m:
        return m.group(1)
    if tok in ("true", "false"):
        return tok == "true"
    try:
        return int(tok)
    except ValueError:
        return float(tok)


def default_config(text: str, cls: str):
    start = text.index(f"class {cls}")
    body = text[text.index("DefaultConfig()", start):]
    body = body[body.index("MakeDict("):]
    depth, end = 0, 0
    for i, ch in enumerate(body):
        depth += ch == "("
        depth -= ch == ")"
        if depth == 0 and i > 8:
            end = i
            break
    inner = body[len("MakeDict("):end]
    return [[k, parse_value(v)] for k, v in re.findall(r'"(\w+)"_\.Bind\(((?:[^()]|\([^()]*\))*)\)', inner)]


out = {}
for rel, classes in FAMILIES.items():
    text = open(os.path.join(REF, rel)).read()
    for cls, fam in classes.items():
        out[fam] = {"source": f"envpool/{rel}", "default_config": default_config(text, cls)}
json.dump(out, open(os.path.join(HERE, "spec_defaults.json"), "w"), indent=1)
for fam, v in out.items():
    print(fam, len(v["default_config"]), v["default_config"][:3])
