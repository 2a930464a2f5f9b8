"""Runs the reference's own registration modules (pure Python) with a recording stub in
place of `envpool.registration.register` and stores every (task_id -> kwargs) in
tests/golden/registry.json.  Pins envpool_amd's registry (ids, max_episode_steps,
per-version overrides) to the reference.  Run in the build container:

    python tests/golden/make_registry_golden.py
"""
import json
import os
import sys
import types

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
recorded = {}


def register(task_id, import_path, spec_cls, dm_cls, gymnasium_cls, **kwargs):
    recorded[task_id] = {"import_path": import_path, "spec_cls": spec_cls, "dm_cls": dm_cls,
                         "gymnasium_cls": gymnasium_cls, "kwargs": kwargs}


stub = types.ModuleType("envpool.registration")
stub.register = register
pkg = types.ModuleType("envpool")
pkg.registration = stub
sys.modules["envpool"] = pkg
sys.modules["envpool.registration"] = stub
for rel in ("envpool/classic_control/registration.py", "envpool/toy_text/registration.py",
            "envpool/mujoco/gym/registration.py"):
    src = open(os.path.join(REF, rel)).read()
    exec(compile(src, rel, "exec"), {"__name__": "golden"})
json.dump(recorded, open(os.path.join(HERE, "registry.json"), "w"), indent=1, sort_keys=True)
print(len(recorded), "task ids")
