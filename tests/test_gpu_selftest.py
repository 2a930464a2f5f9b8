"""Load-time self-test of the MuJoCo kernels (engine.hip, SelfTest): the first epa_create of a MuJoCo family in a
process steps fixed states in every kernel variant of the family and compares with the host instantiation of the same
arithmetic (table generated at build time, envpool_amd/csrc/gen_selftest.cpp); a mismatch, or two launches that
differ, is EPA_ERR_DEVICE.  Correctness of these translation units depends on compiler flags (csrc/Makefile, MJFLAGS):
the library must notice a bad build by itself, not only pytest."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LICM_LIB = os.path.join(ROOT, "envpool_amd", "lib", "libenvpool_amd_licm.so")

_CREATE = """
import sys, numpy as np
from envpool_amd.core.device_pool import DevicePool
for task in sys.argv[1:]:
    try:
        p = DevicePool(task, 128, seed=0, max_episode_steps=100)
    except RuntimeError as e:
        print("REFUSED", task, str(e)[:300].replace("\\n", " "))
        continue
    ids = np.arange(128, dtype=np.int32)
    p.reset(ids); p.recv()
    print("CREATED", task)
"""

MUJOCO = ["HalfCheetah", "Walker2d", "Hopper", "Ant", "Humanoid", "HumanoidStandup"]


def _run(env, *tasks):
    e = dict(os.environ, PYTHONPATH=ROOT, **env)
    r = subprocess.run([sys.executable, "-c", _CREATE, *tasks], capture_output=True, text=True, env=e, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    return r.stdout


def test_product_build_passes_its_self_test():
    out = _run({}, *MUJOCO, "CartPole")
    for task in MUJOCO + ["CartPole"]:
        assert f"CREATED {task}" in out, out


def test_a_mismatch_with_the_host_arithmetic_is_refused():
    """EPA_SELFTEST_CORRUPT shifts the expected values: the refusal path itself (error class and message);
    EPA_SELFTEST=0 or the engine key switch the test off; families without a table are untouched."""
    out = _run({"EPA_SELFTEST_CORRUPT": "1"}, *MUJOCO, "CartPole", "Pusher")
    for task in MUJOCO:
        assert f"REFUSED {task}" in out and "self-test" in out and "refusing to run" in out, out
    assert "CREATED CartPole" in out and "CREATED Pusher" in out, out
    out = _run({"EPA_SELFTEST_CORRUPT": "1", "EPA_SELFTEST": "0"}, "HalfCheetah", "Ant")
    assert "CREATED HalfCheetah" in out and "CREATED Ant" in out, out


def test_a_build_without_the_makefiles_flags():
    """libenvpool_amd_licm.so (tools/build_alt_licm.sh): the MuJoCo translation units compiled WITHOUT MJFLAGS, the
    build the Makefile warns about.  The library must either refuse it, or -- where this compiler happens to get a
    family right -- pass it, in which case its results must equal the product library's."""
    if not os.path.exists(LICM_LIB):
        pytest.skip("libenvpool_amd_licm.so not built (tools/build_alt_licm.sh)")
    out = _run({"ENVPOOL_AMD_LIB": LICM_LIB}, *MUJOCO[:4])
    print(out)
    refused = [t for t in MUJOCO[:4] if f"REFUSED {t}" in out]
    created = [t for t in MUJOCO[:4] if f"CREATED {t}" in out]
    assert sorted(refused + created) == sorted(MUJOCO[:4]), out
    rollout = """
import sys, numpy as np
from envpool_amd.core.device_pool import DevicePool
task, out = sys.argv[1], sys.argv[2]
n = 512
p = DevicePool(task, n, seed=3, max_episode_steps=1000)
ids = np.arange(n, dtype=np.int32); p.reset(ids); p.recv()
adim = int(np.prod(p.action_shape)); rng = np.random.default_rng(0)
obs = []
for t in range(20):
    p.send(ids, rng.uniform(-1, 1, (n, adim))); obs.append(p.recv_dict()["obs"])
np.save(out, np.stack(obs))
"""
    import numpy as np
    import tempfile

    for task in created:  # passed the self-test: then it must BE right
        with tempfile.TemporaryDirectory() as tmp:
            outs = []
            for tag, env in (("product", {}), ("licm", {"ENVPOOL_AMD_LIB": LICM_LIB})):
                f = os.path.join(tmp, tag + ".npy")
                subprocess.run([sys.executable, "-c", rollout, task, f], check=True, timeout=600,
                               env=dict(os.environ, PYTHONPATH=ROOT, **env))
                outs.append(np.load(f))
            np.testing.assert_allclose(outs[1], outs[0], rtol=1e-6, atol=1e-7, err_msg=task)
