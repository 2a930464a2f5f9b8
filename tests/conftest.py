import os
import sys

import pytest

# The oracle (oracle/mjcpu) spreads envs over host cores with OpenMP for the bench's
# cpu_baseline; the tests step a few hundred envs at a time from several oracle
# objects, where hundreds of spinning libgomp workers (256 hardware threads on the
# GPU boxes) cost minutes.  Must be set before libgomp is loaded.
os.environ.setdefault("OMP_NUM_THREADS", "4")
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run via gpurun)")


@pytest.fixture(scope="session", autouse=True)
def _build_port_oracle():
    """The plain-C oracle is test infrastructure; build it on demand."""
    import subprocess

    subprocess.run(
        ["make", "-s", "-C", os.path.join(ROOT, "oracle"), "port"], check=True
    )


@pytest.fixture(scope="session", autouse=True)
def _torch_runtime_first():
    """On a GPU box, initialise torch's HIP runtime BEFORE the product library's.  The torch wheel
    bundles its own ROCm runtime; libenvpool_amd.so links the system one.  Both work in one process
    (the device-path tests rely on it), but a torch runtime that comes up late -- after the other
    one has created and destroyed dozens of pools with six streams each -- was once refused its
    devices ("No HIP GPUs are available", profiles/archive/r3l_*): the order is made deterministic here."""
    if not os.path.exists("/dev/kfd"):
        return
    try:
        import torch

        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:  # no torch / no device: the tests that need them skip or fail on their own
        pass
