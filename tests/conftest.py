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
