import os
import sys

import pytest

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
