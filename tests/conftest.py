import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _ensure_built():
    lib = os.path.join(ROOT, "tray_racing_amd", "libtrx.so")
    orc = os.path.join(ROOT, "oracle", "liboracle.so")
    if not os.path.exists(lib):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tray_racing_amd", "csrc")])
    if not os.path.exists(orc):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])


_ensure_built()


@pytest.fixture(scope="session")
def trx():
    import tray_racing_amd as T
    T.load()
    return T


@pytest.fixture(scope="session")
def orc():
    from oracle import binding
    binding.load()
    return binding


@pytest.fixture(scope="session")
def has_gpu(trx):
    return trx.load().trx_device_count() > 0
