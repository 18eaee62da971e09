import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the CPU oracle is test infrastructure: build it on demand (plain gcc, < 1 s)
    so = os.path.join(ROOT, "oracle", "libkvxoracle.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "libkvxoracle.so"])
    so2 = os.path.join(ROOT, "oracle", "libkvxsupernodal.so")
    if not os.path.exists(so2):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "libkvxsupernodal.so"])
    lib = os.path.join(ROOT, "kvxopt_amd", "libkvxhip.so")
    if not os.path.exists(lib):
        subprocess.check_call(["make", "-s", "-j4", "-C", os.path.join(ROOT, "kvxopt_amd", "csrc")])


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
