import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference (build container only)")
    # the package refuses to import without its HIP library (no CPU fallback): build it if a fresh checkout has none
    lib = os.path.join(ROOT, "pytorch_retinanet_amd", "libretinanet_hip.so")
    if not os.path.exists(lib):
        import subprocess
        subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(ROOT, "pytorch_retinanet_amd", "csrc")])


def pytest_collection_modifyitems(config, items):
    have_ref = os.path.isdir("/root/reference/retinanet")
    for item in items:
        if "reference" in item.keywords and not have_ref:
            item.add_marker(pytest.mark.skip(reason="/root/reference not present"))


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)

    return load


@pytest.fixture(scope="session")
def oracle_lib():
    import oracle
    oracle.build()
    return oracle
