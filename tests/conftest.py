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


# Collection order of the GPU run (the driver runs ``pytest -x``: one red test hides everything collected after it).
# HIP-vs-oracle / HIP-vs-golden parity first, then the assembled model against the reference fixtures, then the
# kernel-level conv / BatchNorm tests, then graph capture and the out-of-bounds guard, and the multi-process
# integration tests (child launchers, run-to-run noise in their trajectories) LAST.
_ORDER = [
    "test_oracle_golden.py", "test_host_surface.py", "test_properties.py", "test_coco_eval.py", "test_parallel_gloo.py",
    "test_hip_parity.py", "test_e2e_gpu.py", "test_model_gpu.py",
    "test_norm_gpu.py", "test_pwconv_gpu.py", "test_dense_conv_gpu.py", "test_narrow_conv_gpu.py",
    "test_graph_gpu.py", "test_guard_gpu.py", "test_traj_gpu.py",
    "test_rccl_world1_gpu.py", "test_ddp_two_rank_gpu.py", "test_bench_launch_gpu.py",
]


def pytest_collection_modifyitems(config, items):
    have_ref = os.path.isdir("/root/reference/retinanet")
    for item in items:
        if "reference" in item.keywords and not have_ref:
            item.add_marker(pytest.mark.skip(reason="/root/reference not present"))
    rank = {name: i for i, name in enumerate(_ORDER)}
    mid = _ORDER.index("test_graph_gpu.py")            # files not listed: after the kernel tests, before the multi-process ones

    def key(item):
        return rank.get(os.path.basename(str(item.fspath)), mid - 0.5)
    items.sort(key=key)                                 # stable: the order inside a file is kept


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
