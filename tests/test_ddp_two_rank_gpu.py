"""GPU: the data-parallel path on the REAL model with 2 ranks -- ``tools/ddp_two_rank.py`` under
``torch.distributed.run``, both ranks on cuda:0, gradients over gloo, ``BucketedGradAllReduce`` + ``MasterSGD``.

The launcher and the ranks are child processes; this process only compares the saved parameters on the CPU.
Bars: ranks bit-equal after 3 steps; equal to a single-process run of the global batch within 1e-5 (fp32),
and within bf16 gradient rounding (2e-3) under bf16 autocast with bf16 conv weights + fp32 masters.
"""
import os
import socket
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "ddp_two_rank.py")
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(cmd):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-4000:]
    return r.stdout


@pytest.mark.parametrize("precision,tol", [("32", 1e-5), ("bf16", 2e-3)])
def test_two_ranks_equal_each_other_and_the_global_batch(tmp_path, precision, tol):
    out = str(tmp_path)
    _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
          "--master-port", str(_free_port()), TOOL, "--out", out, "--precision", precision])
    _run([sys.executable, TOOL, "--single", "--out", out, "--precision", precision])
    r0 = torch.load(os.path.join(out, "rank0.pt"))
    r1 = torch.load(os.path.join(out, "rank1.pt"))
    single = torch.load(os.path.join(out, "single.pt"))
    assert len(r0["buckets"]) >= 3 and all(b % 256 == 0 for b in r0["buckets"])
    moved = 0
    for k, a in r0["params"].items():
        assert torch.equal(a, r1["params"][k]), f"ranks diverged at {k}"
        b = single["params"][k]
        assert torch.allclose(a, b, rtol=0, atol=tol), (k, float((a - b).abs().max()))
        moved += int((a - b).abs().max() < 1.0)
    assert moved == len(r0["params"])
    # the global-batch loss is the mean of the two ranks' local losses (per-image normalisation, equal local batches)
    for s in range(len(single["losses"])):
        assert abs(single["losses"][s] - 0.5 * (r0["losses"][s] + r1["losses"][s])) <= (1e-4 if precision == "32" else 2e-2) * abs(single["losses"][s])
