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


def test_two_ranks_with_live_batchnorm_match_the_emulated_ranks(tmp_path):
    """Q18 (retinanet/backbone.py:348-351: BN is only frozen at construction, DDP training runs it in train mode): per-GPU
    batch statistics and running-stat updates under the bucket hooks.  Ranks stay bit-equal in their parameters; each rank's
    running statistics equal those of a single process that runs that rank's shard with that rank's buffers and steps on the
    averaged gradients."""
    out = str(tmp_path)
    _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
          "--master-port", str(_free_port()), TOOL, "--out", out, "--precision", "32", "--bn", "train"])
    _run([sys.executable, TOOL, "--single", "--out", out, "--precision", "32", "--bn", "train", "--ranks", "2"])
    r = [torch.load(os.path.join(out, f"rank{i}.pt")) for i in range(2)]
    single = torch.load(os.path.join(out, "single.pt"))
    for k, a in r[0]["params"].items():
        assert torch.equal(a, r[1]["params"][k]), f"ranks diverged at {k}"
        b = single["params"][k]
        assert torch.allclose(a, b, rtol=0, atol=2e-5), (k, float((a - b).abs().max()))
    differ = 0
    for i in range(2):
        mine, ref = r[i]["bn_buffers"][0], single["bn_buffers"][i]
        assert mine.keys() == ref.keys() and len(mine) > 0
        for k in mine:
            assert torch.allclose(mine[k], ref[k], rtol=1e-5, atol=2e-5), (i, k, float((mine[k] - ref[k]).abs().max()))   # (the parameters' own tolerance: fp32 convs sum in run-dependent order)
            if "num_batches_tracked" in k:
                assert int(mine[k]) == 3
    for k in r[0]["bn_buffers"][0]:
        differ += int(not torch.equal(r[0]["bn_buffers"][0][k], r[1]["bn_buffers"][0][k]))
    assert differ > 0                                 # the statistics really are per GPU (different shards)
    for s in range(len(single["losses"])):
        assert abs(single["losses"][s] - 0.5 * (r[0]["losses"][s] + r[1]["losses"][s])) <= 1e-4 * abs(single["losses"][s])
