"""CPU suite: the data-parallel gradient exchange (parallel.BucketedGradAllReduce) with 2 gloo ranks."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch import nn


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model():
    torch.manual_seed(7)
    return nn.Sequential(nn.Conv2d(3, 8, 3, padding=1), nn.ReLU(), nn.Conv2d(8, 8, 3, padding=1), nn.ReLU(),
                         nn.Flatten(), nn.Linear(8 * 6 * 6, 5))


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pytorch_retinanet_amd.parallel import BucketedGradAllReduce
    model = _model()
    if rank == 1:                      # start from different weights: sync_parameters must fix that
        with torch.no_grad():
            for p in model.parameters():
                p.add_(1.0)
    ddp = BucketedGradAllReduce(model, bucket_mb=0.002)      # tiny buckets -> several collectives
    assert ddp.num_buckets >= 3
    opt = torch.optim.SGD(model.parameters(), lr=0.1, momentum=0.9)
    g = torch.Generator().manual_seed(100)
    data = torch.randn(3, 4, 3, 6, 6, generator=g)           # [step, global batch 4, ...]
    tgt = torch.randn(3, 4, 5, generator=g)
    for step in range(3):
        x, y = data[step, rank * 2:(rank + 1) * 2], tgt[step, rank * 2:(rank + 1) * 2]
        ddp.zero_grad() if step != 1 else opt.zero_grad(set_to_none=True)    # both zeroing styles
        loss = ((model(x) - y) ** 2).mean()
        loss.backward()
        ddp.finish()
        opt.step()
    torch.save([p.detach().clone() for p in model.parameters()], os.path.join(out_dir, f"rank{rank}.pt"))
    dist.destroy_process_group()


def test_two_rank_gloo_equals_single_process_global_batch(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    p0 = torch.load(tmp_path / "rank0.pt")
    p1 = torch.load(tmp_path / "rank1.pt")
    for a, b in zip(p0, p1):
        assert torch.equal(a, b), "ranks diverged"
    # single process, whole global batch: mean over 4 samples == average of the two ranks' means over 2
    model = _model()
    opt = torch.optim.SGD(model.parameters(), lr=0.1, momentum=0.9)
    g = torch.Generator().manual_seed(100)
    data = torch.randn(3, 4, 3, 6, 6, generator=g)
    tgt = torch.randn(3, 4, 5, generator=g)
    for step in range(3):
        opt.zero_grad()
        ((model(data[step]) - tgt[step]) ** 2).mean().backward()
        opt.step()
    for a, b in zip(p0, model.parameters()):
        assert torch.allclose(a, b.detach(), rtol=1e-5, atol=1e-6)


def test_bucket_layout_single_process():
    from pytorch_retinanet_amd.parallel import BucketedGradAllReduce
    model = _model()
    ddp = BucketedGradAllReduce(model, bucket_mb=0.002)
    params = [p for p in model.parameters()]
    # reverse registration order (~ reverse forward): the last layer's parameters land in the first bucket
    assert ddp.buckets[0].params[0] is params[-1]
    assert sum(ddp.bucket_bytes()) == sum(BucketedGradAllReduce._padded(p.numel()) * 4 for p in params)
    for v in ddp.grad_views().values():    # every view starts 256-byte aligned relative to its bucket (RN_EALIGN)
        b = next(b for b in ddp.buckets if b.flat.data_ptr() <= v.data_ptr() < b.flat.data_ptr() + b.flat.numel() * 4)
        assert (v.data_ptr() - b.flat.data_ptr()) % 256 == 0
    for p in params:                       # grads are views into the buckets
        assert p.grad is not None and p.grad.data_ptr() >= min(b.flat.data_ptr() for b in ddp.buckets)
    model(torch.randn(2, 3, 6, 6)).sum().backward()
    ddp.finish()                           # world size 1: no collective, state resets
    assert all(b.pending == len(b.params) for b in ddp.buckets)
    for p in params:                       # after the exchange the gradients are views into the buckets again
        assert any(b.flat.data_ptr() <= p.grad.data_ptr() < b.flat.data_ptr() + b.flat.numel() * 4 for b in ddp.buckets)
    want = [p.grad.clone() for p in params]
    ddp.zero_grad()                        # drops the gradients (no memset); the next backward assigns fresh ones
    assert all(p.grad is None for p in params)
    model(torch.randn(2, 3, 6, 6)).sum().backward()
    ddp.finish()                           # ... which are gathered into the bucket views, not accumulated onto old values
    model2_grads = [p.grad.clone() for p in params]
    assert all(g.shape == w.shape for g, w in zip(model2_grads, want))
    for p in params:
        assert any(b.flat.data_ptr() <= p.grad.data_ptr() < b.flat.data_ptr() + b.flat.numel() * 4 for b in ddp.buckets)


def test_buckets_exchange_bf16_working_copies_in_fp32():
    """A bf16 parameter that carries an fp32 master (optim.use_bf16_conv_weights) gets an fp32 bucket: its bf16 gradient is
    promoted on the way in, and the optimizer reads it through grad_views()."""
    from pytorch_retinanet_amd.parallel import BucketedGradAllReduce
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3, padding=1), torch.nn.Conv2d(4, 2, 1))
    w = model[0].weight
    master = w.data.clone()
    w.data = master.to(torch.bfloat16)
    w.master = master
    ddp = BucketedGradAllReduce(model, bucket_mb=1.0)
    assert all(b.flat.dtype == torch.float32 for b in ddp.buckets) and w.grad is None
    x = torch.randn(2, 3, 5, 5)
    with torch.autocast("cpu", dtype=torch.bfloat16):
        y = model(x)
    y.float().sum().backward()
    g16 = w.grad.clone()
    assert g16.dtype == torch.bfloat16
    ddp.finish()
    views = ddp.grad_views()
    assert views[w].dtype == torch.float32 and torch.equal(views[w], g16.float())
    assert model[1].weight.grad.data_ptr() == views[model[1].weight].data_ptr()       # fp32 parameters: .grad IS the view
    ddp.zero_grad()
    assert w.grad is None


def test_bucket_views_stay_aligned_after_odd_sized_parameters():
    """9*K-element head biases (K = 90: 810, K = 5: 45) are not multiples of 4 elements: the views packed after them
    must still start on 16-byte boundaries (rn_sgd_master_step rejects anything else with RN_EALIGN)."""
    from pytorch_retinanet_amd.parallel import BucketedGradAllReduce
    model = nn.Sequential(nn.Conv2d(4, 45, 3, padding=1), nn.Conv2d(45, 810, 3, padding=1), nn.Conv2d(810, 3, 1))
    ddp = BucketedGradAllReduce(model, bucket_mb=32.0)
    assert ddp.num_buckets == 1
    base = ddp.buckets[0].flat.data_ptr()
    for p, v in ddp.grad_views().items():
        assert (v.data_ptr() - base) % 256 == 0 and v.shape == p.shape and v.stride() == p.stride()
    model(torch.randn(1, 4, 5, 5)).sum().backward()
    ddp.finish()
    for p, v in ddp.grad_views().items():
        assert p.grad.data_ptr() == v.data_ptr() and torch.isfinite(v).all()


def test_bucket_gather_falls_back_to_copy_on_the_cpu():
    "parallel._gather on CPU tensors (no HIP library involved): plain copies, incl. the widening one and mismatched strides."
    import torch
    from pytorch_retinanet_amd import parallel
    g = torch.Generator().manual_seed(1)
    grads = [torch.randn((4, 3, 2, 2), generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last),
             torch.randn((7,), generator=g), torch.randn((3, 5), generator=g).t()]
    views = [torch.empty((4, 3, 2, 2), dtype=torch.float32).contiguous(memory_format=torch.channels_last), torch.empty((7,)), torch.empty((5, 3))]
    parallel._gather(views, grads)
    for v, gr in zip(views, grads):
        assert torch.equal(v, gr.float())


def _stage_of(name):                   # of _model(): the Linear finishes first in backward, conv 0 last
    return {"5": 0, "2": 1, "0": 2}[name.split(".")[0]]


def _worker_staged(rank, world, port, out_dir):
    """The exchange as graph.CapturedTrainStep's segmented step drives it: stage-aligned buckets, ``deferred`` (the hooks only gather),
    the backward pass cut into three autograd calls at detached leaves, ``issue_ready()`` after each, ``finish()`` before the optimizer."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pytorch_retinanet_amd.parallel import BucketedGradAllReduce
    model = _model()
    ddp = BucketedGradAllReduce(model, bucket_mb=32.0, stage_of=_stage_of)
    assert ddp.num_buckets == 3                                   # one per stage although everything would fit one bucket
    ddp.deferred = True
    opt = torch.optim.SGD(model.parameters(), lr=0.1, momentum=0.9)
    g = torch.Generator().manual_seed(100)
    data = torch.randn(3, 4, 3, 6, 6, generator=g)
    tgt = torch.randn(3, 4, 5, generator=g)
    issued = []
    for step in range(3):
        x, y = data[step, rank * 2:(rank + 1) * 2], tgt[step, rank * 2:(rank + 1) * 2]
        ddp.zero_grad()
        h1 = model[1](model[0](x))
        h1l = h1.detach().requires_grad_(True)
        h2 = model[3](model[2](h1l))
        h2l = h2.detach().requires_grad_(True)
        loss = ((model[5](model[4](h2l)) - y) ** 2).mean()
        loss.backward()
        assert all(b.work is None for b in ddp.buckets)           # nothing was exchanged from the hooks
        issued.append(ddp.issue_ready())
        h2.backward(h2l.grad)
        issued.append(ddp.issue_ready())
        h1.backward(h1l.grad)
        issued.append(ddp.issue_ready())
        ddp.finish()
        opt.step()
    assert issued == [[0], [1], [2]] * 3, issued
    # the replay form: same bucket indices through issue()
    x, y = data[0, rank * 2:(rank + 1) * 2], tgt[0, rank * 2:(rank + 1) * 2]
    ddp.zero_grad()
    ((model(x) - y) ** 2).mean().backward()                      # (all three buckets become ready in one pass)
    ready, ddp.ready = [ddp.buckets.index(b) for b in ddp.ready], []
    before = [b.flat.clone() for b in ddp.buckets]
    ddp.issue(sorted(ready))
    ddp.finish()
    t = torch.stack([bf.abs().sum() for bf in before])
    dist.all_reduce(t)                                            # (just keeps the two ranks in step before saving)
    torch.save([p.detach().clone() for p in model.parameters()], os.path.join(out_dir, f"staged{rank}.pt"))
    dist.destroy_process_group()


def test_staged_deferred_exchange_equals_the_hook_driven_one(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    mp.spawn(_worker_staged, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    ref = torch.load(tmp_path / "rank0.pt")
    s0, s1 = torch.load(tmp_path / "staged0.pt"), torch.load(tmp_path / "staged1.pt")
    for a, b, c in zip(ref, s0, s1):
        assert torch.equal(b, c), "ranks diverged"
        assert torch.allclose(a, b, rtol=1e-6, atol=1e-7)         # (rank 1 of the reference run starts from shifted weights and is synced; same maths after)


def _scaler_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pytorch_retinanet_amd.parallel import BucketedGradAllReduce, ExchangeGradScaler
    model = _model()
    ddp = BucketedGradAllReduce(model, bucket_mb=0.002)
    opt = torch.optim.SGD(model.parameters(), lr=0.1, momentum=0.9)
    scaler = ExchangeGradScaler("cpu", init_scale=256.0, growth_interval=2, backoff_factor=0.5, growth_factor=2.0)
    g = torch.Generator().manual_seed(100)
    data = torch.randn(4, 4, 3, 6, 6, generator=g)
    tgt = torch.randn(4, 4, 5, generator=g)
    log = []
    for step in range(4):
        x, y = data[step, rank * 2:(rank + 1) * 2], tgt[step, rank * 2:(rank + 1) * 2]
        ddp.zero_grad()
        loss = ((model(x) - y) ** 2).mean()
        if step == 1 and rank == 1:                       # ONE rank overflows: its gradients are inf / NaN, the other rank's are finite
            loss = loss * float("inf")
        before = [p.detach().clone() for p in model.parameters()]
        scaler.scale(loss).backward()
        ddp.finish()
        scaler.step_exchanged(opt, ddp)
        scaler.update()
        moved = any(not torch.equal(a, p.detach()) for a, p in zip(before, model.parameters()))
        log.append((moved, float(scaler.get_scale())))
    torch.save({"params": [p.detach().clone() for p in model.parameters()], "log": log}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.destroy_process_group()


def test_exchange_grad_scaler_takes_found_inf_from_the_exchanged_buckets(tmp_path):
    """fp16-style dynamic loss scaling under the exchange: rank 1 overflows at step 1.  The stock GradScaler checks
    the rank-LOCAL gradients, so rank 0 would step and rank 1 would skip; ``ExchangeGradScaler.step_exchanged`` reads the flag off the
    all-reduced buckets (inf + x = inf on every rank): BOTH ranks skip step 1, both halve their scale, both grow it again after two clean
    steps, and the parameters stay bit-equal and equal to a single process that skips the same step."""
    world, port = 2, _free_port()
    mp.spawn(_scaler_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = torch.load(tmp_path / "rank0.pt"), torch.load(tmp_path / "rank1.pt")
    for a, b in zip(r0["params"], r1["params"]):
        assert torch.equal(a, b), "ranks diverged"
    assert r0["log"] == r1["log"]
    assert [m for m, _ in r0["log"]] == [True, False, True, True]                 # step 1 skipped on BOTH ranks
    assert [s for _, s in r0["log"]] == [256.0, 128.0, 128.0, 256.0]              # halved at the overflow, doubled after two clean steps
    model = _model()
    opt = torch.optim.SGD(model.parameters(), lr=0.1, momentum=0.9)
    g = torch.Generator().manual_seed(100)
    data = torch.randn(4, 4, 3, 6, 6, generator=g)
    tgt = torch.randn(4, 4, 5, generator=g)
    for step in (0, 2, 3):                                                         # the single process that skips the same step
        opt.zero_grad()
        ((model(data[step]) - tgt[step]) ** 2).mean().backward()
        opt.step()
    for a, b in zip(r0["params"], model.parameters()):
        assert torch.allclose(a, b.detach(), rtol=1e-5, atol=1e-6)


def test_exchange_plan_is_the_bucket_layout_without_building_it():
    "``plan_for`` (what bench.py prints at N = 1) == ``plan()`` of the built exchange: bucket count, bytes per backward stage, total."
    from pytorch_retinanet_amd.parallel import BucketedGradAllReduce
    stage_of = lambda n: 0 if n.startswith("5") else (1 if n.startswith("2") else 2)
    want = BucketedGradAllReduce.plan_for(_model(), bucket_mb=0.002, stage_of=stage_of)
    ddp = BucketedGradAllReduce(_model(), bucket_mb=0.002, stage_of=stage_of)
    assert ddp.plan() == want and want["buckets"] == ddp.num_buckets >= 3
    assert sum(want["bytes_per_stage"]) == want["bytes_total"] == sum(ddp.bucket_bytes()) and len(want["bytes_per_stage"]) == 3
