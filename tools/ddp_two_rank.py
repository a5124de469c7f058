#!/usr/bin/env python3
"""Real-model data-parallel check on ONE GPU: W ranks share ``cuda:0`` and exchange gradients over gloo
(RCCL refuses two ranks on one device; the bucket / hook / optimizer code is the same for both backends).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P \
        tools/ddp_two_rank.py --out DIR [--precision 32|bf16]
    python tools/ddp_two_rank.py --single --out DIR [--precision 32|bf16]      # same global batch, one process

Model: RetinaNet-R18-FPN, K = 5, 128x160 inputs.  Global batch 4; rank r takes images [2r, 2r+1].  Three steps of
``BucketedGradAllReduce`` + ``MasterSGD.step(grads=grad_views())``.
  --bn frozen (default): BatchNorm on its running statistics, so the 2 x 2 split equals one process on the batch of 4.
  --bn train: BatchNorm in train mode like the reference under DDP (Q18, retinanet/backbone.py:348-351 only freezes at
      construction; Lightning's .train() un-freezes): per-GPU batch statistics and per-GPU running-stat updates under the
      bucket hooks.  The single-process counterpart (``--single``) then EMULATES the ranks: per step it runs each rank's
      shard through the model with that rank's own BN buffers, accumulates loss / W gradients and steps once -- the same
      arithmetic as the exchange, so parameters and each rank's running statistics must agree.
Every rank saves its fp32 parameters (masters) and BN buffers to DIR/rank{r}.pt, the single-process run to DIR/single.pt
(``bn_buffers``: one dict per emulated rank); ``tests/test_ddp_two_rank_gpu.py`` compares them.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("MIOPEN_LOG_LEVEL", "1")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--single", action="store_true")
    ap.add_argument("--precision", default="32", choices=["32", "bf16", "16"],
                    help="16: fp16 autocast + fp16 working copies + dynamic loss scaling (ranks: parallel.ExchangeGradScaler, found_inf from the "
                         "exchanged buckets; --single: torch.amp.GradScaler); --poison-step N makes rank 1 overflow at step N")
    ap.add_argument("--poison-step", type=int, default=-1, help="precision 16: rank 1 (ranks) / the process (--single) multiplies its loss by "
                    "inf at this step: EVERY rank must skip it and halve its scale")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--global-batch", type=int, default=4)
    ap.add_argument("--bn", default="frozen", choices=["frozen", "train"])
    ap.add_argument("--ranks", type=int, default=2, help="--single --bn train: number of ranks to emulate")
    ap.add_argument("--segmented", action="store_true", help="ranks: run the steps through graph.CapturedTrainStep's segmented form (staged "
                    "backward, four hipGraph segments from the third step on, all-reduces issued between the replays)")
    ap.add_argument("--fixture", default=None, choices=[None, "traj"], help="traj: the reference's recorded training trajectory "
                    "(tests/golden/traj.npz, frozen BatchNorm, four images per step): its state dict, inputs and optimizer settings")
    args = ap.parse_args()
    world = 1 if args.single else int(os.environ["WORLD_SIZE"])
    rank = 0 if args.single else int(os.environ["RANK"])
    if not args.single:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")

    import synth
    import pytorch_retinanet_amd as P
    from pytorch_retinanet_amd.optim import MasterSGD, use_16bit_conv_weights

    torch.manual_seed(1234)
    net = P.Retinanet(num_classes=5, backbone_kind="resnet18", pretrained=False, min_size=128, max_size=160)
    if rank == 1:                                  # sync_parameters must repair a rank that starts elsewhere
        with torch.no_grad():
            for p in net.parameters():
                p.add_(0.01)
    if args.fixture == "traj":
        sd = net.state_dict()
        spec = [(k, tuple(v.shape), str(v.dtype).replace("torch.", "")) for k, v in sd.items()]
        for k, v in synth.state_dict_values(spec, seed=4242).items():
            sd[k] = torch.from_numpy(v)
        net.load_state_dict(sd)
        initial = {n: p.detach().clone() for n, p in net.named_parameters()}
        if rank == 1:
            with torch.no_grad():
                for p in net.parameters():
                    p.add_(0.01)
    net = net.to(dev).to(memory_format=torch.channels_last).train()
    live_bn = args.bn == "train"
    if not live_bn:
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.eval()
    bf16 = args.precision != "32"                  # (a 16-bit autocast run: bf16 or fp16)
    amp_dtype = {"32": None, "bf16": torch.bfloat16, "16": torch.float16}[args.precision]
    if bf16:
        use_16bit_conv_weights(net, amp_dtype)
    opt = MasterSGD(net.parameters(), **(synth.TRAJ_OPT if args.fixture == "traj" else dict(lr=1e-2, momentum=0.9, weight_decay=1e-3)))
    emulate = args.ranks if (args.single and live_bn) else 0
    from pytorch_retinanet_amd.graph import CapturedTrainStep, retinanet_stage_of
    # (--single at precision 16: no buckets at all -- the stock torch.amp.GradScaler on MasterSGD is the independent counterpart)
    plain16 = args.single and args.precision == "16"
    ddp = None if (emulate or plain16) else P.BucketedGradAllReduce(net, bucket_mb=8.0, stage_of=retinanet_stage_of if args.segmented else None)   # several buckets for a 20 M-parameter model
    scaler = None
    if args.precision == "16":      # a start value that needs no back-off on this model: every step is a real step unless poisoned
        scaler = (P.ExchangeGradScaler if ddp is not None else torch.amp.GradScaler)("cuda", init_scale=1024.0, growth_interval=10 ** 6)
    stepper = CapturedTrainStep(net, opt, ddp, amp_dtype=amp_dtype, eager_steps=2, scaler=scaler) if (args.segmented and ddp is not None) else None
    assert emulate or plain16 or ddp.num_buckets >= 3

    def bn_buffers():
        return {n: b.detach().clone() for n, b in net.named_buffers() if "running_" in n or "num_batches_tracked" in n}

    def load_bn_buffers(bufs):
        with torch.no_grad():
            for n, b in net.named_buffers():
                if n in bufs:
                    b.copy_(bufs[n])
    rank_bufs = [bn_buffers() for _ in range(emulate)]

    rng = np.random.default_rng(99)
    G = args.global_batch
    per = G // (emulate or world)
    losses, loss_dicts = [], []
    for step in range(args.steps):
        if args.fixture == "traj":
            ims, tgs = synth.traj_inputs("frozen", step)
            assert len(ims) == G
            images = [torch.from_numpy(i) for i in ims]
            targets = [{"boxes": torch.from_numpy(b), "labels": torch.from_numpy(l)} for b, l in tgs]
        else:
            images = [torch.from_numpy(rng.random((3, 128, 160), dtype=np.float32)) for _ in range(G)]
            targets = []
            for _ in range(G):
                b, l = synth.gt_boxes(rng, 3, 128, 160, num_classes=5, wh_lo=20.0, wh_hi=90.0)
                targets.append({"boxes": torch.from_numpy(b), "labels": torch.from_numpy(l)})
        if emulate:
            opt.zero_grad(set_to_none=True)
            tot = 0.0
            for r in range(emulate):
                mine = slice(r * per, (r + 1) * per)
                imgs = [i.to(dev) for i in images[mine]]
                tgts = [{k: v.to(dev) for k, v in t.items()} for t in targets[mine]]
                load_bn_buffers(rank_bufs[r])
                with torch.autocast("cuda", dtype=amp_dtype, enabled=bf16):
                    out = net(imgs, tgts)
                    loss = out["classification_loss"] + out["regression_loss"]
                (loss / emulate).backward()                 # gradients accumulate: the average the exchange computes
                rank_bufs[r] = bn_buffers()
                tot += float(loss.detach()) / emulate
            opt.step()
            losses.append(tot)
            continue
        mine = slice(rank * per, (rank + 1) * per)
        imgs = [i.to(dev) for i in images[mine]]
        tgts = [{k: v.to(dev) for k, v in t.items()} for t in targets[mine]]
        if stepper is not None:
            o = stepper(imgs, tgts)
            losses.append(float(o["loss"]))
            loss_dicts.append([float(o["classification_loss"]), float(o["regression_loss"])])
            continue
        from pytorch_retinanet_amd.losses import grad_prescale, scaler_prescale
        if ddp is not None:
            ddp.zero_grad()
        else:
            opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=amp_dtype, enabled=bf16), grad_prescale(scaler_prescale(scaler, dev)):
            out = net(imgs, tgts)
            loss = out["classification_loss"] + out["regression_loss"]
        poisoned = step == args.poison_step and (args.single or rank == 1)
        if scaler is not None:
            scaler.scale(loss * float("inf") if poisoned else loss).backward()
            if ddp is not None:
                ddp.finish()
                scaler.step_exchanged(opt, ddp)
            else:
                scaler.step(opt)
            scaler.update()
        else:
            loss.backward()
            if ddp is not None:
                ddp.finish()
                opt.step(grads=ddp.grad_views())
            else:
                opt.step()
        losses.append(float(loss.detach()))
        loss_dicts.append([float(out["classification_loss"].detach()), float(out["regression_loss"].detach())])
    torch.cuda.synchronize()
    state = {n: (p.master if hasattr(p, "master") else p.data).detach().float().cpu() for n, p in net.named_parameters()}
    os.makedirs(args.out, exist_ok=True)
    bufs = [{n: b.float().cpu() for n, b in rb.items()} for rb in rank_bufs] if emulate else [{n: b.float().cpu() for n, b in bn_buffers().items()}]
    torch.save({"params": state, "losses": losses, "buckets": ddp.bucket_bytes() if ddp else [], "bn_buffers": bufs, "loss_dicts": loss_dicts,
                "initial": {n: v.float().cpu() for n, v in initial.items()} if args.fixture == "traj" else {},
                "replays": stepper.replays if stepper is not None else 0, "scale": float(scaler.get_scale()) if scaler is not None else None},
               os.path.join(args.out, "single.pt" if args.single else f"rank{rank}.pt"))
    if not args.single:
        dist.barrier()
        dist.destroy_process_group()
    print(f"rank {rank}/{world}: losses {['%.5f' % x for x in losses]} buckets {ddp.num_buckets if ddp else 0} "
          f"graph replays {stepper.replays if stepper is not None else 0}", flush=True)


if __name__ == "__main__":
    main()
