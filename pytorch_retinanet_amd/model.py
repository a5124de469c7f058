"""``RetinaNetModel`` -- the Lightning-style wrapper of the reference (``model.py:18-147``):
same constructor (``RetinaNetModel(conf)``), same hook names, same step-output dict keys
(``loss`` / ``log`` / ``progress_bar``, ``val_loss``, ``AP``).

``pytorch_lightning`` is used as the base class when it is importable; otherwise a minimal
local base provides the two things the hooks rely on (``save_hyperparameters`` and
``nn.Module``), and ``SimpleTrainer`` below drives the hooks (one process per GPU, gradients
exchanged by ``parallel.BucketedGradAllReduce`` over RCCL).

Data: ``dataset.kind: synthetic`` or ``csv`` (the reference's csv format, ``datasets.py``); the COCO-json /
Pascal-xml readers and albumentations pipelines of the reference's ``utils/`` are outside this framework's
scope (``prepare_data`` says so).  ``test_step`` / ``test_epoch_end`` feed the pycocotools-free
``coco_eval.CocoEvaluator`` (reference ``utils/coco/coco_eval.py``).
"""
import argparse
import logging
from typing import Any, Dict, List, Optional, Union

import torch
from torch import nn
from torch.utils.data import DataLoader, Dataset

from .models import Retinanet
from .utils import AttrDict, collate_fn, load_obj

try:                                                   # pragma: no cover - not installed in this image
    import pytorch_lightning as pl
    _Base = pl.LightningModule
except Exception:                                      # noqa: BLE001
    pl = None

    class _Base(nn.Module):
        "The slice of LightningModule the hooks below need."

        def save_hyperparameters(self, conf) -> None:
            self.hparams = conf


class SyntheticDetectionDataset(Dataset):
    """Random images + boxes in the reference's sample format ``(image, target, image_idx)``
    (``utils/pascal/pascal_utils.py:98-142``): image ``F32[3,H,W]`` in 0..1, target
    ``{"boxes": F32[T,4] xyxy, "labels": I64[T] in 1..K, "image_id": I64[1]}``."""

    def __init__(self, length: int = 16, height: int = 800, width: int = 1333, num_classes: int = 90,
                 boxes_per_image: int = 8, seed: int = 0):
        self.length, self.h, self.w, self.k, self.t, self.seed = length, height, width, num_classes, boxes_per_image, seed

    def __len__(self) -> int:
        return self.length

    def __getitem__(self, idx: int):
        g = torch.Generator().manual_seed(self.seed * 1000003 + idx)
        img = torch.rand(3, self.h, self.w, generator=g)
        cx = torch.rand(self.t, generator=g) * self.w
        cy = torch.rand(self.t, generator=g) * self.h
        bw = 16 + torch.rand(self.t, generator=g) * 300
        bh = 16 + torch.rand(self.t, generator=g) * 300
        boxes = torch.stack([(cx - bw / 2).clamp(0, self.w - 2), (cy - bh / 2).clamp(0, self.h - 2),
                             (cx + bw / 2).clamp(0, self.w), (cy + bh / 2).clamp(0, self.h)], 1)
        boxes[:, 2:] = torch.maximum(boxes[:, 2:], boxes[:, :2] + 1.0)
        labels = torch.randint(1, self.k + 1, (self.t,), generator=g)
        return img, {"boxes": boxes, "labels": labels, "image_id": torch.tensor([idx])}, idx


class RetinaNetModel(_Base):
    def __init__(self, conf: Union[AttrDict, Dict[str, Any], argparse.Namespace]):
        super().__init__()
        self.conf = conf
        self.net = Retinanet(**conf.model, logger=logging.getLogger("lightning"))
        self.save_hyperparameters(conf)
        self.trn_ds = self.val_ds = self.test_ds = None
        self.test_evaluator = None

    def forward(self, xb, *args, **kwargs):
        # reference model.py:33-35 calls self.net(xb) with no targets (a TypeError there, Q19);
        # here that means inference.
        return self.net(xb)

    # -- data ----------------------------------------------------------------------------------
    def prepare_data(self):
        d = self.conf.dataset
        if d.kind == "synthetic":
            kw = dict(num_classes=self.net.num_classes)
            kw.update({k: v for k, v in d.items() if k in ("length", "height", "width", "boxes_per_image", "seed")})
            self.trn_ds = SyntheticDetectionDataset(**kw)
            self.val_ds = SyntheticDetectionDataset(**{**kw, "seed": kw.get("seed", 0) + 1})
            self.test_ds = SyntheticDetectionDataset(**{**kw, "seed": kw.get("seed", 0) + 2})
        elif d.kind == "csv":             # README.md:103-125: trn_paths / val_paths (optional) / test_paths are csv files
            from .datasets import CSVDetectionDataset
            self.trn_ds = CSVDetectionDataset(d.trn_paths)
            self.val_ds = CSVDetectionDataset(d.val_paths) if d.get("val_paths") else None
            self.test_ds = CSVDetectionDataset(d.test_paths) if d.get("test_paths") else None
        elif d.kind in ("coco", "pascal"):
            raise NotImplementedError(
                f"dataset.kind={d.kind!r}: the reference's COCO-json / Pascal-xml readers (utils/coco, utils/pascal; "
                "pycocotools, albumentations, cv2) are outside this framework's scope. Convert to the csv format "
                "(kind: csv), assign `trn_ds` / `val_ds` / `test_ds` with your own Dataset yielding "
                "(image, target, image_idx), or use kind: synthetic.")
        else:
            raise ValueError("DATASET_KIND not supported")

    def _loader(self, ds, bs, shuffle=False, shard=False):
        """``shard``: under ``torch.distributed`` the rank reads its own shard (``DistributedSampler``, what Lightning injects
        for the reference's train / val loaders); ``SimpleTrainer`` calls ``sampler.set_epoch`` so the shards reshuffle per
        epoch.  The TEST loader is never sharded: ``test_epoch_end`` scores the detections a rank has seen, a shard would
        report AP on 1/W of the images and the sampler's padding would duplicate image ids; every rank evaluates the full set."""
        import torch.distributed as dist
        sampler = None
        if shard and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            from torch.utils.data.distributed import DistributedSampler
            sampler = DistributedSampler(ds, shuffle=shuffle)
            shuffle = False
        return DataLoader(ds, bs, shuffle=shuffle, sampler=sampler, collate_fn=collate_fn, **dict(self.conf.dataloader.args))

    def train_dataloader(self, *args, **kwargs):
        return self._loader(self.trn_ds, self.conf.dataloader.train_bs, shuffle=True, shard=True)

    def val_dataloader(self, *args, **kwargs):
        # sharded: SimpleTrainer averages the validation loss over ranks (a padded duplicate shifts a mean loss by O(1/len))
        return None if self.val_ds is None else self._loader(self.val_ds, self.conf.dataloader.valid_bs, shard=True)

    def test_dataloader(self, *args, **kwargs):
        from .coco_eval import CocoEvaluator, gt_from_dataset
        loader = self._loader(self.test_ds, self.conf.dataloader.test_bs)
        self.test_evaluator = CocoEvaluator(gt_from_dataset(loader.dataset), ["bbox"])       # reference model.py:105-110
        return loader

    # -- optimisation ----------------------------------------------------------------------------
    def configure_optimizers(self, *args, **kwargs):
        opt_cls = load_obj(self.conf.optimizer.class_name)
        if getattr(opt_cls, "__name__", "") == "MasterSGD" and next(self.net.parameters()).is_cuda:
            # optimizer.class_name: pytorch_retinanet_amd.optim.MasterSGD -- SGD on fp32 masters, conv weights held in bf16
            from .optim import use_16bit_conv_weights
            use_16bit_conv_weights(self.net, getattr(self, "working_dtype", None) or torch.bfloat16)     # (SimpleTrainer sets it from its precision)
        self.optimizer = opt_cls(self.net.parameters(), **dict(self.conf.optimizer.params))
        sched = self.conf.scheduler
        if sched.class_name is None:
            return [self.optimizer]
        params = dict(sched.params)
        if "verbose" in params:
            import inspect
            if "verbose" not in inspect.signature(load_obj(sched.class_name).__init__).parameters:
                params.pop("verbose")               # removed from torch schedulers after the reference was written
        scheduler = load_obj(sched.class_name)(self.optimizer, **params)
        self.scheduler = {"scheduler": scheduler, "interval": sched.interval, "frequency": sched.frequency}
        if sched.monitor:
            self.scheduler["monitor"] = sched.monitor
        return [self.optimizer], [self.scheduler]

    # -- steps (same dict contracts as reference model.py:112-146) ---------------------------------
    def training_step(self, batch, batch_idx, *args, **kwargs):
        images, targets, _ = batch
        targets = [{k: v for k, v in t.items()} for t in targets]
        loss_dict = self.net(images, targets)
        losses = sum(loss for loss in loss_dict.values())
        return {"loss": losses, "log": loss_dict, "progress_bar": loss_dict}

    def validation_step(self, batch, batch_idx, *args, **kwargs):
        images, targets, _ = batch
        targets = [{k: v for k, v in t.items()} for t in targets]
        loss_dict = self.net(images, targets)
        loss = torch.as_tensor(sum(loss for loss in loss_dict.values()))
        logs = {"val_loss": loss}
        return {"val_loss": loss, "log": logs, "progress_bar": logs}

    def test_step(self, batch, batch_idx, *args, **kwargs):
        images, targets, _ = batch
        targets = [{k: v for k, v in t.items()} for t in targets]
        outputs = self.net.predict(images)
        res = {t["image_id"].item(): o for t, o in zip(targets, outputs)}
        if self.test_evaluator is not None:
            self.test_evaluator.update(res)
        return {"detections": res}

    def test_epoch_end(self, outputs, *args, **kwargs):
        if self.test_evaluator is None:
            return {}
        self.test_evaluator.accumulate()
        self.test_evaluator.summarize()
        metric = torch.as_tensor(self.test_evaluator.coco_eval["bbox"].stats[0])
        logs = {"AP": metric}
        return {"AP": metric, "log": logs, "progress_bar": logs}


def _to_device(batch, device):
    images, targets, ids = batch
    images = [im.to(device, non_blocking=True) for im in images]
    targets = [{k: v.to(device, non_blocking=True) for k, v in t.items()} for t in targets]
    return images, targets, ids


class SimpleTrainer:
    """Minimal stand-in for ``pl.Trainer`` driving the hooks above on ONE device per process:
    ``fit`` (train + optional validation, scheduler stepping per the hparams contract) and ``test``.
    Under ``torch.distributed`` every rank runs this loop on its own shard of the dataset (``DistributedSampler``),
    gradients are averaged by ``BucketedGradAllReduce`` and the validation loss is averaged over ranks before it
    reaches a monitoring scheduler.  Validation runs in ``eval()`` mode like Lightning's (BN buffers untouched)."""

    def __init__(self, max_epochs: int = 1, device: Optional[str] = None, precision: str = "bf16",
                 channels_last: bool = True, max_steps: Optional[int] = None, log_every: int = 10, capture: bool = True):
        """``capture``: replay each step as one hipGraph (``graph.CapturedTrainStep`` -- what ``bench.py``'s headline number is
        measured through: ~0.4 ms of host time per step instead of ~20 ms of Python enqueueing ~640 kernels) whenever the step
        is the plain one: one GPU, ``training_step`` not overridden, no scheduler that changes the learning rate every step (a
        scalar passed by value is part of a graph's signature: each new value would re-capture).  Batches of a new shape run
        eagerly twice, then replay; the results are the eager step's (``tests/test_graph_gpu.py``)."""
        self.max_epochs, self.max_steps, self.log_every, self.capture = max_epochs, max_steps, log_every, capture
        self.captured_steps = 0
        self.device = torch.device(device or ("cuda" if torch.cuda.is_available() else "cpu"))
        self.amp_dtype = {"bf16": torch.bfloat16, "16": torch.float16, "32": None}[str(precision)]
        self.channels_last = channels_last
        self.log = logging.getLogger("lightning")

    def _autocast(self):
        return torch.autocast("cuda", dtype=self.amp_dtype, enabled=self.amp_dtype is not None and self.device.type == "cuda")

    def fit(self, model: RetinaNetModel):
        import torch.distributed as dist
        from .parallel import BucketedGradAllReduce
        model.prepare_data() if model.trn_ds is None else None
        model.to(self.device)
        if self.channels_last:
            model.to(memory_format=torch.channels_last)
        model.working_dtype = self.amp_dtype
        opt = model.configure_optimizers()
        optimizers, schedulers = (opt if isinstance(opt, tuple) else (opt, []))
        optimizer = optimizers[0]
        ddp = BucketedGradAllReduce(model.net) if dist.is_available() and dist.is_initialized() else None
        # precision "16" = fp16 autocast WITH dynamic loss scaling, like the reference's native-AMP run (Lightning precision=16):
        # fp16 gradients of a focal loss normalised by num_fg underflow without it
        # (under a gradient exchange: parallel.ExchangeGradScaler -- found_inf from the exchanged buckets, one decision for all ranks)
        from .parallel import ExchangeGradScaler
        scaler = (ExchangeGradScaler("cuda") if ddp is not None else torch.amp.GradScaler("cuda")) \
            if (self.amp_dtype == torch.float16 and self.device.type == "cuda") else None
        stepper = None
        if (self.capture and self.device.type == "cuda" and ddp is None and type(model).training_step is RetinaNetModel.training_step
                and not any(s["interval"] == "step" and "monitor" not in s for s in schedulers)):
            from .graph import CapturedTrainStep
            stepper = CapturedTrainStep(model.net, optimizer, None, amp_dtype=self.amp_dtype, scaler=scaler)
        step = 0
        for epoch in range(self.max_epochs):
            model.train()
            loader = model.train_dataloader()
            if hasattr(getattr(loader, "sampler", None), "set_epoch"):
                loader.sampler.set_epoch(epoch)
            for i, batch in enumerate(loader):
                batch = _to_device(batch, self.device)
                if stepper is not None:
                    # the same step (zero_grad -> autocast forward -> loss = sum of the dict -> backward -> optimizer.step) as ONE graph replay
                    images, targets, _ = batch
                    out = stepper(list(images), [{k: v for k, v in t.items() if isinstance(v, torch.Tensor) and k in ("boxes", "labels")}
                                                 for t in targets])
                    self.captured_steps = stepper.replays
                else:
                    from .losses import grad_prescale, scaler_prescale
                    with self._autocast(), grad_prescale(scaler_prescale(scaler, self.device)):
                        out = model.training_step(batch, i)
                    ddp.zero_grad() if ddp else optimizer.zero_grad(set_to_none=False)
                    if scaler is not None:
                        scaler.scale(out["loss"]).backward()
                        if ddp:
                            ddp.finish()
                            scaler.step_exchanged(optimizer, ddp)
                        else:
                            scaler.step(optimizer)
                        scaler.update()
                    else:
                        out["loss"].backward()
                        if ddp:
                            ddp.finish()
                        if ddp and type(optimizer).__name__ == "MasterSGD":
                            optimizer.step(grads=ddp.grad_views())       # fp32 bucket views of the bf16 working copies
                        else:
                            optimizer.step()
                step += 1
                if step % self.log_every == 0:
                    self.log.info("epoch %d step %d loss %.4f", epoch, step, float(out["loss"]))
                for s in schedulers:
                    if s["interval"] == "step" and "monitor" not in s:
                        s["scheduler"].step()
                if self.max_steps and step >= self.max_steps:
                    return step
            val = self._validate(model)
            for s in schedulers:
                if s["interval"] == "epoch":
                    s["scheduler"].step(val) if "monitor" in s and val is not None else (None if "monitor" in s else s["scheduler"].step())
        return step

    def _validate(self, model):
        loader = model.val_dataloader()
        if loader is None:
            return None
        import torch.distributed as dist
        tot, n = 0.0, 0
        was_training = model.training
        model.eval()                      # Lightning validates in eval mode: BN uses (and does not update) running statistics
        try:
            with torch.no_grad():
                for i, batch in enumerate(loader):
                    with self._autocast():
                        out = model.validation_step(_to_device(batch, self.device), i)
                    tot, n = tot + float(out["val_loss"]), n + 1
        finally:
            model.train(was_training)
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            t = torch.tensor([tot, float(n)], dtype=torch.float64, device=self.device)
            dist.all_reduce(t)            # every rank steps its ReduceLROnPlateau with the same number
            tot, n = float(t[0]), int(t[1])
        return tot / max(n, 1)

    def test(self, model: RetinaNetModel):
        model.to(self.device).eval()
        outs = []
        with torch.no_grad():
            for i, batch in enumerate(model.test_dataloader()):
                with self._autocast():
                    outs.append(model.test_step(_to_device(batch, self.device), i))
        return model.test_epoch_end(outs), outs
