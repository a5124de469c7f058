"""``CapturedTrainStep`` -- one whole train step (zero_grad -> transform -> conv stack -> K1-K3 -> backward -> bucketed
all-reduce -> SGD) captured in a hipGraph and replayed with ONE host call per step.

Why: the step is ~700 kernel launches.  Enqueued one by one from Python they cost 21-26 ms of host time against a GPU
step of ~30 ms (``bench.py``: ``host_enqueue_ms_per_step``), so every millisecond the kernels get faster moves the step
closer to being host-bound.  The library never synchronises with the host and every shape of a step is static once the
image sizes and the GT counts are known, so the launch sequence is captured once per (input signature) and replayed
(reference analogue: none -- the reference launches ~40 torch ops per image from a Python loop with 4 host syncs each,
``retinanet/losses.py:66-126``).

Rules the capture relies on (all true of this package; checked by ``tests/test_graph_gpu.py``):
  * no host synchronisation and no host->device copy from temporary host memory inside the step: the GT offsets come from
    ``ops.gt_offsets`` (cached per count tuple), canvas masks / anchors / zero pages from their caches -- the eager steps
    that precede the capture fill them;
  * every pointer the kernels receive is a static input buffer, a parameter / optimizer state, or memory allocated from
    the graph's private pool during capture (same address at every replay);
  * scalars passed by value (learning rate, momentum, weight decay) are part of the signature: a change re-captures.

``__call__(images, targets)`` performs exactly one optimisation step and returns the loss dict (static tensors: read them
before the next call).  The first ``eager_steps`` calls with a new signature run eagerly (they are real steps and they warm
MIOpen's find, the caches and the optimizer state); the next one captures and replays.  At most ``max_graphs`` signatures
are kept (least recently used goes first); a capture that fails falls back to eager for that signature.
"""
import ctypes as C
import logging
from collections import OrderedDict
from typing import Dict, List, Optional, Sequence, Tuple

import torch
from torch import Tensor

from . import ops
from ._lib import check, lib
from .norm import note_raw_write

_log = logging.getLogger(__name__)


class _Entry:
    __slots__ = ("graph", "images", "targets", "losses", "calls", "failed", "match_state")

    def __init__(self):
        self.graph, self.images, self.targets, self.losses, self.calls, self.failed, self.match_state = None, None, None, None, 0, False, None


class CapturedTrainStep:
    def __init__(self, net, optimizer, ddp=None, amp_dtype: Optional[torch.dtype] = torch.bfloat16, eager_steps: int = 2,
                 max_graphs: int = 4, enabled: bool = True):
        self.net, self.optimizer, self.ddp = net, optimizer, ddp
        self.amp_dtype = amp_dtype
        self.eager_steps, self.max_graphs, self.enabled = max(int(eager_steps), 1), int(max_graphs), enabled
        self._entries: "OrderedDict[tuple, _Entry]" = OrderedDict()
        self.replays = 0          # steps served by a graph replay (diagnostics / tests)
        self.captures = 0

    # -- the step itself (identical in eager mode and under capture) -------------------------------------------------
    def _step(self, images: Sequence[Tensor], targets: Sequence[Dict[str, Tensor]]) -> Dict[str, Tensor]:
        net, opt, ddp = self.net, self.optimizer, self.ddp
        if ddp is not None:
            ddp.zero_grad()
        else:
            opt.zero_grad(set_to_none=True)
        dev_type = images[0].device.type
        with torch.autocast(dev_type, dtype=self.amp_dtype, enabled=self.amp_dtype is not None, cache_enabled=False):
            losses = net(list(images), [dict(t) for t in targets])
            total = losses["classification_loss"] + losses["regression_loss"]
        total.backward()
        if ddp is not None:
            ddp.finish()
            if type(opt).__name__ == "MasterSGD":
                opt.step(grads=ddp.grad_views())
            else:
                opt.step()
        else:
            opt.step()
        return {"classification_loss": losses["classification_loss"].detach(), "regression_loss": losses["regression_loss"].detach(),
                "loss": total.detach()}

    def _signature(self, images, targets) -> tuple:
        groups = tuple((g.get("lr"), g.get("momentum"), g.get("weight_decay"), g.get("dampening"), g.get("nesterov"))
                       for g in self.optimizer.param_groups)
        ims = tuple((tuple(im.shape), im.dtype, im.device) for im in images)
        tgs = tuple(tuple(sorted((k, tuple(v.shape), v.dtype) for k, v in t.items() if isinstance(v, Tensor))) for t in targets)
        mode = tuple(m.training for m in self.net.modules())
        frozen = tuple(p.requires_grad for p in self.net.parameters())       # (freezing / unfreezing layers changes the launch sequence)
        return (ims, tgs, groups, hash(mode), hash(frozen), self.amp_dtype)

    def _capture(self, e: _Entry, images, targets) -> None:
        e.images = [im.clone() for im in images]
        e.targets = [{k: (v.clone() if isinstance(v, Tensor) else v) for k, v in t.items()} for t in targets]
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        # the fused loss kernel's state words: zero-filled here, OUTSIDE the capture, and owned by this entry (ops.use_match_state)
        e.match_state = ops.new_match_state(e.images[0].device)
        with ops.use_match_state(e.match_state), torch.cuda.graph(g, capture_error_mode="thread_local"):
            e.losses = self._step(e.images, e.targets)
        e.graph = g
        self.captures += 1

    def __call__(self, images: Sequence[Tensor], targets: Sequence[Dict[str, Tensor]]) -> Dict[str, Tensor]:
        if not self.enabled or not images or not images[0].is_cuda:
            return self._step(images, targets)
        key = self._signature(images, targets)
        e = self._entries.get(key)
        if e is None:
            e = self._entries[key] = _Entry()
            while len(self._entries) > self.max_graphs:
                self._entries.popitem(last=False)            # drops the graph and its private memory pool
        else:
            self._entries.move_to_end(key)
        e.calls += 1
        if e.failed or e.calls <= self.eager_steps:
            return self._step(images, targets)
        if e.graph is None:
            try:
                self._capture(e, images, targets)
            except Exception as exc:                          # noqa: BLE001 -- a step that cannot be captured still has to run
                _log.warning("train-step capture failed (%s: %s); this input signature runs eagerly", type(exc).__name__, exc)
                e.failed, e.graph, e.images, e.targets, e.losses = True, None, None, None, None
                torch.cuda.synchronize()
                return self._step(images, targets)
        else:
            # the step's inputs into the graph's static buffers: one multi-tensor launch per dtype for what already lives on the
            # device (24 separate copies cost 0.19 ms per step), plain copies for the rest
            dsts, srcs = [], []
            pairs = list(zip(e.images, images)) + [(dt[k], v) for dt, st in zip(e.targets, targets) for k, v in st.items() if isinstance(v, Tensor)]
            for dst, src in pairs:
                if src.device == dst.device and src.dtype == dst.dtype and src.shape == dst.shape and src.is_contiguous() and dst.is_contiguous():
                    dsts.append(dst); srcs.append(src)
                else:
                    dst.copy_(src, non_blocking=True)
            if dsts:
                # (torch._foreach_copy_ still issues one hipMemcpyAsync per tensor: 24 x 11 us of GPU time)
                n = len(dsts)
                check(lib.rn_copy_many((C.c_void_p * n)(*[t.data_ptr() for t in srcs]), (C.c_void_p * n)(*[t.data_ptr() for t in dsts]),
                                       (C.c_int64 * n)(*[t.numel() * t.element_size() for t in dsts]), n,
                                       torch.cuda.current_stream(dsts[0].device).cuda_stream), "rn_copy_many")
        e.graph.replay()
        note_raw_write()                                      # parameters and BN statistics changed behind torch's back
        self.replays += 1
        return e.losses
