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
import os
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
    __slots__ = ("graph", "images", "targets", "losses", "calls", "failed", "match_state", "segments", "bucket_ids", "pool")

    def __init__(self):
        self.graph, self.images, self.targets, self.losses, self.calls, self.failed, self.match_state = None, None, None, None, 0, False, None
        self.segments, self.bucket_ids, self.pool = None, None, None


class MemsetNodeInGraph(RuntimeError):
    pass


class CaptureUnwindError(RuntimeError):
    "A failed capture could not be unwound: some stream of this process is still in capture mode (see ``_device_usable``)."


def _device_usable(dev) -> Optional[BaseException]:
    """After a failed capture: can this process still allocate, launch and run autograd on ``dev``?  A failure raised by the autograd
    engine's worker thread in the middle of a captured backward pass leaves streams the engine pulled into the capture (the legacy
    stream, through AccumulateGrad nodes created by an eager step) in capture mode on ROCm 7 even after the capture has been ended;
    every later allocation in that thread then fails with hipErrorStreamCaptureImplicit.  -> the exception the probe met, or None."""
    try:
        t = torch.ones(8, device=dev, requires_grad=True)
        (t * 2.0).sum().backward()                 # (runs on the engine's device thread)
        torch.cuda.synchronize(dev)
        return None
    except Exception as exc:                       # noqa: BLE001
        return exc


LAST_CENSUS: Dict[str, int] = {}          # node counts of the graphs captured so far in this process (bench.py reports them)


_WARNED_NO_HANDLE = False


def _new_graph() -> "torch.cuda.CUDAGraph":
    try:
        return torch.cuda.CUDAGraph(keep_graph=True)          # (keeps the hipGraph_t for the node census below)
    except TypeError:
        return torch.cuda.CUDAGraph()


def _uninspectable(g) -> None:
    """The raw hipGraph_t is not available (a torch without ``CUDAGraph(keep_graph=True)``): no node census, no memset repair.  On ROCm
    that is exactly the case the repair exists for, so the capture is REFUSED there (the step runs eagerly: slower, never wrong);
    elsewhere it is logged once."""
    global _WARNED_NO_HANDLE
    if getattr(torch.version, "hip", None):
        raise MemsetNodeInGraph("this torch cannot hand out the captured hipGraph_t (CUDAGraph(keep_graph=True) is missing): memset nodes "
                                "of third-party libraries can be neither counted nor replaced, and their replays are not trusted on ROCm")
    if not _WARNED_NO_HANDLE:
        _WARNED_NO_HANDLE = True
        _log.warning("captured graph cannot be inspected (no raw graph handle): node census and memset-node repair skipped")


def _repair_memset_nodes(g) -> None:
    """(Repairs, then refuses what is left.)  A captured step must not hold a MEMSET node: on ROCm 7.0 the memset nodes of a replayed hipGraph write garbage once the
    process has synchronised with the device and enqueued other blit work (round 4: K2's 32-byte ``num_fg`` clear scaled every loss
    after bench.py's warm-up synchronisation by 1 / garbage).  This package issues no memset (kernels clear what needs clearing,
    ``layers.FeaturePyramid._conv_own_bias``); MIOpen still does for some weight-gradient algorithms at some shapes -- such a step
    is refused here and runs eagerly (``CapturedTrainStep.__call__`` catches the exception)."""
    raw = getattr(g, "raw_cuda_graph", None)
    try:
        handle = raw() if raw is not None else None
    except Exception:                    # noqa: BLE001 -- graph not kept (older torch): nothing to inspect
        handle = None
    if handle:
        counts = (C.c_int64 * 4)()
        check(lib.rn_hipgraph_node_census(C.c_void_p(int(handle)), counts), "rn_hipgraph_node_census")
        LAST_CENSUS.update(kernel=LAST_CENSUS.get("kernel", 0) + int(counts[0]), memset=LAST_CENSUS.get("memset", 0) + int(counts[1]),
                           memcpy=LAST_CENSUS.get("memcpy", 0) + int(counts[2]), other=LAST_CENSUS.get("other", 0) + int(counts[3]))
        if counts[1] and not os.environ.get("RN_GRAPH_KEEP_MEMSET_NODES"):
            # the repair: each memset node becomes a kernel node with the same parameters, dependencies and dependents
            n = C.c_int64(0)
            check(lib.rn_hipgraph_replace_memset_nodes(C.c_void_p(int(handle)), C.byref(n)), "rn_hipgraph_replace_memset_nodes")
            LAST_CENSUS["memset_replaced"] = LAST_CENSUS.get("memset_replaced", 0) + int(n.value)
            check(lib.rn_hipgraph_node_census(C.c_void_p(int(handle)), counts), "rn_hipgraph_node_census")
        if counts[1]:
            raise MemsetNodeInGraph(f"the captured step holds {counts[1]} memset node(s) next to {counts[0]} kernels (a third-party "
                                    f"library cleared a buffer with hipMemsetAsync); replays of such a graph are not trusted on this ROCm")
    else:
        _uninspectable(g)
    inst = getattr(g, "instantiate", None)
    if handle and inst is not None:
        inst()


def retinanet_stage_of(name: str) -> int:
    """Backward stage of a ``Retinanet`` parameter (``BucketedGradAllReduce(stage_of=...)``): 0 = head + FPN, 1 = layer4 + layer3,
    2 = layer2 .. stem -- the order in which the staged backward pass finishes their gradients."""
    if name.startswith("backbone.backbone.layer4") or name.startswith("backbone.backbone.layer3"):
        return 1
    if name.startswith("backbone."):
        return 2
    return 0


class CapturedTrainStep:
    def __init__(self, net, optimizer, ddp=None, amp_dtype: Optional[torch.dtype] = torch.bfloat16, eager_steps: int = 2,
                 max_graphs: int = 4, enabled: bool = True, segmented: Optional[bool] = None, scaler=None):
        """``segmented`` (default: on whenever gradients are exchanged): the step with a gradient exchange as FOUR linear hipGraphs --
        forward + head / FPN backward | layer4, layer3 backward | layer2 .. stem backward | optimizer -- with the finished buckets'
        all-reduces issued EAGERLY on the process group's communication stream between the replays and the wait for them in
        front of the last segment.  Why not one graph: torch's process group runs the collectives on its own stream, a capture
        turns that into forked graph branches, and ROCm replays such a graph slower than Python enqueues the same kernels
        (DESIGN.md section 6); why not eager: ~20 ms of host time per 25 ms step.  The backward pass is cut at C3 / C4 / C5
        (``backbone.StageCuts``) and run as separate autograd calls, so each segment's capture begins and ends on this thread."""
        self.net, self.optimizer, self.ddp = net, optimizer, ddp
        # fp16 autocast: a torch.amp.GradScaler (the reference's precision=16 run is native AMP, demo.ipynb).  Its scale / growth
        # tracker are device tensors and optim.MasterSGD takes grad_scale / found_inf on the device, so scale -> backward -> step ->
        # update records into the graph like the rest of the step.  Under a gradient exchange: parallel.ExchangeGradScaler.
        self.scaler = scaler
        if scaler is not None and ddp is not None and not hasattr(scaler, "step_exchanged"):
            raise ValueError("under a gradient exchange the loss scaler must be a parallel.ExchangeGradScaler: found_inf has to come "
                             "from the exchanged buckets, or one rank skips a step the others take")
        self.segmented = (ddp is not None) if segmented is None else (bool(segmented) and ddp is not None)
        if self.segmented:
            ddp.deferred = True
        self.amp_dtype = amp_dtype
        self.eager_steps, self.max_graphs, self.enabled = max(int(eager_steps), 1), int(max_graphs), enabled
        self._entries: "OrderedDict[tuple, _Entry]" = OrderedDict()
        self.replays = 0          # steps served by a graph replay (diagnostics / tests)
        self.captures = 0

    # -- the step itself (identical in eager mode and under capture) -------------------------------------------------
    def _staged(self, images, targets, mark) -> Dict[str, Tensor]:
        """The step with the backward pass in stages; ``mark(i)`` runs after stage i's kernels have been enqueued (i = 0, 1, 2: the
        buckets completed so far may be exchanged; 3: after the optimizer).  ``ddp.finish()`` -- the wait for the exchange -- runs
        between mark(2) and the optimizer, outside any capture."""
        from .backbone import StageCuts
        net, opt, ddp = self.net, self.optimizer, self.ddp
        trunk = net.backbone.backbone
        cuts = StageCuts()
        ddp.zero_grad()
        trunk.stage_cuts = cuts
        from .losses import grad_prescale, scaler_prescale
        pre = scaler_prescale(self.scaler, images[0].device) if images[0].is_cuda else None
        try:
            with torch.autocast(images[0].device.type, dtype=self.amp_dtype, enabled=self.amp_dtype is not None, cache_enabled=False), \
                    grad_prescale(pre):
                losses = net(list(images), [dict(t) for t in targets])
                total = losses["classification_loss"] + losses["regression_loss"]
        finally:
            trunk.stage_cuts = None
        # stage 0: head + FPN; gradients of the cut leaves.  fp16: the scaled loss -- every later stage starts from scaled leaf gradients
        (self.scaler.scale(total) if self.scaler is not None else total).backward()
        mark(0)
        pairs = cuts.pairs                                        # [(C3, leaf), (C4, leaf), (C5, leaf)] in forward order
        # C3 / C4 join the data gradients of their consumers in the receiver's GEMM (pwconv._GradJoin); the FPN lateral's gradient was
        # produced in ANOTHER autograd pass (stage 0), so it is handed to the join here: the receiver accumulates into it (addmm_) and
        # autograd ASSIGNS the result to the leaf -- no 137 / 69 MB add per cut
        for _, leaf in pairs:
            j = getattr(leaf, "_rn_join", None)
            if j is not None and j._has_receiver and not j.recv_done and j.full is None and leaf.grad is not None:
                j.full, leaf.grad = leaf.grad, None
        for out, leaf in reversed(pairs[1:]):                     # stage 1: layer4, then layer3 (each adds to the leaf below it)
            if leaf.grad is not None:
                out.backward(leaf.grad)
                leaf.grad = None
        mark(1)
        if pairs and pairs[0][1].grad is not None:                # stage 2: layer2 .. stem
            pairs[0][0].backward(pairs[0][1].grad)
            pairs[0][1].grad = None
        mark(2)
        ddp.finish()
        if self.scaler is not None:
            self.scaler.step_exchanged(opt, ddp)                  # found_inf from the exchanged buckets: the same on every rank
            self.scaler.update()
        elif type(opt).__name__ == "MasterSGD":
            opt.step(grads=ddp.grad_views())
        else:
            opt.step()
        mark(3)
        return {"classification_loss": losses["classification_loss"].detach(), "regression_loss": losses["regression_loss"].detach(),
                "loss": total.detach()}

    def _step(self, images: Sequence[Tensor], targets: Sequence[Dict[str, Tensor]]) -> Dict[str, Tensor]:
        net, opt, ddp = self.net, self.optimizer, self.ddp
        if self.segmented:
            return self._staged(images, targets, lambda i: ddp.issue_ready() if i < 3 else None)
        if ddp is not None:
            ddp.zero_grad()
        else:
            opt.zero_grad(set_to_none=True)
        dev_type = images[0].device.type
        from .losses import grad_prescale, scaler_prescale
        # fp16: the loss kernel multiplies the GradScaler's scale into its gradients before it rounds them to fp16 (losses.grad_prescale)
        pre = scaler_prescale(self.scaler, images[0].device) if images[0].is_cuda else None
        with torch.autocast(dev_type, dtype=self.amp_dtype, enabled=self.amp_dtype is not None, cache_enabled=False), grad_prescale(pre):
            losses = net(list(images), [dict(t) for t in targets])
            total = losses["classification_loss"] + losses["regression_loss"]
        if self.scaler is not None:
            self.scaler.scale(total).backward()
            if ddp is not None:
                ddp.finish()
                self.scaler.step_exchanged(opt, ddp)
            else:
                self.scaler.step(opt)
            self.scaler.update()
            return {"classification_loss": losses["classification_loss"].detach(), "regression_loss": losses["regression_loss"].detach(),
                    "loss": total.detach()}
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

    def _capture_segments(self, e: _Entry, images, targets) -> None:
        "Four linear graphs sharing one memory pool; the exchange calls between them run eagerly, here as at every replay."
        dev = images[0].device
        e.images = [im.clone() for im in images]
        e.targets = [{k: (v.clone() if isinstance(v, Tensor) else v) for k, v in t.items()} for t in targets]
        e.match_state = ops.new_match_state(dev)
        e.pool = torch.cuda.graph_pool_handle()
        e.segments, e.bucket_ids = [], []
        ddp = self.ddp
        torch.cuda.synchronize()
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        state = {"g": None, "open": False}

        def begin():
            state["g"] = _new_graph()
            state["g"].capture_begin(pool=e.pool, capture_error_mode="thread_local")
            state["open"] = True

        def mark(i):
            state["g"].capture_end()
            state["open"] = False
            _repair_memset_nodes(state["g"])
            e.segments.append(state["g"])
            e.bucket_ids.append(ddp.issue_ready() if i < 3 else [])       # eager: the collectives of the buckets this segment completed
            if i < 2:
                begin()
            # (i == 2: ddp.finish() runs next -- eager: the compute stream waits for the communication stream -- and begins the
            # optimizer's segment, see finish_then_begin below)

        with ops.use_match_state(e.match_state), torch.cuda.stream(side):
            orig_finish = ddp.finish

            def finish_then_begin():
                orig_finish()
                begin()
            ddp.finish = finish_then_begin
            try:
                begin()
                e.losses = self._staged(e.images, e.targets, mark)
            except BaseException:
                # a failure inside an open segment (a MIOpen / check() error in forward or backward) must END that capture before
                # anything else touches the device from this thread: torch.cuda.graph.__exit__ does this for the one-graph path, the
                # hand-driven begin / mark pair has to do it itself.  Then the half-built segments and their pool are dropped.
                if state["open"]:
                    # the autograd engine may have pulled OTHER streams into the capture (an AccumulateGrad node created by an eager
                    # step lives on the stream of that step: the engine makes that stream wait for the capturing one and joins it
                    # back at the end of the pass -- which a failure in the middle of the pass skips).  Join what can be joined, or
                    # the capture ends "unjoined" and that stream stays in capture mode.
                    for other in {torch.cuda.default_stream(dev), torch.cuda.current_stream(dev)} - {side}:
                        try:
                            side.wait_stream(other)
                        except Exception:            # noqa: BLE001
                            pass
                    try:
                        state["g"].capture_end()
                    except Exception:                # noqa: BLE001 -- the capture is already invalid: ending it may raise again
                        pass
                    state["open"] = False
                for g in e.segments + [state["g"]]:
                    try:
                        if g is not None:
                            g.reset()
                    except Exception:                # noqa: BLE001
                        pass
                state["g"] = None
                e.segments, e.bucket_ids, e.pool, e.losses = None, None, None, None
                raise
            finally:
                ddp.finish = orig_finish
                torch.cuda.current_stream(dev).wait_stream(side)
        self.captures += 1

    def _replay_segments(self, e: _Entry) -> None:
        ddp = self.ddp
        for i in range(3):
            e.segments[i].replay()
            ddp.issue(e.bucket_ids[i])
        ddp.finish()
        e.segments[3].replay()

    def _capture(self, e: _Entry, images, targets) -> None:
        if self.segmented:
            return self._capture_segments(e, images, targets)
        e.images = [im.clone() for im in images]
        e.targets = [{k: (v.clone() if isinstance(v, Tensor) else v) for k, v in t.items()} for t in targets]
        torch.cuda.synchronize()
        g = _new_graph()
        # the fused loss kernel's state words: zero-filled here, OUTSIDE the capture, and owned by this entry (ops.use_match_state)
        e.match_state = ops.new_match_state(e.images[0].device)
        with ops.use_match_state(e.match_state), torch.cuda.graph(g, capture_error_mode="thread_local"):
            e.losses = self._step(e.images, e.targets)
        _repair_memset_nodes(g)
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
        if e.graph is None and e.segments is None:
            try:
                self._capture(e, images, targets)
            except Exception as exc:                          # noqa: BLE001 -- a step that cannot be captured still has to run
                _log.warning("train-step capture failed (%s: %s); this input signature runs eagerly", type(exc).__name__, exc)
                e.failed, e.graph, e.images, e.targets, e.losses, e.segments = True, None, None, None, None, None
                e.bucket_ids, e.pool, e.match_state = None, None, None
                torch.cuda.synchronize()                      # (the capture's side stream has been joined by _capture_segments' finally)
                if self.ddp is not None:
                    self.ddp.reset()
                stuck = _device_usable(images[0].device)
                if stuck is not None:
                    raise CaptureUnwindError(f"the failed capture ({type(exc).__name__}: {exc}) left the device in stream-capture mode "
                                             f"({type(stuck).__name__}: {stuck}); this process cannot run further GPU work") from exc
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
        if e.segments is not None:
            self._replay_segments(e)
        else:
            e.graph.replay()
        note_raw_write()                                      # parameters and BN statistics changed behind torch's back
        self.replays += 1
        return e.losses
