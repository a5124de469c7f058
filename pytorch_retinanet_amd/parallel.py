"""Data-parallel gradient exchange for one-process-per-GPU training over RCCL / xGMI.

The reference has no distributed code of its own (multi-GPU would come from
Lightning -> torch DDP -> NCCL); this is the MI355X-first
replacement for that implicit path, not a translation of it:

  * gradients live in a few large flat buckets (default 32 MiB: xGMI is 7
    point-to-point links per GPU, so fewer, larger collectives amortise the per-call
    latency and let RCCL stripe a bucket across all links);
  * after the exchange ``param.grad`` of every parameter is a VIEW into its bucket (no unflatten copy);
    on the way in, the freshly produced gradients of a whole bucket are gathered with ONE multi-tensor
    copy when its last gradient arrives (``zero_grad`` drops the gradients instead of zeroing 150 MB,
    so autograd assigns instead of accumulating -- no per-parameter ``add_`` kernels, no memsets);
  * buckets are filled in reverse-forward order (head -> FPN -> layer4 ... conv1)
    and a bucket's all-reduce is issued from the autograd hook of its last
    gradient, on the process group's communication stream, so it overlaps the
    rest of backward; ``finish()`` makes the compute stream wait before the
    optimizer step;
  * the loss is normalised per image (Q8), so no num_fg exchange is needed, and
    BatchNorm statistics stay per GPU like the reference (Q18).

Works with the "nccl" backend (= RCCL on ROCm) and with "gloo" on CPU (tests).
"""
from typing import List, Optional

import torch
import torch.distributed as dist
from torch import nn


def _flat_pair(view: torch.Tensor, grad: torch.Tensor) -> bool:
    "Same shape and strides, both dense: the copy is a walk over numel elements of the two storages."
    return (grad.is_cuda and grad.device == view.device and grad.shape == view.shape and grad.stride() == view.stride() and
            (grad.is_contiguous() or (grad.dim() == 4 and grad.is_contiguous(memory_format=torch.channels_last))))


def _gather(views, grads) -> None:
    """``views[i].copy_(grads[i])`` for a whole bucket.  On the GPU, 16-bit gradients going into fp32 views take ONE launch per 64
    tensors (``rn_cast_many_to_f32``) and same-dtype ones one launch per 64 (``rn_copy_many``): ``torch._foreach_copy_`` across
    dtypes issues one kernel per tensor -- 161 launches of ~5 us per step for the R50 model, 0.8 ms on the critical path of
    every bucket's all-reduce."""
    rest_v, rest_g, cast, same = [], [], [], []
    for v, g in zip(views, grads):
        if v.is_cuda and _flat_pair(v, g):
            if v.dtype == torch.float32 and g.dtype in (torch.bfloat16, torch.float16):
                cast.append((v, g)); continue
            if v.dtype == g.dtype:
                same.append((v, g)); continue
        rest_v.append(v); rest_g.append(g)
    if cast or same:
        import ctypes as C
        from ._lib import RN_BF16, RN_F16, check, lib
        dev = (cast or same)[0][0].device
        if dev.index != torch.cuda.current_device():
            torch.cuda.set_device(dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        for code, dt in ((RN_BF16, torch.bfloat16), (RN_F16, torch.float16)):
            grp = [(v, g) for v, g in cast if g.dtype == dt]
            if grp:
                n = len(grp)
                check(lib.rn_cast_many_to_f32((C.c_void_p * n)(*[g.data_ptr() for _, g in grp]), (C.c_void_p * n)(*[v.data_ptr() for v, _ in grp]),
                                              (C.c_int64 * n)(*[g.numel() for _, g in grp]), n, code, stream), "rn_cast_many_to_f32")
        if same:
            n = len(same)
            check(lib.rn_copy_many((C.c_void_p * n)(*[g.data_ptr() for _, g in same]), (C.c_void_p * n)(*[v.data_ptr() for v, _ in same]),
                                   (C.c_int64 * n)(*[g.numel() * g.element_size() for _, g in same]), n, stream), "rn_copy_many")
    if rest_v:
        torch._foreach_copy_(rest_v, rest_g)


class _Bucket:
    __slots__ = ("flat", "params", "pending", "work", "launched")

    def __init__(self, flat, params):
        self.flat, self.params = flat, params
        self.pending, self.work, self.launched = len(params), None, False


class BucketedGradAllReduce:
    def __init__(self, module: nn.Module, bucket_mb: float = 32.0, process_group=None, average: bool = True,
                 sync_params: bool = True, stage_of=None):
        """``stage_of(parameter_name) -> int`` (optional): buckets never span two stages -- ``graph.CapturedTrainStep`` runs the backward
        pass stage by stage (head + FPN | layer4, layer3 | layer2 .. stem) and issues a stage's buckets when its graph segment has
        been enqueued (``stage_index`` below; see ``deferred``)."""
        self.module = module
        # deferred = True: a completed bucket is gathered (kernels: capturable) but NOT exchanged from the autograd hook; it waits in
        # ``ready`` until the caller -- outside any stream capture -- calls ``issue_ready()``
        self.deferred = False
        self.ready: List["_Bucket"] = []
        self.group = process_group
        self.average = average
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.backend = dist.get_backend(process_group) if dist.is_initialized() else None
        named = [(n, p) for n, p in module.named_parameters() if p.requires_grad]
        named.reverse()                                      # ~ reverse forward order
        params = [p for _, p in named]
        stage = {id(p): (int(stage_of(n)) if stage_of is not None else 0) for n, p in named}
        self.buckets: List[_Bucket] = []
        self._owner = {}
        cap = int(bucket_mb * (1 << 20))
        group, size = [], 0
        self._bucket_stage: List[int] = []
        for p in params:
            nbytes = self._padded(p.numel()) * self._grad_dtype(p).itemsize
            if group and (size + nbytes > cap or self._grad_dtype(p) != self._grad_dtype(group[0]) or p.device != group[0].device
                          or stage[id(p)] != stage[id(group[0])]):
                self._make_bucket(group)
                self._bucket_stage.append(stage[id(group[0])])
                group, size = [], 0
            group.append(p)
            size += nbytes
        if group:
            self._make_bucket(group)
            self._bucket_stage.append(stage[id(group[0])])
        if sync_params and self.world > 1:
            self.sync_parameters()

    @staticmethod
    def _grad_dtype(p: torch.Tensor) -> torch.dtype:
        # bf16 working copies of fp32 master weights (optim.use_bf16_conv_weights) exchange their gradients in fp32
        return torch.float32 if hasattr(p, "master") else p.dtype

    ALIGN = 64          # elements: every view starts on a 256-byte (fp32) boundary -- rn_sgd_master_step and the
                        # multi-tensor copies want 16-byte-aligned tensors, and 9*K-element biases are not multiples of 4

    @classmethod
    def _padded(cls, n: int) -> int:
        return (n + cls.ALIGN - 1) // cls.ALIGN * cls.ALIGN

    def _make_bucket(self, params) -> None:
        total = sum(self._padded(p.numel()) for p in params)
        flat = torch.zeros(total, dtype=self._grad_dtype(params[0]), device=params[0].device)
        b = _Bucket(flat, params)
        off = 0
        for p in params:
            # same strides as the parameter (channels_last conv weights stay channels_last): autograd's gradient
            # layout contract, and the optimizer's multi-tensor kernels stay on their fast path
            dense = p.is_contiguous() or (p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last))
            view = torch.as_strided(flat, p.size(), p.stride(), off) if dense else flat[off: off + p.numel()].view_as(p)
            if view.dtype == p.dtype:
                p.grad = view
            off += self._padded(p.numel())
            self._owner[p] = (b, view)
            p.register_post_accumulate_grad_hook(self._hook)
        self.buckets.append(b)

    # -- called by autograd once per parameter per backward --------------------------------
    def _hook(self, p: torch.Tensor) -> None:
        b = self._owner[p][0]
        b.pending -= 1
        if b.pending == 0:
            self._launch(b)

    def _launch(self, b: _Bucket) -> None:
        # gather the bucket: gradients autograd assigned as fresh tensors (zero_grad() / set_to_none) are copied
        # into their views in one multi-tensor launch; gradients accumulated in place into the views are already there
        views, grads = [], []
        for p in b.params:
            view = self._owner[p][1]
            if p.grad is None:
                view.fill_(0)                  # parameter without a gradient this step (a fill kernel, not a memset: graph.py)
            elif p.grad.data_ptr() != view.data_ptr():
                views.append(view)
                grads.append(p.grad)
            if view.dtype == p.dtype:
                p.grad = view              # (a bf16 working copy keeps its bf16 .grad; its exchanged gradient is grad_views()[p])
        if views:
            _gather(views, grads)                  # also promotes bf16 gradients into fp32 buckets
        b.launched = True
        if self.deferred:
            self.ready.append(b)
            return
        self._exchange(b)

    def _exchange(self, b: _Bucket) -> None:
        if self.world == 1 and not (dist.is_available() and dist.is_initialized()):
            return
        if self.average and self.backend == "nccl":
            b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.AVG, group=self.group, async_op=True)
        else:
            b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def reset(self) -> None:
        "Forget a half-finished step (a capture that was abandoned): no bucket is pending, ready or in flight."
        self.ready = []
        for b in self.buckets:
            b.pending, b.launched, b.work = len(b.params), False, None

    def issue_ready(self) -> List[int]:
        "Deferred mode: exchange the buckets completed since the last call (NOT under stream capture); -> their indices."
        done = [self.buckets.index(b) for b in self.ready]
        for b in self.ready:
            self._exchange(b)
        self.ready = []
        return done

    def issue(self, indices) -> None:
        "Exchange the given buckets -- the replay of a captured step: the gather kernels ran inside the graph segment just enqueued."
        for i in indices:
            b = self.buckets[i]
            b.launched = True
            self._exchange(b)

    # -- call after backward, before optimizer.step ------------------------------------------
    def finish(self) -> None:
        for b in self.buckets:
            if not b.launched:                 # parameters that received no gradient this step
                self._launch(b)
        if self.deferred and self.ready:
            self.issue_ready()
        for b in self.buckets:
            if b.work is not None:
                b.work.wait()                  # compute stream waits for the communication stream
                b.work = None
                if self.average and self.backend != "nccl":
                    b.flat.div_(self.world)
            b.pending, b.launched = len(b.params), False

    def zero_grad(self) -> None:
        """Drop the gradients (no memset): the next backward assigns fresh tensors, which ``_launch`` gathers
        into the buckets.  ``module.zero_grad(set_to_none=False)`` -- zeroing the views in place -- works too."""
        for b in self.buckets:
            for p in b.params:
                p.grad = None

    def grad_views(self):
        "``{parameter: its (averaged) gradient in the bucket}`` -- what ``optim.MasterSGD.step(grads=...)`` consumes."
        return {p: view for p, (_, view) in self._owner.items()}

    def sync_parameters(self, src: int = 0) -> None:
        "Broadcast parameters and buffers from `src` so every rank starts from the same model."
        with torch.no_grad():
            for t in list(self.module.parameters()) + list(self.module.buffers()):
                if hasattr(t, "master"):
                    dist.broadcast(t.master, src=src, group=self.group)
                    t.data.copy_(t.master)
                else:
                    dist.broadcast(t.data, src=src, group=self.group)
        try:
            from . import biasact
            biasact.invalidate_dgrad_weights()       # (.data writes do not move the parameters' version counters)
        except (ImportError, OSError, RuntimeError):              # CPU-only host (gloo tests): no HIP library, nothing cached
            pass

    def found_inf(self) -> torch.Tensor:
        """f32[1] on the buckets' device: 1 if any EXCHANGED gradient is non-finite, else 0 (call after ``finish()``).  An overflow on ONE
        rank reaches every rank's averaged bucket (inf + x = inf, inf - inf = NaN), so every rank computes the same flag from its own
        copy of the reduced buckets -- one decision per step without a further collective, where the rank-local check of a stock
        ``GradScaler`` (it looks at ``param.grad``, the 16-bit gradients BEFORE the exchange) would let one rank skip a step the
        others take.  One multi-tensor pass over the buckets (``_amp_foreach_non_finite_check_and_unscale_`` with a scale of 1);
        capturable."""
        dev = self.buckets[0].flat.device
        found = torch.zeros(1, dtype=torch.float32, device=dev)
        one = getattr(self, "_one", None)
        if one is None or one.device != dev:
            one = self._one = torch.ones(1, dtype=torch.float32, device=dev)
        by_dtype = {}
        for b in self.buckets:
            by_dtype.setdefault(b.flat.dtype, []).append(b.flat)
        for flats in by_dtype.values():
            torch._amp_foreach_non_finite_check_and_unscale_(flats, found, one)
        return found

    def plan(self) -> dict:
        "The exchange as the N-rank run issues it: bucket count, bytes per backward stage and in total (bench.py prints it)."
        per_stage = {}
        for b, st in zip(self.buckets, getattr(self, "_bucket_stage", [0] * len(self.buckets))):
            per_stage[st] = per_stage.get(st, 0) + b.flat.numel() * b.flat.element_size()
        return {"buckets": len(self.buckets), "bytes_total": sum(self.bucket_bytes()), "bytes_per_stage": [per_stage[k] for k in sorted(per_stage)],
                "bucket_bytes": self.bucket_bytes()}

    @classmethod
    def plan_for(cls, module: nn.Module, bucket_mb: float = 32.0, stage_of=None) -> dict:
        """``plan()`` of the exchange this module WOULD get, without building it (no buffers, no hooks): what an N-rank run of the same
        model exchanges per step.  ``bench.py`` prints it in the single-GPU line, where no exchange exists."""
        named = [(n, p) for n, p in module.named_parameters() if p.requires_grad]
        named.reverse()
        cap = int(bucket_mb * (1 << 20))
        buckets, stages, size, cur = [], [], 0, None
        for n, p in named:
            nbytes = cls._padded(p.numel()) * cls._grad_dtype(p).itemsize
            key = (cls._grad_dtype(p), p.device, int(stage_of(n)) if stage_of is not None else 0)
            if cur is not None and (size + nbytes > cap or key != cur):
                buckets.append(size); stages.append(cur[2]); size = 0
            cur = key
            size += nbytes
        if cur is not None:
            buckets.append(size); stages.append(cur[2])
        per_stage = {}
        for b, st in zip(buckets, stages):
            per_stage[st] = per_stage.get(st, 0) + b
        return {"buckets": len(buckets), "bytes_total": sum(buckets), "bytes_per_stage": [per_stage[k] for k in sorted(per_stage)], "bucket_bytes": buckets}

    @property
    def num_buckets(self) -> int:
        return len(self.buckets)

    def bucket_bytes(self) -> List[int]:
        return [b.flat.numel() * b.flat.element_size() for b in self.buckets]


class ExchangeGradScaler(torch.amp.GradScaler):
    """``torch.amp.GradScaler`` for steps whose gradients go through ``BucketedGradAllReduce`` (fp16 autocast under data parallelism: the
    reference's only published run is Lightning ``precision=16``, ``demo.ipynb``; under DDP Lightning drives the same GradScaler).

        scaler.scale(loss).backward(); ddp.finish(); scaler.step_exchanged(optimizer, ddp); scaler.update()

    ``step_exchanged`` takes found_inf from the EXCHANGED buckets (``BucketedGradAllReduce.found_inf``: identical on every rank, so all
    ranks skip or step together and ``update()`` moves every rank's scale the same way -- one scale per step on every rank without a
    second collective) and steps on the buckets' views.  With ``optim.MasterSGD`` the unscale and the skip happen on the device
    (``rn_sgd_master_step_ex`` reads scale and flag): nothing synchronises and the whole sequence captures into the optimizer segment of
    ``graph.CapturedTrainStep``.  Other optimizers: the buckets are unscaled in place and the flag is read on the host, as the stock
    scaler does."""

    def step_exchanged(self, optimizer, ddp: BucketedGradAllReduce):
        if not self._enabled:
            return optimizer.step(grads=ddp.grad_views()) if type(optimizer).__name__ == "MasterSGD" else optimizer.step()
        from torch.amp.grad_scaler import OptState
        self._check_scale_growth_tracker("step_exchanged")
        state = self._per_optimizer_states[id(optimizer)]
        if state["stage"] is OptState.STEPPED:
            raise RuntimeError("step_exchanged() has already been called since the last update().")
        found = ddp.found_inf()
        state["found_inf_per_device"] = {found.device: found}
        retval = None
        if getattr(optimizer, "_step_supports_amp_scaling", False):
            optimizer.grad_scale, optimizer.found_inf = self._scale, found
            try:
                retval = optimizer.step(grads=ddp.grad_views()) if type(optimizer).__name__ == "MasterSGD" else optimizer.step()
            finally:
                del optimizer.grad_scale, optimizer.found_inf
        else:
            inv = self._scale.double().reciprocal().float()
            by_dtype = {}
            for b in ddp.buckets:
                by_dtype.setdefault(b.flat.dtype, []).append(b.flat)
            dummy = torch.zeros_like(found)
            for flats in by_dtype.values():
                torch._amp_foreach_non_finite_check_and_unscale_(flats, dummy, inv)
            if not float(found.item()):
                retval = optimizer.step()
        state["stage"] = OptState.STEPPED
        return retval
