#!/usr/bin/env python3
"""Headline benchmark: images/sec of a full RetinaNet-R50-FPN train step @800x1333 (BASELINE.json).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = transform -> ResNet50+FPN+heads forward (bf16 autocast, channels_last, MIOpen) ->
K1 anchors / K2 iou_match / K3 fused loss+grad (hand-written HIP) -> backward -> bucketed RCCL
all-reduce (N > 1) -> SGD step, on a per-GPU batch of 8 synthetic 3x800x1333 images resident in HBM.
Weak scaling: per-GPU batch is fixed, `value` = all ranks' images / max-over-ranks wall time.

The JSON line also carries
  roofline     -- the dominant dense-head kernel (K3 loss fwd+bwd): algorithmic bytes per launch /
                  its average duration measured with HIP events right around the kernel on the launch stream, in eager
                  replays of the same step run right after the timed region (the timed steps are hipGraph replays), plus
                  the same kernel on cold logits (`frac_cold`);
  cpu_baseline -- the CPU oracle (oracle/rn_oracle.c, OpenMP) timed on this box's host cores on the
                  same dense-head workload (rank 0, N=1 only; bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("MIOPEN_LOG_LEVEL", "1")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8, help="per-GPU batch (BASELINE configs[1]: 8)")
    ap.add_argument("--backbone", default="resnet50")
    ap.add_argument("--gt", type=int, default=8, help="GT boxes per image")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-ddp", action="store_true", help="use the bucketed all-reduce path even at world size 1 (validation)")
    ap.add_argument("--torch-sgd", action="store_true", help="fp32 parameters + torch.optim.SGD instead of fp32 masters + bf16 conv weights (optim.MasterSGD); same arithmetic")
    ap.add_argument("--cpu-baseline-reps", type=int, default=30, help="batches of the dense-head workload timed on the host (about 10 s of CPU work)")
    ap.add_argument("--no-detect", action="store_true", help="skip the inference-chain (decode + NMS + top-k) roofline line (BASELINE configs[3] shape)")
    ap.add_argument("--ddp-mode", default="segmented", choices=["segmented", "eager", "onegraph"],
                    help="how the step is launched when gradients are exchanged: segmented (default) = four linear hipGraphs with the buckets' "
                         "all-reduces issued eagerly between the replays (graph.CapturedTrainStep); eager = every kernel enqueued from Python; "
                         "onegraph = one capture incl. the collectives (forked stream branches: replays slower than eager on ROCm 7, DESIGN.md section 6)")
    ap.add_argument("--ddp-graph", action="store_true", help="same as --ddp-mode onegraph (round-3 spelling)")
    ap.add_argument("--set", default="", metavar="MODULE.NAME=VALUE[,...]",
                    help="A/B runs: set module-level switches of the package after import, e.g. --set biasact.NARROW_FWD=False,pwconv.EVAL_1X1_FUSED=False "
                         "(the switches README.md lists; the JSON line records what was set)")
    ap.add_argument("--no-graph", action="store_true", help="enqueue every kernel of every step from Python instead of replaying the captured hipGraph of the step (graph.CapturedTrainStep)")
    ap.add_argument("--mode", default="train", choices=["train", "predict"], help="predict: only the BASELINE configs[3] end-to-end inference line "
                    "(R101-FPN, 16 x 3x1333x1333, eval-mode folded BN, conv stack + decode + NMS + top-100 + rescale) as the JSON line")
    ap.add_argument("--no-predict", action="store_true", help="skip the configs[3] end-to-end predict line (roofline_other.predict_e2e)")
    ap.add_argument("--predict-backbone", default="resnet101")
    ap.add_argument("--predict-batch", type=int, default=16)
    ap.add_argument("--predict-size", type=int, default=1333)
    ap.add_argument("--predict-dtype", default="bf16", choices=["bf16", "fp16"], help="autocast dtype of the predict line (bf16: the head runs on "
                    "the hand-written MFMA kernels; fp16: MIOpen convolutions, same detect chain)")
    ap.add_argument("--amp", default="bf16", choices=["bf16", "fp16"], help="autocast dtype of the train step.  fp16 = BASELINE configs[4] / the reference's own "
                    "published run (native AMP, precision=16): the same MFMA kernels on v_mfma_*_f16, fp16 working copies of the conv weights, "
                    "torch.amp.GradScaler with optim.MasterSGD unscaling / skipping on the device; use --gt 500 for configs[4]'s GT count")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="process-group backend of the gradient exchange.  gloo: the launch path of "
                    "the N-GPU run (torch.distributed.run spawn before any GPU call, per-rank MIOpen db, bucket hooks, segmented graphs, rank-0 JSON with "
                    "the MAX-reduced time) on a box that cannot run RCCL with N ranks -- e.g. with --share-gpu")
    ap.add_argument("--share-gpu", action="store_true", help="every rank uses cuda:0 (RCCL refuses two ranks on one device: needs --backend gloo); "
                    "the line then reports launch-path correctness, its images/sec is N ranks time-slicing one GPU")
    ap.add_argument("--timing-steps", type=int, default=5, help="eager steps run AFTER the timed region with HIP events around the hand-written kernels (per-kernel figures of the JSON line)")
    return ap.parse_args()


class ClockSampler:
    """Shader clock / power / temperature of the GPU while the timed region runs, from the amdgpu sysfs files (no child process, no
    HIP call): `pp_dpm_sclk` (the level marked `*`), hwmon `freq1_input` (Hz), `power1_average` / `power1_input` (uW), `temp1_input`.
    A thread samples every `period` seconds; `summary()` gives min / mean / max and the sample count, or says what was missing.
    Evidence for "the part holds N GHz under MFMA load" statements and for the box-to-box spread of the headline number."""

    def __init__(self, pci=None, period=0.05):
        """``pci``: "dddd:bb:dd.f" of the GPU (the box shows every GPU of the node in sysfs, HIP only the leased one)."""
        import glob
        import threading
        self.period, self.samples, self._stop, self._thr = period, [], threading.Event(), None
        cards = []
        for d in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
            try:
                if open(os.path.join(d, "vendor")).read().strip() == "0x1002":
                    cards.append(d)
            except OSError:
                pass
        self.dev = None
        if pci:
            for d in cards:
                if os.path.basename(os.path.realpath(d)).lower() == pci.lower():
                    self.dev = d
        self.pci = pci
        self.hwmon = None
        if self.dev:
            hm = sorted(glob.glob(os.path.join(self.dev, "hwmon", "hwmon*")))
            self.hwmon = hm[0] if hm else None

    @staticmethod
    def _read(path):
        try:
            return open(path).read()
        except OSError:
            return None

    def _sample(self):
        rec = {}
        if self.dev:
            t = self._read(os.path.join(self.dev, "pp_dpm_sclk"))
            if t:
                for ln in t.splitlines():
                    if ln.rstrip().endswith("*"):
                        try:
                            rec["sclk_mhz"] = float(ln.split(":")[1].lower().replace("mhz", "").replace("*", "").strip())
                        except (IndexError, ValueError):
                            pass
        if self.dev:
            t = self._read(os.path.join(self.dev, "gpu_busy_percent"))
            if t:
                try:
                    rec["busy_pct"] = float(t)
                except ValueError:
                    pass
        if self.hwmon:
            f = self._read(os.path.join(self.hwmon, "freq1_input"))
            if f and "sclk_mhz" not in rec:
                try:
                    rec["sclk_mhz"] = float(f) / 1e6
                except ValueError:
                    pass
            for name in ("power1_average", "power1_input"):
                w = self._read(os.path.join(self.hwmon, name))
                if w:
                    try:
                        rec["power_w"] = float(w) / 1e6
                        break
                    except ValueError:
                        pass
            c = self._read(os.path.join(self.hwmon, "temp1_input"))
            if c:
                try:
                    rec["temp_c"] = float(c) / 1e3
                except ValueError:
                    pass
        return rec

    def start(self):
        import threading

        def run():
            while not self._stop.is_set():
                r = self._sample()
                if r:
                    self.samples.append(r)
                self._stop.wait(self.period)
        self._thr = threading.Thread(target=run, daemon=True)
        self._thr.start()
        return self

    def stop(self):
        self._stop.set()
        if self._thr:
            self._thr.join(timeout=2.0)
        return self.summary()

    def summary(self):
        out = {"source": f"sysfs {self.dev} (pci {self.pci})" if self.dev else f"no amdgpu sysfs device for pci {self.pci}",
               "samples": len(self.samples), "period_s": self.period}
        for key in ("sclk_mhz", "power_w", "temp_c", "busy_pct"):
            v = [r[key] for r in self.samples if key in r]
            if v:
                out[key] = {"min": round(min(v), 1), "mean": round(sum(v) / len(v), 1), "max": round(max(v), 1)}
        return out


def newest_pmc(root):
    "profiles/rNN_k3_pmc.json of the highest round (the counter passes are a separate rocprofv3 run; the file says which commit)."
    import glob
    import re
    best = None
    for f in glob.glob(os.path.join(root, "profiles", "r*_k3_pmc.json")):
        m = re.match(r"r(\d+)_k3_pmc\.json$", os.path.basename(f))
        if m and (best is None or int(m.group(1)) > best[0]):
            best = (int(m.group(1)), f)
    return best[1] if best else None


def synth_batch(batch, gt, seed, device):
    import synth
    rng = np.random.default_rng(seed)
    g = torch.Generator().manual_seed(seed)
    images = [torch.rand(3, 800, 1333, generator=g).to(device) for _ in range(batch)]
    targets = []
    for _ in range(batch):
        b, l = synth.gt_boxes(rng, gt, 800, 1333)
        targets.append({"boxes": torch.from_numpy(b).to(device), "labels": torch.from_numpy(l).to(device)})
    return images, targets


def k3_bytes(B, A, K, T, s):
    "Algorithmic bytes of one K3 launch (SURVEY 8d): logits r+w, box r+w, matches, GT rows."
    return B * (2 * A * K * s + 2 * A * 4 * s + A * 8 + T * 24)


def cpu_baseline(args, A, K):
    """Dense-head path (K1 anchors + K2 match + K3 loss fwd+bwd) of ONE batch of `args.batch` images on the host."""
    import oracle
    import synth
    oracle.build()
    rng = np.random.default_rng(0)
    B = args.batch
    cls, box = synth.head_outputs(rng, B, A, K)
    gtb, gtl = zip(*[synth.gt_boxes(rng, args.gt, 800, 1333) for _ in range(B)])
    cells = [oracle.cell_anchors(s, synth.ANCHOR_RATIOS) for s in synth.ANCHOR_SIZES]
    levels = synth.levels_for(800, 1344)
    times = []
    for _ in range(1 + args.cpu_baseline_reps):
        t0 = time.perf_counter()
        anc = oracle.anchors_emit(levels, cells, 0.0)
        m, _ = oracle.iou_match(anc, list(gtb))
        oracle.loss_fwd_bwd(cls, box, anc, list(gtb), list(gtl), m)
        times.append(time.perf_counter() - t0)
    t = float(np.median(times[1:]))
    return {"value": round(B / t, 3), "unit": "images/sec (dense-head K1-K3 only, fp32)", "cores": oracle.num_threads(),
            "kind": "port",
            "sample": f"{args.cpu_baseline_reps} reps of one batch of {B} images: anchors + iou_match + loss fwd/bwd at "
                      f"A={A}, K={K}, T={args.gt} (oracle/rn_oracle.c, OpenMP); conv stack NOT included",
            "ms_per_batch": round(t * 1e3, 2)}


def self_launch(args) -> int:
    """``python bench.py --gpus N`` without a launcher: start N ranks with torch.distributed.run from THIS process,
    which has not touched the GPU yet (a process that has initialised HIP must not exec / fork GPU children on this
    pool), relay rank 0's JSON line and return the launcher's exit code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def cpu_train_step_baseline(args):
    """The WHOLE train step on the host, one image: transform -> R50-FPN + heads (PyTorch's CPU kernels, fp32 -- what the
    reference itself runs on a CPU) -> anchors + IoU match + focal / smooth-L1 loss with gradients (the CPU oracle) ->
    backward through the conv stack -> SGD(momentum) step.  One warm-up step, then the median of 4 timed steps (8-9 s each:
    the >= 10 reps of SURVEY 8d would add two minutes to every bench run; the unit string says what was done)."""
    import oracle
    import synth
    import pytorch_retinanet_amd as P
    oracle.build()
    torch.manual_seed(0)
    net = P.Retinanet(num_classes=90, backbone_kind=args.backbone, pretrained=False, min_size=800, max_size=1333).train()
    opt = torch.optim.SGD(net.parameters(), lr=1e-3, weight_decay=1e-3, momentum=0.9)
    rng = np.random.default_rng(0)
    img = torch.rand(3, 800, 1333)
    gtb, gtl = synth.gt_boxes(rng, args.gt, 800, 1333)
    cells = [oracle.cell_anchors(sz, synth.ANCHOR_RATIOS) for sz in synth.ANCHOR_SIZES]
    times = []
    for _ in range(5):
        t0 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        il, _ = net.transform([img], None)
        fmaps, out = net._features(il.tensors)
        anc = oracle.anchors_emit(synth.levels_for(il.tensors.shape[-2], il.tensors.shape[-1]), cells, 0.0)
        m, _ = oracle.iou_match(anc, [gtb])
        o = oracle.loss_fwd_bwd(out["cls_preds"].detach().numpy(), out["bbox_preds"].detach().numpy(), anc, [gtb], [gtl], m)
        torch.autograd.backward([out["cls_preds"], out["bbox_preds"]], [torch.from_numpy(o["gcls"]), torch.from_numpy(o["gbox"])])
        opt.step()
        times.append(time.perf_counter() - t0)
    t = float(np.median(times[1:]))
    return {"value": round(1.0 / t, 4), "unit": "images/sec (whole train step, fp32, batch 1; median of 4 steps after 1 warm-up)",
            "cores": torch.get_num_threads(), "kind": "port",
            "sample": "median of 4 steps (after 1 warm-up) of ONE 3x800x1333 image: PyTorch CPU conv stack forward + backward + SGD, "
                      "dense head on oracle/rn_oracle.c", "s_per_step": round(t, 3)}


MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: bf16 dense MFMA peak ~2.5 PFLOP/s
FWD_GFLOP_PER_IMAGE = 510.3  # R50-FPN + heads forward @800x1344; x3 for forward + data + weight gradients


def detect_chain_line(device, with_cpu, regime="sparse"):
    """BASELINE configs[3] shape: B = 16 images, A = 338 454 anchors (1344 x 1344 padded input), K = 90, fp16 head outputs:
    rn_detect = score scan + decode + per-class NMS + top-100, events around the call.  SURVEY 8d's two regimes: ``sparse`` = logits
    N(-7, 1.2), about 11 k candidates per image (~120 per class); ``stress`` = N(-6, 1.5), about 2 % of all (anchor, class) pairs =
    ~640 k candidates per image (~7 k per class: the reference has no pre-NMS top-k, models.py:193-219, Q14).  CPU: the oracle's
    process_detections on ONE image of the same batch."""
    import synth
    from pytorch_retinanet_amd import ops
    from pytorch_retinanet_amd.anchors import AnchorGenerator
    B, A, K = 16, 338454, 90
    mean, std = (-7.0, 1.2) if regime == "sparse" else (-6.0, 1.5)
    ag = AnchorGenerator().to(device)
    anc = ops.anchors_emit(synth.levels_for(1344, 1344), list(ag.cell_anchors), 0.0)
    g = torch.Generator(device=device).manual_seed(1)
    cls = (torch.randn((B, A, K), device=device, generator=g) * std + mean).to(torch.float16)
    box = (torch.randn((B, A, 4), device=device, generator=g) * 0.1).to(torch.float16)
    hw = [(1333, 1333)] * B
    ncand = sum(int((torch.sigmoid(cls[b].float()) > 0.05).sum()) for b in range(B))
    cap = 1 << 18 if regime == "sparse" else 1 << 20            # per-image candidate capacity of the workspace (overflow would repeat the call)
    for _ in range(3):
        ops.detect(cls, box, anc, hw, 0.05, 1e-2, 0.5, 100, max_candidates=cap)
    torch.cuda.synchronize()
    ops.enable_timing(True)
    for _ in range(10):
        dets = ops.detect(cls, box, anc, hw, 0.05, 1e-2, 0.5, 100, max_candidates=cap)
    torch.cuda.synchronize()
    ev = ops.timing_events()["detect"]
    ops.enable_timing(False)
    ms = float(np.median([a.elapsed_time(b) for a, b in ev]))
    # the same six launches as hipGraph replays (20 back to back between one event pair): no launch gaps, no per-call events
    handle = []
    ops.detect_levels([cls], [box], anc, hw, 0.05, 1e-2, 0.5, 100, max_candidates=cap, enqueue_only=handle)
    ms_graph = graph_replay_ms(handle[0])
    nbytes = B * (A * K * 2 + A * 4 * 2 + A * 16)              # logits + deltas + anchors per image
    # bytes that can reach HBM: the logits once, ONE shared anchor set and one delta row only at the candidate anchors
    unique = B * A * K * 2 + ncand * (4 * 2 + 16)
    line = {"bound": "hbm", "kernel": "rn_detect chain (score_scan + seg_count + seg_scatter + nms_mask + nms_large + topk)",
            "workload": f"B={B} A={A} K={K} fp16, logits N({mean:g}, {std:g}) ({regime} regime), {ncand // B} candidates/image", "achieved": round(nbytes / (ms * 1e-3) / 1e9, 1),
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "avg_call_ms": round(ms, 4),
            "algorithmic_bytes_per_call": nbytes, "unique_bytes_per_call": unique,
            "frac_on_unique_bytes": round(unique / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "graph_replay_ms": round(ms_graph, 4), "frac_graph_replay": round(nbytes / (ms_graph * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "timing": "avg_call_ms: events around the eager rn_detect call (six launches from the host, as Retinanet.predict issues them); "
                      "graph_replay_ms: the same launches captured in a hipGraph and replayed 20 x back to back (no launch gaps)"}
    del handle
    cpu = None
    if with_cpu:
        import oracle
        oracle.build()
        c1, b1, a1 = cls[:1].float().cpu().numpy(), box[:1].float().cpu().numpy(), anc.cpu().numpy()
        times = []
        n_rep, n_warm = (13, 3) if regime == "sparse" else (4, 1)
        for _ in range(n_rep):
            t0 = time.perf_counter()
            ref = oracle.detect(c1, b1, a1, hw[:1])
            times.append(time.perf_counter() - t0)
        t = float(np.median(times[n_warm:]))
        same = np.array_equal(dets[0]["labels"].cpu().numpy(), ref[0]["labels"])       # the checker, checking
        assert same or regime != "sparse"      # (stress: ~7 k boxes per class -- a 1-ulp difference of exp() may flip one IoU > 0.5 test; reported, and held in tests/)
        cpu = {"labels_equal_oracle": bool(same), "value": round(1.0 / t, 3), "unit": "images/sec (decode + NMS + top-100 chain only, fp32 oracle)", "cores": oracle.num_threads(),
               "kind": "port", "sample": f"median of {n_rep - n_warm} reps after {n_warm} warm-up(s) of ONE image of the batch (A={A}, K={K}, {regime} regime); the GPU line processes 16 per call",
               "ms_per_image": round(t * 1e3, 2), "gpu_images_per_sec": round(B / (ms * 1e-3), 1)}
    del cls, box
    return line, cpu


R101_FWD_GFLOP_1344 = 1124.0   # R101-FPN + heads forward @1344x1344 per image


def predict_e2e_line(args, device, with_cpu):
    """BASELINE configs[3] END TO END ("conv stack ... included in an end-to-end figure"; reference path
    retinanet/models.py:245-272): ``Retinanet.predict`` on B synthetic 3 x S x S images resident in HBM -- fused transform ->
    R101-FPN + heads in eval mode (frozen BatchNorm folded into the convolutions) under 16-bit autocast -> per-level
    rn_detect_levels (score scan, decode, per-class NMS, top-100) -> rescale to the original sizes -> the <= 100 rows per image.
    Random-init weights (prior bias -4.6): the candidate count of the detect chain is whatever those logits give and is stated;
    the chain under a stated candidate load is `roofline_other.detect_chain`.  CPU leg: the same model's forward on ONE image
    with PyTorch's CPU kernels (fp32) + the oracle's detect chain on its head outputs."""
    import pytorch_retinanet_amd as P
    from pytorch_retinanet_amd.optim import use_bf16_conv_weights
    B, S = args.predict_batch, args.predict_size
    dt = torch.bfloat16 if args.predict_dtype == "bf16" else torch.float16
    torch.manual_seed(0)
    net = P.Retinanet(num_classes=90, backbone_kind=args.predict_backbone, pretrained=False, min_size=S, max_size=S)
    net = net.to(device).to(memory_format=torch.channels_last)
    g = torch.Generator().manual_seed(3)
    images = [torch.rand(3, S, S, generator=g).to(device) for _ in range(B)]
    # Random-init weights need two calibrations to stand in for a trained model.  (1) Eval-mode BatchNorm on identity running
    # statistics lets the activations of a 101-layer trunk explode (logits of +-2500, half of all (anchor, class) pairs become
    # candidates -- the case that exposed the seg_count race, tests/test_hip_parity.py): one train-mode pass over two of the images
    # with momentum 1 sets every layer's running statistics to its batch statistics.  (2) The class-output conv is rescaled so that
    # the logits are ~N(-7, 1.2): SURVEY 8d's sparse regime, ~11 k candidates per image for the decode + NMS + top-100 chain.
    with torch.no_grad(), torch.autocast("cuda", dtype=dt):
        bns = [m for m in net.modules() if isinstance(m, torch.nn.BatchNorm2d)]
        old = [m.momentum for m in bns]
        for m in bns:
            m.momentum = 1.0
        net.train()
        il, _ = net.transform(images[:2], None, **net._batch_layout())
        net._features(il.tensors)
        for m, mo in zip(bns, old):
            m.momentum = mo
        net.eval()
        il, _ = net.transform(images[:2], None, **net._batch_layout())
        _, out = net._features(il.tensors)
        head = net.retinanet_head.classification_head.class_subnet_output
        raw = out["cls_preds"].float() - float(head.bias.float().mean())          # (the bias is one constant: the prior, layers.py:175-178)
        f = 1.2 / max(float(raw.std()), 1e-12)
        head.weight.mul_(f)
        head.bias.fill_(-7.0 - f * float(raw.mean()))
        del out, raw, il
    if dt == torch.bfloat16:
        use_bf16_conv_weights(net)                       # (no per-forward weight casts)
    with torch.autocast("cuda", dtype=dt):
        for _ in range(2):
            dets = net.predict(images)
        torch.cuda.synchronize()
        times = []
        for _ in range(5):
            t0 = time.perf_counter()
            dets = net.predict(images)                   # (ends with the host read of the <= 100 rows per image: synchronous)
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
    t = float(np.median(times))
    with torch.no_grad(), torch.autocast("cuda", dtype=dt):
        il, _ = net.transform(images[:2], None, **net._batch_layout())
        _, out = net._features(il.tensors)
        ncand = int((torch.sigmoid(out["cls_preds"].float()) > 0.05).sum()) // 2
        del out, il
    Sp = (S + 31) // 32 * 32
    gflop = R101_FWD_GFLOP_1344 * (Sp * Sp) / (1344.0 * 1344.0) if args.predict_backbone == "resnet101" else None
    line = {"metric": f"images/sec RetinaNet-{args.predict_backbone.replace('resnet', 'R')}-FPN predict() end to end",
            "value": round(B / t, 2), "unit": "images/sec", "ms_per_batch": round(t * 1e3, 2),
            "workload": f"B={B} x 3x{S}x{S} (padded {Sp}x{Sp}), eval-mode folded BN, {args.predict_dtype} autocast, K=90, random-init weights "
                        f"(BN statistics calibrated on the batch, class logits rescaled to ~N(-7, 1.2)); {ncand} candidates/image into NMS, "
                        f"{int(np.mean([len(d['scores']) for d in dets]))} detections/image kept",
            "sample": "median of 5 predict() calls after 2 warm-ups, wall clock incl. the final device-to-host read",
            "conv_tflops": round(gflop * B / t / 1e3, 1) if gflop else None,
            "conv_frac_of_mfma_peak": round(gflop * B / t / 1e3 / MFMA_PEAK_TFLOPS, 4) if gflop else None}
    if with_cpu:
        import oracle
        import synth
        oracle.build()
        cpu_net = P.Retinanet(num_classes=90, backbone_kind=args.predict_backbone, pretrained=False, min_size=S, max_size=S).eval()
        img = images[0].cpu()
        ctimes = []
        with torch.no_grad():
            for _ in range(3):
                t0 = time.perf_counter()
                il, _ = cpu_net.transform([img], None)
                fmaps, out = cpu_net._features(il.tensors)
                anc = oracle.anchors_emit(synth.levels_for(il.tensors.shape[-2], il.tensors.shape[-1]),
                                          [oracle.cell_anchors(sz, synth.ANCHOR_RATIOS) for sz in synth.ANCHOR_SIZES], 0.0)
                oracle.detect(out["cls_preds"].numpy(), out["bbox_preds"].numpy(), anc, [tuple(il.image_sizes[0])])
                ctimes.append(time.perf_counter() - t0)
        ct = float(np.median(ctimes[1:]))
        line["cpu"] = {"value": round(1.0 / ct, 4), "unit": "images/sec (PyTorch CPU conv stack fp32 + oracle detect chain, batch 1)",
                       "cores": torch.get_num_threads(), "kind": "port", "sample": "median of 2 single-image passes after 1 warm-up",
                       "s_per_image": round(ct, 3)}
        del cpu_net
    del net
    torch.cuda.empty_cache()
    return line


K3_FORM_NAMES = {0: "rn_loss_fwd_bwd_levels_fin (one launch, special rows chunk by chunk, in-kernel finalize)",
                 1: "rn_loss_fwd_bwd_levels_rp form 1 (background stream + repair kernel + finalize: two launches)",
                 2: "rn_loss_fwd_bwd_levels_rp form 2 (one launch, special rows through one compact list, in-kernel finalize)"}
VALU_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32 vector peak (256 CUs x 128 lanes x 2 flop x 2.4 GHz)


def graph_replay_ms(fn, reps=20, rounds=5):
    """Device time of ``fn()`` (which enqueues kernels on the current stream and allocates its outputs) WITHOUT per-call event or launch
    overhead: one call is captured in a hipGraph, the graph is replayed ``reps`` times back to back between ONE pair of events, and the
    median over ``rounds`` such loops / reps is returned.  (Per-call event pairs in eager steps add ~5 us each and expose the launch gap:
    K2's 14 us kernel reads 36 us that way.)"""
    from pytorch_retinanet_amd import ops
    dev = torch.device("cuda", torch.cuda.current_device())
    fn()
    torch.cuda.synchronize()
    state = ops.new_match_state(dev)                         # the loss kernels' state words: zeroed once, outside the capture
    g = torch.cuda.CUDAGraph()
    with ops.use_match_state(state), torch.cuda.graph(g, capture_error_mode="thread_local"):
        keep = fn()
    g.replay()
    torch.cuda.synchronize()
    out = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / reps)
    del keep
    return float(np.median(out))


def match_plus_loss_line(device, B, T, dtype, label):
    """K2 + K3 as the pair the north-star names ("focal-loss + IoU-match kernels"), at the train shape (A = 201 600, K = 90, per-level
    tensors as the head writes them): ``gt_pack`` -> ``rn_iou_match_special_ex`` (flagged rows only) -> K3 (background stream + repair
    kernel incl. the finalize), timed as graph replays (``graph_replay_ms``).  Two figures: ``isolated`` -- the three calls back to
    back, logits as the previous replay left them --, and ``after_writer`` -- a copy kernel rewrites the logits in front of every
    replay (standing in for the class-output conv, which leaves their tail in the 256 MiB Infinity Cache exactly like that) and the
    writer's own replay time is subtracted: the in-step condition without a profiler."""
    import synth
    from pytorch_retinanet_amd import losses as L, ops
    from pytorch_retinanet_amd.anchors import AnchorGenerator
    K = 90
    shapes = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
    A = sum(h * w * 9 for h, w in shapes)
    s = 2 if dtype != torch.float32 else 4
    ag = AnchorGenerator().to(device)
    anc = ops.anchors_emit(synth.levels_for(800, 1344), list(ag.cell_anchors), 0.0)
    g = torch.Generator(device=device).manual_seed(2)
    src = [(torch.randn((B, h * w * 9, K), device=device, generator=g) - 4.6).to(dtype) for h, w in shapes]
    cls = [t.clone() for t in src]
    box = [(torch.randn((B, h * w * 9, 4), device=device, generator=g) * 0.1).to(dtype) for h, w in shapes]
    rng = np.random.default_rng(0)
    gtb, gtl = zip(*[synth.gt_boxes(rng, T, 800, 1333) for _ in range(B)])
    gb = [torch.from_numpy(b).to(device) for b in gtb]
    gl = [torch.from_numpy(l).to(device) for l in gtl]
    params = ops.make_loss_params(0.25, 2.0, 0.1)
    ops.gt_offsets([T] * B, device)                           # (cached: no upload inside the capture)

    def writer():
        for d, t in zip(cls, src):
            d.copy_(t)

    def k2_only():
        gt_boxes, gt_labels, off, nfg0 = ops.gt_pack(gb, gl, device)
        return ops.iou_match(anc, gt_boxes, off, B, 0.5, 0.4, want_special=True, flagged_only=True, zeroed_num_fg=nfg0), gt_boxes, gt_labels, off

    def pair():
        (m, nfg, sp), gt_boxes, gt_labels, off = k2_only()
        return ops.loss_fwd_bwd_levels(cls, box, anc, gt_boxes, gt_labels, off, m, nfg, params, True, special=sp,
                                       in_kernel_finalize=L.IN_KERNEL_FINALIZE, form=L.k3_form(B * T, B))

    def writer_then_pair():
        writer()
        return pair()

    (m, nfg, sp), gt_boxes, gt_labels, off = k2_only()
    n_fg = int(nfg.sum())
    n_special = int(sum(bin(int(w) & 0xFFFFFFFFFFFFFFFF).count("1") for w in sp.flatten().tolist())) if sp.numel() < 200000 else None

    def k3_only():
        return ops.loss_fwd_bwd_levels(cls, box, anc, gt_boxes, gt_labels, off, m, nfg, params, True, special=sp,
                                       in_kernel_finalize=L.IN_KERNEL_FINALIZE, form=L.k3_form(B * T, B))

    t_k2 = graph_replay_ms(k2_only)
    t_k3 = graph_replay_ms(k3_only)
    t_pair = graph_replay_ms(pair)
    t_w = graph_replay_ms(writer)
    t_wp = graph_replay_ms(writer_then_pair)
    k3b = k3_bytes(B, A, K, T, s)
    k2_alg = B * (A * 16 + T * 16 + A * 8)
    frac = lambda nb, ms: round(nb / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    pairs = B * A * T
    return {"bound": "hbm (K2 at T = 500: fp32 VALU)", "workload": f"{label}: B={B} A={A} K={K} T={T} {str(dtype).replace('torch.', '')}, {n_fg // B} matched rows/image"
            + (f", {n_special // B} special rows/image" if n_special is not None else ""),
            "kernels": "rn_copy_many (gt_pack) + rn_iou_match_special_ex (flagged rows only) + K3 " + K3_FORM_NAMES[L.k3_form(B * T, B)],
            "timing": "hipGraph of the calls replayed 20 x back to back between one event pair, median of 5 loops / 20",
            "k2_ms": round(t_k2, 4), "k3_ms": round(t_k3, 4), "pair_ms": round(t_pair, 4),
            "pair_ms_after_writer": round(t_wp - t_w, 4), "writer_ms": round(t_w, 4),
            "survey_8d_bytes": k2_alg + k3b, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac_on_survey_8d_bytes": frac(k2_alg + k3b, t_pair), "frac_after_writer": frac(k2_alg + k3b, t_wp - t_w),
            "k3_frac": frac(k3b, t_k3), "k3_algorithmic_bytes": k3b,
            "k2_pairs_per_sec": round(pairs / (t_k2 * 1e-3), 0),
            # ~25 fp32 VALU instructions per (anchor, GT box) pair ("K2 is VALU-bound at T = 500"): the share of the vector peak they are
            "k2_frac_of_fp32_valu_peak_at_25_flop_per_pair": round(25.0 * pairs / (t_k2 * 1e-3) / 1e12 / VALU_PEAK_TFLOPS, 4)}


def vendor_gemm_yardstick(device):
    """What the vendor GEMM (hipBLASLt through torch.mm, bf16) reaches on THIS box: a large square, and the head's 3x3 convolutions written as
    the explicit GEMMs they are -- positions x 2304 -> 256 / 810, the im2col matrix given for free -- beside `conv_mfma.own_kernels` (which gather
    their operand from the canvas).  Graph-replayed; tools/gemm_ceiling_probe.py is the longer table."""
    g = torch.Generator(device=device).manual_seed(0)
    out = {"timing": "torch.mm(bf16) captured in a hipGraph, 10 replays between one event pair, median of 3"}
    for key, M, K, N in (("square_8192", 8192, 8192, 8192), ("tower_pair_as_gemm", 2 * 179200, 2304, 256), ("cls_output_as_gemm", 179200, 2304, 810)):
        a = torch.randn((M, K), device=device, generator=g).to(torch.bfloat16)
        b = (torch.randn((N, K), device=device, generator=g) * 0.05).to(torch.bfloat16)
        ms = graph_replay_ms(lambda: torch.mm(a, b.t()), reps=10, rounds=3)
        out[key] = {"M": M, "K": K, "N": N, "ms": round(ms, 4), "tflops": round(2.0 * M * K * N / (ms * 1e-3) / 1e12, 1),
                    "frac_of_mfma_peak": round(2.0 * M * K * N / (ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4)}
        del a, b
    torch.cuda.empty_cache()
    return out


def k3_cold_line(device, B, T, nbytes):
    """K3 at the train shape on COLD logits (the isolated kernel: the in-step figure of `roofline` reads logits the
    class-output conv has just left in the 256 MiB Infinity Cache, walking them back to front).  Per-level bf16 tensors as
    the head writes them; a 1 GiB fill between launches evicts the cache; events right around the streaming kernel."""
    import synth
    from pytorch_retinanet_amd import losses as L, ops
    from pytorch_retinanet_amd.anchors import AnchorGenerator
    K = 90
    shapes = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
    ag = AnchorGenerator().to(device)
    anc = ops.anchors_emit(synth.levels_for(800, 1344), list(ag.cell_anchors), 0.0)
    g = torch.Generator(device=device).manual_seed(2)
    cls = [(torch.randn((B, h * w * 9, K), device=device, generator=g) - 4.6).to(torch.bfloat16) for h, w in shapes]
    box = [(torch.randn((B, h * w * 9, 4), device=device, generator=g) * 0.1).to(torch.bfloat16) for h, w in shapes]
    rng = np.random.default_rng(0)
    gtb, gtl = zip(*[synth.gt_boxes(rng, T, 800, 1333) for _ in range(B)])
    gt_boxes = torch.from_numpy(np.concatenate(gtb)).to(device)
    gt_labels = torch.from_numpy(np.concatenate(gtl)).to(device)
    off = ops.gt_offsets([T] * B, device)
    matches, num_fg, special = ops.iou_match(anc, gt_boxes, off, B, 0.5, 0.4, want_special=True)
    params = ops.make_loss_params(0.25, 2.0, 0.1)                      # config.py: alpha, gamma, smooth-L1 beta
    evict = torch.empty((1 << 30,), dtype=torch.uint8, device=device)
    ops.enable_timing(True)
    for _ in range(13):
        evict.fill_(1)
        ops.loss_fwd_bwd_levels(cls, box, anc, gt_boxes, gt_labels, off, matches, num_fg, params, special=special,
                                in_kernel_finalize=L.IN_KERNEL_FINALIZE, form=L.k3_form(B * T, B))
    torch.cuda.synchronize()
    ev = ops.timing_events()["loss_stream_kernel"][3:]
    ops.enable_timing(False)
    ms = float(np.median([a.elapsed_time(b) for a, b in ev]))
    return {"avg_launch_ms_cold": round(ms, 4), "achieved_cold": round(nbytes / (ms * 1e-3) / 1e9, 1),
            "frac_cold": round(nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "cold_sample": "median of 10 isolated launches after 3 warm-ups, 1 GiB fill between launches (Infinity Cache evicted)"}


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if rank != 0:
        os.dup2(2, 1)                            # only rank 0 owns stdout (see the end of main)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the dense-head path)")
    if args.share_gpu:
        if args.backend != "gloo":
            raise SystemExit("bench.py --share-gpu needs --backend gloo (RCCL refuses two ranks on one device)")
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1 or args.force_ddp:
        if "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29577", RANK="0", WORLD_SIZE="1")
        if args.backend == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)
    if args.gpus != world or (dist.is_initialized() and dist.get_world_size() != args.gpus):
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")

    import pytorch_retinanet_amd as P
    from pytorch_retinanet_amd import ops, tuning

    if args.set:
        import ast
        import importlib
        for item in args.set.split(","):
            path, value = item.split("=", 1)
            mod, name = path.strip().rsplit(".", 1)
            m = importlib.import_module("pytorch_retinanet_amd." + mod)
            if not hasattr(m, name):
                raise SystemExit(f"bench.py --set: pytorch_retinanet_amd.{mod} has no switch {name}")
            setattr(m, name, ast.literal_eval(value.strip()))
    tuning.use_shipped_miopen_db(rank)        # before the first conv: skip MIOpen's exhaustive search on a cold box
    tuning.enable_conv_autotune()             # channels_last needs find-mode picks (tuning.py has the numbers)
    if args.mode == "predict":
        if world != 1:
            raise SystemExit("bench.py --mode predict is a single-GPU line (images are independent: run N replicas for N GPUs)")
        line = predict_e2e_line(args, device, not args.no_cpu_baseline)
        line.update({"n_gpus": 1, "higher_is_better": True, "vs_baseline": None, "dtype": args.predict_dtype, "data": "synthetic",
                     "config": {"workload": line["workload"]}})
        print(json.dumps(line), flush=True)
        return
    torch.manual_seed(0)
    amp_dtype = torch.float16 if args.amp == "fp16" else torch.bfloat16
    # fp16: dynamic loss scaling like the reference's precision=16 run; the scale, the growth tracker, the unscale and the skip all
    # live on the device (optim.MasterSGD._step_supports_amp_scaling), so the step captures like the bf16 one
    # under a gradient exchange: parallel.ExchangeGradScaler (found_inf from the exchanged buckets: one decision for all ranks)
    scaler = None
    if args.amp == "fp16":
        if args.torch_sgd:
            raise SystemExit("bench.py --amp fp16 runs on optim.MasterSGD (fp16 working copies of fp32 masters)")
        scaler = P.ExchangeGradScaler("cuda", init_scale=4096.0) if (world > 1 or args.force_ddp) else torch.amp.GradScaler("cuda", init_scale=4096.0)
    net = P.Retinanet(num_classes=90, backbone_kind=args.backbone, pretrained=False, min_size=800, max_size=1333)
    net = net.to(device).to(memory_format=torch.channels_last).train()
    if args.torch_sgd:
        optimizer = torch.optim.SGD(net.parameters(), lr=1e-3, weight_decay=1e-3, momentum=0.9)   # hparams.yaml:63-68
    else:   # the same SGD on fp32 masters; conv weights live in the autocast dtype (what autocast would feed the convs anyway)
        from pytorch_retinanet_amd.optim import MasterSGD, use_16bit_conv_weights
        use_16bit_conv_weights(net, amp_dtype)
        optimizer = MasterSGD(net.parameters(), lr=1e-3, weight_decay=1e-3, momentum=0.9)
    from pytorch_retinanet_amd.graph import CapturedTrainStep, retinanet_stage_of
    ddp_mode = "onegraph" if args.ddp_graph else args.ddp_mode
    ddp = P.BucketedGradAllReduce(net, stage_of=retinanet_stage_of if ddp_mode == "segmented" else None) if (world > 1 or args.force_ddp) else None
    images, targets = synth_batch(args.batch, args.gt, seed=rank, device=device)

    # One step = graph.CapturedTrainStep: zero_grad -> autocast forward -> backward -> (bucketed all-reduce) -> SGD.  The first
    # two calls run eagerly (MIOpen find, caches, optimizer state), the third captures the step in a hipGraph, and every
    # later call is one graph replay (--no-graph: every call enqueues its ~700 kernels from Python).
    # With a gradient exchange the step is captured as FOUR linear graphs (forward + head / FPN backward | layer4, layer3 | layer2 .. stem |
    # optimizer) and the buckets' all-reduces are issued eagerly on the process group's stream between the replays: ONE capture would
    # turn the collectives into forked graph branches, which ROCm replays slower than Python enqueues the same kernels (world 1, same
    # box, round 3: 279 captured / 289 eager / 302 without the exchange), and eager steps cost ~20 ms of host time each.
    use_graph = not args.no_graph and (ddp is None or ddp_mode != "eager")
    stepper = CapturedTrainStep(net, optimizer, ddp, amp_dtype=amp_dtype, eager_steps=2, enabled=use_graph,
                                segmented=(ddp is not None and ddp_mode == "segmented"), scaler=scaler)
    n_warm = max(args.warmup, 3)               # (MIOpen's find, the caches and -- when capturing -- the capture itself stay out of the timed region)
    for _ in range(n_warm):
        stepper(images, targets)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    clocks = None
    if rank == 0:
        pr = torch.cuda.get_device_properties(local_rank)
        pci = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", 0), getattr(pr, "pci_device_id", 0))
        clocks = ClockSampler(pci).start()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = stepper(images, targets)
    host_enqueue = time.perf_counter() - t0      # (diagnostic: the host is done enqueueing here; the GPU usually is not)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    gpu_state = clocks.stop() if clocks else None
    final_loss = float(out["loss"])
    graph_replays = stepper.replays
    # per-kernel figures: the same step, enqueued eagerly with a pair of HIP events around every hand-written kernel (events
    # recorded inside a captured graph are dependency markers, not timestamps), right after the timed region, same data
    ops.enable_timing(True)
    for _ in range(max(args.timing_steps, 1)):
        stepper._step(images, targets)
    torch.cuda.synchronize()
    ev = ops.timing_events() or {}
    ops.enable_timing(False)
    timing_steps = max(args.timing_steps, 1)

    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    if rank == 0:
        A = sum(h * w * 9 for h, w in [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)])
        K, s = 90, 2
        kms = {k: float(np.mean([a.elapsed_time(b) for a, b in v])) for k, v in ev.items() if v}
        nbytes = k3_bytes(args.batch, A, K, args.gt, s)
        # columns the loss kernel really streams per anchor row: K when the class-output conv writes dense logits (the MFMA
        # path), the next multiple of 8 when MIOpen runs it with dead classes (RN_CLS_OUTPUT=miopen)
        K_run = K if net.retinanet_head.mfma_cls_output else net.retinanet_head.classification_head.padded_classes
        streamed = k3_bytes(args.batch, A, K_run, args.gt, s)
        k3_ms = kms.get("loss_stream_kernel")             # events recorded by the library right around the streaming kernel
        traffic, traffic_source = None, None
        pmc = newest_pmc(ROOT)
        if pmc:
            try:
                rec = json.load(open(pmc))
                traffic = rec.get("hbm_bytes_per_launch")
                traffic_source = (f"profiles/{os.path.basename(pmc)} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command at commit "
                                  f"{rec.get('commit', 'unrecorded')}; not measured in this run)")
            except Exception:          # noqa: BLE001
                traffic = None
        from pytorch_retinanet_amd import losses as L_
        form_run = L_.k3_form(args.batch * args.gt, args.batch)
        k3_name = (f"loss_bg_kernel<{args.amp}> + loss_repair_kernel<{args.amp}> (K3 focal + smooth-L1 loss, forward + gradients + finalize: background stream, "
                   f"then the special rows; HIP events around the PAIR of kernels)") if form_run == 1 else \
            f"loss_stream_kernel<{args.amp}{', list form' if form_run == 2 else ''}> (K3 focal + smooth-L1 loss, forward + gradients + in-kernel finalize; HIP events right around the kernel)"
        roof = {"bound": "hbm", "kernel": k3_name,
                "achieved": round(nbytes / (k3_ms * 1e-3) / 1e9, 1) if k3_ms else None, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(nbytes / (k3_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if k3_ms else None,
                "traffic": traffic, "traffic_source": traffic_source,
                "algorithmic_bytes_per_launch": nbytes, "avg_launch_ms": round(k3_ms, 4) if k3_ms else None,
                "streamed_bytes_per_launch": streamed, "classes_streamed": K_run,
                "call_ms_with_finalize": round(kms["loss_fwd_bwd"], 4) if "loss_fwd_bwd" in kms else None,
                "other_kernels_ms": {k: round(v, 4) for k, v in kms.items() if k in ("transform_batch", "iou_match", "gt_pack")}}
        # K2 (iou_match): one anchor set shared by the batch -> the anchors are read once per launch, so the bytes that can
        # reach HBM are A*16 + B*(T*16 + A*8); SURVEY 8d's per-image figure (anchors counted per image) is beside it
        k2_ms = kms.get("iou_match")
        k2_unique = A * 16 + args.batch * (args.gt * 16 + A * 8)
        k2_alg = args.batch * (A * 16 + args.gt * 16 + A * 8)
        roof_other = {"iou_match": {
            "bound": "hbm (launch-latency-bound at this size: DESIGN.md section 3, K2)", "kernel": "iou_match_batch_kernel + the num_fg memset (events around rn_iou_match_ex)",
            "achieved": round(k2_unique / (k2_ms * 1e-3) / 1e9, 1) if k2_ms else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(k2_unique / (k2_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if k2_ms else None,
            "unique_bytes_per_launch": k2_unique, "survey_8d_bytes_per_launch": k2_alg,
            "frac_on_survey_8d_bytes": round(k2_alg / (k2_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if k2_ms else None,
            "avg_call_ms": round(k2_ms, 4) if k2_ms else None, "pairs_per_sec": round(args.batch * A * args.gt / (k2_ms * 1e-3), 0) if k2_ms else None}}
        # the pair the north-star names ("focal-loss + IoU-match kernels"): K2's call + K3's call INCLUDING its one-block finalize, in the
        # step (events around the two library calls), on SURVEY 8d's algorithmic bytes and on the bytes that can reach HBM
        pair_ms = (k2_ms + kms["loss_fwd_bwd"]) if (k2_ms and "loss_fwd_bwd" in kms) else None
        if pair_ms:
            k2_now = A * 16 + args.batch * (args.gt * 16 + ((A + 63) // 64) * 8)        # anchors once + GT + the flag words (matches: flagged rows only)
            roof_other["match_plus_loss"] = {
                "bound": "hbm", "kernels": "rn_iou_match_special_ex (flagged rows only) + K3's call (events around each library call in EAGER steps: ~5 us of "
                                           "event / launch-gap overhead per call -- `graph_replay` below is the same pair without it)",
                "call_ms": round(pair_ms, 4), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "survey_8d_bytes": k2_alg + nbytes, "frac_on_survey_8d_bytes": round((k2_alg + nbytes) / (pair_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "bytes_r04_accounting": k2_unique + nbytes, "frac_on_r04_accounting": round((k2_unique + nbytes) / (pair_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "bytes_moved_now": k2_now + nbytes, "frac_on_bytes_moved_now": round((k2_now + nbytes) / (pair_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        # MFMA share: whole step on the model's useful flops, and the hand-written conv kernels alone (events around each launch)
        from pytorch_retinanet_amd import biasact
        step_s = elapsed / args.steps
        own = {k: (kms[k], biasact.MFMA_FLOP[k], len(ev[k]) / timing_steps) for k in kms if k.startswith("mfma_") and k in biasact.MFMA_FLOP}
        own_flop = sum(f * n for _, f, n in own.values())
        own_ms = sum(ms * n for ms, _, n in own.values())
        conv_mfma = {"peak_tflops": MFMA_PEAK_TFLOPS,
                     "whole_step_tflops": round(3 * FWD_GFLOP_PER_IMAGE * args.batch / step_s / 1e3, 1),
                     "whole_step_frac": round(3 * FWD_GFLOP_PER_IMAGE * args.batch / step_s / 1e3 / MFMA_PEAK_TFLOPS, 4),
                     "own_kernels_tflops": round(own_flop / (own_ms * 1e-3) / 1e12, 1) if own_ms else None,
                     "own_kernels_frac": round(own_flop / (own_ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4) if own_ms else None,
                     "own_kernels_ms_per_step": round(own_ms, 3), "own_kernels_share_of_step_flop": round(own_flop / (3 * FWD_GFLOP_PER_IMAGE * 1e9 * args.batch), 3),
                     "own_kernels": {k: {"ms": round(ms, 4), "calls_per_step": round(n, 2), "tflops": round(f / (ms * 1e-3) / 1e12, 1)} for k, (ms, f, n) in sorted(own.items())}}
        line = {
            "metric": "images/sec RetinaNet-R50-FPN train step @800x1333",
            "value": round(world * args.batch * args.steps / elapsed, 3),
            "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": n_warm,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "host_enqueue_ms_per_step": round(host_enqueue / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.amp, "data": "synthetic",
            "config": {"workload": f"RetinaNet-{args.backbone.replace('resnet', 'R')}-FPN {args.amp} train step, per-GPU batch "
                                   f"{args.batch} x 3x800x1333 (padded 800x1344), A=201600 anchors, K=90, T={args.gt} GT/img, "
                                   f"SGD(momentum{'' if args.torch_sgd else f', fp32 masters + {args.amp} conv weights'})"
                                   f"{', GradScaler (scale %g at the end)' % scaler.get_scale() if scaler is not None else ''}; random-init weights",
                       "global_batch": world * args.batch, "parallelism": f"dp{world}", "final_loss": round(final_loss, 4),
                       **({"switches_set": args.set} if args.set else {})},
            "roofline": roof, "roofline_other": roof_other, "conv_mfma": conv_mfma,
            "rccl_ranks": dist.get_world_size() if (dist.is_initialized() and args.backend == "nccl") else 0,
            "exchange_backend": (args.backend + (" (all ranks on cuda:0)" if args.share_gpu else "")) if dist.is_initialized() else None,
            # the gradient exchange of an N-rank run of this model (stage-aligned 32 MiB fp32 buckets: head + FPN | layer4, layer3 |
            # layer2 .. stem): what travels over xGMI per step and rank -- printed at N = 1 too, where nothing is exchanged
            "exchange_plan": ddp.plan() if ddp is not None else P.BucketedGradAllReduce.plan_for(net, stage_of=retinanet_stage_of),
            "gpu_state_during_timed_region": gpu_state,
            "step_launch": {"mode": ("4 linear hipGraph segments + eager all-reduces between them" if stepper.segmented else "hipGraph replay") if graph_replays else "eager",
                            "graph_replays_in_run": graph_replays, "buckets": ddp.num_buckets if ddp is not None else 0,
                            "graph_nodes": dict(__import__("pytorch_retinanet_amd.graph", fromlist=["LAST_CENSUS"]).LAST_CENSUS),
                            "per_kernel_events": f"{timing_steps} eager steps after the timed region"},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args, A, K)
            line["cpu_baseline"]["train_step"] = cpu_train_step_baseline(args)
        if world == 1 and not args.no_detect:
            del net, optimizer, stepper
            torch.cuda.empty_cache()
            line["roofline"].update(k3_cold_line(device, args.batch, args.gt, nbytes))
            line["conv_mfma"]["vendor_gemm_yardstick"] = vendor_gemm_yardstick(device)
            # the pair as graph replays (no per-call events): the train shape, and BASELINE configs[4] (fp16, 500 GT boxes per image)
            line["roofline_other"].setdefault("match_plus_loss", {})["graph_replay"] = match_plus_loss_line(device, args.batch, args.gt, amp_dtype, "train shape")
            line["roofline_other"]["cfg5"] = match_plus_loss_line(device, args.batch, 500, torch.float16, "BASELINE configs[4] (IoU-matcher stress)")
            det_line, det_cpu = detect_chain_line(device, not args.no_cpu_baseline)
            line["roofline_other"]["detect_chain"] = det_line
            if det_cpu is not None:
                line["cpu_baseline"]["detect_chain"] = det_cpu
            det_line, det_cpu = detect_chain_line(device, not args.no_cpu_baseline, regime="stress")
            line["roofline_other"]["detect_chain_stress"] = det_line
            if det_cpu is not None:
                line["cpu_baseline"]["detect_chain_stress"] = det_cpu
            if not args.no_predict:
                line["roofline_other"]["predict_e2e"] = predict_e2e_line(args, device, not args.no_cpu_baseline)
    else:
        line = None
    # The JSON line must be the LAST line of the job's stdout.  RCCL writes a "Librccl path" line to C stdout when it
    # unloads, and torch.distributed.run merges every rank's stdout: so the process group goes first, ranks != 0 never write to
    # stdout (fd 1 was pointed at stderr right after argument parsing), and rank 0 flushes C stdio, prints, and leaves without
    # running any library destructor.
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    if line is not None:
        sys.stdout.flush()
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:          # noqa: BLE001
            pass
        print(json.dumps(line), flush=True)
        if (world > 1 or args.force_ddp) and not os.environ.get("RN_BENCH_NORMAL_EXIT"):      # (a profiler needs the normal exit to write its files)
            os._exit(0)


if __name__ == "__main__":
    main()
