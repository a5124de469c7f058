"""The TRAINING LOOP against the reference's: five optimisation steps (``/root/reference/model.py:112-119`` ``training_step`` = sum of the
loss dict, ``model.py:76-78`` + ``hparams.yaml:63-68`` = ``torch.optim.SGD`` with momentum 0.9 and weight decay 1e-3 over
``net.parameters()``) of the reference's own ``Retinanet`` on a seed-reproducible state dict, recorded by ``tests/golden/gen_golden.py traj``
into ``tests/golden/traj.npz``: per-step loss dicts, and for every parameter / BatchNorm buffer a fingerprint of final - initial (norm,
projection on a seeded direction, 16 seeded elements).  Two recorded runs: ``live`` (module in train(): BatchNorm on batch statistics,
Q18) and ``frozen`` (the module as constructed, ``backbone.py:347-351``: BatchNorm in eval()) on four same-size images, which a 2 x 2
data-parallel split reproduces exactly (per-image normaliser, Q8).  The learning rate of the fixture is 2e-5, not hparams' 1e-3
(``synth.TRAJ_OPT``: 1e-3 diverges on the synthetic weights and makes the 5-step map chaotic).

Held to it: (a) this package's eager step, (b) ``graph.CapturedTrainStep`` (2 eager steps + capture + 2 replays),
fp32, bf16 autocast on bf16 working copies and fp16 autocast on fp16 working copies under a GradScaler, all with ``MasterSGD``, (c) two ranks on the split batch through the segmented graphs
(``tools/ddp_two_rank.py --fixture traj``).  Tolerances: per-step losses fp32 rel <= 1e-4, 16-bit <= 2e-2; parameter
movement: see ``_BOUNDS`` (measured on MI355X, then fixed with margin).
"""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import synth

DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# Bounds (measured on MI355X, then fixed with margin; the numbers of a run are printed with -s):
#   total   relative L2 error of the WHOLE parameter movement, estimated from the fingerprints: a projection error (d - d_ref) . r on a
#           random N(0, 1) direction r has expectation |d - d_ref|^2, so sqrt(sum_k proj_err_k^2 / sum_k |d_ref_k|^2) estimates
#           |delta - delta_ref| / |delta_ref| over all parameters at once;
#   norm    worst per-parameter | |d| - |d_ref| | / |d_ref|;
#   proj, sample (fp32 with frozen BatchNorm only: the well-conditioned leg): worst per-parameter projection error / |d_ref| and sampled
#           element error / rms(d_ref).
# With live BatchNorm the deep layers see 4 x 5 positions x 2 images: their batch statistics amplify rounding differences (MIOpen's fp32
# algorithm choice, 16-bit activations) in a handful of BN parameters, which the per-parameter projections of small tensors show at once.
# fp32 / frozen has TWO outcomes, picked per process by MIOpen's find step (it times its fp32 solvers and keeps the fastest; a Winograd pick
# carries ~1e-4 of its own): 1.3e-5 / 6.4e-5 / 1.1e-3 / 4.9e-3 or 1.9e-4 / 1.4e-3 / 1.0e-2 / 0.19 (seen for the eager and for the captured
# leg, 3 of 8 processes).  The bounds admit both; the per-step losses (1e-4, measured 1e-6) do not move with it.
_BOUNDS = {("32", "frozen"): dict(total=1e-3, norm=5e-3, proj=5e-2, sample=0.5),
           ("32", "live"): dict(total=5e-3, norm=2e-2),                                  # measured 4.0e-4 / 6.3e-3
           ("bf16", "frozen"): dict(total=5e-2, norm=5e-2), ("bf16", "live"): dict(total=0.12, norm=0.3),      # 1.4e-2 / 1.3e-2; 4.0e-2 / 0.12
           ("16", "frozen"): dict(total=4e-2, norm=5e-2), ("16", "live"): dict(total=5e-2, norm=0.15)}         # 1.2e-2 / 1.3e-2; 1.3e-2 / 5.0e-2
_LOSS_RTOL = {"32": 1e-4, "bf16": 2e-2, "16": 2e-2}              # measured 1.4e-6 / 5.9e-3 / 4.7e-3 (VERDICT r4 asked 1e-3 / 2e-2)


def _fixture():
    return np.load(os.path.join(ROOT, "tests", "golden", "traj.npz"), allow_pickle=False)


def build_model(device, kind):
    "This package's Retinanet with the fixture's state dict, in the fixture's mode (live: train(); frozen: BatchNorm in eval())."
    import pytorch_retinanet_amd as P
    net = P.Retinanet(**synth.TRAJ)
    sd = net.state_dict()
    spec = [(k, tuple(v.shape), str(v.dtype).replace("torch.", "")) for k, v in sd.items()]
    for k, v in synth.state_dict_values(spec, seed=4242).items():
        sd[k] = torch.from_numpy(v)
    net.load_state_dict(sd)
    net = net.to(device).to(memory_format=torch.channels_last).train()
    if kind == "frozen":
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.eval()
    return net


def masters(net):
    return {n: (p.master if hasattr(p, "master") else p.data).detach().double().cpu().numpy().copy() for n, p in net.named_parameters()}


def buffers(net):
    return {n: b.detach().double().cpu().numpy().copy() for n, b in net.named_buffers() if "running_" in n}


def batch(kind, step, device, lo=0, hi=None):
    images, targets = synth.traj_inputs(kind, step)
    hi = len(images) if hi is None else hi
    timgs = [torch.from_numpy(i).to(device) for i in images[lo:hi]]
    ttgts = [{"boxes": torch.from_numpy(b).to(device), "labels": torch.from_numpy(l).to(device)} for b, l in targets[lo:hi]]
    return timgs, ttgts


def movement_errors(g, kind, tag, initial, final):
    "-> dict(total, norm, proj, sample) of final - initial against the fixture's fingerprints (see _BOUNDS) and the keys of the worst ones."
    keys = [str(k) for k in g[f"{kind}_{tag}_keys"]]
    assert keys == [k for k in initial if k in set(keys)] and len(keys) == len(initial), "parameter / buffer names differ from the reference's"
    worst, where = dict(norm=0.0, proj=0.0, sample=0.0), dict(norm="", proj="", sample="")
    se, sn = 0.0, 0.0
    for i, k in enumerate(keys):
        norm, proj, samp = float(g[f"{kind}_{tag}_norm"][i]), float(g[f"{kind}_{tag}_proj"][i]), g[f"{kind}_{tag}_samples"][i]
        d = (final[k] - initial[k]).reshape(-1)
        gn, gp, _, gs = synth.fingerprint(k, d)
        assert norm > 0, k
        rms = norm / np.sqrt(d.size)
        errs = dict(norm=abs(gn - norm) / norm, proj=abs(gp - proj) / norm, sample=float(np.max(np.abs(gs - samp))) / rms)
        se, sn = se + (gp - proj) ** 2, sn + norm ** 2
        for j in errs:
            if errs[j] > worst[j]:
                worst[j], where[j] = errs[j], k
    worst["total"] = float(np.sqrt(se / sn))
    return worst, where


def check_run(g, kind, precision, losses, initial, final, buf0=None, buf1=None):
    ref = g[f"{kind}_losses"]
    got = np.array(losses, np.float64)
    assert got.shape == ref.shape
    np.testing.assert_allclose(got, ref, rtol=_LOSS_RTOL[precision], err_msg=f"{kind} {precision}: per-step loss dict")
    worst, where = movement_errors(g, kind, "param", initial, final)
    print(f"[traj] {kind} {precision}: loss rel err {np.max(np.abs(got - ref) / np.abs(ref)):.2e}; parameter movement: total {worst['total']:.3e}, "
          f"worst norm {worst['norm']:.3e} ({where['norm']}), proj {worst['proj']:.3e} ({where['proj']}), sample {worst['sample']:.3e} ({where['sample']})")
    for name, bound in _BOUNDS[(precision, kind)].items():
        assert worst[name] <= bound, (kind, precision, name, worst, where)
    if buf0 is not None:
        wb, whb = movement_errors(g, kind, "buf", buf0, buf1)
        print(f"[traj] {kind} {precision}: BN running statistics: total {wb['total']:.3e}, worst norm {wb['norm']:.3e} ({whb['norm']})")
        b = _BOUNDS[(precision, kind)]
        assert wb["total"] <= b["total"] and wb["norm"] <= b["norm"], (kind, precision, "running statistics", wb, whb)


def test_fixture_is_the_reference_optimizer_and_covers_every_parameter():
    "CPU: the fixture carries hparams' momentum / weight decay, five steps, and one fingerprint per parameter of this package's model."
    g = _fixture()
    assert int(g["steps"]) == synth.TRAJ_STEPS == 5 and g["opt"].tolist() == [synth.TRAJ_OPT["lr"], 1e-3, 0.9]
    import yaml
    hp = yaml.safe_load(open(os.path.join(ROOT, "pytorch_retinanet_amd", "hparams.yaml")))["optimizer"]["params"]
    assert hp["weight_decay"] == synth.TRAJ_OPT["weight_decay"] and hp["momentum"] == synth.TRAJ_OPT["momentum"]
    import pytorch_retinanet_amd as P
    net = P.Retinanet(**synth.TRAJ)
    for kind in ("live", "frozen"):
        assert [str(k) for k in g[f"{kind}_param_keys"]] == [n for n, _ in net.named_parameters()]
        assert g[f"{kind}_losses"].shape == (5, 2) and np.all(g[f"{kind}_param_norm"] > 0)
        assert g[f"{kind}_losses"][:, 0].min() > 1.0 and g[f"{kind}_losses"].sum(1)[-1] < g[f"{kind}_losses"].sum(1)[0]      # it trains
    assert [str(k) for k in g["live_buf_keys"]] == [n for n, _ in net.named_buffers() if "running_" in n]


@pytest.mark.reference
def test_fixture_is_current_with_the_reference():
    "Build container: the first two steps of the live trajectory, re-run with the reference itself, equal the committed fixture."
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import _tv_standin
    R = _tv_standin.import_reference()
    g = _fixture()
    torch.manual_seed(0)
    ref = R.Retinanet(**synth.TRAJ)
    sd = ref.state_dict()
    spec = [(k, tuple(v.shape), str(v.dtype).replace("torch.", "")) for k, v in sd.items()]
    for k, v in synth.state_dict_values(spec, seed=4242).items():
        sd[k] = torch.from_numpy(v)
    ref.load_state_dict(sd)
    ref.train()
    opt = torch.optim.SGD(ref.parameters(), **synth.TRAJ_OPT)
    for step in range(2):
        images, targets = synth.traj_inputs("live", step)
        out = ref([torch.from_numpy(i) for i in images], [{"boxes": torch.from_numpy(b), "labels": torch.from_numpy(l)} for b, l in targets])
        total = sum(out.values())
        opt.zero_grad()
        total.backward()
        opt.step()
        np.testing.assert_allclose([float(out["classification_loss"]), float(out["regression_loss"])], g["live_losses"][step], rtol=1e-5)


def _run_single(kind, precision, captured):
    from pytorch_retinanet_amd.graph import CapturedTrainStep
    from pytorch_retinanet_amd.optim import MasterSGD, use_16bit_conv_weights
    net = build_model(DEV, kind)
    amp = {"32": None, "bf16": torch.bfloat16, "16": torch.float16}[precision]
    if amp is not None:
        use_16bit_conv_weights(net, amp)
    opt = MasterSGD(net.parameters(), **synth.TRAJ_OPT)
    initial, buf0 = masters(net), buffers(net)
    # fp16: dynamic loss scaling like the reference's precision=16 run; a start value that needs no back-off on this model, so that
    # all five steps are real steps (a skipped step would be a different trajectory) -- checked below
    scaler = torch.amp.GradScaler("cuda", init_scale=1024.0, growth_interval=10 ** 6) if precision == "16" else None
    step = CapturedTrainStep(net, opt, amp_dtype=amp, eager_steps=2, enabled=captured, scaler=scaler)
    losses = []
    for s in range(synth.TRAJ_STEPS):
        out = step(*batch(kind, s, DEV))
        losses.append([float(out["classification_loss"]), float(out["regression_loss"])])
    torch.cuda.synchronize()
    assert step.replays == (synth.TRAJ_STEPS - 2 if captured else 0) and step.captures == int(captured)
    if scaler is not None:
        assert float(scaler.get_scale()) == 1024.0, "a step was skipped (found_inf): the loss scale backed off"
    return losses, initial, masters(net), buf0, buffers(net)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["live", "frozen"])
@pytest.mark.parametrize("precision", ["32", "bf16", "16"])
@pytest.mark.parametrize("captured", [False, True], ids=["eager", "captured"])
def test_five_steps_reproduce_the_reference_trajectory(kind, precision, captured):
    g = _fixture()
    losses, p0, p1, b0, b1 = _run_single(kind, precision, captured)
    check_run(g, kind, precision, losses, p0, p1, *((b0, b1) if kind == "live" else ()))
    if kind == "frozen":
        for k in b0:
            assert np.array_equal(b0[k], b1[k]), k                        # BatchNorm in eval(): running statistics untouched


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["32", "bf16", "16"])
def test_two_ranks_on_the_split_batch_reproduce_the_reference_trajectory(tmp_path, precision):
    """(c): the frozen-BatchNorm trajectory of the reference on four images per step == two ranks with two images each through
    ``BucketedGradAllReduce`` + the segmented ``CapturedTrainStep`` (2 eager staged steps + capture + 2 replays), ranks bit-equal."""
    g = _fixture()
    out = str(tmp_path)
    env = dict(os.environ, MIOPEN_LOG_LEVEL="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tools", "ddp_two_rank.py"), "--out", out, "--precision", precision,
           "--segmented", "--fixture", "traj", "--steps", str(synth.TRAJ_STEPS)]
    r = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:]
    r0, r1 = torch.load(os.path.join(out, "rank0.pt")), torch.load(os.path.join(out, "rank1.pt"))
    assert r0["replays"] >= 2 and r1["replays"] >= 2
    for k, a in r0["params"].items():
        assert torch.equal(a, r1["params"][k]), f"ranks diverged at {k}"
    # the reference's loss at a step = the mean over the four images = the mean of the two ranks' losses (each the mean of its two)
    losses = 0.5 * (np.array(r0["loss_dicts"]) + np.array(r1["loss_dicts"]))
    initial = {k: v.double().numpy() for k, v in r0["initial"].items()}
    final = {k: v.double().numpy() for k, v in r0["params"].items()}
    check_run(g, "frozen", precision, losses, initial, final)
    if precision == "16":
        # fp16 under the exchange: parallel.ExchangeGradScaler (found_inf from the exchanged buckets); no step skipped, one scale on both ranks
        assert r0["scale"] == r1["scale"] == 1024.0, (r0["scale"], r1["scale"])
