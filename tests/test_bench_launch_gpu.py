"""GPU: the N-GPU launch path of ``bench.py`` on the one GPU the driver's test box has.

``python bench.py --gpus 2 --backend gloo --share-gpu`` goes through exactly what the 8-GPU run goes through -- ``self_launch``
(``torch.distributed.run`` spawned before the parent touches the GPU), one process per rank, per-rank MIOpen user db, the process group,
``BucketedGradAllReduce`` with stage-aligned buckets and parameter broadcast, the step as four hipGraph segments with the all-reduces
issued between the replays, the barrier + MAX-reduced time, rank 0's JSON as the LAST stdout line -- with two differences forced by the
box: both ranks use ``cuda:0`` and the exchange runs over gloo (RCCL refuses two ranks on one device).  What is asserted is the launch
path and the line's bookkeeping, not a throughput.
"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("amp", ["bf16", "fp16"])
def test_bench_two_ranks_through_self_launch_on_one_gpu(amp):
    "(fp16: the same path with parallel.ExchangeGradScaler -- found_inf from the exchanged buckets -- captured into the optimizer segment)"
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-gpu", "--steps", "2", "--warmup", "3",
           "--no-cpu-baseline", "--no-detect", "--timing-steps", "1", "--amp", amp]      # (a gloo step moves 153 MB through the host: ~7 s)
    r = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1800)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [l for l in r.stdout.strip().splitlines() if l.strip()]
    line = json.loads(lines[-1])                                   # the JSON line is the LAST line of the job's stdout
    assert line["n_gpus"] == 2 and line["config"]["parallelism"] == "dp2" and line["config"]["global_batch"] == 16
    assert line["scaling"] == "weak" and line["steps"] == 2 and line["value"] > 0 and line["ms_per_step"] > 0
    assert abs(line["value"] - 16 * 1e3 / line["ms_per_step"]) <= 1e-2 * line["value"]      # value = ALL ranks' images / the max-over-ranks time
    assert line["exchange_backend"].startswith("gloo") and line["rccl_ranks"] == 0
    sl = line["step_launch"]
    assert sl["mode"].startswith("4 linear hipGraph segments") and sl["graph_replays_in_run"] >= 2 and sl["buckets"] >= 4
    gn = sl["graph_nodes"]                                          # (census BEFORE the repair: every memset node found was replaced by a kernel node)
    assert gn.get("memset", 0) == gn.get("memset_replaced", 0) and gn.get("kernel", 0) > 500
    assert 0 < line["config"]["final_loss"] < 100
    assert "cpu_baseline" not in line                               # rank 0 at N = 1 only
    assert line["dtype"] == amp and line["exchange_plan"]["buckets"] == sl["buckets"] and line["exchange_plan"]["bytes_total"] > 150e6
    if amp == "fp16":
        assert "GradScaler (scale 4096" in line["config"]["workload"]  # no step was skipped: every rank kept the initial scale
    assert line["roofline"]["frac"] and line["roofline"]["bound"] == "hbm"


def test_bench_fp16_matcher_stress_line_with_the_cpu_baseline():
    """BASELINE configs[4] as the documented command (`bench.py --amp fp16 --gt 500`) WITH the CPU baseline: round 5's
    line crashed with a NameError behind the timed region (a pasted block in `cpu_train_step_baseline`), and no test ran the flag."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--amp", "fp16", "--gt", "500", "--steps", "3", "--warmup", "3", "--no-detect",
           "--cpu-baseline-reps", "2", "--timing-steps", "1"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=2400)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    line = json.loads([l for l in r.stdout.strip().splitlines() if l.strip()][-1])
    assert line["dtype"] == "fp16" and "T=500" in line["config"]["workload"] and "GradScaler" in line["config"]["workload"]
    assert "fp16" in line["roofline"]["kernel"] and "bf16" not in line["roofline"]["kernel"]
    assert line["roofline"]["frac"] > 0.2 and 0 < line["config"]["final_loss"] < 100
    cb = line["cpu_baseline"]
    assert cb["value"] > 0 and cb["kind"] == "port" and cb["train_step"]["value"] > 0 and "T=500" in cb["sample"]
