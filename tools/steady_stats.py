#!/usr/bin/env python3
"""Steady-state per-kernel stats from a rocprofv3 kernel_trace.csv: only the dispatches of the last
N train steps (delimited by the K3 stream kernel) are kept, so MIOpen's find-mode/JIT warm-up
kernels do not pollute the table.  ``skip``: trailing steps to leave out (bench.py's --timing-steps run EAGERLY with an event pair
around every hand-written launch: same kernels, but 5-6 us of idle time at every pair -- they are not steady-state replays).
``marker``: substring of the kernel that delimits one iteration -- default the K3 stream kernel (train steps); the inference line
(``bench.py --mode predict``) is delimited by ``score_scan_kernel`` (one per ``predict()`` call: an iteration then runs from one call's
scan to the next call's, i.e. the detect tail of call i + the conv stack of call i + 1 -- the same kernels as one whole call).
usage: steady_stats.py <kernel_trace.csv> <out.csv> [steps] [skip] [marker]"""
import csv
import sys
from collections import defaultdict

src, dst = sys.argv[1], sys.argv[2]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
skip = int(sys.argv[4]) if len(sys.argv) > 4 else 0
rows = list(csv.DictReader(open(src)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marker = sys.argv[5] if len(sys.argv) > 5 else "loss_stream_kernel"
marks = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
assert len(marks) > steps + skip, "not enough steps in the trace"
lo, hi = marks[-steps - 1 - skip], marks[-1 - skip]          # [K3 of step n-steps-1, K3 of the last kept step): `steps` whole steps
sel = rows[lo:hi]
t0, t1 = int(sel[0]["Start_Timestamp"]), int(sel[-1]["End_Timestamp"])
agg = defaultdict(list)
for r in sel:
    agg[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
busy = sum(sum(v) for v in agg.values())
with open(dst, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "CallsPerStep", "TotalNsPerStep", "AverageNs", "PercentOfGpuBusy"])
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        w.writerow([k[:160], round(len(v) / steps, 2), round(sum(v) / steps), round(sum(v) / len(v)), round(100 * sum(v) / busy, 3)])
print(f"steps={steps} wall_ms_per_step={(t1 - t0) / steps / 1e6:.3f} gpu_busy_ms_per_step={busy / steps / 1e6:.3f} "
      f"busy_fraction={busy / (t1 - t0):.3f} kernels_per_step={len(sel) / steps:.0f}")
