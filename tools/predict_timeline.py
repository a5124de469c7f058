#!/usr/bin/env python3
"""One steady-state predict() call as a kernel timeline (start offset, duration, gap before, grid, name) from a rocprofv3 kernel_trace.csv of
`bench.py --mode predict`; a call = the kernels between two score_scan_kernel launches (the scan closes the conv stack of its call).
usage: predict_timeline.py <kernel_trace.csv> <out.csv>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "score_scan_kernel" in r["Kernel_Name"]]
sel = rows[marks[-2] + 1:marks[-1] + 1]
t0 = int(sel[0]["Start_Timestamp"])
prev_end = t0
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["start_us", "dur_us", "gap_us", "grid", "wg", "name"])
    for r in sel:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        grid = "x".join(r.get(k, "") for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"))
        wg = "x".join(r.get(k, "") for k in ("Workgroup_Size_X", "Workgroup_Size_Y", "Workgroup_Size_Z"))
        w.writerow([round((s - t0) / 1e3, 1), round((e - s) / 1e3, 1), round((s - prev_end) / 1e3, 1), grid, wg, r["Kernel_Name"][:140]])
        prev_end = max(prev_end, e)
print("kernels in the call:", len(sel), " wall ms:", (int(sel[-1]["End_Timestamp"]) - t0) / 1e6)
