#!/bin/bash
# STEADY-STATE per-kernel table of the cfg-4 predict line (bench.py --mode predict) under rocprofv3: only the last 3 predict() calls
# (delimited by score_scan_kernel) are kept, so MIOpen's find-mode kernels (naive_conv_*: 98 % of the raw table) stay out.
# usage: tools/prof_predict.sh [N] [out.csv]
n=${1:-30}; out=${2:-/tmp/predict_steady.csv}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/pp; rocprofv3 --kernel-trace --output-format csv -d /tmp/pp -- python bench.py --mode predict --no-cpu-baseline > /tmp/pp.log 2>&1
tail -1 /tmp/pp.log | cut -c1-400
python tools/steady_stats.py /tmp/pp/*/*kernel_trace.csv "$out" 3 0 score_scan_kernel
python - "$n" "$out" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[2])))
for r in rows[:int(sys.argv[1])]:
    print("%8.3f ms/call %6.2f%% calls=%7.1f avg=%9.1fus  %s" % (float(r["TotalNsPerStep"]) / 1e6, float(r["PercentOfGpuBusy"]), float(r["CallsPerStep"]), float(r["AverageNs"]) / 1e3, r["Name"][:100]))
PY
