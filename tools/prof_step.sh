#!/bin/bash
# steady-state per-kernel table of the train step (top N kernels); usage: tools/prof_step.sh [N] [pattern]
n=${1:-25}; pat=${2:-.}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/pb; rocprofv3 --kernel-trace --output-format csv -d /tmp/pb -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-detect --timing-steps 1 > /tmp/pb.log 2>&1
python tools/steady_stats.py /tmp/pb/*/*kernel_trace.csv /tmp/steady.csv 3 1
python - "$n" "$pat" <<'PY'
import csv, sys, re
n, pat = int(sys.argv[1]), sys.argv[2]
rows = [r for r in csv.DictReader(open("/tmp/steady.csv")) if re.search(pat, r["Name"])]
for r in rows[:n]:
    print("%8.3f ms/step %6.2f%% calls=%7.1f avg=%9.1fus  %s" % (float(r["TotalNsPerStep"]) / 1e6, float(r["PercentOfGpuBusy"]), float(r["CallsPerStep"]), float(r["AverageNs"]) / 1e3, r["Name"][:100]))
PY
