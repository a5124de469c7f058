#!/bin/bash
# usage: tools/prof_kernels.sh <bench_kernels args...> : per-kernel avg/min us from rocprofv3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/ks; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python tools/bench_kernels.py "$@" > /tmp/ks.log 2>&1
python - <<PY
import csv,glob
f=glob.glob("/tmp/ks/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:28]:
    print(r["Calls"].rjust(5), str(round(float(r["AverageNs"])/1e3,1)).rjust(8), "us  min", str(round(float(r["MinNs"])/1e3,1)).rjust(8), r["Name"][:100])
PY
