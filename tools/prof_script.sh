#!/bin/bash
# usage: tools/prof_script.sh <python script + args...> : per-kernel calls / avg / min us from a rocprofv3 kernel trace of the command
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/ks; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python3 "$@" > /tmp/ks.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("/tmp/ks/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:${PROF_ROWS:-28}]:
    print(r["Calls"].rjust(6), str(round(float(r["AverageNs"])/1e3,1)).rjust(8), "us  min", str(round(float(r["MinNs"])/1e3,1)).rjust(8), " max", str(round(float(r["MaxNs"])/1e3,1)).rjust(8), r["Name"][:110])
PY
