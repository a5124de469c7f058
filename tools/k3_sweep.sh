#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for pf in 2 4 8; do for nt in 0 1 2 3; do
  export RN_K3_PF=$pf RN_K3_NT=$nt
  rm -rf /tmp/ks; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python tools/bench_kernels.py k3 --reps 60 > /tmp/ks.log 2>&1
  python - <<PY
import csv,glob
f=glob.glob("/tmp/ks/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "loss_stream" in r["Name"]:
        print("PF=$pf NT=$nt", round(float(r["AverageNs"])/1e3,1), "us min", round(float(r["MinNs"])/1e3,1), "max", round(float(r["MaxNs"])/1e3,1))
PY
done; done
