#!/bin/bash
# per-kernel table of the cfg-4 predict line (bench.py --mode predict) under rocprofv3; usage: tools/prof_predict.sh [N]
n=${1:-30}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/pp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -- python bench.py --mode predict > /tmp/pp.log 2>&1
python - "$n" <<'PY'
import csv, glob, sys
f = glob.glob("/tmp/pp/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total GPU ms", round(tot / 1e6, 2), "kernels", sum(int(r["Calls"]) for r in rows))
for r in rows[:int(sys.argv[1])]:
    print("%8.2f ms %5.1f%% calls=%5s avg=%8.1fus %s" % (float(r["TotalDurationNs"]) / 1e6, 100 * float(r["TotalDurationNs"]) / tot, r["Calls"], float(r["AverageNs"]) / 1e3, r["Name"][:110]))
PY
