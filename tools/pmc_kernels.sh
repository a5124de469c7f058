#!/bin/bash
# usage: tools/pmc_kernels.sh <kernel-name-substring> <bench_kernels args...>
# Separate rocprofv3 --pmc passes (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE cannot share a pass).
pat=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "GRBM_GUI_ACTIVE"; do
  rm -rf /tmp/pm; rocprofv3 --pmc $ctrs --output-format csv -d /tmp/pm -- python tools/bench_kernels.py "$@" --reps 6 > /tmp/pm.log 2>&1
  python - "$pat" <<PY
import csv,glob,sys,collections
pat=sys.argv[1]
fs=glob.glob("/tmp/pm/*/*counter_collection.csv")
acc=collections.defaultdict(list)
for f in fs:
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items():
    v=v[len(v)//2:]   # skip warm-up launches
    print(f"{pat} {k}: mean={sum(v)/len(v):.4g} n={len(v)}")
PY
done
