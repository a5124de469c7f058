#!/bin/bash
# rocprofv3 summaries of the headline bench for profiles/: kernel-trace stats, then separate PMC passes
# for the K3 kernel (FETCH_SIZE / WRITE_SIZE cannot share a pass; MI355X_MICROARCH.md, HBM section).
out=$GRAFT_REPO_ROOT/gpurun_out/profile_r01
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_kernel_trace.log 2>&1
cp /tmp/pb/*/*kernel_stats.csv $out/bench_kernel_stats.csv
python tools/steady_stats.py /tmp/pb/*/*kernel_trace.csv $out/bench_kernel_stats_steady.csv 3 | tee $out/steady_summary.txt
grep "^{\"metric" $out/bench_kernel_trace.log | tail -1 > $out/bench_line_under_rocprof.json
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pc; rocprofv3 --pmc $ctr --output-format csv -d /tmp/pc -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline > /tmp/pc.log 2>&1
  python - $ctr >> $out/k3_pmc.txt <<PY
import csv,glob,sys
vals=[]
for f in glob.glob("/tmp/pc/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "loss_stream_kernel" in r["Kernel_Name"] and r["Counter_Name"]==sys.argv[1]:
            vals.append(float(r["Counter_Value"]))
print(sys.argv[1], "per launch (KB) over", len(vals), "launches: mean", sum(vals)/max(len(vals),1), "min", min(vals), "max", max(vals))
PY
done
cat $out/k3_pmc.txt
