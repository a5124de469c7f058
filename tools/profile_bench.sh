#!/bin/bash
# rocprofv3 summaries of the headline bench for profiles/: kernel-trace stats, then separate PMC passes
# for the K3 kernel (FETCH_SIZE / WRITE_SIZE cannot share a pass; MI355X_MICROARCH.md, HBM section).
# The GPU box has no .git: pass the commit the snapshot was taken at as RN_COMMIT (gpurun -- 'RN_COMMIT=<hash> tools/profile_bench.sh r05');
# it is written into k3_pmc.json, which bench.py quotes as the source of roofline.traffic.
tag=${1:-r03}
out=$GRAFT_REPO_ROOT/gpurun_out/profile_$tag
mkdir -p $out; rm -f $out/k3_pmc.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/pb; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb -- python bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-detect --timing-steps 2 > $out/bench_kernel_trace.log 2>&1
# (a helper process of the run can leave a second, tiny trace: take the largest file of each kind)
cp "$(ls -S /tmp/pb/*/*kernel_stats.csv | head -1)" $out/bench_kernel_stats.csv
python tools/steady_stats.py "$(ls -S /tmp/pb/*/*kernel_trace.csv | head -1)" $out/bench_kernel_stats_steady.csv 4 2 | tee $out/steady_summary.txt
grep "^{\"metric" $out/bench_kernel_trace.log | tail -1 > $out/bench_line_under_rocprof.json
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -f $out/k3_pmc.txt.tmp; rm -rf /tmp/pc; rocprofv3 --pmc $ctr --output-format csv -d /tmp/pc -- python bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-detect --timing-steps 1 > /tmp/pc.log 2>&1
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
python - $out <<PY
import json, os, re, sys, csv
out = sys.argv[1]
txt = open(out + "/k3_pmc.txt").read()
f = float(re.search(r"FETCH_SIZE per launch \(KB\) over \d+ launches: mean ([0-9.]+)", txt).group(1))
w = float(re.search(r"WRITE_SIZE per launch \(KB\) over \d+ launches: mean ([0-9.]+)", txt).group(1))
avg = None
for r in csv.DictReader(open(out + "/bench_kernel_stats_steady.csv")):
    if "loss_stream_kernel" in r["Name"]:
        avg = float(r["AverageNs"]) / 1e3
line = json.loads(open(out + "/bench_line_under_rocprof.json").read())
json.dump({"kernel": "loss_stream_kernel<bf16,gamma2,grad> (per-level rn_loss_fwd_bwd_levels, dense K = 90 logits)",
           "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w,
           "correction": "FETCH_SIZE x2 on gfx950 for 16 B/lane coalesced streams (MI355X_MICROARCH.md, HBM); WRITE_SIZE exact",
           "hbm_bytes_per_launch": int(round((2 * f + w) * 1024)),
           "algorithmic_bytes_per_launch": line["roofline"]["algorithmic_bytes_per_launch"],
           "kernel_avg_us_rocprof": avg, "commit": os.environ.get("RN_COMMIT", "unrecorded"),
           "source": "tools/profile_bench.sh (rocprofv3 --pmc, separate passes), bench.py --steps 3 --warmup 3 --no-detect (hipGraph replays + 1 eager step)"}, open(out + "/k3_pmc.json", "w"), indent=1)
print(open(out + "/k3_pmc.json").read())
PY
