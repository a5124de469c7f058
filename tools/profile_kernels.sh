#!/bin/bash
# rocprofv3 kernel-trace summary of the dense-head microbenchmarks (K2 at T=8 and T=500, K3 bf16/f32, the
# inference chain at the cfg-4 shape) -> gpurun_out/profile_r01/kernels_*; copy into profiles/.
tag=${1:-r03}; shift
out=$GRAFT_REPO_ROOT/gpurun_out/profile_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/pk; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -- python tools/bench_kernels.py ${@:-k2 k2_500 k3 k3_500 k3f32 detect} > $out/kernels_bench.log 2>&1
python - $out <<'PY'
import csv, glob, sys
out = sys.argv[1]
f = glob.glob("/tmp/pk/*/*kernel_stats.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "at::native" not in r["Name"] and "rocclr" not in r["Name"]]
with open(out + "/kernels_stats.csv", "w", newline="") as g:
    w = csv.writer(g); w.writerow(["Name", "Calls", "AverageNs", "MinNs", "MaxNs"])
    for r in rows: w.writerow([r["Name"][:120], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"]])
for r in rows: print(r["Calls"].rjust(5), str(round(float(r["AverageNs"]) / 1e3, 1)).rjust(8), "us  min", str(round(float(r["MinNs"]) / 1e3, 1)).rjust(8), r["Name"][:90])
PY
grep '^{"kernel"' $out/kernels_bench.log > $out/kernels_bench_lines.json
