#!/bin/bash
# SQ counters of the hand-written conv kernels INSIDE the train step (separate rocprofv3 --pmc passes, no trace domains):
# MFMA-pipe busy cycles against SIMD busy cycles, LDS waits and bank conflicts -> gpurun_out/profile_$tag/conv_pmc.txt
tag=${1:-r03}
out=$GRAFT_REPO_ROOT/gpurun_out/profile_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
: > $out/conv_pmc.txt
for set in "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
  rm -rf /tmp/pc; rocprofv3 --pmc $set --output-format csv -d /tmp/pc -- python bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-detect --timing-steps 1 > /tmp/pc.log 2>&1
  python - >> $out/conv_pmc.txt <<'PY'
import csv, glob
from collections import defaultdict
agg = defaultdict(lambda: defaultdict(list))
for f in glob.glob("/tmp/pc/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if any(k in n for k in ("conv3x3_", "stem_fwd_kernel", "stem_wgrad_kernel", "wgrad3x3_kernel", "pw_gemm_kernel", "pw_wgrad_kernel", "pw_conv3_bwd_kernel",
                                "pw_block_out_conv1_kernel", "pw_dgrad_sums_kernel")):
            agg[n[:95]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(agg.items()):
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:28s} mean {sum(v) / len(v):16.0f}  (n={len(v)})")
PY
done
python - $out/conv_pmc.txt <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
cur, vals = None, {}
for line in txt.splitlines():
    if not line.startswith("   "):
        cur = line.strip(); vals.setdefault(cur, {})
    else:
        m = re.match(r"\s+(\S+)\s+mean\s+(\d+)", line)
        if m: vals[cur][m.group(1)] = float(m.group(2))
print("\n# MFMA pipe busy = (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs) / (SQ_BUSY_CYCLES / 32 shader engines): SQ_BUSY_CYCLES is summed over the 32")
print("# SEs (its per-SE value x the held clock = the kernel's duration), the MFMA counter over all 1024 SIMDs (cross-check: the tower launch")
print("# issues 29.9 M v_mfma_f32_16x16x32_bf16 x 16 cycles = 478 M SIMD-cycles)")
for k, d in sorted(vals.items()):
    if "SQ_VALU_MFMA_BUSY_CYCLES" in d and d.get("SQ_BUSY_CYCLES"):
        print(f"{k[:80]:80s}  mfma_busy {d['SQ_VALU_MFMA_BUSY_CYCLES'] / (32.0 * d['SQ_BUSY_CYCLES']):.3f}")
PY
cat $out/conv_pmc.txt | tail -30
