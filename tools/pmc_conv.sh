#!/bin/bash
# PMC counters of the MFMA conv probe kernels (separate rocprofv3 pass; no trace domains)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/pc; rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --output-format csv -d /tmp/pc -- ./tools/conv_mfma_probe.bin 196608 > /tmp/pc.log 2>&1
python - <<'PY'
import csv, glob
from collections import defaultdict
agg = defaultdict(lambda: defaultdict(list))
for f in glob.glob("/tmp/pc/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:28s} mean {sum(v) / len(v):16.0f}  (n={len(v)})")
PY
