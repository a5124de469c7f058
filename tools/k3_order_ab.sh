#!/bin/bash
# same-box A/B of K3's range order (RN_K3_ORDER=0 front to back, 1 back to front): K3 kernel time and the class-output data gradient that reads its output
cd $GRAFT_REPO_ROOT
for o in 0 1 0 1; do
  RN_K3_ORDER=$o python bench.py --no-detect --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('order $o', d['value'], d['ms_per_step'], 'K3 ms', d['roofline']['avg_launch_ms'], 'frac', d['roofline']['frac'], 'cls dgrad ms', d['conv_mfma']['own_kernels']['mfma_cls_output_dgrad']['ms'], 'cls wgrad', d['conv_mfma']['own_kernels']['mfma_cls_output_wgrad']['ms'])"
done
RN_K3_ORDER=1 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "loss" 2>&1 | tail -1
python bench.py --no-detect --no-cpu-baseline --force-ddp 2>/dev/null | tail -2 | cut -c1-120
