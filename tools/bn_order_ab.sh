#!/bin/bash
# A/B of the BN reduction kernels' row traversal (RN_BN_ORDER = 0 block ranges, 1 sweep, 2 reverse sweep; unset = by size)
cd $GRAFT_REPO_ROOT
for o in 0 1 2 ""; do
  echo "== tests RN_BN_ORDER=$o"; RN_BN_ORDER=$o python -m pytest tests/test_norm_gpu.py -x -q -m gpu 2>&1 | tail -1
done
for o in 0 2; do
  echo "== RN_BN_ORDER=$o"; RN_BN_ORDER=$o python tools/bn_fused_probe.py 2>&1 | grep "res=" 
done
for o in 0 2 "" 0 2 ""; do
  echo "== bench RN_BN_ORDER=$o"; RN_BN_ORDER=$o python bench.py --no-detect --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"
done
