#!/bin/bash
# per-kernel time of the fused BN kernels on ONE layer shape (rocprofv3): usage tools/bn_kernel_bw.sh N C H W res
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cat > /tmp/bn_one.py <<PY
import sys, torch
sys.path.insert(0, ".")
from pytorch_retinanet_amd.norm import FusedBatchNorm2d
N, C, H, W, res = $1, $2, $3, $4, $5
bn = FusedBatchNorm2d(C).cuda().train()
x = torch.randn(N, C, H, W, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
r = torch.randn_like(x).requires_grad_(True) if res else None
g = torch.randn_like(x)
for _ in range(12):
    y = bn(x, relu=True, residual=r); y.backward(g); x.grad = None
torch.cuda.synchronize()
PY
rm -rf /tmp/bk; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bk -- python /tmp/bn_one.py > /tmp/bk.log 2>&1
python - $1 $2 $3 $4 $5 <<'PY'
import csv, glob, sys
N, C, H, W, res = map(int, sys.argv[1:])
mb = N * C * H * W * 2 / 1e6
passes = {"bn_stats_partial": 1, "bn_apply": 3 if res else 2, "bn_bwd_partial": 2 + 1 / 16 if res else 2, "bn_bwd_apply": (4 + 1 / 16) if res else 3}
f = glob.glob("/tmp/bk/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    for k, p in passes.items():
        if k + "_kernel" in r["Name"]:
            us = float(r["AverageNs"]) / 1e3
            print(f"[{N},{C},{H},{W}] res={res} {k:18s} {us:8.1f} us  {p:4.2f} passes x {mb:6.1f} MB -> {p * mb / us / 1e3:5.2f} TB/s")
PY
