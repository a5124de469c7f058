"""In-kernel timeline of nms_mask_kernel (csrc/nms.hip) at the detect chain's sparse-regime shape: 1 440 segments of ~120 boxes.
tools/probes/make_nms_stamped.sh builds tools/probes/libnms_stamped.so (the library with clock64() stamps in the kernel); this script runs
rn_nms_segments on it and prints the phases of a workgroup: segment header + keys (global), rank sort, box gather (global),
suppression matrix, greedy scan, compaction + stores -- and the launch's span."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lib = C.CDLL(os.path.join(ROOT, "tools", "probes", "libnms_stamped.so"))
S, per = 1440, 123
rng = np.random.default_rng(0)
lens = rng.poisson(per, S).clip(1, 250)
off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
N = int(off[-1])
c = rng.uniform(0, 1333, (N, 2)); wh = rng.uniform(16, 300, (N, 2))
boxes = torch.from_numpy(np.concatenate([c - wh / 2, c + wh / 2], 1).astype(np.float32)).cuda()
scores = torch.from_numpy(rng.uniform(0.05, 1, N).astype(np.float32)).cuda()
seg = torch.from_numpy(off).cuda()
keep = torch.empty(N, dtype=torch.int64, device="cuda"); cnt = torch.empty(S, dtype=torch.int32, device="cuda")
lib.rn_nms_workspace_bytes.restype = C.c_size_t
lib.rn_nms_workspace_bytes.argtypes = [C.c_int64, C.c_int]
wsb = lib.rn_nms_workspace_bytes(N, S)
ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
lib.rn_nms_segments.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
for _ in range(5):
    rc = lib.rn_nms_segments(boxes.data_ptr(), scores.data_ptr(), seg.data_ptr(), S, N, 0.5, keep.data_ptr(), cnt.data_ptr(), ws.data_ptr(), wsb, None)
    assert rc == 0, rc
torch.cuda.synchronize()
out = (C.c_ulonglong * (S * 8))()
lib.rn_debug_nms_stamps.argtypes = [C.c_void_p, C.c_int]
assert lib.rn_debug_nms_stamps(out, S * 8) == 0
st = np.array(out, dtype=np.int64).reshape(S, 8)[:, :7].astype(np.float64)
names = ["header + keys (global)", "rank sort", "box gather (global)", "suppression matrix", "greedy scan", "compaction + stores"]
d = np.diff(st, axis=1)
GHZ = 1e-3 * float(sys.argv[1]) if len(sys.argv) > 1 else 2.1      # clock64: shader cycles; MHz as argv[1]
for i, nm in enumerate(names):
    print(f"{nm:28s} median {np.median(d[:, i]) / GHZ / 1e3:7.2f} us   p90 {np.percentile(d[:, i], 90) / GHZ / 1e3:7.2f}   max {d[:, i].max() / GHZ / 1e3:7.2f}")
life = st[:, 6] - st[:, 0]
print(f"workgroup lifetime           median {np.median(life) / GHZ / 1e3:7.2f} us   max {life.max() / GHZ / 1e3:7.2f}")
print(f"launch span (first entry -> last exit) {(st[:, 6].max() - st[:, 0].min()) / GHZ / 1e3:7.2f} us; entries spread over {(st[:, 0].max() - st[:, 0].min()) / GHZ / 1e3:7.2f} us")
