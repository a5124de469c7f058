import ctypes, torch, numpy as np, collections
from pytorch_retinanet_amd import biasact
from pytorch_retinanet_amd._lib import lib
dev = torch.device("cuda:0")
raw = ctypes.CDLL(lib._name) if hasattr(lib, "_name") else None
import pytorch_retinanet_amd._lib as L
so = ctypes.CDLL(L.__file__.replace("_lib.py", "libretinanet_hip.so"))
so.rn_debug_set_stamps.argtypes = [ctypes.c_void_p]
shapes = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
N = 8
feats = [torch.randn(N, 256, h, w, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last) for h, w in shapes]
cv = biasact.Canvas.of(feats, pad=1)
x = biasact.pack_levels(cv, feats)
ws = [(torch.randn(256, 256, 3, 3, device=dev) * 0.02).to(torch.bfloat16).contiguous(memory_format=torch.channels_last) for _ in range(2)]
bs = [torch.randn(256, device=dev) * 0.1 for _ in range(2)]
stamps = torch.zeros((8192, 12), dtype=torch.int64, device=dev)
import sys
variant = sys.argv[1] if len(sys.argv) > 1 else "plain"
ws2 = [(torch.randn(256, 256, 3, 3, device=dev) * 0.02).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True) for _ in range(2)]
bs2 = [(torch.randn(256, device=dev) * 0.1).requires_grad_(True) for _ in range(2)]


def run_once(arm):
    """plain: forward without ReLU bits (no grad).  fwd_bits: the forward of a linked layer (writes the ReLU bits of its output).
    dgrad_relu: the data gradient of the layer above it (applies those bits, sums the columns for the bias gradient)."""
    if variant == "plain":
        with torch.no_grad():
            if arm:
                so.rn_debug_set_stamps(ctypes.c_void_p(stamps.data_ptr()))
            biasact.tower_conv_pair(x, x, ws[0], ws[1], bs[0], bs[1], cv.mask)
        return
    link = biasact.TowerLink()
    w0 = [w.detach().requires_grad_(True) for w in ws]
    a, b = biasact.tower_conv_pair(x, x.clone(), w0[0], w0[1], bs2[0], bs2[1], cv.mask, None, link)      # layer 0: inputs need no gradient
    if variant == "fwd_bits":
        return                                                      # (the caller armed the stamps before: layer 0's forward is the last tile launch)
    a2, b2 = biasact.tower_conv_pair(a, b, ws2[0], ws2[1], bs2[0], bs2[1], cv.mask, link, None)
    loss = (a2.float().sum() + b2.float().sum())
    if arm:
        torch.cuda.synchronize()
        so.rn_debug_set_stamps(ctypes.c_void_p(stamps.data_ptr()))
    loss.backward()                                                 # tile launches in backward: layer 1's data gradient only (layer 0's inputs are leaves without grad)


for _ in range(3):
    run_once(False)
torch.cuda.synchronize()
if variant == "fwd_bits":
    so.rn_debug_set_stamps(ctypes.c_void_p(stamps.data_ptr()))
run_once(True)
torch.cuda.synchronize()
so.rn_debug_set_stamps(ctypes.c_void_p(0))
print("variant", variant)
s = stamps.cpu().numpy()
s = s[s[:, 4] != 0]
print("workgroups stamped", len(s), "canvas", tuple(x.shape))
t0 = s[:, 0].min()
us = lambda v: v / 100.0
pro, main, stage, store = us(s[:, 1] - s[:, 0]), us(s[:, 2] - s[:, 1]), us(s[:, 3] - s[:, 2]), us(s[:, 4] - s[:, 3])
print(f"  walk end -> mask prefetch issued {us(s[:,10]-s[:,2]).mean():7.2f} us")
print(f"  barrier after the walk          mean {us(s[:,8]-s[:,2]).mean():7.2f} us;  accumulators -> LDS {us(s[:,9]-s[:,8]).mean():7.2f} us;  second barrier {us(s[:,3]-s[:,9]).mean():7.2f} us")
for name, v in (("prologue (entry -> first K-tile)", pro), ("K walk", main), ("accumulators -> LDS + barrier", stage), ("LDS -> global stores issued", store), ("whole", us(s[:, 4] - s[:, 0]))):
    print(f"{name:34s} mean {v.mean():7.2f} us  p10 {np.percentile(v,10):7.2f}  p50 {np.percentile(v,50):7.2f}  p90 {np.percentile(v,90):7.2f}")
print("kernel span (first entry -> last exit) us", us(s[:, 4].max() - t0))
cus = collections.defaultdict(list)
for r in s:
    cus[(int(r[6]) & 0xf, (int(r[5]) >> 8) & 0xff)].append((r[0], r[4]))
gaps = []
per = []
for k, v in cus.items():
    v.sort()
    per.append(len(v))
    for a, b in zip(v[:-1], v[1:]):
        gaps.append(us(b[0] - a[1]))
gaps = np.array(gaps)
print("CUs seen", len(cus), "tiles per CU min/max", min(per), max(per))
print(f"gap exit -> next entry on the same CU: mean {gaps.mean():.2f} us p10 {np.percentile(gaps,10):.2f} p50 {np.percentile(gaps,50):.2f} p90 {np.percentile(gaps,90):.2f}")
first = us(np.array([v[0][0] for v in cus.values()]) - t0)
print(f"first entry per CU after kernel start: mean {first.mean():.2f} max {first.max():.2f}")
