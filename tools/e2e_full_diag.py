import sys, zlib
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
import synth
import test_e2e_gpu as T
g = np.load("tests/golden/e2e_full.npz")
net = T._model(g, "cuda:0", T.FIXTURES["e2e_full.npz"]).train()
images, targets = T._inputs(g, "cuda:0", "e2e_full.npz")
import torch.backends.cudnn as cudnn
print("allow_tf32 conv", cudnn.allow_tf32, "matmul", torch.backends.cuda.matmul.allow_tf32)
if len(sys.argv) > 1 and sys.argv[1] == "notf32":
    cudnn.allow_tf32 = False; torch.backends.cuda.matmul.allow_tf32 = False
out = net(images, targets)
print([float(out["classification_loss"]), float(out["regression_loss"])], g["train_losses"])
(out["classification_loss"] + out["regression_loss"]).backward()
named = dict(net.named_parameters())
rows = []
for k, norm, proj, samp, pos in zip(g["grad_all_keys"], g["grad_all_norms"], g["grad_all_proj"], g["grad_all_samples"], g["grad_all_pos"]):
    flat = named[str(k)].grad.reshape(-1).double().cpu().numpy()
    r = np.random.default_rng(zlib.crc32(str(k).encode())).standard_normal(flat.size)
    rms = norm / np.sqrt(flat.size)
    rows.append((abs(np.linalg.norm(flat) - norm) / norm, abs(flat @ r - proj) / norm, float(np.max(np.abs(flat[pos] - samp) - 5e-2 * np.abs(samp)) / rms), str(k)))
a = np.array([r[:3] for r in rows])
for i, nm in enumerate(("norm", "proj/norm", "samp/rms")):
    print(nm, "median %.2e p90 %.2e max %.2e" % (np.median(a[:, i]), np.percentile(a[:, i], 90), a[:, i].max()), rows[int(a[:, i].argmax())][3])
worst = sorted(rows, key=lambda r: -r[1])[:8]
for w in worst: print("  %.3e %.3e %.3e %s" % w)
