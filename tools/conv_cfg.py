import os, sys, time
cl = int(sys.argv[1]); bm = int(sys.argv[2]); fm = sys.argv[3]; dt = sys.argv[4] if len(sys.argv) > 4 else "bf16"
if fm != "default": os.environ["MIOPEN_FIND_MODE"] = fm
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
torch.backends.cudnn.benchmark = bool(bm)
import pytorch_retinanet_amd as P
from bench import synth_batch
dev = torch.device("cuda:0")
net = P.Retinanet(num_classes=90, backbone_kind="resnet50", pretrained=False, min_size=800, max_size=1333).to(dev)
if cl: net = net.to(memory_format=torch.channels_last)
net.train()
opt = torch.optim.SGD(net.parameters(), lr=1e-3, momentum=0.9)
images, targets = synth_batch(8, 8, 0, dev)
amp = {"bf16": torch.bfloat16, "f16": torch.float16}.get(dt)
def step():
    opt.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=amp, enabled=amp is not None):
        l = net(images, targets); loss = l["classification_loss"] + l["regression_loss"]
    loss.backward(); opt.step()
t0 = time.time(); step(); torch.cuda.synchronize(); t1 = time.time()
step(); torch.cuda.synchronize(); t2 = time.time()
for _ in range(3): step()
torch.cuda.synchronize(); t3 = time.time()
print(f"cl={cl} benchmark={bm} find={fm} dt={dt}: first={t1-t0:.1f}s second={t2-t1:.2f}s steady={(t3-t2)/3*1e3:.1f} ms/step", flush=True)
