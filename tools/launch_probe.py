"""How much of a train step is host launch time?  Enqueue K steps without synchronising and compare
the host time to enqueue with the wall time to finish."""
import os, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
import pytorch_retinanet_amd as P
from pytorch_retinanet_amd import tuning
from bench import synth_batch
tuning.use_shipped_miopen_db(0); tuning.enable_conv_autotune()
dev = torch.device("cuda:0")
net = P.Retinanet(num_classes=90, backbone_kind="resnet50", pretrained=False, min_size=800, max_size=1333).to(dev).to(memory_format=torch.channels_last).train()
opt = torch.optim.SGD(net.parameters(), lr=1e-3, weight_decay=1e-3, momentum=0.9)
images, targets = synth_batch(8, 8, 0, dev)
def step():
    opt.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        l = net(images, targets); loss = l["classification_loss"] + l["regression_loss"]
    loss.backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
K = 10
t0 = time.perf_counter()
for _ in range(K): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3*(t1-t0)/K:.2f} ms/step ; wall {1e3*(t2-t0)/K:.2f} ms/step")
