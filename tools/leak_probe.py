"""Run many train steps and watch allocated / reserved memory and the step time (leak / fragmentation check)."""
import sys, time, torch
sys.path.insert(0, ".")
import pytorch_retinanet_amd as P
from pytorch_retinanet_amd import tuning
from pytorch_retinanet_amd.optim import MasterSGD, use_bf16_conv_weights
tuning.use_shipped_miopen_db(); tuning.enable_conv_autotune()
dev = torch.device("cuda")
torch.manual_seed(0)
net = P.Retinanet(num_classes=90, backbone_kind="resnet50", pretrained=False, min_size=800, max_size=1333).to(dev).to(memory_format=torch.channels_last).train()
use_bf16_conv_weights(net)
opt = MasterSGD(net.parameters(), lr=1e-3, weight_decay=1e-3, momentum=0.9)
g = torch.Generator(device=dev).manual_seed(0)
images = [torch.rand((3, 800, 1333), device=dev, generator=g) for _ in range(8)]
targets = [{"boxes": torch.tensor([[100.0, 100.0, 400.0, 300.0], [500.0, 200.0, 900.0, 700.0]], device=dev), "labels": torch.tensor([1, 7], device=dev)} for _ in range(8)]
t0 = time.perf_counter()
for step in range(1, 121):
    opt.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = net(images, targets)
        loss = out["classification_loss"] + out["regression_loss"]
    loss.backward()
    opt.step()
    if step in (20, 60, 120):
        torch.cuda.synchronize()
        print(f"step {step}: loss {float(loss):.4f} allocated {torch.cuda.memory_allocated() / 2**30:.2f} GiB reserved {torch.cuda.memory_reserved() / 2**30:.2f} GiB "
              f"peak {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB  {((time.perf_counter() - t0) / step) * 1e3:.1f} ms/step avg")
