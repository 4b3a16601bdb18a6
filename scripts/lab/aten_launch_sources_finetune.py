"""Which ATen kernels one finetune micro-step (configs[4]: bench.py --mode finetune) still launches (run on the GPU box)."""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import bench
import ofb_amd
from ofb_amd.losses import DistillationLoss
from ofb_amd.optim import AdamW
from ofb_amd.utils import ModelEma
from torch.profiler import profile, ProfilerActivity

dev = torch.device('cuda', 0)
torch.manual_seed(0)
model, _, _ = bench.build_finetune_subnet(ofb_amd, dev, 1000)
model.train(False)
opt = AdamW(model.parameters(), None, lr=1e-4, weight_decay=0.05)
crit = DistillationLoss(ofb_amd.SoftTargetCrossEntropy(), None, 'none', 0.5, 1.0)
mix = ofb_amd.Mixup(mixup_alpha=0.8, cutmix_alpha=1.0, label_smoothing=0.1, num_classes=1000)
np.random.seed(1)
ema = ModelEma(model, decay=0.99996)
imgs = torch.randn(256, 3, 224, 224, device=dev)
labels = torch.randint(0, 1000, (256,), device=dev)

def step():
    x, soft = mix(imgs, labels)
    loss = crit(x, model(x), soft)
    loss.backward()
    opt.step(); opt.zero_grad(set_to_none=True)
    ema.update(model)

for _ in range(5):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
by = collections.Counter()
for ev in prof.events():
    if ev.device_type != torch.autograd.DeviceType.CPU or not ev.name.startswith('aten::') or not ev.kernels:
        continue
    by[(ev.name, str(ev.input_shapes)[:110], len(ev.kernels))] += 1
tot = 0
for (name, shp, nk), n in sorted(by.items(), key=lambda kv: -kv[1] * kv[0][2]):
    print(f'{n * nk:4d} launches  {name:24s} {shp}')
    tot += n * nk
print('total ATen launches attributed:', tot)
