"""Which ATen / runtime kernels one search step of the PRUNED search model (bench.py --pruned) still launches (run on the GPU box)."""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
import ofb_amd
from ofb_amd import engine
from ofb_amd.losses import DistillationLoss, LabelSmoothingCrossEntropy, OFBSearchLOSS
from torch.profiler import profile, ProfilerActivity

dev = torch.device('cuda', 0)
torch.manual_seed(0)
model, _ = bench.build_pruned_search(ofb_amd, dev, 1000)
model.adjust_masking_ratio(0, 20, 100)
for m in model.searchable_modules:
    m.update_w(0, 20)
model.to(dev).train()
opts = engine.build_optimizers(model, 2.5e-4 * 128 / 256)
crit = OFBSearchLOSS(DistillationLoss(LabelSmoothingCrossEntropy(0.1), None, 'none', 0.5, 1.0), dev, attn_w=0.5, mlp_w=0.5,
                     patch_w=0.0, embedding_w=0.5, flops_w=5.0)
imgs = torch.randn(128, 3, 224, 224, device=dev)
labels = torch.randint(0, 1000, (128,), device=dev)

def step():
    engine.search_step(model, crit, imgs, labels, 1.0, opts)

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
mem = collections.Counter()
for ev in prof.events():
    if ev.device_type == torch.autograd.DeviceType.CPU and ('Memcpy' in ev.name or 'Memset' in ev.name or 'hipMem' in ev.name):
        mem[ev.name] += 1
print('runtime copies / fills:', dict(mem))
