"""Which source lines of the package still launch ATen kernels inside one DeiT-S bs128 search step (run on the GPU box).
torch.profiler with stacks: every aten op that launches a device kernel is attributed to the innermost once-for-both_amd frame."""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import ofb_amd
from ofb_amd import engine
from ofb_amd.losses import OFBSearchLOSS, DistillationLoss, LabelSmoothingCrossEntropy
from torch.profiler import profile, ProfilerActivity

dev = torch.device('cuda', 0)
torch.manual_seed(0)
model = ofb_amd.create_model('deit_small_patch16_224_mim', method='search', num_classes=1000, drop_path_rate=0.1, attn_search=True,
                             mlp_search=True, embed_search=True, patch_search=False, mae=True, mask_ratio=1.0)
model.correct_require_grad(0.5, 0.5, 0, 0.5)
model.adjust_masking_ratio(0.0, 20, 100)
model.to(dev).train()
opts = engine.build_optimizers(model, 2.5e-4 * 128 / 256)
crit = OFBSearchLOSS(DistillationLoss(LabelSmoothingCrossEntropy(0.1), None, 'none', 0.5, 1.0), dev, attn_w=0.5, mlp_w=0.5, patch_w=0.0,
                     embedding_w=0.5, flops_w=5.0)
imgs = torch.randn(128, 3, 224, 224, device=dev)
labels = torch.randint(0, 1000, (128,), device=dev)
for _ in range(6):
    engine.search_step(model, crit, imgs, labels, 1.0, opts)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    engine.search_step(model, crit, imgs, labels, 1.0, opts)
    torch.cuda.synchronize()
by = collections.Counter()
for ev in prof.events():
    if ev.device_type != torch.autograd.DeviceType.CPU or not ev.name.startswith('aten::'):
        continue
    if not ev.kernels:
        continue
    frames = [f for f in (ev.stack or []) if 'once-for-both_amd' in f or 'ofb_amd' in f]
    where = (frames[0].strip() if frames else '') + ' shapes ' + str(ev.input_shapes)[:120]
    by[(ev.name, where, len(ev.kernels))] += 1
tot = 0
for (name, where, nk), n in sorted(by.items(), key=lambda kv: -kv[1] * kv[0][2]):
    print(f'{n * nk:4d} launches  {name:28s} {where}')
    tot += n * nk
print('total ATen launches attributed:', tot)
