"""which Python lines launch the small ATen kernels of a search step (torch.profiler with stacks)"""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import ofb_amd
from ofb_amd import engine
from ofb_amd.losses import OFBSearchLOSS, DistillationLoss, LabelSmoothingCrossEntropy
from torch.profiler import profile, ProfilerActivity
dev = torch.device('cuda:0')
torch.manual_seed(0)
model = ofb_amd.create_model('deit_small_patch16_224_mim', method='search', num_classes=1000, drop_path_rate=0.1, attn_search=True,
                             mlp_search=True, embed_search=True, patch_search=False, mae=True, mask_ratio=1.0)
model.correct_require_grad(0.5, 0.5, 0, 0.5); model.adjust_masking_ratio(0.0, 20, 100); model.to(dev).train()
opts = engine.build_optimizers(model, 1e-3)
crit = OFBSearchLOSS(DistillationLoss(LabelSmoothingCrossEntropy(0.1), None, 'none', 0.5, 1.0), dev, attn_w=0.5, mlp_w=0.5, patch_w=0.0,
                     embedding_w=0.5, flops_w=5.0)
imgs = torch.randn(16, 3, 224, 224, device=dev); labels = torch.randint(0, 1000, (16,), device=dev)
for _ in range(3): engine.search_step(model, crit, imgs, labels, 1.0, opts)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    engine.search_step(model, crit, imgs, labels, 1.0, opts)
    torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in ('aten::fill_', 'aten::zero_', 'aten::zeros', 'aten::copy_', 'aten::sum', 'aten::add', 'aten::mul', 'aten::repeat', 'aten::contiguous', 'aten::clone'):
        st = [s for s in ev.stack if 'once-for-both_amd' in s or 'ofb_amd' in s]
        cnt[(ev.name, st[0] if st else (ev.stack[0] if ev.stack else '?'))] += 1
for (n, s), c in cnt.most_common(40):
    print(f'{c:4d}  {n:18s} {s[-110:]}')
