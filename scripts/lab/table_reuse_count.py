"""Lab (GPU box): how often the address-keyed tables of a search step repeat from one step to the next."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, ofb_amd
from ofb_amd import engine, hip
from ofb_amd.losses import OFBSearchLOSS, DistillationLoss, LabelSmoothingCrossEntropy
dev = torch.device('cuda')
torch.manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
m = ofb_amd.create_model('deit_small_patch16_224_mim', method='search', num_classes=1000, drop_path_rate=0.1, patch_search=False, mask_ratio=1.0)
m.correct_require_grad(0.5, 0.5, 0, 0.5); m.adjust_masking_ratio(0.0, 20, 100); m.to(dev).train()
opts = engine.build_optimizers(m, 1e-4)
crit = OFBSearchLOSS(DistillationLoss(LabelSmoothingCrossEntropy(0.1), None, 'none', 0.5, 1.0), dev, 0.5, 0.5, 0.0, 0.5, 5.0)
imgs = torch.randn(B, 3, 224, 224, device=dev); labels = torch.randint(0, 1000, (B,), device=dev)
n_up = [0]
orig = hip.upload_structs
def counting(array, device, site=None):
    n_up[0] += 1
    return orig(array, device, site)
hip.upload_structs = counting
import ofb_amd.ops as ops, ofb_amd.optim as optim
for _ in range(10): engine.search_step(m, crit, imgs, labels, 1.0, opts)
h0, u0 = [o.table_hits for o in opts], n_up[0]
seen = {}
for _ in range(20):
    engine.search_step(m, crit, imgs, labels, 1.0, opts)
    for i, o in enumerate(opts):
        for slot, ent in o._tables.items():
            seen.setdefault((i, slot), set()).add(hash(ent[0]))
torch.cuda.synchronize()
print('AdamW launches that reused their table in 20 steps:', [o.table_hits - h for o, h in zip(opts, h0)], '(of 20 x launches per optimizer)')
print('distinct address tuples seen per table in 20 steps:', {k: len(v) for k, v in seen.items()})
print('upload_structs calls per step:', (n_up[0] - u0) / 20)
