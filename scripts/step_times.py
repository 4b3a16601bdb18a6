"""per-step wall time (device-synchronised) of the first steps of the bench workload: shows how long the warm-up tail is."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ofb_amd
from ofb_amd import engine
from ofb_amd.losses import OFBSearchLOSS, DistillationLoss, LabelSmoothingCrossEntropy
dev = torch.device('cuda')
torch.manual_seed(0)
m = ofb_amd.create_model('deit_small_patch16_224_mim', method='search', num_classes=1000, drop_path_rate=0.1, patch_search=False, mask_ratio=1.0)
m.correct_require_grad(0.5, 0.5, 0, 0.5); m.adjust_masking_ratio(0.0, 20, 100); m.to(dev).train()
opts = engine.build_optimizers(m, 1e-4)
crit = OFBSearchLOSS(DistillationLoss(LabelSmoothingCrossEntropy(0.1), None, 'none', 0.5, 1.0), dev, 0.5, 0.5, 0.0, 0.5, 5.0)
imgs = torch.randn(128, 3, 224, 224, device=dev); labels = torch.randint(0, 1000, (128,), device=dev)
ts = []
for i in range(60):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    engine.search_step(m, crit, imgs, labels, 1.0, opts)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    ts.append((1e3 * (t1 - t0), 1e3 * (t2 - t0)))
print(" ".join(f"{i}:{a:.0f}/{b:.0f}" for i, (a, b) in enumerate(ts) if b > 36 or i < 6), "(host enqueue ms / wall ms; steps over 36 ms)"); print("median wall", sorted(b for a, b in ts)[len(ts)//2])
print('allocator', torch.cuda.memory_reserved() / 2**30, 'GiB reserved')
