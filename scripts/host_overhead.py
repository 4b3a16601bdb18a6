"""How long does the host take to ENQUEUE one search step (no device sync)?  If close to the step time, the step is host-bound."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ofb_amd
from ofb_amd import engine
from ofb_amd.losses import OFBSearchLOSS, DistillationLoss, LabelSmoothingCrossEntropy
dev = torch.device('cuda')
torch.manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
m = ofb_amd.create_model('deit_small_patch16_224_mim', method='search', num_classes=1000, drop_path_rate=0.1, patch_search=False, mask_ratio=1.0)
m.correct_require_grad(0.5, 0.5, 0, 0.5); m.adjust_masking_ratio(0.0, 20, 100); m.to(dev).train()
opts = engine.build_optimizers(m, 1e-4)
crit = OFBSearchLOSS(DistillationLoss(LabelSmoothingCrossEntropy(0.1), None, 'none', 0.5, 1.0), dev, 0.5, 0.5, 0.0, 0.5, 5.0)
imgs = torch.randn(B, 3, 224, 224, device=dev); labels = torch.randint(0, 1000, (B,), device=dev)
for _ in range(12): engine.search_step(m, crit, imgs, labels, 1.0, opts)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(10): engine.search_step(m, crit, imgs, labels, 1.0, opts)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f'B={B}: host enqueue {1e3*(t1-t0)/10:.2f} ms/step, wall {1e3*(t2-t0)/10:.2f} ms/step')
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(3): engine.search_step(m, crit, imgs, labels, 1.0, opts)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
pstats.Stats(pr).sort_stats('tottime').print_stats(45)
