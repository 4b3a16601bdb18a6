"""Lab (GPU box): where the HOST's time of one search step goes (cProfile over 30 steps at a small batch, where the device is never the
limit; the autograd engine runs the backward on the calling thread so that the profile sees it).  usage: host_profile.py [batch] [--pruned]"""
import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, ofb_amd, bench
from ofb_amd import engine
from ofb_amd.losses import OFBSearchLOSS, DistillationLoss, LabelSmoothingCrossEntropy
dev = torch.device('cuda')
torch.manual_seed(0)
args = [a for a in sys.argv[1:] if not a.startswith('--')]
B = int(args[0]) if args else 8
finetune = '--finetune' in sys.argv
if finetune:
    import numpy as np
    from ofb_amd.optim import AdamW
    from ofb_amd.utils import ModelEma
    m, _, _ = bench.build_finetune_subnet(ofb_amd, dev, 1000)
    m.train(False)
    opt = AdamW(m.parameters(), None, lr=1e-4, weight_decay=0.05)
    crit = DistillationLoss(ofb_amd.SoftTargetCrossEntropy(), None, 'none', 0.5, 1.0)
    mix = ofb_amd.Mixup(mixup_alpha=0.8, cutmix_alpha=1.0, label_smoothing=0.1, num_classes=1000)
    np.random.seed(1)
    ema = ModelEma(m, decay=0.99996)
else:
    if '--pruned' in sys.argv:
        m, _ = bench.build_pruned_search(ofb_amd, dev, 1000)
    else:
        m = ofb_amd.create_model('deit_small_patch16_224_mim', method='search', num_classes=1000, drop_path_rate=0.1, patch_search=False, mask_ratio=1.0)
        m.correct_require_grad(0.5, 0.5, 0, 0.5)
    m.adjust_masking_ratio(0.0, 20, 100); m.to(dev).train()
    opts = engine.build_optimizers(m, 1e-4)
    crit = OFBSearchLOSS(DistillationLoss(LabelSmoothingCrossEntropy(0.1), None, 'none', 0.5, 1.0), dev, 0.5, 0.5, 0.0, 0.5, 5.0)
imgs = torch.randn(B, 3, 224, 224, device=dev); labels = torch.randint(0, 1000, (B,), device=dev)
def ft_step():
    x, soft = mix(imgs, labels)
    loss = crit(x, m(x), soft)
    engine.run_backward(loss)
    opt.step(); opt.zero_grad(set_to_none=True)
    ema.update(m)
def run(n):
    for _ in range(n):
        if finetune: ft_step()
        else: engine.search_step(m, crit, imgs, labels, 1.0, opts)
run(12); torch.cuda.synchronize()
for rep in range(3):
    for on_caller in (False, True):
        engine._BACKWARD_ON_CALLER = on_caller
        run(3); torch.cuda.synchronize()
        t0 = time.perf_counter(); run(20); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f'B={B}: backward on the {"calling thread" if on_caller else "engine thread "}: host enqueue {1e3*(t1-t0)/20:.2f} ms/step, wall {1e3*(t2-t0)/20:.2f} ms/step')
engine._BACKWARD_ON_CALLER = True
if True:
    pr = cProfile.Profile(); pr.enable(); run(30); pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr)
print('(30 steps: divide by 30 for per-step seconds)')
st.sort_stats('tottime').print_stats(70)
st.sort_stats('cumulative').print_stats(60)
