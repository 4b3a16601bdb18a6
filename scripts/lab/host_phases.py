"""host wall time of the phases of one search step at a host-bound size (DeiT-T bs 8: the GPU keeps up, so phase walls = host cost)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import ofb_amd
from ofb_amd import engine
from ofb_amd.losses import OFBSearchLOSS, DistillationLoss, LabelSmoothingCrossEntropy
name, bs = (sys.argv[1], int(sys.argv[2])) if len(sys.argv) > 2 else ('deit_tiny', 8)
dev = torch.device('cuda:0')
torch.manual_seed(0)
model = ofb_amd.create_model(f'{name}_patch16_224_mim', method='search', num_classes=1000, drop_path_rate=0.1, attn_search=True,
                             mlp_search=True, embed_search=True, patch_search=False, mae=True, mask_ratio=1.0)
model.correct_require_grad(0.5, 0.5, 0, 0.5)
model.adjust_masking_ratio(0.0, 20, 100)
model.to(dev).train()
opts = engine.build_optimizers(model, 2.5e-4 * bs / 256)
crit = OFBSearchLOSS(DistillationLoss(LabelSmoothingCrossEntropy(0.1), None, 'none', 0.5, 1.0), dev, attn_w=0.5, mlp_w=0.5, patch_w=0.0,
                     embedding_w=0.5, flops_w=5.0)
imgs = torch.randn(bs, 3, 224, 224, device=dev)
labels = torch.randint(0, 1000, (bs,), device=dev)
T = [0.0] * 5
def step(rec):
    t0 = time.perf_counter()
    outputs, (dl, _) = model(imgs)
    t1 = time.perf_counter()
    loss = crit(imgs, outputs, labels, model, 'arch', 1.0, False)
    base, arch, total = engine.mix_losses(loss, dl)
    t2 = time.perf_counter()
    total.backward()
    t3 = time.perf_counter()
    for o in opts: o.step()
    for o in opts: o.zero_grad(set_to_none=True)
    t4 = time.perf_counter()
    if rec:
        for i, d in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3)): T[i] += d
for _ in range(10): step(False)
torch.cuda.synchronize()
n = 40
t0 = time.perf_counter()
for _ in range(n): step(True)
torch.cuda.synchronize()
tot = time.perf_counter() - t0
print(f'{name} bs {bs}: {tot / n * 1e3:.2f} ms/step  forward {T[0] / n * 1e3:.2f}  loss {T[1] / n * 1e3:.2f}  backward {T[2] / n * 1e3:.2f}  optimizer {T[3] / n * 1e3:.2f}')
