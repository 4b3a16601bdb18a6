"""hipGraph capture of the whole search step (engine.GraphedStep): eager vs replayed step time"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import ofb_amd
from ofb_amd import engine
from ofb_amd.losses import OFBSearchLOSS, DistillationLoss, LabelSmoothingCrossEntropy
name, bs = (sys.argv[1], int(sys.argv[2])) if len(sys.argv) > 2 else ('deit_tiny', 8)
dev = torch.device('cuda:0')
torch.cuda.set_stream(torch.cuda.Stream())      # the whole job on one non-default stream (capturable)
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
def step():
    return engine.search_step(model, crit, imgs, labels, 1.0, opts)
for _ in range(8): out = step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): out = step()
torch.cuda.synchronize()
print(f'eager   {(time.perf_counter() - t0) / 20 * 1e3:.2f} ms/step  loss {float(out[3]):.4f}')
gs = engine.GraphedStep(step, opts)
gs.capture()
print('captured; slots per optimizer:', [len(o._cap['slots']) for o in opts])
for _ in range(3): out = gs()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): out = gs()
torch.cuda.synchronize()
print(f'graphed {(time.perf_counter() - t0) / 20 * 1e3:.2f} ms/step  loss {float(out[3]):.4f}')
