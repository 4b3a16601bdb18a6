"""which part of the step breaks hipGraph capture: python graph_bisect.py <fwd|fwdloss|bwd|opt|upload|rand>"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import ofb_amd
from ofb_amd import engine, hip
from ofb_amd.losses import OFBSearchLOSS, DistillationLoss, LabelSmoothingCrossEntropy
what = sys.argv[1]
name, bs = 'deit_tiny', 8
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
for _ in range(4): engine.search_step(model, crit, imgs, labels, 1.0, opts)
torch.cuda.synchronize()
def fwd():
    with torch.no_grad():
        return model(imgs)[0]
def fwdloss():
    outputs, (dl, _) = model(imgs)
    loss = crit(imgs, outputs, labels, model, 'arch', 1.0, False)
    return engine.mix_losses(loss, dl)[2]
def bwd():
    t = fwdloss()
    t.backward()
    return t
def bwd_cls():
    outputs, (dl, _) = model(imgs)
    t = outputs.float().sum(); t.backward(); return t
def bwd_dec():
    outputs, (dl, _) = model(imgs)
    dl.backward(); return dl
def bwd_arch():
    outputs, (dl, _) = model(imgs)
    loss = crit(imgs, outputs, labels, model, 'arch', 1.0, False)
    base, arch, total = engine.mix_losses(loss, dl)
    arch.backward(); return arch
def bwd_gates():
    model._gate_out = None
    sp = model.get_sparsity_loss(dev)
    t = sp[0] + sp[1] + sp[3]; t.backward(); return t
def bwd_flops():
    model._gate_out = None
    model.get_sparsity_loss(dev)
    t = model.get_flops_loss(dev, 1.0) if hasattr(model, 'get_flops_loss') else None
    t.backward(); return t
def fwd_gates():
    model._gate_out = None
    sp = model.get_sparsity_loss(dev)
    return sp[0] + sp[1] + sp[3]
def opt():
    for o in opts: o.step()
def upload():
    tab = (hip.AdamwTensor * 4)()
    d, h = hip.upload_structs(tab, dev)
    keep.append((d, h))
    return d
def rand():
    return torch.rand(8, 196, device=dev)
keep = []
fn = dict(bwd_gates=bwd_gates, bwd_flops=bwd_flops, fwd_gates=fwd_gates, bwd_cls=bwd_cls, bwd_dec=bwd_dec, bwd_arch=bwd_arch, fwd=fwd, fwdloss=fwdloss, bwd=bwd, opt=opt, upload=upload, rand=rand)[what]
if what == 'opt':
    bwd()
arena = hip.begin_capture_arena()
if what in ('opt',):
    for o in opts: o.begin_capture()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = fn()
print(what, 'captured')
if what == 'opt':
    for o in opts: o.end_capture(); o.refresh_hyper(0)
g.replay(); torch.cuda.synchronize()
print(what, 'replayed OK')
