"""Lab: the bound passes (hstat_kernel through hip.to_hformat, hstat_multi through the model's weight refresh) at the step's shapes.
usage: python scripts/lab/hstat_bench.py [label]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import ofb_amd
from ofb_amd import hip

def t(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps

label = sys.argv[1] if len(sys.argv) > 1 else 'build'
x = torch.randn(128 * 197, 384, device='cuda')
small = torch.randn(1280, 384, device='cuda')
print(f'{label:20s} to_hformat [25216][384] (stat + split) {t(lambda: hip.to_hformat(x)):6.1f} us   [1280][384] {t(lambda: hip.to_hformat(small)):6.1f} us')
torch.manual_seed(0)
m = ofb_amd.create_model('deit_small_patch16_224_mim', method='search', num_classes=1000, drop_path_rate=0.1, patch_search=False, mask_ratio=1.0).cuda()
ws = [p for p in m.parameters() if p.dim() == 2]
for w in ws: hip.weight_h(w)
def refresh():
    hip.bump_weight_epoch()
    hip.weight_h(ws[0])
print(f'{label:20s} weight planes refresh ({len(ws)} matrices: hstat_multi + to_hformat_multi) {t(refresh):6.1f} us')
