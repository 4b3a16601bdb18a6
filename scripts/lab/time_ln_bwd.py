"""Times ofb_layernorm_bwd_h (bound pass + main kernel) on the DeiT-S bs 128 activation shape (GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ofb_amd import hip

M, D = 128 * 197, 384
x = torch.randn(M, D, device='cuda'); dy = torch.randn(M, D, device='cuda') * 1e-3; dres = torch.randn(M, D, device='cuda') * 1e-3
g = torch.randn(D, device='cuda'); b = torch.randn(D, device='cuda')
mean = torch.empty(M, device='cuda'); rstd = torch.empty(M, device='cuda'); y = torch.empty(M, D, device='cuda')
yP = hip.HMat(M, D, 'cuda')
hip.layernorm_fwd_h(x, g, b, y, yP, mean, rstd, M, D, 1e-6)
rs = torch.rand(128, device='cuda')
dx = torch.empty(M, D, device='cuda'); dxP = hip.HMat(M, D, 'cuda')
part = torch.empty(hip.layernorm_bwd_blocks(M) * 3 * D, device='cuda')
def run(tag, fn, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    print(f'{tag:40s} {e0.elapsed_time(e1) / iters * 1e3:8.1f} us')
run('ln bwd f32 (no planes)', lambda: hip.layernorm_bwd(dy, x, g, mean, rstd, dres, dx, part, M, D))
run('ln bwd planes, dres + rowscale', lambda: hip.layernorm_bwd_h(dy, x, g, mean, rstd, dres, dx, part, dxP, rs, 197, M, D))
run('ln bwd planes, no dres', lambda: hip.layernorm_bwd_h(dy, x, g, mean, rstd, None, dx, part, dxP, rs, 197, M, D))
