"""why does bias + gate cost 30 us on the fc1 shape?  (run on the GPU box)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ofb_amd import hip
M, D, HID = 128 * 197, 384, 1536
def run(tag, fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    print(f'{tag:44s} {e0.elapsed_time(e1) / iters * 1e3:8.1f} us')
r = lambda *s: torch.randn(*s, device='cuda')
x = r(M, D); xp = hip.to_pformat(x)
w3, b3, g3 = r(HID, D), r(HID), r(HID); w3p = hip.to_pformat(w3)
y = torch.empty(M, HID, device='cuda')
one, zero = torch.ones(HID, device='cuda'), torch.zeros(HID, device='cuda')
for rep in range(2):
    run('plain', lambda: hip.gemm_p(xp, w3p, 1, 1, M, HID, D, C_out=y, ldc=HID))
    run('bias (randn)', lambda: hip.gemm_p(xp, w3p, 1, 1, M, HID, D, C_out=y, ldc=HID, bias=b3))
    run('colscale (randn)', lambda: hip.gemm_p(xp, w3p, 1, 1, M, HID, D, C_out=y, ldc=HID, colscale=g3))
    run('bias + colscale (randn)', lambda: hip.gemm_p(xp, w3p, 1, 1, M, HID, D, C_out=y, ldc=HID, bias=b3, colscale=g3))
    run('bias = 0, colscale = 1', lambda: hip.gemm_p(xp, w3p, 1, 1, M, HID, D, C_out=y, ldc=HID, bias=zero, colscale=one))
    run('plain again', lambda: hip.gemm_p(xp, w3p, 1, 1, M, HID, D, C_out=y, ldc=HID))
