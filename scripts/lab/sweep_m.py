import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ofb_amd import hip
def run(M, N, K, iters=10, **kw):
    x = torch.randn(M, K, device='cuda'); w = torch.randn(N, K, device='cuda'); y = torch.empty(M, N, device='cuda')
    f = lambda: hip.gemm(x, w, y, M, N, K, K, K, N, 1, 1, **kw)
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return ms * 1e3, 2.0 * M * N * K / ms / 1e9
for M in (8192, 16384, 24576, 25216):
    for K in (128, 384, 768, 1536, 4096):
        us, tf = run(M, 1536, K)
        print(f'M {M:6d} N 1536 K {K:5d}: tiles {M//128*12:5d} ({M//128*12/512:5.2f} rounds) {us:9.1f} us {tf:6.1f} TF')
