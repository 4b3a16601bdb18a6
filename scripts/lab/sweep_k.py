import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ofb_amd import hip
def run(M, N, K, iters=10):
    x = torch.randn(M, K, device='cuda'); w = torch.randn(N, K, device='cuda'); y = torch.empty(M, N, device='cuda')
    f = lambda: hip.gemm(x, w, y, M, N, K, K, K, N, 1, 1)
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return ms * 1e3, 2.0 * M * N * K / ms / 1e9
for (M, N, K) in [(16384, 1536, 4096), (16384, 2048, 2048), (25216, 1152, 384), (25216, 1536, 384), (25216, 384, 1536)]:
    us, tf = run(M, N, K)
    print(f'M {M:6d} N {N:5d} K {K:5d}: {us:9.1f} us {tf:6.1f} TF')
