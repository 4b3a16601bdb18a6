"""Token-row GEMMs of the configs[4] finetune subnet (B = 256: M = 50432 rows) per shape; run with OFB_GEMM_P_TILE=128 / 0 to A/B the
256 x 96 tile against the 128 x 192 one."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ofb_amd import hip
M = 256 * 197
def run(tag, fn, flops, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f'{tag:40s} {ms*1e3:8.1f} us {flops/ms/1e9:7.1f} TF')
    return ms
r = lambda *s: torch.randn(*s, device='cuda')
tot = 0
for (N, K) in [(264, 192), (264, 576), (264, 768), (264, 960), (480, 264), (672, 264), (576, 264), (160, 264), (224, 264), (192, 264), (768, 264), (960, 264)]:
    x, w = hip.to_pformat(r(M, K)), r(N, K)
    wp, wt = hip.to_pformat(w), hip.to_pformat(w.t().contiguous())
    y = torch.empty(M, N, device='cuda')
    tot += run(f'kc,kc N={N} K={K} -> f32', lambda: hip.gemm_p(x, wp, 1, 1, M, N, K, C_out=y, ldc=N), 2. * M * N * K)
    tot += run(f'kc,kr N={N} K={K} -> f32', lambda: hip.gemm_p(x, wt, 1, 0, M, N, K, C_out=y, ldc=N), 2. * M * N * K)
print(f'sum {tot:.2f} ms')
