"""Times the attention kernels at the DeiT-S bs-128 shape (B 128, N 197, H 6, d 64) through the product bindings (run on the GPU box).
OFB_LIB_PATH=<other build> times another library: python scripts/att_perf.py [label]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ofb_amd import hip
B, N, H, dh = 128, 197, 6, 64
M, Hd = B * N, H * dh
g = torch.Generator(device='cuda').manual_seed(1)
qkv = torch.randn(M, 3 * Hd, device='cuda', generator=g)
o, lse = torch.empty(M, Hd, device='cuda'), torch.empty(2 * B * H, N, device='cuda')
do, dqkv = torch.randn(M, Hd, device='cuda', generator=g), torch.empty(M, 3 * Hd, device='cuda')
qb, dob, amax = qkv.abs().max().reshape(1) * 1.5, do.abs().max().reshape(1) * 1.5, torch.zeros(1, device='cuda')
oP = hip.HMat.for_rows_written_by_kernel(M, Hd, 'cuda')
fwd = lambda: hip.attention_fwd_h(qkv, o, oP, lse, B, N, H, dh, 0.125, qb)
bwd = lambda: hip.attention_bwd(qkv, o, lse, do, dqkv, B, N, H, dh, 0.125, qb, dob, amax)
def run(f, n=60):
    for _ in range(20): f()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best
print(f'{(sys.argv[1] if len(sys.argv) > 1 else "build"):24s} attention fwd (+planes) {run(fwd):7.1f} us   bwd {run(bwd):7.1f} us')
