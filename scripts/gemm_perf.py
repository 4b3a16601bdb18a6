"""Micro-benchmark (conversion included) of hip.gemm and the attention kernels on the DeiT-S bs=128 layer shapes (run on the GPU box); scripts/gemm_step_shapes.py times the pre-split path the model uses."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ofb_amd import hip

M, D = 128 * 197, 384
def run(tag, fn, flops, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f'{tag:28s} {ms*1e3:8.1f} us  {flops/ms/1e9:7.1f} TFLOP/s')

for N, K in [(1152, 384), (384, 384), (1536, 384), (384, 1536), (768, 384)]:
    x = torch.randn(M, K, device='cuda'); w = torch.randn(N, K, device='cuda') * 0.05
    y = torch.empty(M, N, device='cuda'); b = torch.randn(N, device='cuda')
    run(f'NT  {M}x{N}x{K}', lambda: hip.gemm(x, w, y, M, N, K, K, K, N, 1, 1, bias=b), 2.0 * M * N * K)
    dy = torch.randn(M, N, device='cuda'); dx = torch.empty(M, K, device='cuda')
    run(f'NN  {M}x{K}x{N}', lambda: hip.gemm(dy, w, dx, M, K, N, N, K, K, 1, 0), 2.0 * M * N * K)
    dw = torch.empty(N, K, device='cuda')
    run(f'TN  {N}x{K}x{M}', lambda: hip.gemm(dy, x, dw, N, K, M, N, K, K, 0, 0), 2.0 * M * N * K)
B, N, H, dh = 128, 197, 6, 64
qkv = torch.randn(B * N, 3 * H * dh, device='cuda'); o = torch.empty(B * N, H * dh, device='cuda')
lse = torch.empty(2 * B * H, N, device="cuda"); do = torch.randn_like(o); dqkv = torch.empty_like(qkv)
run('attn fwd', lambda: hip.attention_fwd(qkv, o, lse, B, N, H, dh, 0.125), 4.0 * B * H * N * N * dh)
run('attn bwd', lambda: hip.attention_bwd(qkv, o, lse, do, dqkv, B, N, H, dh, 0.125), 10.0 * B * H * N * N * dh)
xx = torch.randn(M, D, device='cuda'); g = torch.ones(D, device='cuda'); bb = torch.zeros(D, device='cuda')
yy = torch.empty_like(xx); mean = torch.empty(M, device='cuda'); rstd = torch.empty(M, device='cuda')
run('ln fwd (GB/s in TF col x1e3)', lambda: hip.layernorm_fwd(xx, g, bb, yy, mean, rstd, M, D, 1e-6), 8.0 * M * D)
part = torch.empty(hip.layernorm_bwd_blocks(M), 2, D, device='cuda')
run('ln bwd (16B/elem)', lambda: hip.layernorm_bwd(yy, xx, g, mean, rstd, xx, yy, part, M, D), 16.0 * M * D)
