"""Lab: the two LayerNorm plane kernels of the step at its shape ([25216][384]) against plain copies of the same byte counts.
`cold`: eight buffer sets in rotation (1.2 GB: nothing is left in the Infinity Cache); `warm`: one set (what a micro-benchmark loop sees).
usage: python scripts/lab/ln_bench.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ofb_amd import hip

M, D = 128 * 197, 384
dev = 'cuda'
NSET = 8
sets = []
for i in range(NSET):
    x, dy = torch.randn(M, D, device=dev), torch.randn(M, D, device=dev)
    sets.append(dict(x=x, dy=dy, dx=torch.empty_like(x), y=torch.empty_like(x), yP=hip.HMat(M, D, dev), dxP=hip.HMat(M, D, dev),
                     mean=torch.empty(M, device=dev), rstd=torch.empty(M, device=dev)))
g, b = torch.rand(D, device=dev) + 0.5, torch.randn(D, device=dev)
nblk = hip.layernorm_bwd_blocks(M)
part = torch.empty(nblk, 3, D, device=dev)
rs = torch.rand(128, device=dev)
# the row-norm ingredients the producing GEMM hands to the backward (any positive numbers do for timing; use a real product once)
xp = hip.to_hformat(sets[0]['x'])
wp = hip.to_hformat(torch.randn(D, D, device=dev) * 0.05)
out = torch.empty(M, D, device=dev)
rn = hip.gemm_h(xp, wp, 1, 1, M, D, D, C_out=out, ldc=D, rn=(g, rs.repeat_interleave(197)))
for s in sets:
    hip.layernorm_fwd_h(s['x'], g, b, None, s['yP'], s['mean'], s['rstd'], M, D, 1e-6)


def timeit(fn, n_sets, reps=40):
    for i in range(8):
        fn(sets[i % n_sets])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        fn(sets[i % n_sets])
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def fwd(s): hip.layernorm_fwd_h(s['x'], g, b, None, s['yP'], s['mean'], s['rstd'], M, D, 1e-6)
def bwd(s): hip.layernorm_bwd_h_rn(s['dy'], s['x'], g, s['mean'], s['rstd'], s['dx'], part, s['dxP'], rs, 197, M, D, rn[0], rn[1])
def copy1(s): s['y'].copy_(s['x'])                                   # 38.7 MB read + 38.7 MB written: the forward's bytes
def copy2(s): s['y'].copy_(s['x']); s['dx'].copy_(s['dy'])           # twice that: the backward's bytes (x, dy read; dx, planes written)

mb = M * D * 4 / 1e6
for name, fn, bytes_mb in (('ln_fwd_h (planes only)', fwd, 2 * mb), ('ln_bwd_h_rn', bwd, 4 * mb), ('copy 1x', copy1, 2 * mb), ('copy 2x (two launches)', copy2, 4 * mb)):
    cold, warm = timeit(fn, NSET), timeit(fn, 1)
    print(f'{name:26s} cold {cold:6.1f} us = {bytes_mb / cold / 1e3 * 1e3:5.2f} TB/s   warm {warm:6.1f} us = {bytes_mb / warm:5.2f} TB/s   (blocks {nblk})')
