import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ofb_amd import hip
M, N, K = 25216, 1152, 384
x = torch.randn(M, K, device='cuda'); w = torch.randn(N, K, device='cuda'); y = torch.empty(M, N, device='cuda')
b = torch.randn(N, device='cuda'); g = torch.randn(N, device='cuda'); res = torch.randn(M, N, device='cuda')
rs = torch.rand(M, device='cuda')
def run(tag, **kw):
    f = lambda: hip.gemm(x, w, y, M, N, K, K, K, N, 1, 1, **kw)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f'{tag:34s} {ms*1e3:8.1f} us {2.0*M*N*K/ms/1e9:6.1f} TF')
run('plain')
run('bias', bias=b)
run('bias+colscale', bias=b, colscale=g)
run('resid', resid=res, ldr=N)
run('rowscale', rowscale=rs, rs_div=1)
run('bias+rowscale+resid', bias=b, rowscale=rs, rs_div=1, resid=res, ldr=N)
run('plain again')
