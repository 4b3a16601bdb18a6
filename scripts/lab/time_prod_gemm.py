"""times the production ofb_gemm_f32 on the lab's shapes (same-box comparison for scripts/lab/gemm_p_lab.hip)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ofb_amd import hip

def t(form, M, N, K, iters=20):
    a_kc, b_kc = form != 'tn', form == 'nt'
    A = torch.randn((M, K) if a_kc else (K, M), device='cuda')
    B = torch.randn((N, K) if b_kc else (K, N), device='cuda')
    C = torch.empty(M, N, device='cuda')
    f = lambda: hip.gemm(A, B, C, M, N, K, K if a_kc else M, K if b_kc else N, N, a_kc, b_kc)
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    print(f'prod {form} M {M:6d} N {N:5d} K {K:5d}: {us:8.1f} us  {2.0*M*N*K/us/1e6:7.1f} TFLOP/s')

for args in [('nt', 32768, 2048, 384), ('nt', 32768, 1024, 1536), ('nn', 32768, 2048, 384), ('tn', 1536, 512, 25216),
             ('nt', 25216, 1536, 384), ('nt', 25216, 384, 1536)]:
    t(*args)
