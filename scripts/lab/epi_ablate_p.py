"""epilogue ablation of the P engine on the fc1 shape (M 25216, N 1536, K 384) and the fc2 shape"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ofb_amd import hip
M, D, HID = 128 * 197, 384, 1536
def run(tag, fn, flops, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f'{tag:52s} {ms*1e3:8.1f} us {flops/ms/1e9:7.1f} TF')
r = lambda *s: torch.randn(*s, device='cuda')
x = r(M, D); xp = hip.to_pformat(x)
w3, b3, g3 = r(HID, D), r(HID), r(HID); w3p = hip.to_pformat(w3)
y = torch.empty(M, HID, device='cuda'); aux = torch.empty(M, HID, device='cuda'); hP = hip.PMat(M, HID, 'cuda')
F = 2. * M * HID * D
run('fc1 shape: plain f32 out', lambda: hip.gemm_p(xp, w3p, 1, 1, M, HID, D, C_out=y, ldc=HID), F)
run('fc1 shape: bias+gate f32 out', lambda: hip.gemm_p(xp, w3p, 1, 1, M, HID, D, C_out=y, ldc=HID, bias=b3, colscale=g3), F)
run('fc1 shape: P out only', lambda: hip.gemm_p(xp, w3p, 1, 1, M, HID, D, Cp=hP), F)
run('fc1 shape: GELU f32 out, no aux', lambda: hip.gemm_p(xp, w3p, 1, 1, M, HID, D, C_out=y, ldc=HID, bias=b3, colscale=g3, act=hip.ACT_GELU), F)
run('fc1 shape: GELU + aux, f32 out', lambda: hip.gemm_p(xp, w3p, 1, 1, M, HID, D, C_out=y, ldc=HID, bias=b3, colscale=g3, act=hip.ACT_GELU, aux=aux, ldaux=HID), F)
run('fc1 shape: GELU + aux, P out (the real fc1)', lambda: hip.gemm_p(xp, w3p, 1, 1, M, HID, D, Cp=hP, bias=b3, colscale=g3, act=hip.ACT_GELU, aux=aux, ldaux=HID), F)
run('fc1 shape: dGELU(aux) P out (the real dH)', lambda: hip.gemm_p(xp, w3p, 1, 1, M, HID, D, Cp=hP, act=hip.ACT_DGELU, aux=aux, ldaux=HID), F)
run('fc1 shape: GELU_GRAD + aux, P out (the real fc1 now)', lambda: hip.gemm_p(xp, w3p, 1, 1, M, HID, D, Cp=hP, bias=b3, colscale=g3, act=hip.ACT_GELU_GRAD, aux=aux, ldaux=HID), F)
run('fc1 shape: MULAUX P out (the real dH now)', lambda: hip.gemm_p(xp, w3p, 1, 1, M, HID, D, Cp=hP, act=hip.ACT_MULAUX, aux=aux, ldaux=HID), F)
cs = torch.empty(HID, device='cuda')
run('fc1 shape: MULAUX P out + column sums', lambda: hip.gemm_p(xp, w3p, 1, 1, M, HID, D, Cp=hP, act=hip.ACT_MULAUX, aux=aux, ldaux=HID, colsum_out=cs), F)
run('fc1 shape: P out + column sums', lambda: hip.gemm_p(xp, w3p, 1, 1, M, HID, D, Cp=hP, colsum_out=cs), F)
w4 = r(D, HID); w4p = hip.to_pformat(w4); y2 = torch.empty(M, D, device='cuda'); rs = torch.rand(M, device='cuda')
run('fc2 shape: plain f32 out', lambda: hip.gemm_p(hP, w4p, 1, 1, M, D, HID, C_out=y2, ldc=D), F)
run('fc2 shape: bias+rowscale+resid (real)', lambda: hip.gemm_p(hP, w4p, 1, 1, M, D, HID, C_out=y2, ldc=D, bias=b3[:D].contiguous(), rowscale=rs, resid=x, ldr=D), F)
