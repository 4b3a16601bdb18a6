"""Lab: the DeiT-B (configs[3], bs 64) GEMM forms with their epilogues, for comparing tile configurations."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ofb_amd import hip
lib = C.CDLL(sys.argv[1]); lib.ofb_gemm_f32.restype = C.c_int; lib.ofb_gemm_workspace_bytes.restype = C.c_int64
M, D, HID = 64 * 197, 768, 3072
r = lambda *s: torch.randn(*s, device='cuda')
keep = []
def case(A, B, Cm, m, n, k, lda, ldb, ldc, a_kc, b_kc, **kw):
    g = hip.GemmArgs()
    g.A, g.B, g.C = A.data_ptr(), B.data_ptr(), Cm.data_ptr()
    g.M, g.N, g.K, g.lda, g.ldb, g.ldc, g.a_kc, g.b_kc, g.alpha = m, n, k, lda, ldb, ldc, a_kc, b_kc, 1.0
    g.rs_div = g.ks_div = 1
    for key, v in kw.items():
        setattr(g, key, v.data_ptr() if isinstance(v, torch.Tensor) else v)
        keep.append(v)
    need = lib.ofb_gemm_workspace_bytes(C.byref(g))
    ws = torch.empty(max(need, 4) // 4, device='cuda'); keep.append(ws)
    g.workspace, g.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    return g
def time(g, flops):
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(60): assert lib.ofb_gemm_f32(C.byref(g), st) == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): lib.ofb_gemm_f32(C.byref(g), st)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 30
    return f'{ms*1e3:6.1f}us {flops/ms/1e9:5.1f}TF'
x, h, hp = r(M, D), r(M, HID), r(M, HID)
w1, b1, g1, w2, bD, wq = r(HID, D), r(HID), r(HID), r(D, HID), r(D), r(3 * D, D)
rs = torch.rand(64, device='cuda').repeat_interleave(197)
y, yh, yq = torch.empty(M, D, device='cuda'), torch.empty(M, HID, device='cuda'), torch.empty(M, 3 * D, device='cuda')
F1, FQ, FP = 2. * M * HID * D, 2. * M * 3 * D * D, 2. * M * D * D
out = [
    ('qkv', time(case(x, wq, yq, M, 3 * D, D, D, D, 3 * D, 1, 1, bias=r(3 * D), colscale=r(3 * D)), FQ)),
    ('proj', time(case(x, r(D, D), y, M, D, D, D, D, D, 1, 1, bias=bD, rowscale=rs, resid=x, ldr=D), FP)),
    ('fc1', time(case(x, w1, yh, M, HID, D, D, D, HID, 1, 1, bias=b1, colscale=g1, act=hip.ACT_GELU, aux=hp, ldaux=HID), F1)),
    ('fc2', time(case(h, w2, y, M, D, HID, HID, HID, D, 1, 1, bias=bD, rowscale=rs, resid=x, ldr=D), F1)),
    ('dH', time(case(x, w2, yh, M, HID, D, D, HID, HID, 1, 0, rowscale=rs, act=hip.ACT_DGELU, aux=hp, ldaux=HID), F1)),
    ('dXfc1', time(case(h, w1, y, M, D, HID, HID, D, D, 1, 0, resid=x, ldr=D), F1)),
    ('dXqkv', time(case(yq, wq, y, M, D, 3 * D, 3 * D, D, D, 1, 0, resid=x, ldr=D), FQ)),
    ('dW1', time(case(h, x, torch.empty(HID, D, device='cuda'), HID, D, M, HID, D, D, 0, 0, a_colsum=torch.empty(HID, device='cuda')), F1)),
]
print(f'{sys.argv[2]:12s} ' + ' | '.join(f'{k} {v}' for k, v in out))
