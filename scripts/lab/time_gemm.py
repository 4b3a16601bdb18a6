import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ofb_amd import hip
lib = C.CDLL(sys.argv[1]); lib.ofb_gemm_f32.restype = C.c_int
M, D = 128 * 197, 384
def run(N, K, a_kc, b_kc):
    x = torch.randn(M, K, device='cuda'); w = torch.randn(N, K, device='cuda'); y = torch.empty(M, N, device='cuda')
    g = hip.GemmArgs()
    g.A, g.B, g.C = x.data_ptr(), w.data_ptr(), y.data_ptr()
    g.M, g.N, g.K, g.lda, g.ldb, g.ldc, g.a_kc, g.b_kc, g.alpha = M, N, K, K, K, N, a_kc, b_kc, 1.0
    lib.ofb_gemm_workspace_bytes.restype = C.c_int64
    need = lib.ofb_gemm_workspace_bytes(C.byref(g))
    ws = torch.empty(max(need, 4) // 4, device='cuda')
    g.workspace, g.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(3): lib.ofb_gemm_f32(C.byref(g), st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): lib.ofb_gemm_f32(C.byref(g), st)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    return ms * 1e3, 2.0 * M * N * K / ms / 1e9
r1 = run(1152, 384, 1, 1); r2 = run(1536, 384, 1, 1)
print(f'{sys.argv[2]:50s} NT 1152: {r1[0]:7.1f} us {r1[1]:6.1f} TF | NT 1536: {r2[0]:7.1f} us {r2[1]:6.1f} TF')
