"""Lab: cost of the guarded (ragged-shape) GEMM path next to the unguarded one."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ofb_amd import hip
lib = hip.lib(); lib.ofb_gemm_workspace_bytes.restype = C.c_int64
def run(M, N, K, a_kc=1, b_kc=1):
    x = torch.randn(M, K, device='cuda'); w = torch.randn(N, K, device='cuda') if b_kc else torch.randn(K, N, device='cuda')
    y = torch.empty(M, N, device='cuda')
    g = hip.GemmArgs()
    g.A, g.B, g.C = x.data_ptr(), w.data_ptr(), y.data_ptr()
    g.M, g.N, g.K, g.lda, g.ldb, g.ldc, g.a_kc, g.b_kc, g.alpha = M, N, K, K, (K if b_kc else N), N, a_kc, b_kc, 1.0
    g.rs_div = g.ks_div = 1
    need = lib.ofb_gemm_workspace_bytes(C.byref(g))
    ws = torch.empty(max(need, 4) // 4, device='cuda')
    g.workspace, g.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(5): assert lib.ofb_gemm_f32(C.byref(g), st) == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): lib.ofb_gemm_f32(C.byref(g), st)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f'M={M} N={N} K={K} kc=({a_kc},{b_kc}): {ms*1e3:7.1f} us {2.0*M*N*K/ms/1e9:6.1f} TF')
for shp in [(25216, 1536, 384), (25216, 1536, 392), (25216, 1528, 384), (25220, 1536, 384), (50432, 792, 264), (50432, 264, 768), (50432, 576, 264)]:
    run(*shp)
run(50432, 264, 768, 1, 0)
