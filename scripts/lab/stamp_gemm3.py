"""Lab: 3-stamp variant (top, loads issued, MFMA+staging block done) for the interleaved-staging loop."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ofb_amd import hip
lib = C.CDLL(sys.argv[1]); lib.ofb_gemm_f32.restype = C.c_int; lib.ofb_gemm_workspace_bytes.restype = C.c_int64
M, N, K = 128 * 197, 1536, 384
x, w, y = torch.randn(M, K, device='cuda'), torch.randn(N, K, device='cuda'), torch.empty(M, N, device='cuda')
g = hip.GemmArgs()
g.A, g.B, g.C = x.data_ptr(), w.data_ptr(), y.data_ptr()
g.M, g.N, g.K, g.lda, g.ldb, g.ldc, g.a_kc, g.b_kc, g.alpha = M, N, K, K, K, N, 1, 1, 1.0
g.rs_div = g.ks_div = 1
need = lib.ofb_gemm_workspace_bytes(C.byref(g))
ws = torch.empty(max(need, 4) // 4, device='cuda'); g.workspace, g.workspace_bytes = ws.data_ptr(), ws.numel() * 4
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(30): assert lib.ofb_gemm_f32(C.byref(g), st) == 0
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): lib.ofb_gemm_f32(C.byref(g), st)
e1.record(); torch.cuda.synchronize()
wall_us = e0.elapsed_time(e1) * 100
print(f'wall per launch {wall_us:.1f} us')
buf = (C.c_ulonglong * 8192)()
assert lib.ofb_diag_gemm_stamps(buf) == 0
a = np.frombuffer(buf, dtype=np.uint64).astype(np.int64).reshape(2, 4096)
for wg in range(2):
    s = a[wg][a[wg] > 0]
    n = (len(s) // 3) * 3
    s = s[:n].reshape(-1, 3)
    d = np.diff(s, axis=1)
    bar = s[1:, 0] - s[:-1, 2]
    it = s[1:, 0] - s[:-1, 0]
    ok = it < 20000
    span = int(s[-1, 2] - s[0, 0])
    print(f'wg {wg}: iterations {len(s)}  per-iteration cycles median {np.median(it[ok]):.0f};  first->last stamp {span} ticks = {span / wall_us / 1e3:.2f} GHz-equivalent over ~the whole launch')
    for name, col in zip(['issue global loads', 'reads + MFMA + staging'], d[:-1].T):
        print(f'   {name:28s} median {np.median(col[ok]):7.0f}   mean {col[ok].mean():7.0f}')
    print(f'   {"epilogue/barrier":28s} median {np.median(bar[ok]):7.0f}   mean {bar[ok].mean():7.0f}')
