"""Lab: per-phase cycle breakdown of the GEMM K-iteration from in-kernel s_memtime stamps (library built with -DOFB_GEMM_STAMPS).
Stamps per iteration (wave 0 of workgroups 8 and 264, which share a CU only by luck): 0 top of iteration, 1 global loads issued,
2 MFMAs issued (+ fragment reads waited), 3 prefetched tile arrived (vmcnt 0), 4 split + LDS writes done; the gap to the next
iteration's stamp 0 is the barrier."""
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
buf = (C.c_ulonglong * 8192)()
assert lib.ofb_diag_gemm_stamps(buf) == 0
a = np.frombuffer(buf, dtype=np.uint64).astype(np.int64).reshape(2, 4096)
for wg in range(2):
    s = a[wg][a[wg] > 0]
    n = (len(s) // 5) * 5
    s = s[:n].reshape(-1, 5)
    d = np.diff(s, axis=1)                       # 0->1 load issue, 1->2 reads+MFMA, 2->3 vmcnt wait, 3->4 split+write
    bar = s[1:, 0] - s[:-1, 4]                   # barrier (+ loop overhead)
    it = s[1:, 0] - s[:-1, 0]
    ok = it < 20000                              # drop tile boundaries (epilogue)
    print(f'wg {wg}: iterations {len(s)}  per-iteration cycles median {np.median(it[ok]):.0f}')
    for name, col in zip(['issue global loads', 'fragment reads + 24 MFMA', 'wait prefetched tile', 'split + ds_write'], d[:-1].T):
        print(f'   {name:28s} median {np.median(col[ok]):7.0f}   mean {col[ok].mean():7.0f}')
    print(f'   {"barrier":28s} median {np.median(bar[ok]):7.0f}   mean {bar[ok].mean():7.0f}')
