"""Times ofb_layernorm_bwd_h_rn / ofb_layernorm_bwd_h on the DeiT-S bs-128 token matrix (run on the GPU box; OFB_LIB_PATH = another build)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ofb_amd import hip
rows, D = 128 * 197, int(sys.argv[1]) if len(sys.argv) > 1 else 384
r = lambda *s: torch.randn(*s, device='cuda')
dy, x, gamma, dres = r(rows, D), r(rows, D), r(D), r(rows, D)
mean, rstd = r(rows), torch.rand(rows, device='cuda') + 0.5
dx = torch.empty(rows, D, device='cuda')
parts = torch.empty(hip.layernorm_bwd_blocks(rows), 3, D, device='cuda')
dxP = hip.HMat.for_rows_written_by_kernel(rows, D, 'cuda')
rs = torch.rand(128, device='cuda')
rn = torch.rand(394, device='cuda')
# keep the caches honest: a 300-MB buffer is streamed between calls
junk = torch.empty(75_000_000, device='cuda')
def t(fn, n=20):
    best = 1e9
    for _ in range(3):
        tot = 0.
        for _ in range(n):
            junk.add_(1.0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize()
            tot += e0.elapsed_time(e1)
        best = min(best, tot / n)
    return best * 1e3
print(f'D {D}: bwd_h_rn (no dres) {t(lambda: hip.layernorm_bwd_h_rn(dy, x, gamma, mean, rstd, dx, parts, dxP, rs, 197, rows, D, rn, 1.41)):.1f} us   '
      f'bwd_h (dres, own bound pass) {t(lambda: hip.layernorm_bwd_h(dy, x, gamma, mean, rstd, dres, dx, parts, dxP, rs, 197, rows, D)):.1f} us   '
      f'bwd_h (no dres) {t(lambda: hip.layernorm_bwd_h(dy, x, gamma, mean, rstd, None, dx, parts, dxP, rs, 197, rows, D)):.1f} us')
