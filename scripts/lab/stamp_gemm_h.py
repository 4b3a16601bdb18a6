"""Lab: where a GEMM unit's cycles go (s_memtime stamps of wave 0 of every workgroup; build with -DOFB_H_STAMPS, see stamp_gemm_h.sh).
usage: OFB_LIB_PATH=/tmp/libofb_stamps.so python scripts/lab/stamp_gemm_h.py [qkv|proj|fc1|fc2|dh|dxfc1|dw]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from ofb_amd import hip
which = sys.argv[1] if len(sys.argv) > 1 else 'qkv'
M, D, H3, HID = 128 * 197, 384, 1152, 1536
r = lambda *s: torch.randn(*s, device='cuda')
P, G = hip.to_hformat, hip.gemm_h
x, rs = r(M, D), torch.rand(128, device="cuda").repeat_interleave(197)
xp = P(x)
hpre = torch.rand(M, HID, device='cuda'); hP = hip.HMat(M, HID, 'cuda'); dhP = hip.HMat(M, HID, 'cuda')
y, y2 = torch.empty(M, H3, device='cuda'), torch.empty(M, D, device='cuda')
w, w2, w3, w4 = P(r(H3, D)), P(r(D, D)), P(r(HID, D)), P(r(D, HID))
b, b2, b3, g3 = r(H3), r(D), r(HID), torch.rand(HID, device='cuda')
G(xp, w3, 1, 1, M, HID, D, Cp=hP, bias=b3, colscale=g3, act=hip.ACT_GELU_GRAD, aux=hpre, ldaux=HID)
G(xp, w4, 1, 0, M, HID, D, Cp=dhP, act=hip.ACT_MULAUX, aux=hpre, ldaux=HID)
dw3 = torch.empty(HID, D, device='cuda')
hpre_t = hip.aux_t(M, HID, 'cuda')
G(xp, w3, 1, 1, M, HID, D, Cp=hP, bias=b3, colscale=g3, act=hip.ACT_GELU_GRAD_T, aux=hpre_t)
calls = {
    'fc1t': (lambda: G(xp, w3, 1, 1, M, HID, D, Cp=hP, bias=b3, colscale=g3, act=hip.ACT_GELU_GRAD_T, aux=hpre_t), 2. * M * HID * D),
    'dht': (lambda: G(xp, w4, 1, 0, M, HID, D, Cp=dhP, act=hip.ACT_MULAUX_T, aux=hpre_t, want_colpart=True), 2. * M * HID * D),
    'qkv': (lambda: G(xp, w, 1, 1, M, H3, D, C_out=y, ldc=H3, bias=b), 2. * M * H3 * D),
    'proj': (lambda: G(xp, w2, 1, 1, M, D, D, C_out=y2, ldc=D, bias=b2, rowscale=rs, resid=x, ldr=D), 2. * M * D * D),
    'fc1': (lambda: G(xp, w3, 1, 1, M, HID, D, Cp=hP, bias=b3, colscale=g3, act=hip.ACT_GELU_GRAD, aux=hpre, ldaux=HID), 2. * M * HID * D),
    'fc2': (lambda: G(hP, w4, 1, 1, M, D, HID, C_out=y2, ldc=D, bias=b2, rowscale=rs, resid=x, ldr=D), 2. * M * D * HID),
    'dh': (lambda: G(xp, w4, 1, 0, M, HID, D, Cp=dhP, act=hip.ACT_MULAUX, aux=hpre, ldaux=HID), 2. * M * HID * D),
    'dxfc1': (lambda: G(dhP, w3, 1, 0, M, D, HID, C_out=y2, ldc=D, resid=x, ldr=D), 2. * M * HID * D),
    'dw': (lambda: G(dhP, xp, 0, 0, HID, D, M, C_out=dw3, ldc=D), 2. * M * HID * D),
}
fn, flops = calls[which]
for _ in range(200): fn()                    # warm: the clock settles under load
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); fn(); e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3
buf = (C.c_ulonglong * (1024 * 8 * 4))()
rc = hip.lib().ofb_diag_h_stamps(buf)
assert rc == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 8, 4).astype(np.int64)
nwg = 512
st = st[:nwg]
rt0, rt1, hw, mt0 = st[:, 7, 0], st[:, 7, 1], st[:, 7, 2], st[:, 7, 3]
print(f'{which}: wall {us:.1f} us = {flops / us / 1e6:.1f} TFLOP/s; kernel span by s_memrealtime (100 MHz): {np.median(rt1 - rt0) / 100:.1f} us')
units = []
for u in range(7):
    ok = st[:, u, 3] > st[:, u, 0]
    ok &= st[:, u, 0] > 0
    if ok.sum() < 8: break
    a = st[ok, u]
    units.append((u, ok.sum(), np.median(a[:, 1] - a[:, 0]), np.median(a[:, 2] - a[:, 1]), np.median(a[:, 3] - a[:, 2])))
for u, n, pro, kl, ep in units:
    print(f'  unit {u}: {n} workgroups; prologue {pro:.0f}  K loop {kl:.0f}  epilogue {ep:.0f} cycles (median)')
# clock: shader cycles per 10 ns over the whole kernel of each workgroup
last = np.max(st[:, :7, 3], axis=1)
clk = (last - mt0) / np.maximum(rt1 - rt0, 1) * 100e6 / 1e9
print(f'  in-kernel clock: median {np.median(clk):.2f} GHz (min {clk.min():.2f}, max {clk.max():.2f})')

# co-residency: workgroups per CU key (XCC | SE, SH, CU of HW_ID) and how much of a pair's epilogues overlap in time
key = ((hw >> 32) & 15) * 256 + ((hw >> 8) & 255)
uniq, cnt = np.unique(key, return_counts=True)
print(f'  CU keys: {len(uniq)} distinct; workgroups per key: ' + ', '.join(f'{c}: {int((cnt == c).sum())}' for c in sorted(set(cnt))))
for u in range(min(len(units), 3)):
    ov, tot = 0, 0
    for k in uniq[cnt == 2]:
        i, j = np.nonzero(key == k)[0]
        a0, a1, b0, b1 = st[i, u, 2], st[i, u, 3], st[j, u, 2], st[j, u, 3]
        if min(a0, b0) <= 0: continue
        ov += max(0, min(a1, b1) - max(a0, b0)); tot += min(a1 - a0, b1 - b0)
    if tot: print(f'  unit {u}: epilogue overlap inside a CU pair: {ov / tot:.2f} of the shorter epilogue')
if os.environ.get('OFB_STAMP_DUMP'):
    base = st[:, 0, 0].min()
    for k in uniq[cnt == 2][:6]:
        i, j = np.nonzero(key == k)[0]
        for n in (i, j):
            print(f'   key {k:4d} wg {n:3d}: ' + ' | '.join(' '.join(f'{int(st[n, u, s] - base):7d}' for s in range(4)) for u in range(3)))
