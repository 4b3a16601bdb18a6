"""in-kernel cycle shares of the P engine (build: hipcc -DOFB_P_STAMPS ... -> /tmp/libofb_stamps.so): per unit K-loop vs epilogue"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
from ofb_amd import hip
hip.LIB_PATH = sys.argv[1]
M, D, HID = 128 * 192, 384, 1536        # 1536 tiles: whole rounds at W = 512 and 256
r = lambda *s: torch.randn(*s, device='cuda')
x = r(M, D); xp = hip.to_pformat(x)
w3, b3, g3 = r(HID, D), r(HID), r(HID); w3p = hip.to_pformat(w3)
y = torch.empty(M, HID, device='cuda'); aux = torch.empty(M, HID, device='cuda'); hP = hip.PMat(M, HID, 'cuda')
W = int(os.environ.get('OFB_GEMM_P_WCAP', 512))
def stamps(tag, fn, units=3):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * (1024 * 8 * 4))()
    assert hip.lib().ofb_diag_p_stamps(buf) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 8, 4).astype(np.int64)[:W, :units]
    if os.environ.get('OFB_GEMM_P_STAGGER'):
        hw = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 8, 4)[:W, 7, 3].astype(np.int64)
        print('   TG_ID histogram', np.bincount((hw >> 16) & 15), ' CU ids', len(set((hw >> 8) & 0xff)), ' SE', len(set((hw >> 13) & 7)))
    pro, kl, ep = a[:, :, 1] - a[:, :, 0], a[:, :, 2] - a[:, :, 1], a[:, :, 3] - a[:, :, 2]
    gap = a[:, 1:, 0] - a[:, :-1, 3]
    tot = a[:, units - 1, 3] - a[:, 0, 0]
    skew = a[:, 0, 0] - a[:, 0, 0].min()
    us = tot / 1e2
    print(f'{tag:40s} prologue {np.median(pro):7.0f}  K loop {np.median(kl):7.0f}  epilogue {np.median(ep):7.0f}  gap {np.median(gap):5.0f} cycles (100 MHz ticks?); '
          f'total/WG {np.median(tot):8.0f}; start skew max {skew.max()}')
stamps('fc1 plain f32', lambda: hip.gemm_p(xp, w3p, 1, 1, M, HID, D, C_out=y, ldc=HID))
stamps('fc1 real (GELU aux P)', lambda: hip.gemm_p(xp, w3p, 1, 1, M, HID, D, Cp=hP, bias=b3, colscale=g3, act=hip.ACT_GELU, aux=aux, ldaux=HID))
w4 = r(D, HID); w4p = hip.to_pformat(w4); y2 = torch.empty(M, D, device='cuda')
stamps('fc2 plain', lambda: hip.gemm_p(hP, w4p, 1, 1, M, D, HID, C_out=y2, ldc=D), units=1)
import time
for tag, fn in [('fc1 plain', lambda: hip.gemm_p(xp, w3p, 1, 1, M, HID, D, C_out=y, ldc=HID)), ('fc1 real', lambda: hip.gemm_p(xp, w3p, 1, 1, M, HID, D, Cp=hP, bias=b3, colscale=g3, act=hip.ACT_GELU, aux=aux, ldaux=HID))]:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    print(tag, 'wall us per call', e0.elapsed_time(e1) * 100)
