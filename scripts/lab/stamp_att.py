"""In-kernel s_memtime stamps of the split-engine attention backward (lab build with -DOFB_ATT_STAMPS):
    bash scripts/lab/build_att_stamps.sh && OFB_LIB_PATH=scripts/lab/bin/libofb_attstamps.so python scripts/lab/stamp_att.py
Prints, per wave of workgroup 0, the cycles spent in each section of every query block."""
import ctypes as C
import sys
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ofb_amd import hip
B, N, H, dh = 128, 197, 6, 64
torch.manual_seed(0)
qkv = torch.randn(B * N, 3 * H * dh, device='cuda')
o = torch.empty(B * N, H * dh, device='cuda'); lse = torch.empty(2 * B * H, N, device='cuda'); do = torch.randn_like(o)
dq = torch.empty_like(qkv)
hip.attention_fwd(qkv, o, lse, B, N, H, dh, 0.125)
for _ in range(5):
    hip.attention_bwd(qkv, o, lse, do, dq, B, N, H, dh, 0.125)
torch.cuda.synchronize()
buf = (C.c_ulonglong * (8 * 64))()
hip.lib().ofb_diag_att_stamps(buf)
st = [[buf[w * 64 + i] for i in range(64)] for w in range(8)]
t0 = min(s[0] for s in st)
names = ['S/dP', 'P/dS', 'dV/dK', 'wait Y', 'dQ+store', 'st.store', 'wait X']
print('wave  prologue ' + ' '.join(f'{n:>9s}' for n in names) + '   (cycles, summed over the 7 query blocks)   total   epilogue')
for w in range(8):
    s = st[w]
    tot = [0] * 7
    for qb in range(7):
        b = 2 + 8 * qb
        seg = [s[b + 1] - s[b], s[b + 2] - s[b + 1], s[b + 3] - s[b + 2], s[b + 4] - s[b + 3], s[b + 5] - s[b + 4], s[b + 6] - s[b + 5], s[b + 7] - s[b + 6]]
        tot = [a + c for a, c in zip(tot, seg)]
    print(f'{w:4d}  {s[1] - s[0]:8d} ' + ' '.join(f'{v:9d}' for v in tot) + f'   {s[60] - s[0]:8d} {s[60] - s[2 + 8 * 6 + 7]:8d}')
print('prologue of wave 0: loads issued', st[0][58] - st[0][0], ' K planes written', st[0][59] - st[0][58], ' lse + stage 0 stored', st[0][61] - st[0][59], ' V split', st[0][62] - st[0][61], ' to barrier', st[0][1] - st[0][62])
print('block 0 of wave 0:', [st[0][2 + i + 1] - st[0][2 + i] for i in range(7)], ' wave 3:', [st[3][2 + i + 1] - st[3][2 + i] for i in range(7)])
