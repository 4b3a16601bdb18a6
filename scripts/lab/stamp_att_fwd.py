"""In-kernel s_memtime stamps of the attention forward (python scripts/lab/make_attention_fwd_stamps.py writes the stamped copy of the product kernel and builds scripts/lab/bin/libofb_attfstamps.so):
    OFB_LIB_PATH=scripts/lab/bin/libofb_attfstamps.so python scripts/lab/stamp_att_fwd.py
Per wave of one mid-grid workgroup: cycles per phase, summed over the seven key blocks."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from ofb_amd import hip
B, N, H, dh = 128, 197, 6, 64
M, Hd = B * N, H * dh
g = torch.Generator(device='cuda').manual_seed(1)
qkv = torch.randn(M, 3 * Hd, device='cuda', generator=g)
o, lse = torch.empty(M, Hd, device='cuda'), torch.empty(2 * B * H, N, device='cuda')
qb = qkv.abs().max().reshape(1) * 1.5
oP = hip.HMat.for_rows_written_by_kernel(M, Hd, 'cuda')
for _ in range(50):
    hip.attention_fwd_h(qkv, o, oP, lse, B, N, H, dh, 0.125, qb)
torch.cuda.synchronize()
buf = (C.c_ulonglong * (13 * 40))()
assert hip.lib().ofb_diag_attf_stamps(buf) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(13, 40).astype(np.int64)
t0 = st[:, 0].min()
print('wave  start  prologue |    S-tiles  softmax+split   P.V   stage_store  barrier-wait | loop total | f32 stores  plane patch | total')
for w in range(13):
    s = st[w]
    seg = np.zeros(5, dtype=np.int64)
    prev = s[1]
    for kb in range(7):
        b = 2 + 5 * kb
        seg += np.array([s[b] - prev, s[b + 1] - s[b], s[b + 2] - s[b + 1], s[b + 3] - s[b + 2], s[b + 4] - s[b + 3]])
        prev = s[b + 4]
    print(f'{w:4d} {s[0] - t0:6d} {s[1] - s[0]:9d} | ' + ' '.join(f'{v:10d}' for v in seg) + f' | {s[37] - s[1]:10d} | {s[38] - s[37]:10d} {s[39] - s[38]:11d} | {s[39] - s[0]:6d}')
print('block by block, wave 5: ' + ' | '.join(' '.join(str(int(st[5][2 + 5 * kb + i] - (st[5][1] if (kb == 0 and i == 0) else st[5][2 + 5 * kb + i - 1]))) for i in range(5)) for kb in range(7)))
