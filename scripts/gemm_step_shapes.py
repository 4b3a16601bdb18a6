"""Times every GEMM configuration of one DeiT-S bs=128 search step on the H-format engine (operands converted beforehand), with its
real epilogue and output form (run on the GPU box).  OFB_LIB_PATH=<other build> runs the same shapes on another library."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ofb_amd import hip
M, D, H3, HID = 128 * 197, 384, 1152, 1536
# --dims=D,H3,HID[,M]: another block geometry (e.g. a post-compress block of the configs[4] subnet: --dims=264,576,768)
for a in sys.argv[1:]:
    if a.startswith('--dims='):
        v = [int(x) for x in a[7:].split(',')]
        D, H3, HID = v[:3]
        if len(v) > 3: M = v[3]
# --ab KEY=V1,V2[,V3]: every product is timed under each value of the run-time switch hip.TUNE_<KEY> (interleaved rounds in this one
# process, MI355X guide rule 24); the table then carries one column per value
AB = None
for a in sys.argv[1:]:
    if a.startswith('--ab='):
        k, vs = a[5:].split('=')
        AB = (getattr(hip, 'TUNE_' + k.upper()), [int(v) for v in vs.split(',')])
TOT = {}
def run(tag, fn, flops, count, iters=10):
    variants = AB[1] if AB else [None]
    best = {v: 1e9 for v in variants}
    for v in variants:
        if AB: hip.tune(AB[0], v)
        for _ in range(2): fn()
    torch.cuda.synchronize()
    for rnd in range(4):
        for v in (variants if rnd % 2 == 0 else variants[::-1]):     # (the first variant after the idle sync runs on a cooler chip: alternate the order)
            if AB: hip.tune(AB[0], v)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters): fn()
            e1.record(); torch.cuda.synchronize()
            best[v] = min(best[v], e0.elapsed_time(e1) / iters)
    cols = '  |  '.join(f'{best[v]*1e3:8.1f} us {flops/best[v]/1e9:7.1f} TF' for v in variants)
    print(f'{tag:46s} {cols}   x{count:2d}/step = ' + ' / '.join(f'{best[v]*count:6.2f}' for v in variants) + ' ms')
    for v in variants: TOT[v] = TOT.get(v, 0.) + best[v] * count
    return best[variants[0]] * count
r = lambda *s: torch.randn(*s, device='cuda')
P, G = hip.to_hformat, hip.gemm_h
x, rs = r(M, D), torch.rand(128, device="cuda").repeat_interleave(197)
xp = P(x)
tot = 0
w, b, g = r(H3, D), r(H3), torch.rand(H3, device='cuda'); y = torch.empty(M, H3, device='cuda'); wp = P(w)
qb = torch.empty(1, device='cuda')
tot += run('fwd qkv  KC,KC bias+gate -> f32 (+bound)', lambda: G(xp, wp, 1, 1, M, H3, D, C_out=y, ldc=H3, bias=b, colscale=g, cbound_out=qb), 2.*M*H3*D, 12)
w2, b2 = r(D, D), r(D); y2 = torch.empty(M, D, device='cuda'); w2p = P(w2)
tot += run('fwd proj KC,KC bias+rowscale+resid -> f32', lambda: G(xp, w2p, 1, 1, M, D, D, C_out=y2, ldc=D, bias=b2, rowscale=rs, resid=x, ldr=D), 2.*M*D*D, 12)
w3, b3, g3 = r(HID, D), r(HID), torch.rand(HID, device='cuda'); hpre = hip.aux_t(M, HID, 'cuda'); hP = hip.HMat(M, HID, 'cuda'); w3p = P(w3)
tot += run("fwd fc1  KC,KC bias+gate+GELU' aux -> planes", lambda: G(xp, w3p, 1, 1, M, HID, D, Cp=hP, bias=b3, colscale=g3, act=hip.ACT_GELU_GRAD_T, aux=hpre, ldaux=HID), 2.*M*HID*D, 12)
w4 = r(D, HID); w4p = P(w4)
tot += run('fwd fc2  KC,KC bias+rowscale+resid -> f32', lambda: G(hP, w4p, 1, 1, M, D, HID, C_out=y2, ldc=D, bias=b2, rowscale=rs, resid=x, ldr=D), 2.*M*D*HID, 12)
dq = r(M, H3); dqp = P(dq)
tot += run('bwd dX qkv  KC,KR +resid (K=1152) -> f32', lambda: G(dqp, wp, 1, 0, M, D, H3, C_out=y2, ldc=D, resid=x, ldr=D), 2.*M*H3*D, 12)
tot += run('bwd dO proj KC,KR (K=384) -> f32 (+bound)', lambda: G(xp, w2p, 1, 0, M, D, D, C_out=y2, ldc=D, cbound_out=qb), 2.*M*D*D, 12)
dhP = hip.HMat(M, HID, 'cuda')
tot += run('bwd dH fc2  KC,KR x aux -> planes + colsums', lambda: G(xp, w4p, 1, 0, M, HID, D, Cp=dhP, act=hip.ACT_MULAUX_T, aux=hpre, ldaux=HID, want_colpart=True), 2.*M*HID*D, 12)
tot += run('bwd dX fc1  KC,KR +resid (K=1536) -> f32', lambda: G(dhP, w3p, 1, 0, M, D, HID, C_out=y2, ldc=D, resid=x, ldr=D), 2.*M*HID*D, 12)
dw = torch.empty(H3, D, device='cuda')
tot += run('bwd dW qkv  KR,KR', lambda: G(dqp, xp, 0, 0, H3, D, M, C_out=dw, ldc=D), 2.*M*H3*D, 12)
dw2 = torch.empty(D, D, device='cuda')
tot += run('bwd dW proj KR,KR', lambda: G(xp, xp, 0, 0, D, D, M, C_out=dw2, ldc=D), 2.*M*D*D, 12)
dw3 = torch.empty(HID, D, device='cuda')
tot += run('bwd dW fc1  KR,KR', lambda: G(dhP, xp, 0, 0, HID, D, M, C_out=dw3, ldc=D), 2.*M*HID*D, 12)
dw4 = torch.empty(D, HID, device='cuda')
tot += run('bwd dW fc2  KR,KR', lambda: G(xp, hP, 0, 0, D, HID, M, C_out=dw4, ldc=HID), 2.*M*D*HID, 12)
print('sum over 12 blocks: ' + ' / '.join(f'{TOT[v]:.2f} ms' + (f' ({v})' if v is not None else '') for v in TOT)); TOT.clear()
tot2 = 0
tot2 += run('convert x [M][384] -> planes (stat + split)', lambda: P(x), 0, 1)
hh = r(M, HID)
tot2 += run('convert h [M][1536] -> planes (stat + split)', lambda: P(hh), 0, 1)
