"""Round-4 lab: the H-format engine (two f16 planes, three terms) against fp64 and against the P-format engine (three bf16 planes,
six terms): accuracy on the three operand-mode pairs, then the twelve GEMM shapes of a DeiT-S bs-128 search step on both."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ofb_amd import hip

torch.manual_seed(0)
def rel(got, exp):
    return ((got.double().cpu() - exp).abs().max() / exp.abs().max()).item()
def rms(got, exp):
    return ((got.double().cpu() - exp).pow(2).mean().sqrt() / exp.pow(2).mean().sqrt()).item()

if '--time-only' not in sys.argv:
    x = torch.randn(1000, 264) * torch.pow(torch.tensor(2.0), torch.randint(-6, 6, (1000, 264)).float())
    hm = hip.to_hformat(x.cuda())
    back = hm.to_f32().cpu()
    print('round trip max rel err', ((back - x).abs() / x.abs().clamp_min(1e-30)).max().item(), 'header', hm.header(), 'amax', x.abs().max().item(),
          'rn2sq', x.pow(2).sum(1).max().item())
    for (M, N, K) in [(256, 256, 64), (394, 384, 384), (591, 1152, 384), (130, 70, 36), (77, 13, 5), (591, 264, 200), (2600, 520, 48),
                      (1100, 264, 200), (300, 200, 16), (300, 200, 48), (300, 200, 80), (25216, 384, 1536)]:
        a, b = torch.randn(M, K), torch.randn(N, K)
        exact = a.double() @ b.double().t()
        ad, bd = a.cuda(), b.cuda()
        hk_a, hk_b = hip.to_hformat(ad), hip.to_hformat(bd)
        hr_a, hr_b = hip.to_hformat(ad.t().contiguous()), hip.to_hformat(bd.t().contiguous())
        pk_a, pk_b = hip.to_pformat(ad), hip.to_pformat(bd)
        for name, (A, B, akc, bkc) in {'kc,kc': (hk_a, hk_b, 1, 1), 'kc,kr': (hk_a, hr_b, 1, 0), 'kr,kr': (hr_a, hr_b, 0, 0)}.items():
            out = torch.full((M, N), float('nan'), device='cuda')
            outp = hip.HMat(M, N, 'cuda')
            hip.gemm_h(A, B, akc, bkc, M, N, K, C_out=out, ldc=N, Cp=outp)
            o2 = outp.to_f32()
            print(f'{M}x{N}x{K} {name}: max rel {rel(out, exact):.2e} rms {rms(out, exact):.2e} | planes out vs f32 out {rel(o2, out.double().cpu()):.2e} hdr {outp.header()[:2]} amax {out.abs().max().item():.3g}')
        outp6 = torch.empty(M, N, device='cuda')
        hip.gemm_p(pk_a, pk_b, 1, 1, M, N, K, C_out=outp6, ldc=N)
        f32 = (ad @ bd.t())
        print(f'   six-term bf16: max rel {rel(outp6, exact):.2e} rms {rms(outp6, exact):.2e};  torch f32 matmul: rms {rms(f32, exact):.2e}')

# ---- timing of the step shapes, both engines interleaved ----
M, D, H3, HID = 128 * 197, 384, 1152, 1536
def run(fn, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
r = lambda *s: torch.randn(*s, device='cuda')
tot = {'p': 0.0, 'h': 0.0}
def both(tag, mk, flops, count):
    res = {}
    for rnd in range(2):
        for eng in ('p', 'h'):
            ms = run(mk(eng))
            res[eng] = min(res.get(eng, 1e9), ms)
    for eng in ('p', 'h'):
        tot[eng] += res[eng] * count
    print(f'{tag:44s} P {res["p"]*1e3:7.1f} us {flops/res["p"]/1e9:6.1f} TF | H {res["h"]*1e3:7.1f} us {flops/res["h"]/1e9:6.1f} TF  x{res["p"]/res["h"]:.2f}')
E = {'p': (hip.to_pformat, hip.gemm_p, hip.PMat), 'h': (hip.to_hformat, hip.gemm_h, hip.HMat)}
x, rs = r(M, D), torch.rand(128, device="cuda").repeat_interleave(197)
w, b, g = r(H3, D), r(H3), r(H3); y = torch.empty(M, H3, device='cuda')
w2, b2 = r(D, D), r(D); y2 = torch.empty(M, D, device='cuda')
w3, b3, g3 = r(HID, D), r(HID), torch.rand(HID, device='cuda'); hpre = torch.empty(M, HID, device='cuda')
w4 = r(D, HID); dq = r(M, H3)
ops = {}
for eng, (P, G, MatT) in E.items():
    o = ops[eng] = {}
    o['xp'], o['wp'], o['w2p'], o['w3p'], o['w4p'], o['dqp'] = P(x), P(w), P(w2), P(w3), P(w4), P(dq)
    o['hP'], o['dhP'] = MatT(M, HID, 'cuda'), MatT(M, HID, 'cuda')
    G(o['xp'], o['w3p'], 1, 1, M, HID, D, Cp=o['hP'], bias=b3, colscale=g3, act=hip.ACT_GELU_GRAD, aux=hpre, ldaux=HID)
    G(o['xp'], o['w4p'], 1, 0, M, HID, D, Cp=o['dhP'], act=hip.ACT_MULAUX, aux=hpre, ldaux=HID)
dw, dw2, dw3, dw4 = torch.empty(H3, D, device='cuda'), torch.empty(D, D, device='cuda'), torch.empty(HID, D, device='cuda'), torch.empty(D, HID, device='cuda')
def G(eng): return E[eng][1]
both('fwd qkv  KC,KC bias+gate -> f32', lambda e: (lambda: G(e)(ops[e]['xp'], ops[e]['wp'], 1, 1, M, H3, D, C_out=y, ldc=H3, bias=b, colscale=g)), 2.*M*H3*D, 12)
both('fwd proj KC,KC bias+rowscale+resid -> f32', lambda e: (lambda: G(e)(ops[e]['xp'], ops[e]['w2p'], 1, 1, M, D, D, C_out=y2, ldc=D, bias=b2, rowscale=rs, resid=x, ldr=D)), 2.*M*D*D, 12)
both('fwd fc1  KC,KC bias+gate+GELU\' aux -> planes', lambda e: (lambda: G(e)(ops[e]['xp'], ops[e]['w3p'], 1, 1, M, HID, D, Cp=ops[e]['hP'], bias=b3, colscale=g3, act=hip.ACT_GELU_GRAD, aux=hpre, ldaux=HID)), 2.*M*HID*D, 12)
both('fwd fc2  KC,KC bias+rowscale+resid -> f32', lambda e: (lambda: G(e)(ops[e]['hP'], ops[e]['w4p'], 1, 1, M, D, HID, C_out=y2, ldc=D, bias=b2, rowscale=rs, resid=x, ldr=D)), 2.*M*D*HID, 12)
both('bwd dX qkv  KC,KR +resid (K=1152) -> f32', lambda e: (lambda: G(e)(ops[e]['dqp'], ops[e]['wp'], 1, 0, M, D, H3, C_out=y2, ldc=D, resid=x, ldr=D)), 2.*M*H3*D, 12)
both('bwd dO proj KC,KR (K=384) -> f32', lambda e: (lambda: G(e)(ops[e]['xp'], ops[e]['w2p'], 1, 0, M, D, D, C_out=y2, ldc=D)), 2.*M*D*D, 12)
both('bwd dH fc2  KC,KR x aux -> planes', lambda e: (lambda: G(e)(ops[e]['xp'], ops[e]['w4p'], 1, 0, M, HID, D, Cp=ops[e]['dhP'], act=hip.ACT_MULAUX, aux=hpre, ldaux=HID)), 2.*M*HID*D, 12)
both('bwd dX fc1  KC,KR +resid (K=1536) -> f32', lambda e: (lambda: G(e)(ops[e]['dhP'], ops[e]['w3p'], 1, 0, M, D, HID, C_out=y2, ldc=D, resid=x, ldr=D)), 2.*M*HID*D, 12)
both('bwd dW qkv  KR,KR', lambda e: (lambda: G(e)(ops[e]['dqp'], ops[e]['xp'], 0, 0, H3, D, M, C_out=dw, ldc=D)), 2.*M*H3*D, 12)
both('bwd dW proj KR,KR', lambda e: (lambda: G(e)(ops[e]['xp'], ops[e]['xp'], 0, 0, D, D, M, C_out=dw2, ldc=D)), 2.*M*D*D, 12)
both('bwd dW fc1  KR,KR', lambda e: (lambda: G(e)(ops[e]['dhP'], ops[e]['xp'], 0, 0, HID, D, M, C_out=dw3, ldc=D)), 2.*M*HID*D, 12)
both('bwd dW fc2  KR,KR', lambda e: (lambda: G(e)(ops[e]['xp'], ops[e]['hP'], 0, 0, D, HID, M, C_out=dw4, ldc=HID)), 2.*M*D*HID, 12)
print(f'sum over 12 blocks: P {tot["p"]:.2f} ms | H {tot["h"]:.2f} ms')
