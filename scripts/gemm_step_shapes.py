"""Times every GEMM configuration of one DeiT-S bs=128 search step with its real epilogue (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ofb_amd import hip
M, D, H3, HID = 128 * 197, 384, 1152, 1536
def run(tag, fn, flops, count, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f'{tag:44s} {ms*1e3:8.1f} us {flops/ms/1e9:7.1f} TF   x{count:2d}/step = {ms*count:6.2f} ms')
    return ms * count
r = lambda *s: torch.randn(*s, device='cuda')
x, rs = r(M, D), torch.rand(128, device="cuda").repeat_interleave(197)
tot = 0
# forward
w, b, g = r(H3, D), r(H3), r(H3); y = torch.empty(M, H3, device='cuda')
tot += run('fwd qkv  NT bias+gate', lambda: hip.gemm(x, w, y, M, H3, D, D, D, H3, 1, 1, bias=b, colscale=g), 2.*M*H3*D, 12)
w2, b2 = r(D, D), r(D); y2 = torch.empty(M, D, device='cuda')
tot += run('fwd proj NT bias+rowscale+resid', lambda: hip.gemm(x, w2, y2, M, D, D, D, D, D, 1, 1, bias=b2, rowscale=rs, rs_div=1, resid=x, ldr=D), 2.*M*D*D, 12)
w3, b3, g3 = r(HID, D), r(HID), r(HID); h = torch.empty(M, HID, device='cuda'); hp = torch.empty(M, HID, device='cuda')
tot += run('fwd fc1  NT bias+gate+GELU+aux', lambda: hip.gemm(x, w3, h, M, HID, D, D, D, HID, 1, 1, bias=b3, colscale=g3, act=hip.ACT_GELU, aux=hp, ldaux=HID), 2.*M*HID*D, 12)
w4 = r(D, HID)
tot += run('fwd fc2  NT bias+rowscale+resid', lambda: hip.gemm(h, w4, y2, M, D, HID, HID, HID, D, 1, 1, bias=b2, rowscale=rs, rs_div=1, resid=x, ldr=D), 2.*M*D*HID, 12)
# backward input grads
dq = r(M, H3)
tot += run('bwd dX qkv  NN +resid (K=1152)', lambda: hip.gemm(dq, w, y2, M, D, H3, H3, D, D, 1, 0, resid=x, ldr=D), 2.*M*H3*D, 12)
tot += run('bwd dO proj NN rowscale (K=384)', lambda: hip.gemm(x, w2, y2, M, D, D, D, D, D, 1, 0, rowscale=rs, rs_div=1), 2.*M*D*D, 12)
tot += run('bwd dH fc2  NN rowscale+dgelu', lambda: hip.gemm(x, w4, h, M, HID, D, D, HID, HID, 1, 0, rowscale=rs, rs_div=1, act=hip.ACT_DGELU, aux=hp, ldaux=HID), 2.*M*HID*D, 12)
tot += run('bwd dX fc1  NN +resid (K=1536)', lambda: hip.gemm(h, w3, y2, M, D, HID, HID, D, D, 1, 0, resid=x, ldr=D), 2.*M*HID*D, 12)
# backward weight grads (+ fused bias grads)
dw, db = torch.empty(H3, D, device='cuda'), torch.empty(H3, device='cuda')
tot += run('bwd dW qkv  TN (+db)', lambda: hip.gemm(dq, x, dw, H3, D, M, H3, D, D, 0, 0, a_colsum=db), 2.*M*H3*D, 12)
dw2, dbd = torch.empty(D, D, device='cuda'), torch.empty(D, device='cuda')
tot += run('bwd dW proj TN kscale (+db)', lambda: hip.gemm(x, x, dw2, D, D, M, D, D, D, 0, 0, kscale=rs, ks_div=1, a_colsum=dbd), 2.*M*D*D, 12)
dw3, db3 = torch.empty(HID, D, device='cuda'), torch.empty(HID, device='cuda')
tot += run('bwd dW fc1  TN (+db)', lambda: hip.gemm(h, x, dw3, HID, D, M, HID, D, D, 0, 0, a_colsum=db3), 2.*M*HID*D, 12)
dw4 = torch.empty(D, HID, device='cuda')
tot += run('bwd dW fc2  TN kscale (+db)', lambda: hip.gemm(x, h, dw4, D, HID, M, D, HID, HID, 0, 0, kscale=rs, ks_div=1, a_colsum=dbd), 2.*M*D*HID, 12)
print(f'sum over 12 blocks: {tot:.2f} ms   (ideal at 157.3 TF: {12*6*2.*M*D*(H3+D+2*HID)/157.3e9/1e0*1e-3*3/6:.2f} ms)')
