"""single kernels under hipGraph capture: python graph_bisect2.py <attn_bwd|attn_fwd|gemm_tail|ln_bwd|autograd>"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ofb_amd import hip, ops
what = sys.argv[1]
dev = 'cuda'
B, N, H, dh = 8, 197, 3, 64
qkv = torch.randn(B * N, 3 * H * dh, device=dev); o = torch.empty(B * N, H * dh, device=dev); lse = torch.empty(2 * B * H, N, device=dev)
do = torch.randn(B * N, H * dh, device=dev); dqkv = torch.empty_like(qkv)
x = torch.randn(B * N, 192, device=dev); w = torch.randn(576, 192, device=dev); dy = torch.randn(B * N, 576, device=dev)
def attn_fwd(): hip.attention_fwd(qkv, o, lse, B, N, H, dh, 0.125)
def attn_bwd(): hip.attention_bwd(qkv, o, lse, do, dqkv, B, N, H, dh, 0.125)
def attn_bwd_p():
    dP = hip.PMat.for_rows_written_by_kernel(B * N, 3 * H * dh, dev); cp = torch.empty(B, 3 * H * dh, device=dev)
    hip.attention_bwd_p(qkv, o, lse, do, dP, cp, B, N, H, dh, 0.125)
def gemm_tail():
    dw = torch.empty(576, 192, device=dev)
    hip.gemm_p(hip.to_pformat(dy), hip.to_pformat(x), 0, 0, 576, 192, B * N, C_out=dw, ldc=192)
def ln_bwd():
    xx = x.clone().requires_grad_(True); g = torch.ones(192, device=dev, requires_grad=True); b = torch.zeros(192, device=dev, requires_grad=True)
    y = ops.layer_norm(xx, g, b, 1e-6); y.sum().backward()
class UpFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a):
        return a * 2
    @staticmethod
    def backward(ctx, g):
        tab = (hip.AdamwTensor * 4)()
        d, h = hip.upload_structs(tab, g.device)
        keep.append((d, h))
        return g * 2
class JoinFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a):
        return a * 2
    @staticmethod
    def backward(ctx, g):
        hip.join_side()
        return g * 2
keep = []
def up_bwd():
    xx = x.clone().requires_grad_(True); UpFn.apply(xx).sum().backward()
def join_bwd():
    xx = x.clone().requires_grad_(True); JoinFn.apply(xx).sum().backward()
def autograd():
    xx = x.clone().requires_grad_(True); (xx * 2).sum().backward()
fn = dict(up_bwd=up_bwd, join_bwd=join_bwd, attn_fwd=attn_fwd, attn_bwd=attn_bwd, attn_bwd_p=attn_bwd_p, gemm_tail=gemm_tail, ln_bwd=ln_bwd, autograd=autograd)[what]
attn_fwd()
for _ in range(2): fn()
torch.cuda.synchronize()
arena = hip.begin_capture_arena()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    fn()
print(what, 'captured'); g.replay(); torch.cuda.synchronize(); print(what, 'replayed OK')
