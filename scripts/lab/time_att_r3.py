import sys, torch
sys.path.insert(0, '/root/repo')
from ofb_amd import hip
B, N, H, dh = 128, 197, 6, 64
torch.manual_seed(0)
qkv = torch.randn(B * N, 3 * H * dh, device='cuda')
o = torch.empty(B * N, H * dh, device='cuda'); lse = torch.empty(2 * B * H, N, device='cuda'); do = torch.randn_like(o)
oP = hip.PMat.for_rows_written_by_kernel(B * N, H * dh, 'cuda')
dP = hip.PMat.for_rows_written_by_kernel(B * N, 3 * H * dh, 'cuda'); cp = torch.empty(B, 3 * H * dh, device='cuda')
dq = torch.empty_like(qkv)
def t(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
print('fwd_p  us', t(lambda: hip.attention_fwd_p(qkv, o, oP, lse, B, N, H, dh, 0.125)))
print('bwd_p  us', t(lambda: hip.attention_bwd_p(qkv, o, lse, do, dP, cp, B, N, H, dh, 0.125)))
print('bwd    us', t(lambda: hip.attention_bwd(qkv, o, lse, do, dq, B, N, H, dh, 0.125)))
