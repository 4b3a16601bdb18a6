import ctypes as C, sys, torch
lib = C.CDLL(sys.argv[1])
B, N, H, dh = 128, 197, 6, 64
qkv = torch.randn(B * N, 3 * H * dh, device='cuda'); o = torch.empty(B * N, H * dh, device='cuda')
lse = torch.empty(2 * B * H, N, device="cuda"); do = torch.randn_like(o); dqkv = torch.empty_like(qkv)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: C.c_void_p(t.data_ptr())
def fwd(): lib.ofb_attention_fwd(P(qkv), P(o), P(lse), B, N, H, dh, C.c_float(0.125), st)
def bwd(): lib.ofb_attention_bwd(P(qkv), P(o), P(lse), P(do), P(dqkv), B, N, H, dh, C.c_float(0.125), st)
def run(f):
    for _ in range(200): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 50 * 1e3
print(f'{sys.argv[2]:48s} fwd {run(fwd):7.1f} us  bwd {run(bwd):7.1f} us')
