"""HIP f32-MFMA GEMM (through the C ABI) vs an fp64 CPU product on the same seeded inputs."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _mk(shape, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g, dtype=torch.float32)


def _check(got, exp, tol=2e-6, what=''):
    err = (got.double().cpu() - exp).abs().max().item()
    scale = exp.abs().max().item() + 1e-30
    print(f'{what}: max abs err {err:.3e} (scale {scale:.3e})')
    assert err <= tol * scale * 40, what


# (M, N, K): exact tiles, ragged edges, K not multiple of 16, tiny, pruned-style dims (multiples of 12 / 6), scalar path
SHAPES = [(256, 256, 64), (394, 384, 384), (197 * 3, 1152, 384), (130, 70, 36), (128, 1000, 384), (77, 13, 5),
          (591, 288, 204), (200, 102, 102)]


@pytest.mark.parametrize('M,N,K', SHAPES)
def test_gemm_nt_epilogues(M, N, K):
    from ofb_amd import hip
    x, w, b = _mk((M, K), 1), _mk((N, K), 2), _mk((N,), 3)
    cs, res = _mk((N,), 4), _mk((M, N), 5)
    rs = _mk(((M + 196) // 197,), 6)
    xd, wd, bd, csd, resd, rsd = (t.cuda() for t in (x, w, b, cs, res, rs))
    out = torch.empty(M, N, device='cuda')
    # plain x @ W^T + b
    hip.gemm(xd, wd, out, M, N, K, K, K, N, 1, 1, bias=bd)
    ref = x.double() @ w.double().t() + b.double()
    _check(out, ref, what=f'nt bias {M}x{N}x{K}')
    # gate column scale + GELU (+ pre-activation side output)
    aux = torch.empty(M, N, device='cuda')
    hip.gemm(xd, wd, out, M, N, K, K, K, N, 1, 1, bias=bd, colscale=csd, aux=aux, ldaux=N, act=hip.ACT_GELU)
    pre = ref * cs.double()
    _check(aux, pre, what='gelu pre-activation')
    _check(out, torch.nn.functional.gelu(pre), what='gelu out')
    # residual + per-sample row scale
    hip.gemm(xd, wd, out, M, N, K, K, K, N, 1, 1, bias=bd, rowscale=rsd, rs_div=197, resid=resd, ldr=N)
    rows = torch.arange(M) // 197
    _check(out, ref * rs.double()[rows].unsqueeze(1) + res.double(), what='residual+rowscale')
    # DGELU multiply
    hip.gemm(xd, wd, out, M, N, K, K, K, N, 1, 1, aux=aux, ldaux=N, act=hip.ACT_DGELU)
    p = pre.clone().requires_grad_(True)
    torch.nn.functional.gelu(p).sum().backward()
    _check(out, (x.double() @ w.double().t()) * p.grad, what='dgelu')


@pytest.mark.parametrize('M,N,K', SHAPES)
def test_gemm_nn_and_tn(M, N, K):
    """dX = dY @ W (a_kc=1,b_kc=0) and dW = dY^T @ X (a_kc=0,b_kc=0, split-K over tokens)."""
    from ofb_amd import hip
    dy, w, x = _mk((M, N), 7), _mk((N, K), 8), _mk((M, K), 9)
    dyd, wd, xd = dy.cuda(), w.cuda(), x.cuda()
    dx = torch.empty(M, K, device='cuda')
    hip.gemm(dyd, wd, dx, M, K, N, N, K, K, 1, 0)
    _check(dx, dy.double() @ w.double(), what=f'nn {M}x{K}x{N}')
    ks = _mk(((M + 196) // 197,), 10)
    rows = torch.arange(M) // 197
    exp = (dy.double() * ks.double()[rows].unsqueeze(1)).t() @ x.double()
    dw, db = torch.empty(N, K, device='cuda'), torch.full((N,), float('nan'), device='cuda')
    hip.gemm(dyd, xd, dw, N, K, M, N, K, K, 0, 0, kscale=ks.cuda(), ks_div=197, a_colsum=db)
    _check(dw, exp, what=f'tn {N}x{K}x{M}')
    _check(db, (dy.double() * ks.double()[rows].unsqueeze(1)).sum(0), what='fused bias grad', tol=1e-5)


@pytest.mark.parametrize('tokens,N,K', [(1000, 512, 256), (197 * 8, 768, 512), (5000, 256, 256)])
def test_gemm_weight_gradient_on_the_256_tile(tokens, N, K):
    """dW = dY^T @ X whose output is a whole number of 256x256 tiles takes the 8-wave T256 configuration (csrc/gemm.hip: use_t256):
    with the per-sample K scale, the fused bias gradient and a token count that is not a multiple of the K-step."""
    from ofb_amd import hip
    dy, x = _mk((tokens, N), 31), _mk((tokens, K), 32)
    ks = _mk(((tokens + 196) // 197,), 33)
    rows = torch.arange(tokens) // 197
    scaled = dy.double() * ks.double()[rows].unsqueeze(1)
    dw, db = torch.empty(N, K, device='cuda'), torch.full((N,), float('nan'), device='cuda')
    hip.gemm(dy.cuda(), x.cuda(), dw, N, K, tokens, N, K, K, 0, 0, kscale=ks.cuda(), ks_div=197, a_colsum=db)
    _check(dw, scaled.t() @ x.double(), what=f'tn/T256 {N}x{K}x{tokens}')
    _check(db, scaled.sum(0), what='fused bias grad', tol=1e-5)
    dw2 = torch.empty(N, K, device='cuda')
    hip.gemm(dy.cuda(), x.cuda(), dw2, N, K, tokens, N, K, K, 0, 0, kscale=ks.cuda(), ks_div=197)
    assert torch.equal(dw, dw2), 'deterministic partial sums'


def test_gemm_stream_k_tail_shapes():
    """tile counts around the workgroup count (full rounds + streamed tail, tail only, exact rounds)."""
    from ofb_amd import hip
    for (M, N, K) in [(128 * 40, 128 * 13, 80), (128 * 64, 128 * 8, 48), (128 * 33, 128 * 16, 200), (25216, 384, 64)]:
        x, w, b, res = _mk((M, K), 21), _mk((N, K), 22), _mk((N,), 23), _mk((M, N), 24)
        out = torch.empty(M, N, device='cuda')
        hip.gemm(x.cuda(), w.cuda(), out, M, N, K, K, K, N, 1, 1, bias=b.cuda(), resid=res.cuda(), ldr=N)
        rows = torch.arange(0, M, 37)
        ref = x[rows].double() @ w.double().t() + b.double() + res[rows].double()
        _check(out[rows.cuda()], ref, what=f'stream-k {M}x{N}x{K}')
        out2 = torch.empty(M, N, device='cuda')
        hip.gemm(x.cuda(), w.cuda(), out2, M, N, K, K, K, N, 1, 1, bias=b.cuda(), resid=res.cuda(), ldr=N)
        assert torch.equal(out, out2), 'stream-K partial sums must be summed in a fixed order (deterministic)'


def test_gemm_deit_small_shapes():
    """The real bs=128 DeiT-S layer shapes (25216 tokens): compare a strided sample of rows with fp64."""
    from ofb_amd import hip
    M, D = 128 * 197, 384
    for N in (1152, 384, 1536):
        x, w, b = _mk((M, D), 11), _mk((N, D), 12) * 0.05, _mk((N,), 13)
        out = torch.empty(M, N, device='cuda')
        hip.gemm(x.cuda(), w.cuda(), out, M, N, D, D, D, N, 1, 1, bias=b.cuda())
        rows = torch.arange(0, M, 97)
        ref = x[rows].double() @ w.double().t() + b.double()
        _check(out[rows.cuda()], ref, what=f'deit-s {N}')


def test_gemm_rejects_bad_arguments():
    from ofb_amd import hip
    a = torch.zeros(4, 4, device='cuda')
    with pytest.raises(hip.OfbError):
        hip.gemm(a, a, a, 4, 4, 4, 2, 4, 4, 1, 1)        # lda < K
    with pytest.raises(hip.OfbError):
        hip.gemm(a.cpu(), a, a, 4, 4, 4, 4, 4, 4, 1, 1)  # CPU tensor: no fallback
