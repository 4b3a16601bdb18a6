"""LayerNorm / column-sum / gate-folding kernels vs fp64 CPU formulas (oracle definitions)."""
import pytest
import torch

from oracle import ofb_oracle as O

pytestmark = pytest.mark.gpu


def _mk(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g, dtype=torch.float32) * scale


def _close(got, exp, tol, what):
    err = (got.double().cpu() - exp.double()).abs().max().item()
    scale = exp.abs().max().item() + 1e-30
    print(f'{what}: max abs err {err:.3e} (scale {scale:.3e})')
    assert err <= tol * scale, what


@pytest.mark.parametrize('rows,D', [(394, 384), (197 * 8, 192), (50, 64), (33, 102), (200, 768), (7, 1024), (5000, 384)])
def test_layernorm_fwd_bwd(rows, D):
    from ofb_amd import hip
    x, g, b = _mk((rows, D), 1) * 2 + 0.3, _mk((D,), 2) * 0.2 + 1, _mk((D,), 3) * 0.1
    dy, dres = _mk((rows, D), 4), _mk((rows, D), 5)
    xd = x.double().requires_grad_(True)
    gd, bd = g.double().requires_grad_(True), b.double().requires_grad_(True)
    y_ref = O.layer_norm(xd, gd, bd, 1e-6)
    y_ref.backward(dy.double())
    y = torch.empty(rows, D, device='cuda')
    mean, rstd = torch.empty(rows, device='cuda'), torch.empty(rows, device='cuda')
    hip.layernorm_fwd(x.cuda(), g.cuda(), b.cuda(), y, mean, rstd, rows, D, 1e-6)
    _close(y, y_ref.detach(), 2e-6, f'ln fwd {rows}x{D}')
    nb = hip.layernorm_bwd_blocks(rows)
    part = torch.empty(nb, 2, D, device='cuda')
    dx = torch.empty(rows, D, device='cuda')
    hip.layernorm_bwd(dy.cuda(), x.cuda(), g.cuda(), mean, rstd, dres.cuda(), dx, part, rows, D)
    _close(dx, xd.grad + dres.double(), 5e-6, 'ln dx (+residual grad)')
    dgb = torch.empty(2 * D, device='cuda')
    hip.colsum(part, 2 * D, nb, 2 * D, dgb)
    _close(dgb[:D], gd.grad, 2e-5, 'ln dgamma')
    _close(dgb[D:], bd.grad, 2e-5, 'ln dbeta')
    hip.layernorm_bwd(dy.cuda(), x.cuda(), g.cuda(), mean, rstd, None, dx, part, rows, D)
    _close(dx, xd.grad, 5e-6, 'ln dx')


@pytest.mark.parametrize('M,N', [(25216, 1152), (1576, 384), (130, 70), (3, 5), (128, 1000)])
def test_colsum(M, N):
    from ofb_amd import hip
    x, rs = _mk((M, N), 6), _mk(((M + 196) // 197,), 7)
    out = torch.empty(N, device='cuda')
    hip.colsum(x.cuda(), N, M, N, out)
    _close(out, x.double().sum(0), 1e-5, f'colsum {M}x{N}')
    hip.colsum(x.cuda(), N, M, N, out, rowscale=rs.cuda(), rs_div=197)
    rows = torch.arange(M) // 197
    _close(out, (x.double() * rs.double()[rows].unsqueeze(1)).sum(0), 1e-5, 'colsum rowscale')


@pytest.mark.parametrize('N,K', [(1152, 384), (96, 102), (1536, 384)])
def test_gate_fold(N, K):
    from ofb_amd import hip
    W, g, b = _mk((N, K), 8), _mk((N,), 9), _mk((N,), 10)
    dWraw, dbraw = _mk((N, K), 11), _mk((N,), 12)
    out = torch.empty(N, K, device='cuda')
    hip.scale_rows(W.cuda(), g.cuda(), out, N, K)
    _close(out, W.double() * g.double().unsqueeze(1), 1e-6, 'scale_rows')
    dW, db, dg = torch.empty(N, K, device='cuda'), torch.empty(N, device='cuda'), torch.empty(N, device='cuda')
    hip.gate_fold_bwd(dWraw.cuda(), W.cuda(), g.cuda(), dbraw.cuda(), b.cuda(), dW, db, dg, N, K)
    _close(dW, dWraw.double() * g.double().unsqueeze(1), 1e-6, 'fold dW')
    _close(db, dbraw.double() * g.double(), 1e-6, 'fold db')
    _close(dg, (dWraw.double() * W.double()).sum(1) + dbraw.double() * b.double(), 1e-5, 'fold dg')
