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


@pytest.mark.parametrize('rows,D', [(394, 384), (197 * 8, 192), (33, 102), (200, 768), (7, 1024), (5000, 384), (20, 77), (64, 96)])
def test_layernorm_hformat_outputs(rows, D):
    """the H-format variants (planes written by the LayerNorm kernels themselves): same f32 rows as the plain kernels (up to the
    compiler's fma contraction), planes == the kernel's own f32 rows (x the DropPath row scale in backward) to 2^-23 of the header's
    bound, the bound really bounds, third partial section = the column sums of the scaled rows"""
    from ofb_amd import hip
    x, g, b = (_mk((rows, D), 1) * 2 + 0.3).cuda(), (_mk((D,), 2) * 0.2 + 1).cuda(), (_mk((D,), 3) * 0.1).cuda()
    dy, dres = _mk((rows, D), 4).cuda(), _mk((rows, D), 5).cuda()
    y0, y1 = torch.empty(rows, D, device='cuda'), torch.empty(rows, D, device='cuda')
    m0, r0, m1, r1 = (torch.empty(rows, device='cuda') for _ in range(4))
    hip.layernorm_fwd(x, g, b, y0, m0, r0, rows, D, 1e-6)
    yP = hip.HMat.for_rows_written_by_kernel(rows, D, 'cuda')
    hip.layernorm_fwd_h(x, g, b, y1, yP, m1, r1, rows, D, 1e-6)
    _close(y1, y0.cpu(), 1e-6, 'ln fwd rows of the H variant')
    _close(m1, m0.cpu(), 1e-6, 'mean')
    _close(r1, r0.cpu(), 1e-6, 'rstd')
    e, bound, rn2sq, _ = yP.header()
    assert y1.abs().max().item() <= bound <= (D ** 0.5 * g.abs().max().item() + b.abs().max().item()) * 1.001, 'analytic bound of |LN(x)|'
    assert y1.double().pow(2).sum(1).max().item() <= rn2sq, 'row-norm bound'
    assert 2.0 ** 14 <= bound * 2.0 ** e < 2.0 ** 15
    assert (yP.to_f32() - y1).abs().max().item() <= 2.0 ** -23 * bound
    y0 = y1
    # as a GEMM operand (reduction along the columns and along the rows): padding rows / columns must be zero
    w = _mk((48, D), 8).cuda()
    out = torch.empty(rows, 48, device='cuda')
    hip.gemm_h(yP, hip.to_hformat(w), 1, 1, rows, 48, D, C_out=out, ldc=48)
    _close(out, y0.double().cpu() @ w.double().cpu().t(), 2e-6, 'LN planes as GEMM operand (K along columns)')
    out2 = torch.empty(D, 48, device='cuda')
    z = _mk((rows, 48), 9).cuda()
    hip.gemm_h(yP, hip.to_hformat(z), 0, 0, D, 48, rows, C_out=out2, ldc=48)
    _close(out2, y0.double().cpu().t() @ z.double().cpu(), 2e-6, 'LN planes as GEMM operand (K along rows)')

    nb = hip.layernorm_bwd_blocks(rows)
    for rs_div, use_res in ((1, True), (rows if rows % 197 else 197, False), (0, True)):
        rowscale = None if rs_div == 0 else (torch.rand((rows + rs_div - 1) // rs_div, device='cuda') + 0.5)
        dr = dres if use_res else None
        part2, dx0 = torch.empty(nb, 2, D, device='cuda'), torch.empty(rows, D, device='cuda')
        hip.layernorm_bwd(dy, x, g, m0, r0, dr, dx0, part2, rows, D)
        part3, dx1 = torch.empty(nb, 3, D, device='cuda'), torch.empty(rows, D, device='cuda')
        dxP = hip.HMat.for_rows_written_by_kernel(rows, D, 'cuda')
        hip.layernorm_bwd_h(dy, x, g, m0, r0, dr, dx1, part3, dxP, rowscale, max(rs_div, 1), rows, D)
        _close(dx1, dx0.cpu(), 2e-6, 'ln dx of the H variant')
        sc = torch.ones(rows, device='cuda') if rowscale is None else rowscale[torch.arange(rows, device='cuda') // rs_div]
        scaled = dx1 * sc.unsqueeze(1)
        e, bound, rn2sq, _ = dxP.header()
        amax = scaled.abs().max().item()
        assert amax <= bound <= 64 * amax, f'the projection bound of |dx| must hold and stay useful: {bound:.3e} vs max {amax:.3e}'
        assert scaled.double().pow(2).sum(1).max().item() <= rn2sq
        assert (dxP.to_f32() - scaled).abs().max().item() <= 2.0 ** -23 * bound
        s2, s3 = torch.empty(2 * D, device='cuda'), torch.empty(3 * D, device='cuda')
        hip.colsum(part2, 2 * D, nb, 2 * D, s2)
        hip.colsum(part3, 3 * D, nb, 3 * D, s3)
        _close(s3[:2 * D], s2.cpu(), 2e-6, 'dgamma | dbeta of the H variant')
        _close(s3[2 * D:], scaled.double().sum(0).cpu(), 1e-5, 'column sums of the scaled dx rows')


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
    full = (dWraw.double() * W.double()).sum(1) + dbraw.double() * b.double()
    _close(dg, full, 1e-5, 'fold dg')
    if N % 3 == 0:                                         # q | k | v share one gate: g tiled three times, dg summed over the three groups
        g3 = g[:N // 3].repeat(3)
        dW3, db3, dg3 = torch.empty(N, K, device='cuda'), torch.empty(N, device='cuda'), torch.empty(N // 3, device='cuda')
        hip.gate_fold_bwd(dWraw.cuda(), W.cuda(), g3.cuda(), dbraw.cuda(), b.cuda(), dW3, db3, dg3, N, K, fold=3)
        _close(dW3, dWraw.double() * g3.double().unsqueeze(1), 1e-6, 'fold dW (tiled gate)')
        _close(db3, dbraw.double() * g3.double(), 1e-6, 'fold db (tiled gate)')
        _close(dg3, full.view(3, N // 3).sum(0), 1e-5, 'fold dg summed over q | k | v')


@pytest.mark.parametrize('M,N,K,with_resid', [(1970, 384, 200, True), (788, 264, 72, False), (2100, 768, 96, True), (300, 100, 40, True),
                                                (2100, 264, 96, True), (1300, 480, 72, False)])   # (widths the 96-column tile would take without rn)
def test_gemm_row_norm_handover_to_layernorm_backward(M, N, K, with_resid):
    """the input-gradient GEMM's epilogue leaves, per output tile, max_rows rstd |gamma (.) dy, the tile's columns|_2 (ofb_gemm_h
    rn_out); ofb_layernorm_bwd_h_rn bounds its result with sqrt(column tiles) x their maximum instead of a pass over dy: the values
    against torch, the resulting bound >= the exact one and within sqrt(column tiles) x 1.01 of it, the planes and dx as the plain path's"""
    from ofb_amd import hip
    g = torch.Generator().manual_seed(3)
    dyq = torch.randn(M, K, generator=g).cuda()
    w = (torch.randn(K, N, generator=g) * 0.1).cuda()
    resid = torch.randn(M, N, generator=g).cuda() if with_resid else None
    gamma = (1.0 + 0.3 * torch.randn(N, generator=g)).cuda()
    x = torch.randn(M, N, generator=g).cuda()
    mean, var = x.mean(1), x.var(1, unbiased=False)
    rstd = (var + 1e-6).rsqrt()
    dyP, wP = hip.to_hformat(dyq), hip.to_hformat(w)
    dy = torch.empty(M, N, device='cuda')
    rn = hip.gemm_h(dyP, wP, 1, 0, M, N, K, C_out=dy, ldc=N, resid=resid, ldr=N, rn=(gamma, rstd))
    if rn is None:                                           # a shape the library runs on its 256 x 96 tile: the hand-over steps aside
        assert M >= 1024 and (N + 95) // 96 * 96 < (N + 191) // 192 * 192
        return
    vals, fac = rn
    ref = dyq.double() @ w.double() + (resid.double() if with_resid else 0)
    assert float((dy.double() - ref).abs().max()) < 3e-6 * float(ref.abs().max())
    # per-tile reference: the tile geometry follows the value count (128 x 192 or 256 x 96 tiles)
    nt = int(round(fac * fac))
    bn = 192 if nt == (N + 191) // 192 else 96
    bm = 128 if bn == 192 else 256
    assert vals.numel() == ((M + bm - 1) // bm) * nt
    t = (gamma.double() * ref) ** 2
    exp = torch.zeros(vals.numel(), dtype=torch.float64)
    for mi in range((M + bm - 1) // bm):
        for ni in range(nt):
            part = t[mi * bm:(mi + 1) * bm, ni * bn:(ni + 1) * bn].sum(1).sqrt() * rstd.double()[mi * bm:(mi + 1) * bm]
            exp[mi * nt + ni] = float(part.max())
    assert float((vals.double().cpu() - exp).abs().max()) < 1e-5 * float(exp.max()), (vals[:4], exp[:4])
    # LayerNorm backward with the hand-over against the plain plane-writing backward
    rowscale = (torch.rand(10, generator=g) + 0.5).cuda()
    rs_div = (M + 9) // 10
    nb = hip.layernorm_bwd_blocks(M)
    outs = []
    for use_rn in (False, True):
        dx, part, dxP = torch.empty(M, N, device='cuda'), torch.empty(nb, 3 * N, device='cuda'), hip.HMat.for_rows_written_by_kernel(M, N, 'cuda')
        if use_rn:
            hip.layernorm_bwd_h_rn(dy, x, gamma, mean, rstd, dx, part, dxP, rowscale, rs_div, M, N, vals, fac)
        else:
            hip.layernorm_bwd_h(dy, x, gamma, mean, rstd, None, dx, part, dxP, rowscale, rs_div, M, N)
        outs.append((dx, part, dxP))
    (dx0, part0, p0), (dx1, part1, p1) = outs
    assert torch.equal(dx0, dx1) and torch.allclose(part0, part1, rtol=1e-6, atol=1e-6)
    b0, b1 = p0.header()[1], p1.header()[1]
    scaled = dx0 * rowscale[torch.arange(M, device='cuda') // rs_div].unsqueeze(1)
    true_max = float(scaled.abs().max())
    assert b1 >= true_max and b1 >= 0.99 * b0 and b1 <= fac * 1.01 * float(rowscale.abs().max()) / float(rowscale.abs().min()) * b0 + 1e-30, (b0, b1, true_max)
    err = (p1.to_f32() - scaled).abs()
    assert bool((err <= 2.0 ** -23 * scaled.abs() + 2.0 ** -39 * b1).all())
