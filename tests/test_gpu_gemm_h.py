"""H-format engine (csrc/gemm_h.hip, csrc/hformat.h) through the C ABI: f32 <-> plane conversion and its accuracy contract, the
three operand-mode pairs, f32 and H-format outputs with the fused epilogues, ragged shapes and stream-K tails, against fp64 on the
same seeded inputs."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _mk(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g, dtype=torch.float32) * scale


def _close(got, exp, what, tol=3e-6):
    err = (got.double().cpu() - exp).abs().max().item()
    scale = exp.abs().max().item() + 1e-30
    print(f'{what}: max abs err {err:.3e} (scale {scale:.3e})')
    assert err <= tol * scale, what


def _planes_close(pm, x, what='planes'):
    """the format's contract: |X - (h1 + h2) 2^-e| <= 2^-23 |X| + 2^-39 bound per element, bound 2^e in [2^14, 2^15)"""
    e, bound, _, _ = pm.header()
    assert x.abs().max().item() <= bound and 2.0 ** 14 <= bound * 2.0 ** e < 2.0 ** 15, what
    err = (pm.to_f32().double() - x.double()).abs()
    tol = 2.0 ** -23 * x.double().abs() + 2.0 ** -39 * bound
    assert (err <= tol).all(), f'{what}: {(err > tol).sum().item()} elements outside the format contract (max err {err.max().item():.3e})'


@pytest.mark.parametrize('R,C', [(4, 16), (197, 384), (130, 70), (33, 17), (1000, 264)])
def test_hformat_round_trip_contract(R, C):
    """x 2^e = h1 + h2 with e from the measured maximum: 2^-23 relative inside the window, 2^-39 of the maximum below it; the header
    holds the measured maximum and the largest squared row norm"""
    from ofb_amd import hip
    x = _mk((R, C), 1)
    x[0, 0], x[-1, -1] = 0.0, -3.0e-20
    x = (x * torch.pow(torch.tensor(2.0), torch.randint(-30, 30, (R, C)).float())).cuda()
    pm = hip.to_hformat(x)
    _planes_close(pm, x)
    e, amax, rn2sq, _ = pm.header()
    assert amax == x.abs().max().item()
    assert abs(rn2sq - x.double().pow(2).sum(1).max().item()) <= 1e-5 * rn2sq
    # narrow-range data (what the model produces): plain 2^-23 relative
    y = (_mk((R, C), 3) + 3.0 * torch.sign(_mk((R, C), 4))).cuda()
    back = hip.to_hformat(y).to_f32()
    assert ((back - y).abs() <= 2.0 ** -23 * y.abs()).all()
    rs = _mk(((R + 6) // 7,), 2).cuda()
    pm = hip.to_hformat(x, rowscale=rs, rs_div=7)
    _planes_close(pm, x * rs[torch.arange(R, device='cuda') // 7].unsqueeze(1), 'row-scaled planes')
    # a caller-supplied (loose) bound instead of the statistics pass
    b = (x.abs().max() * 7.0).reshape(1)
    pm = hip.to_hformat(x, bound=b)
    assert pm.header()[1] == float(b)
    _planes_close(pm, x, 'planes under a loose bound')
    # all-zero and non-finite-free corner: zeros stay zeros
    z = torch.zeros(R, C, device='cuda')
    assert torch.equal(hip.to_hformat(z).to_f32(), z)


def test_hformat_with_column_sums():
    """the conversion pass of a gradient also yields its column sums (bias gradient), DropPath row scale included"""
    from ofb_amd import hip
    R, C = 1970, 264
    x = _mk((R, C), 5).cuda()
    rs = _mk((10,), 6).cuda()
    out = torch.full((C,), float('nan'), device='cuda')
    pm = hip.to_hformat(x, rowscale=rs, rs_div=197, colsum_out=out)
    scaled = x * rs[torch.arange(R, device='cuda') // 197].unsqueeze(1)
    _planes_close(pm, scaled)
    _close(out, scaled.double().sum(0).cpu(), 'column sums', tol=2e-6)
    out2 = torch.empty(C, device='cuda')
    hip.colsum_h(pm, out2)
    _close(out2, scaled.double().sum(0).cpu(), 'column sums of an H-format matrix', tol=2e-6)


SHAPES = [(256, 256, 64), (394, 384, 384), (591, 1152, 384), (130, 70, 36), (128, 1000, 384), (77, 13, 5), (591, 264, 200),
          (2600, 520, 48), (8192, 1024, 80),
          # widths that pad badly on 192 columns and M >= 1024: the 256 x 96 tile (C96) takes the token-row forms (kc,kc / kc,kr);
          # the pruned / finetune widths of configs[4] (264, 480, 672, 160, 224, 576, 960) and odd ones
          (1100, 264, 200), (2048, 480, 264), (1500, 96, 64), (1300, 672, 264), (1024, 100, 40), (5000, 224, 160), (1234, 77, 264),
          (1027, 576, 264), (1500, 960, 264),
          # more than half a round but at most one round of 128-row tiles (2 x CUs workgroups): the token-row forms run on the
          # 112 x 192 tile (C112F: 7 x 3 blocks per wave, a short second epilogue pass), incl. ragged widths and a last tile that is cut
          (19200, 384, 64), (19000, 264, 72), (17001, 576, 40)]


@pytest.fixture(params=[0, 96, 112], ids=['tile128', 'tile96', 'tile112'])
def tile_choice(request):
    """every shape on the default tile and, where it is legal (token-row forms, M >= 1024), on the 256 x 96 tile (32x32x16, K16 stages);
    112 = the 112 x 192 tile switched on (the host picks it for the token-row forms of the shapes marked below, the others skip)"""
    from ofb_amd import hip
    if request.param == 112:
        hip.tune(hip.TUNE_GEMM_T112, 1)
    else:
        hip.tune(hip.TUNE_GEMM_TILE, request.param)
    yield request.param
    hip.tune(hip.TUNE_GEMM_TILE, 0)
    hip.tune(hip.TUNE_GEMM_T112, 0)


@pytest.mark.parametrize('M,N,K', SHAPES)
def test_gemm_h_modes_and_outputs(M, N, K, tile_choice):
    from ofb_amd import hip
    if tile_choice == 96 and M < 1024:
        pytest.skip('the 256 x 96 tile needs M >= 1024')
    if tile_choice == 112 and M < 17000:
        pytest.skip('the 112 x 192 tile is only chosen between half a round and one round of 128-row tiles')
    a, b = _mk((M, K), 3), _mk((N, K), 4)                       # logical A[M][K], B[N][K]
    exact = a.double() @ b.double().t()
    ad, bd = a.cuda(), b.cuda()
    pa_kc, pb_kc = hip.to_hformat(ad), hip.to_hformat(bd)                                    # P matrices [rows][K]
    pa_kr, pb_kr = hip.to_hformat(ad.t().contiguous()), hip.to_hformat(bd.t().contiguous())   # P matrices [K][rows]
    for name, (A, B, akc, bkc) in {'kc,kc': (pa_kc, pb_kc, 1, 1), 'kc,kr': (pa_kc, pb_kr, 1, 0), 'kr,kr': (pa_kr, pb_kr, 0, 0)}.items():
        out = torch.full((M, N), float('nan'), device='cuda')
        outp = hip.HMat(M, N, 'cuda')
        hip.gemm_h(A, B, akc, bkc, M, N, K, C_out=out, ldc=N, Cp=outp)
        _close(out, exact, f'{name} {M}x{N}x{K} f32 out')
        e, bound, _, _ = outp.header()
        assert out.abs().max().item() <= bound, 'the Cauchy-Schwarz bound of the output'
        assert (outp.to_f32() - out).abs().max().item() <= 2.0 ** -23 * bound, 'the H-format output carries the f32 result'
        out2 = torch.empty(M, N, device='cuda')
        hip.gemm_h(A, B, akc, bkc, M, N, K, C_out=out2, ldc=N)
        assert torch.equal(out, out2), 'deterministic (fixed-order partial sums)'
    # the zero padding of an H-format OUTPUT must be real zeros: feed it to a product that reduces over its padded rows
    outp = hip.HMat(M, N, 'cuda')
    outp.buf.fill_(0x7f)                                           # poison (NaN patterns) before the kernel writes it
    hip.gemm_h(pa_kc, pb_kc, 1, 1, M, N, K, Cp=outp)
    y = torch.empty(N, K, device='cuda')
    hip.gemm_h(outp, pa_kc, 0, 0, N, K, M, C_out=y, ldc=K)          # y = out^T @ a   (reduction over M, padded to 16)
    _close(y, exact.t() @ a.double(), 'H-format output as KR operand (zero padding)', tol=2e-5)


@pytest.mark.parametrize('M,N,K', [(394, 384, 384), (300, 1536, 384), (130, 70, 36), (1200, 264, 72), (1400, 480, 264), (2100, 672, 100),
                                   (19200, 384, 48), (18999, 264, 40)])          # (the last two: the 112 x 192 tile)
def test_gemm_h_epilogues(M, N, K, tile_choice):
    from ofb_amd import hip
    if tile_choice == 96 and M < 1024:
        pytest.skip('the 256 x 96 tile needs M >= 1024')
    if tile_choice == 112 and M < 17000:
        pytest.skip('the 112 x 192 tile is only chosen between half a round and one round of 128-row tiles')
    x, w, b = _mk((M, K), 5), _mk((N, K), 6, 0.1), _mk((N,), 7)
    cs, res = _mk((N,), 8), _mk((M, N), 9)
    rs = _mk(((M + 196) // 197,), 10)
    xp, wp = hip.to_hformat(x.cuda()), hip.to_hformat(w.cuda())
    bd, csd, resd, rsd = b.cuda(), cs.cuda(), res.cuda(), rs.cuda()
    ref = x.double() @ w.double().t() + b.double()
    pre = ref * cs.double()
    # fc1 form: bias, gate column scale, GELU; f32 pre-activation to aux, gelu output as H-format only
    aux, hp = torch.empty(M, N, device='cuda'), hip.HMat(M, N, 'cuda')
    hip.gemm_h(xp, wp, 1, 1, M, N, K, Cp=hp, bias=bd, colscale=csd, aux=aux, ldaux=N, act=hip.ACT_GELU)
    _close(aux, pre, 'gelu pre-activation')
    _close(hp.to_f32(), torch.nn.functional.gelu(pre), 'gelu out (H-format)')
    # proj / fc2 form: residual + per-sample row scale
    out = torch.empty(M, N, device='cuda')
    hip.gemm_h(xp, wp, 1, 1, M, N, K, C_out=out, ldc=N, bias=bd, rowscale=rsd, rs_div=197, resid=resd, ldr=N)
    rows = torch.arange(M) // 197
    _close(out, ref * rs.double()[rows].unsqueeze(1) + res.double(), 'residual + rowscale')
    # dGELU form with H-format output
    dp, cs_out = hip.HMat(M, N, 'cuda'), torch.full((N,), float('nan'), device='cuda')
    hip.gemm_h(xp, wp, 1, 1, M, N, K, Cp=dp, aux=aux, ldaux=N, act=hip.ACT_DGELU, colsum_out=cs_out)
    p = pre.clone().requires_grad_(True)
    torch.nn.functional.gelu(p).sum().backward()
    dref = (x.double() @ w.double().t()) * p.grad
    _close(dp.to_f32(), dref, 'dgelu (H-format)')
    _close(cs_out, dref.sum(0), 'column sums of the output from the fused epilogue', tol=1e-5)
    # the pair the MLP branch uses: the forward saves GELU'(pre-activation), the backward epilogue multiplies by it
    gaux, hp2 = torch.full((M, N), float('nan'), device='cuda'), hip.HMat(M, N, 'cuda')
    hip.gemm_h(xp, wp, 1, 1, M, N, K, Cp=hp2, bias=bd, colscale=csd, aux=gaux, ldaux=N, act=hip.ACT_GELU_GRAD)
    _close(gaux, p.grad, 'saved gelu derivative')
    assert torch.equal(hp2.to_f32(), hp.to_f32())                       # same values, same bound, same exponent
    dp2, cs2 = hip.HMat(M, N, 'cuda'), torch.full((N,), float('nan'), device='cuda')
    hip.gemm_h(xp, wp, 1, 1, M, N, K, Cp=dp2, aux=gaux, ldaux=N, act=hip.ACT_MULAUX, colsum_out=cs2)
    _close(dp2.to_f32(), dref, 'value x saved derivative (H-format)')
    _close(cs2, dref.sum(0), 'column sums (saved-derivative form)', tol=1e-5)


@pytest.mark.parametrize('M,N,K', [(394, 384, 384), (300, 1536, 384), (130, 70, 36), (1200, 264, 72), (1400, 480, 264), (512, 768, 96),
                                   (19200, 384, 48), (18999, 264, 40)])
def test_gemm_h_saved_derivative_in_t_layout(M, N, K, tile_choice):
    """OFB_ACT_GELU_GRAD_T / OFB_ACT_MULAUX_T (include/ofb_hip.h): the saved GELU derivative in the layout of the 128 x 192 tile's
    accumulators.  Same numbers as the row-major pair on every epilogue form - interior tiles straight from the registers (direct
    epilogue), edge tiles / other tile shapes / unaligned launches / the fix-up kernel through the index map - and the direct
    epilogue switched off (OFB_TUNE_GEMM_DIRECT = 0) gives bit-identical planes and derivative."""
    from ofb_amd import hip
    if tile_choice == 96 and M < 1024:
        pytest.skip('the 256 x 96 tile needs M >= 1024')
    if tile_choice == 112 and M < 17000:
        pytest.skip('the 112 x 192 tile is only chosen between half a round and one round of 128-row tiles')
    x, w, b, cs = _mk((M, K), 5), _mk((N, K), 6, 0.1), _mk((N,), 7), _mk((N,), 8)
    xp, wp, bd, csd = hip.to_hformat(x.cuda()), hip.to_hformat(w.cuda()), b.cuda(), cs.cuda()
    pre = ((x.double() @ w.double().t() + b.double()) * cs.double()).requires_grad_(True)
    torch.nn.functional.gelu(pre).sum().backward()
    dref = (x.double() @ w.double().t()) * pre.grad
    # row-major reference pair
    gaux, hp = torch.empty(M, N, device='cuda'), hip.HMat(M, N, 'cuda')
    hip.gemm_h(xp, wp, 1, 1, M, N, K, Cp=hp, bias=bd, colscale=csd, aux=gaux, ldaux=N, act=hip.ACT_GELU_GRAD)
    dp, csum = hip.HMat(M, N, 'cuda'), torch.empty(N, device='cuda')
    hip.gemm_h(xp, wp, 1, 1, M, N, K, Cp=dp, aux=gaux, ldaux=N, act=hip.ACT_MULAUX, colsum_out=csum)
    got = {}
    for direct in (1, 0):
        hip.tune(hip.TUNE_GEMM_DIRECT, direct)
        try:
            taux = hip.aux_t(M, N, 'cuda')
            taux.fill_(float('nan'))
            hp_t = hip.HMat(M, N, 'cuda')
            hp_t.buf.fill_(0x7f)
            hip.gemm_h(xp, wp, 1, 1, M, N, K, Cp=hp_t, bias=bd, colscale=csd, aux=taux, act=hip.ACT_GELU_GRAD_T)
            dp_t, cs_t = hip.HMat(M, N, 'cuda'), torch.full((N,), float('nan'), device='cuda')
            dp_t.buf.fill_(0x7f)
            hip.gemm_h(xp, wp, 1, 1, M, N, K, Cp=dp_t, aux=taux, act=hip.ACT_MULAUX_T, colsum_out=cs_t)
            # without the column sums (the form of a bias-less fc1)
            dp_n = hip.HMat(M, N, 'cuda')
            hip.gemm_h(xp, wp, 1, 1, M, N, K, Cp=dp_n, aux=taux, act=hip.ACT_MULAUX_T)
        finally:
            hip.tune(hip.TUNE_GEMM_DIRECT, 1)
        rows = hip.aux_t_to_rows(taux, M, N)
        assert torch.equal(rows, gaux), f'saved derivative (direct {direct})'
        assert torch.equal(hp_t.to_f32(), hp.to_f32()) and torch.equal(dp_t.to_f32(), dp.to_f32()), f'planes (direct {direct})'
        _close(dp_n.to_f32(), dref, 'value x saved derivative, no column sums')       # (may run on another tile than the colsum form)
        _close(cs_t, dref.sum(0), f'column sums (direct {direct})', tol=1e-5)
        got[direct] = (hp_t.buf.clone(), dp_t.buf.clone(), cs_t.clone())
    _close(gaux, pre.grad, 'saved gelu derivative')
    _close(dp.to_f32(), dref, 'value x saved derivative')
    # the padding of the H-format outputs is written the same way by both forms (real zeros: checked by the KR-operand test above)
    ncb = (N + 15) // 16
    for k in (0, 1):
        a, b2 = got[1][k], got[0][k]
        assert torch.equal(a[256:256 + ((M + 15) // 16) * 4 * ncb * 256], b2[256:256 + ((M + 15) // 16) * 4 * ncb * 256]), 'plane bytes incl. padding'


def test_gemm_h_direct_epilogue_full_size_fc1_and_dh():
    """the two launches the direct epilogue exists for, at the DeiT-S bs-128 size (25216 x 1536, K = 384: 1576 interior tiles, three rounds
    + a partial one; every unit but a workgroup's first starts from stages its predecessor's epilogue requested): a strided row sample
    against fp64, the column sums against the planes, run-to-run bit-identity."""
    from ofb_amd import hip
    M, D, N = 128 * 197, 384, 1536
    x, w, b, cs = _mk((M, D), 21), _mk((N, D), 22, 0.05), _mk((N,), 23), _mk((N,), 24)
    xp, wp = hip.to_hformat(x.cuda()), hip.to_hformat(w.cuda())
    rows = torch.arange(0, M, 61)
    taux, hp = hip.aux_t(M, N, 'cuda'), hip.HMat(M, N, 'cuda')
    hip.gemm_h(xp, wp, 1, 1, M, N, D, Cp=hp, bias=b.cuda(), colscale=cs.cuda(), aux=taux, act=hip.ACT_GELU_GRAD_T)
    pre = ((x[rows].double() @ w.double().t() + b.double()) * cs.double()).requires_grad_(True)
    act = torch.nn.functional.gelu(pre)
    act.sum().backward()
    _close(hp.to_f32()[rows.cuda()], act.detach(), 'gelu planes, full size')
    _close(hip.aux_t_to_rows(taux, M, N)[rows.cuda()], pre.grad, 'saved derivative, full size')
    dy = _mk((M, D), 25)
    dyp, w2p = hip.to_hformat(dy.cuda()), hip.to_hformat(w.cuda().t().contiguous())          # dH = dY W2, W2 [D][hid] read along its rows
    dp, csum = hip.HMat(M, N, 'cuda'), torch.full((N,), float('nan'), device='cuda')
    hip.gemm_h(dyp, w2p, 1, 0, M, N, D, Cp=dp, aux=taux, act=hip.ACT_MULAUX_T, colsum_out=csum)
    got = dp.to_f32()
    _close(got[rows.cuda()], (dy[rows].double() @ w.double().t()) * pre.grad, 'dH planes, full size')
    _close(csum, got.double().sum(0).cpu(), 'column sums of dH', tol=2e-6)
    dp2, csum2 = hip.HMat(M, N, 'cuda'), torch.empty(N, device='cuda')
    hip.gemm_h(dyp, w2p, 1, 0, M, N, D, Cp=dp2, aux=taux, act=hip.ACT_MULAUX_T, colsum_out=csum2)
    assert torch.equal(dp.to_f32(), dp2.to_f32()) and torch.equal(csum, csum2)


def test_gemm_h_deit_small_layer_shapes():
    """the bs-128 DeiT-S shapes (25216 tokens): full rounds + streamed tail; strided row sample against fp64"""
    from ofb_amd import hip
    M, D = 128 * 197, 384
    x = _mk((M, D), 11)
    xp = hip.to_hformat(x.cuda())
    rows = torch.arange(0, M, 97)
    for N in (1152, 384, 1536):
        w, b = _mk((N, D), 12, 0.05), _mk((N,), 13)
        out = torch.empty(M, N, device='cuda')
        hip.gemm_h(xp, hip.to_hformat(w.cuda()), 1, 1, M, N, D, C_out=out, ldc=N, bias=b.cuda())
        _close(out[rows.cuda()], x[rows].double() @ w.double().t() + b.double(), f'deit-s N {N}')
    # the fc2 input-gradient form at full size: 1576 tiles = 3 rounds + 5 whole tile rows in the streamed tail; the column sums of the
    # H-format-only output come from both (fused epilogue of the rounds, fix-up kernel of the tail)
    N = 1536
    w, aux = _mk((N, D), 15, 0.05), _mk((M, N), 16)
    dp, cs = hip.HMat(M, N, 'cuda'), torch.full((N,), float('nan'), device='cuda')
    hip.gemm_h(xp, hip.to_hformat(w.cuda()), 1, 1, M, N, D, Cp=dp, aux=aux.cuda(), ldaux=N, act=hip.ACT_MULAUX, colsum_out=cs)
    got = dp.to_f32()
    _close(got[rows.cuda()], (x[rows].double() @ w.double().t()) * aux[rows].double(), 'value x aux, planes out, full size')
    _close(cs, got.double().sum(0).cpu(), 'column sums over rounds + tail', tol=2e-6)
    # weight gradient: dW[N][K] = dY^T X over all tokens (12 tiles, K = 25216: tail only)
    dy = _mk((M, 1536), 14)
    dw = torch.empty(1536, D, device='cuda')
    hip.gemm_h(hip.to_hformat(dy.cuda()), xp, 0, 0, 1536, D, M, C_out=dw, ldc=D)
    _close(dw, dy.double().t() @ x.double(), 'dW 1536x384x25216', tol=2e-5)


@pytest.mark.parametrize('M,N,K', [(25216, 384, 384), (25216, 384, 1536), (19200, 384, 1152), (32896, 192, 400), (16500, 576, 392),
                                   (27800, 384, 384)])
def test_gemm_h_between_one_and_two_rounds(M, N, K):
    """between one and two tiles per CU slot pair (394 tiles at DeiT-S bs 128 and N = 384; 300; 257; 387 with a ragged last row tile and
    an odd number of K16 steps; 436): the shapes of five of the eight activation products of a block.  All epilogue forms these products
    take in the model + an H-format output; strided row sample against fp64; deterministic.  (Written for the round-4 lab form that ran
    the remainder tiles as streamed K ranges on the second workgroup of every CU, scripts/lab/gemm_h_ab_launch_inkernel_fixup.patch.txt;
    kept because nothing else covered these tile counts at full width.)"""
    from ofb_amd import hip
    x, w, b = _mk((M, K), 21), _mk((N, K), 22, 0.05), _mk((N,), 23)
    res, rs = _mk((M, N), 24), _mk(((M + 196) // 197,), 25)
    rows = torch.arange(0, M, 61)
    xd, wd = x.cuda(), w.cuda()
    xp, wp, wpr = hip.to_hformat(xd), hip.to_hformat(wd), hip.to_hformat(wd.t().contiguous())
    ref = x[rows].double() @ w.double().t()
    out = torch.full((M, N), float('nan'), device='cuda')
    hip.gemm_h(xp, wp, 1, 1, M, N, K, C_out=out, ldc=N, bias=b.cuda(), rowscale=rs.cuda(), rs_div=197, resid=res.cuda(), ldr=N)
    exp = (ref + b.double()) * rs.double()[rows // 197].unsqueeze(1) + res[rows].double()
    _close(out[rows.cuda()], exp, 'kc,kc bias + rowscale + residual')
    assert not torch.isnan(out).any()
    out2 = torch.empty_like(out)
    hip.gemm_h(xp, wp, 1, 1, M, N, K, C_out=out2, ldc=N, bias=b.cuda(), rowscale=rs.cuda(), rs_div=197, resid=res.cuda(), ldr=N)
    assert torch.equal(out, out2), 'deterministic'
    # input-gradient form (kc,kr) with the fused residual-gradient add, and the bound scalar the attention kernels take
    cb = torch.zeros(1, device='cuda')
    hip.gemm_h(xp, wpr, 1, 0, M, N, K, C_out=out, ldc=N, resid=res.cuda(), ldr=N)
    _close(out[rows.cuda()], ref + res[rows].double(), 'kc,kr + residual')
    hip.gemm_h(xp, wpr, 1, 0, M, N, K, C_out=out, ldc=N, cbound_out=cb)
    _close(out[rows.cuda()], ref, 'kc,kr plain')
    assert out.abs().max().item() <= float(cb)
    # H-format output (bound folded into the product, published by workgroup 0; the fix-up kernel reads it for the remainder tiles)
    outp = hip.HMat(M, N, 'cuda')
    hip.gemm_h(xp, wp, 1, 1, M, N, K, C_out=out, ldc=N, Cp=outp, bias=b.cuda())
    _close(out[rows.cuda()], ref + b.double(), 'kc,kc bias, f32 + planes')
    assert (outp.to_f32() - out).abs().max().item() <= 2.0 ** -23 * outp.header()[1]


def test_gemm_h_rejects_bad_arguments():
    from ofb_amd import hip
    a = hip.to_hformat(torch.zeros(32, 32, device='cuda'))
    out = torch.zeros(32, 32, device='cuda')
    with pytest.raises(hip.OfbError):
        hip.gemm_h(a, a, 0, 1, 32, 32, 32, C_out=out, ldc=32)          # A^T B^T is not on the path
    with pytest.raises(hip.OfbError):
        hip.gemm_h(a, a, 1, 1, 32, 32, 32)                             # no output at all
    with pytest.raises(hip.OfbError):
        hip.gemm_h(a, a, 1, 1, 32, 32, 64, C_out=out, ldc=32)          # K beyond the operand's granule columns
    with pytest.raises(hip.OfbError):
        hip.gemm_h(a, a, 1, 1, 32, 32, 32, C_out=out, ldc=32, act=hip.ACT_DGELU)   # dGELU without aux
    with pytest.raises(hip.OfbError):
        hip.gemm_h(a, a, 1, 1, 32, 32, 32, C_out=out, ldc=32, act=hip.ACT_GELU_GRAD)   # save-derivative form without aux


def test_weight_planes_are_refreshed_together():
    """hip.weight_h: every registered weight is converted by ONE multi-tensor launch pair per epoch (ofb_to_hformat_multi); the planes
    equal a single conversion, follow raw-pointer updates after bump_weight_epoch() and torch in-place edits (_version)"""
    from ofb_amd import hip
    shapes = [(1152, 384), (384, 384), (1536, 384), (384, 1536), (1000, 384), (70, 36), (33, 100)]
    ws = [torch.nn.Parameter(_mk(s, 40 + i).cuda()) for i, s in enumerate(shapes)]
    def same(pm, w):
        return torch.equal(pm.to_f32(), hip.to_hformat(w.detach().contiguous()).to_f32()) and pm.header() == hip.to_hformat(w.detach().contiguous()).header()
    for w in ws:
        assert same(hip.weight_h(w), w)
        assert ((hip.weight_h(w).to_f32() - w.detach()).abs() <= 2.0 ** -23 * w.detach().abs().max()).all()
    bufs = [hip.weight_h(w).buf.data_ptr() for w in ws]
    with torch.no_grad():
        for w in ws:
            w.mul_(1.5)                                       # torch in-place edit: _version moves
    assert same(hip.weight_h(ws[3]), ws[3])
    hip.lib().ofb_scale_rows                                  # (a raw-pointer edit below: only the epoch tells)
    x = ws[0].detach().clone()
    hip.scale_rows(x, torch.full((1152,), 2.0, device='cuda'), ws[0].data, 1152, 384)
    assert not same(hip.weight_h(ws[0]), ws[0])      # stale by design until the epoch is bumped
    hip.bump_weight_epoch()
    for w, b in zip(ws, bufs):
        pm = hip.weight_h(w)
        assert same(pm, w) and pm.buf.data_ptr() == b                # same persistent planes, fresh content
    # as GEMM operands (padding columns / rows of the multi-tensor conversion must be zero)
    a = _mk((50, 36), 60).cuda()
    out = torch.empty(50, 70, device='cuda')
    hip.gemm_h(hip.to_hformat(a), hip.weight_h(ws[5]), 1, 1, 50, 70, 36, C_out=out, ldc=70)
    _close(out, a.double().cpu() @ ws[5].detach().double().cpu().t(), 'multi-converted planes as GEMM operand', tol=2e-6)


@pytest.mark.parametrize('B,Cin,S,patch', [(3, 3, 224, 16), (2, 3, 64, 16), (5, 1, 48, 8)])
def test_patchify_planes_hold_the_patch_matrix(B, Cin, S, patch):
    """ofb_patchify_hformat: the planes of the patch-embedding conv's GEMM operand written straight from the images hold exactly the
    patch matrix of models/layers.py:177 (Conv2d with kernel = stride = patch): rows (b, py, px), columns (c, i, j)"""
    from ofb_amd import hip
    g = torch.Generator().manual_seed(B + S)
    imgs = torch.randn(B, Cin, S, S, generator=g).cuda()
    gh = S // patch
    ref = imgs.reshape(B, Cin, gh, patch, gh, patch).permute(0, 2, 4, 1, 3, 5).reshape(B * gh * gh, Cin * patch * patch)
    pm = hip.patchify_hformat(imgs, patch)
    assert (pm.R, pm.C) == tuple(ref.shape)
    assert pm.header()[1] == imgs.abs().max().item()
    assert ((pm.to_f32() - ref).abs() <= 2.0 ** -23 * imgs.abs().max()).all()


def test_single_round_scheduling_switches_move_tiles_not_bits():
    """A launch of ONE partial round (300 tiles on 512 slots): the XCD-balanced tile order (OFB_TUNE_GEMM_SCHED 2) and the yielding first
    workgroup (OFB_TUNE_GEMM_YIELD) only decide WHICH workgroup computes a tile and when - the f32 output, the planes and the row norms
    handed to LayerNorm backward are bit-identical to the contiguous order without yield."""
    from ofb_amd import hip
    M, N, K = 19200, 384, 200
    x, w, res = _mk((M, K), 31).cuda(), _mk((N, K), 32, 0.1).cuda(), _mk((M, N), 33).cuda()
    xp, wp = hip.to_hformat(x), hip.to_hformat(w)
    gamma, rowfac = _mk((N,), 34).cuda(), _mk((M,), 35).abs().cuda()
    outs = []
    try:
        for sched, yl in ((1, 0), (2, 0), (2, 4), (2, 7)):
            hip.tune(hip.TUNE_GEMM_SCHED, sched)
            hip.tune(hip.TUNE_GEMM_YIELD, yl)
            out = torch.full((M, N), float('nan'), device='cuda')
            rn = hip.gemm_h(xp, wp, 1, 1, M, N, K, C_out=out, ldc=N, resid=res, ldr=N, rn=(gamma, rowfac))
            outs.append((out, None if rn is None else rn[0].clone()))
    finally:
        hip.tune(hip.TUNE_GEMM_SCHED, 2)
        hip.tune(hip.TUNE_GEMM_YIELD, 4)
    ref = x.double().cpu() @ w.double().cpu().t() + res.double().cpu()
    _close(outs[0][0], ref, 'single-round product')
    for out, rn in outs[1:]:
        assert torch.equal(out, outs[0][0])
        assert (rn is None) == (outs[0][1] is None) and (rn is None or torch.equal(rn, outs[0][1]))


@pytest.mark.parametrize('cus', [224, 200, 131])
def test_gemm_h_full_size_products_on_fewer_cus(cus):
    """OFB_TUNE_GEMM_CUS (what `dp.GradAllReducer` sets in a group of more than one rank: 256 - 32 = 224 CUs for the GEMM plans while
    RCCL's kernels hold the rest; 200 and the odd 131 stand for other reservations): the persistent grid shrinks to 2 x cus workgroups,
    every plan - rounds, streamed tails, the one-round balance, the direct epilogue's unit chain, the weight gradients' K cuts - is
    another one.  The full-size checks of this file must hold unchanged on them."""
    from ofb_amd import hip
    try:
        hip.tune(hip.TUNE_GEMM_CUS, cus)
        test_gemm_h_direct_epilogue_full_size_fc1_and_dh()
        test_gemm_h_deit_small_layer_shapes()
        test_gemm_h_between_one_and_two_rounds(25216, 384, 384)
        test_gemm_h_between_one_and_two_rounds(27800, 384, 384)
    finally:
        hip.tune(hip.TUNE_GEMM_CUS, 0)


def test_gemm_h_cached_argument_struct_takes_every_call_s_own_pointers():
    """hip.gemm_h keeps one argument struct per call configuration: calls that differ only in their tensors must each see their own
    operands, side vectors and outputs; a switch change (hip.tune) drops the cached plans"""
    from ofb_amd import hip
    M, N, K = 300, 200, 96
    outs = []
    for rep in range(3):
        a, w = _mk((M, K), 10 + rep).cuda(), _mk((N, K), 20 + rep).cuda()
        bias, cs, res = _mk((N,), 30 + rep).cuda(), _mk((N,), 40 + rep).cuda(), _mk((M, N), 50 + rep).cuda()
        aP, wP = hip.to_hformat(a), hip.to_hformat(w)
        y, colsum = torch.empty(M, N, device='cuda'), torch.empty(N, device='cuda')
        hip.gemm_h(aP, wP, 1, 1, M, N, K, C_out=y, ldc=N, bias=bias, colscale=cs, resid=res, ldr=N, colsum_out=colsum)
        exp = ((a.double() @ w.double().t() + bias.double()) * cs.double() + res.double()).cpu()
        _close(y, exp, f'call {rep}: product + bias, column scale, residual')
        _close(colsum, exp.sum(0), f'call {rep}: column sums', tol=2e-5)
        outs.append(y)
        if rep == 1:
            hip.tune(hip.TUNE_GEMM_YIELD, 4)                  # (its default value: only the cache epoch moves)
    assert outs[0].data_ptr() != outs[1].data_ptr() and not torch.equal(outs[0], outs[1])
    # same shapes, fewer side inputs: a different configuration, not the struct of the calls above with stale pointers
    a, w = _mk((M, K), 60).cuda(), _mk((N, K), 61).cuda()
    y = torch.empty(M, N, device='cuda')
    hip.gemm_h(hip.to_hformat(a), hip.to_hformat(w), 1, 1, M, N, K, C_out=y, ldc=N)
    _close(y, (a.double() @ w.double().t()).cpu(), 'plain product after the decorated ones')
