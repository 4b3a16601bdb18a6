"""Fused attention core (forward + backward) vs the fp64 definition softmax(q k^T * scale) v."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(qkv, B, N, H, dh, scale, dout):
    x = qkv.double().reshape(B, N, 3, H, dh).permute(2, 0, 3, 1, 4).detach().requires_grad_(True)
    q, k, v = x[0], x[1], x[2]
    s = (q @ k.transpose(-2, -1)) * scale
    p = torch.softmax(s, -1)
    o = (p @ v).transpose(1, 2).reshape(B * N, H * dh)
    lse = torch.logsumexp(s, -1)
    o.backward(dout.double())
    dqkv = x.grad.permute(1, 3, 0, 2, 4).reshape(B * N, 3 * H * dh)
    return o.detach(), lse.detach().reshape(B * H, N), dqkv


# (B, N, H, dh): DeiT-S/T/B head shapes, pruned head dims, short and maximal sequences
# N > 208 runs chunked (forward: query chunks per workgroup, backward: one launch per 224 keys): 384 px (577), patch 8 (785), chunk edges
CASES = [(2, 197, 6, 64), (3, 197, 3, 64), (2, 197, 4, 40), (1, 197, 2, 16), (2, 50, 2, 32), (1, 208, 2, 64), (2, 33, 1, 24),
         (1, 1, 1, 8), (2, 577, 3, 64), (3, 785, 2, 40), (2, 209, 1, 64), (1, 224, 2, 32), (3, 225, 1, 16), (2, 448, 1, 64), (1, 1025, 1, 64)]


@pytest.mark.parametrize('B,N,H,dh', CASES)
def test_attention_fwd_bwd(B, N, H, dh):
    from ofb_amd import hip
    g = torch.Generator().manual_seed(B * 1000 + N + H + dh)
    qkv = torch.randn(B * N, 3 * H * dh, generator=g) * 1.5
    dout = torch.randn(B * N, H * dh, generator=g)
    scale = 0.125
    o_ref, lse_ref, dqkv_ref = _ref(qkv, B, N, H, dh, scale, dout)
    qd = qkv.cuda()
    o = torch.empty(B * N, H * dh, device='cuda')
    lse = torch.empty(2 * B * H, N, device='cuda')
    hip.attention_fwd(qd, o, lse, B, N, H, dh, scale)
    e_o = (o.cpu().double() - o_ref).abs().max().item()
    e_l = (lse[:B * H].cpu().double() - lse_ref).abs().max().item()
    print(f'attn fwd B{B} N{N} H{H} d{dh}: out err {e_o:.2e} lse err {e_l:.2e}')
    assert e_o < 2e-5 and e_l < 2e-5
    # the H-format form of the output: the planes carry the kernel's own f32 rows to 2^-22 of the tensor bound; image 0 (no tile
    # shift) == the plain kernel bit for bit
    o2, lse2 = torch.full_like(o, float('nan')), torch.full_like(lse, float('nan'))
    oP = hip.HMat.for_rows_written_by_kernel(B * N, H * dh, 'cuda')
    qb = hip.attention_fwd_h(qd, o2, oP, lse2, B, N, H, dh, scale)
    assert float(qb) == qd.abs().max().item() and oP.header()[1] == float(qb)
    assert (oP.to_f32() - o2).abs().max().item() <= 2.0 ** -22 * float(qb)
    assert torch.equal(o2[:N], o[:N]) and torch.equal(lse2[:H], lse[:H])
    assert (o2.cpu().double() - o_ref).abs().max().item() < 2e-5 and (lse2[:B * H].cpu().double() - lse_ref).abs().max().item() < 2e-5
    dqkv = torch.full((B * N, 3 * H * dh), float('nan'), device='cuda')
    hip.attention_bwd(qd, o, lse, dout.cuda(), dqkv, B, N, H, dh, scale)
    err = (dqkv.cpu().double() - dqkv_ref).abs()
    Hd = H * dh
    print(f'attn bwd: dq {err[:, :Hd].max():.2e} dk {err[:, Hd:2*Hd].max():.2e} dv {err[:, 2*Hd:].max():.2e} '
          f'(scale {dqkv_ref.abs().max():.2e})')
    assert not torch.isnan(dqkv).any()
    assert err.max().item() < 5e-5 * max(1.0, dqkv_ref.abs().max().item())
    # deterministic, and the reported maximum is the maximum of what was stored
    again, amax = torch.empty_like(dqkv), torch.zeros(1, device='cuda')
    hip.attention_bwd(qd, o, lse, dout.cuda(), again, B, N, H, dh, scale, dqkv_amax=amax)
    assert torch.equal(again, dqkv)
    if N <= 208:
        assert float(amax) == dqkv.abs().max().item()
    else:                                                   # chunked: the running dq sums of the earlier launches are included (an upper bound)
        assert dqkv.abs().max().item() <= float(amax) <= 4.0 * dqkv.abs().max().item()
    # the per-workgroup form of the maximum (no atomics, no memset node): word (b, h) = max |dq|dk|dv| of that head; the conversion pass
    # that takes the vector as its bound gives the same planes and column sums as the one that gets the scalar
    third, wg = torch.empty_like(dqkv), torch.full((B * H,), float('nan'), device='cuda')
    hip.attention_bwd(qd, o, lse, dout.cuda(), third, B, N, H, dh, scale, wg_amax=wg)
    assert torch.equal(third, dqkv) and float(wg.max()) == float(amax)
    if N <= 208:
        per_head = dqkv.view(B, N, 3, H, dh).abs().amax(dim=(1, 2, 4)).reshape(-1)
        assert torch.equal(wg, per_head)
    cs_a, cs_b = torch.empty(3 * Hd, device='cuda'), torch.empty(3 * Hd, device='cuda')
    pa = hip.to_hformat(dqkv, B * N, 3 * Hd, 3 * Hd, colsum_out=cs_a, bound=amax)
    pb = hip.to_hformat(dqkv, B * N, 3 * Hd, 3 * Hd, colsum_out=cs_b, bound=wg)
    assert pa.header() == pb.header() and torch.equal(pa.to_f32(), pb.to_f32()) and torch.equal(cs_a, cs_b)
    # a LOOSE bound (what the model hands over: Cauchy-Schwarz bounds, 4-30x the maximum) costs no accuracy
    loose_q, loose_d = hip.amax(qd) * 16.0, hip.amax(dout.cuda()) * 16.0
    o3, lse3, dq3 = torch.empty_like(o), torch.empty_like(lse), torch.empty_like(dqkv)
    hip.attention_fwd(qd, o3, lse3, B, N, H, dh, scale, loose_q)
    hip.attention_bwd(qd, o3, lse3, dout.cuda(), dq3, B, N, H, dh, scale, loose_q, loose_d)
    assert (o3.cpu().double() - o_ref).abs().max().item() < 2e-5
    assert (dq3.cpu().double() - dqkv_ref).abs().max().item() < 5e-5 * max(1.0, dqkv_ref.abs().max().item())


def test_attention_peaked_softmax():
    """one key dominates each row (large logits): exercises the max-subtraction path."""
    from ofb_amd import hip
    B, N, H, dh = 1, 197, 2, 64
    g = torch.Generator().manual_seed(5)
    qkv = torch.randn(B * N, 3 * H * dh, generator=g)
    qkv[:, :H * dh] *= 12.0
    dout = torch.randn(B * N, H * dh, generator=g)
    o_ref, lse_ref, dqkv_ref = _ref(qkv, B, N, H, dh, 0.125, dout)
    o = torch.empty(B * N, H * dh, device='cuda')
    lse = torch.empty(2 * B * H, N, device='cuda')
    hip.attention_fwd(qkv.cuda(), o, lse, B, N, H, dh, 0.125)
    assert (o.cpu().double() - o_ref).abs().max().item() < 5e-5
    dqkv = torch.empty(B * N, 3 * H * dh, device='cuda')
    hip.attention_bwd(qkv.cuda(), o, lse, dout.cuda(), dqkv, B, N, H, dh, 0.125)
    assert (dqkv.cpu().double() - dqkv_ref).abs().max().item() < 2e-4 * dqkv_ref.abs().max().item()


def test_attention_rejects_unsupported():
    from ofb_amd import hip
    t = torch.zeros(300 * 192, device='cuda')
    with pytest.raises(hip.OfbError):
        hip.attention_fwd(t, t, t, 1, 5000, 1, 64, 0.125)    # N > 4096
    with pytest.raises(hip.OfbError):
        hip.attention_fwd(t, t, t, 1, 100, 1, 68, 0.125)     # dh > 64
    with pytest.raises(hip.OfbError):
        hip.attention_fwd(t, t, t, 1, 100, 1, 30, 0.125)     # dh % 4 != 0


@pytest.mark.parametrize('B,N', [(3, 197), (2, 207), (2, 50), (2, 577)])
def test_attention_branch_planes(B, N):
    """ops.attn_branch end to end (qkv GEMM, attention, projection, residual) against fp64 autograd: the attention forward writes the
    projection's operand planes (shifted tile origin; N = 207 with two images spills into a second query chunk, N = 577 is the
    384-px sequence)"""
    from ofb_amd import ops
    H, dh, D = 2, 32, 64
    g = torch.Generator().manual_seed(N)
    x = torch.randn(B, N, D, generator=g)
    wq, bq = torch.randn(3 * H * dh, D, generator=g) * 0.1, torch.randn(3 * H * dh, generator=g) * 0.1
    wp, bp = torch.randn(D, H * dh, generator=g) * 0.1, torch.randn(D, generator=g) * 0.1
    dout = torch.randn(B, N, D, generator=g)
    assert ops._att_planes_ok(B, N)
    tens = [t.cuda().requires_grad_(True) for t in (x, wq, bq, wp, bp)]
    out = ops.attn_branch(tens[0], None, tens[1], tens[2], tens[3], tens[4], None, None, H, dh ** -0.5)
    out.backward(dout.cuda())
    ref = [t.double().requires_grad_(True) for t in (x, wq, bq, wp, bp)]
    qkv = (ref[0] @ ref[1].t() + ref[2]).reshape(B, N, 3, H, dh).permute(2, 0, 3, 1, 4)
    att = torch.softmax(qkv[0] @ qkv[1].transpose(-2, -1) * dh ** -0.5, -1) @ qkv[2]
    o_ref = ref[0] + att.transpose(1, 2).reshape(B, N, H * dh) @ ref[3].t() + ref[4]
    o_ref.backward(dout.double())
    assert (out.detach().cpu().double() - o_ref.detach()).abs().max().item() < 1e-5
    for t, r, name in zip(tens, ref, ('dx', 'dWqkv', 'dbqkv', 'dWproj', 'dbproj')):
        err = (t.grad.cpu().double() - r.grad).abs().max().item() / (r.grad.abs().max().item() + 1e-30)
        print(f'  {name}: rel err {err:.2e}')
        assert err < 2e-5, name


def test_attention_branch_without_qkv_bias():
    """`Attention`, `Block` and `MAEBlock` default to qkv_bias=False (reference layers.py:369, vision_transformer.py:145,174): the backward
    then has no consumer for the column sums of dq|dk|dv but still hands the per-workgroup maxima to the conversion pass
    (that combination raised)."""
    from ofb_amd import ops
    B, N, H, dh, D = 3, 197, 2, 32, 64
    g = torch.Generator().manual_seed(7)
    x = torch.randn(B, N, D, generator=g)
    wq = torch.randn(3 * H * dh, D, generator=g) * 0.1
    wp, bp = torch.randn(D, H * dh, generator=g) * 0.1, torch.randn(D, generator=g) * 0.1
    dout = torch.randn(B, N, D, generator=g)
    tens = [t.cuda().requires_grad_(True) for t in (x, wq, wp, bp)]
    out = ops.attn_branch(tens[0], None, tens[1], None, tens[2], tens[3], None, None, H, dh ** -0.5)
    out.backward(dout.cuda())
    ref = [t.double().requires_grad_(True) for t in (x, wq, wp, bp)]
    qkv = (ref[0] @ ref[1].t()).reshape(B, N, 3, H, dh).permute(2, 0, 3, 1, 4)
    att = torch.softmax(qkv[0] @ qkv[1].transpose(-2, -1) * dh ** -0.5, -1) @ qkv[2]
    o_ref = ref[0] + att.transpose(1, 2).reshape(B, N, H * dh) @ ref[2].t() + ref[3]
    o_ref.backward(dout.double())
    assert (out.detach().cpu().double() - o_ref.detach()).abs().max().item() < 1e-5
    for t, r, name in zip(tens, ref, ('dx', 'dWqkv', 'dWproj', 'dbproj')):
        err = (t.grad.cpu().double() - r.grad).abs().max().item() / (r.grad.abs().max().item() + 1e-30)
        assert err < 2e-5, name
    # and through the module API with the reference's default constructor arguments
    import ofb_amd
    blk = ofb_amd.Block(D, H).cuda()
    assert blk.attn.qkv.bias is None
    y = blk(tens[0].detach().requires_grad_(True))
    y.sum().backward()
    assert torch.isfinite(blk.attn.qkv.weight.grad).all()
