"""norm_targets / PMIM loss / CE / patch mask kernels vs oracle definitions and the reference golden."""
import numpy as np
import pytest
import torch

from oracle import fill
from oracle import ofb_oracle as O
from tests.golden_util import GOLDEN_DIR

pytestmark = pytest.mark.gpu


def test_norm_targets_golden_and_oracle():
    from ofb_amd import ops
    z = np.load(f'{GOLDEN_DIR}/norm_targets.npz')
    imgs = torch.from_numpy(fill.images(1, tag='nt'))
    imgs[0, 2, :40, :] = 0.25
    t = ops.norm_targets(imgs.cuda()).cpu()
    exp = torch.from_numpy(z['out'])
    err = (t[0][:, z['rows'], :] - exp).abs()
    print(f'norm_targets vs reference: max {err[:2].max():.2e} (textured planes), median {err.median():.2e}')
    assert float(err[:2].max()) < 2e-4 and float(err.median()) < 1e-5
    g = torch.Generator().manual_seed(3)
    x = torch.randn(3, 3, 224, 224, generator=g)
    got = ops.norm_targets(x.cuda()).cpu()
    assert float((got - O.norm_targets(x)).abs().max()) < 1e-4


def test_norm_targets_masked_patches_only():
    """the fused kernel over the masked patches' windows: those pixels equal the full-image kernels' values (same summation
    order), corner / edge patches (partial windows, count_include_pad=False) included; nothing else is written"""
    from ofb_amd import hip, ops
    g = torch.Generator().manual_seed(11)
    x = torch.randn(3, 3, 224, 224, generator=g).cuda()
    full = ops.norm_targets(x)
    L, P = 196, 16
    ids = torch.tensor([0, 13, 14 * 13, 195, 196 + 7, 196 + 100, 2 * 196 + 97, 2 * 196 + 195], dtype=torch.int32, device='cuda')
    out = torch.full_like(x, float('nan'))
    hip.norm_targets_masked(x, ids, out, 3, 3, L, P, 224, 224)
    seen = torch.zeros(3, 224, 224, dtype=torch.bool, device='cuda')
    for pid in ids.tolist():
        b, l = divmod(pid, L)
        py, px = divmod(l, 14)
        sl = (b, slice(None), slice(py * P, py * P + P), slice(px * P, px * P + P))
        assert float((out[sl] - full[sl]).abs().max()) <= 1e-6 * float(full[sl].abs().max())
        seen[b, py * P:py * P + P, px * P:px * P + P] = True
    assert torch.isnan(out[:, 0][~seen]).all()            # the rest of the buffer is untouched
    assert float((out.cpu()[0, :, :16, :16] - O.norm_targets(x.cpu())[0, :, :16, :16]).abs().max()) < 1e-4


def test_patch_mask_and_ce():
    from ofb_amd import hip, ops
    n = torch.from_numpy(fill.patch_noise(5))
    for keep in (0.95, 0.75, 0.5):
        lk = int(196 * keep)
        mask = torch.empty(5, 196, device='cuda')
        ids = torch.empty(5 * (196 - lk), device='cuda', dtype=torch.int32)
        hip.patch_mask(n.cuda(), mask, 5, 196, lk, ids)
        ref = O.keep_mask_from_noise(n, lk)
        assert torch.equal(mask.cpu(), ref)
        assert sorted(ids.cpu().tolist()) == torch.nonzero(ref.reshape(-1)).reshape(-1).tolist()
    g = torch.Generator().manual_seed(4)
    logits = (torch.randn(37, 1000, generator=g) * 3).requires_grad_(True)
    labels = torch.randint(0, 1000, (37,), generator=g)
    ref = O.label_smoothing_ce(logits.double(), labels)
    ref.backward()
    ld = logits.detach().cuda().requires_grad_(True)
    loss = ops.LabelSmoothingCE.apply(ld, labels.cuda(), 0.1)
    (loss * 1.7).backward()
    assert abs(float(loss) - float(ref)) < 1e-5
    assert float((ld.grad.cpu() / 1.7 - logits.grad).abs().max()) < 1e-7


def test_loss_mixing_is_the_reference_expression():
    """ops.ArchLoss / ops.TotalLoss (ofb_loss_mix: one launch each) against the expressions of reference losses.py:97-104 and
    engine.py:134-144 written out in torch: values and every gradient, with and without the arch term / the decoder loss."""
    import ofb_amd
    from ofb_amd import ops, engine
    dev = torch.device('cuda')
    torch.manual_seed(5)
    w = (0.5, 0.25, 0.125, 5.0)
    for with_arch, with_dec in ((True, True), (True, False), (False, True)):
        leaves = [torch.rand(3, device=dev, requires_grad=True), torch.rand((), device=dev, requires_grad=True),
                  (torch.rand((), device=dev) + 3).requires_grad_(True), (torch.rand((), device=dev) + 0.5).requires_grad_(True)]
        res = []
        for fused in (True, False):
            sp, fl, base, dec = [x.detach().clone().requires_grad_(True) for x in leaves]
            if with_arch:
                arch = ops.ArchLoss.apply(sp, fl, *w) if fused else (sp * torch.tensor(w[:3], device=dev)).sum() + w[3] * fl
            else:
                arch = None
            d = dec if with_dec else 0.
            if fused:
                b2, a2, total = engine.mix_losses((base, arch) if with_arch else base, d)
            else:
                total = base if arch is None else base + arch
                if with_dec:
                    total = total + (base / dec).detach() * dec
            (total * 1.7).backward()
            res.append([total.detach()] + [None if x.grad is None else x.grad.clone() for x in (sp, fl, base, dec)])
        for a, b in zip(*res):
            assert (a is None) == (b is None)
            if a is not None:
                assert torch.allclose(a, b, rtol=1e-6, atol=1e-7), (with_arch, with_dec, a, b)


@pytest.mark.parametrize('B,L,D,n_per', [(5, 196, 384, 10), (3, 9, 66, 4), (2, 16, 40, 0)])
def test_token_taps_gather_and_scatter(B, L, D, n_per):
    """ops.TokenTaps (ofb_token_taps_fwd / _bwd): the cls rows and the masked patches' token rows of the final stream, and the stream
    gradient with exactly those rows set - against index arithmetic written out in torch (reference vision_transformer.py:735-744)."""
    from ofb_amd import ops
    dev = torch.device('cuda')
    g = torch.Generator(device=dev).manual_seed(B * 1000 + L)
    T = L + 1
    latent = torch.randn(B, T, D, device=dev, generator=g, requires_grad=True)
    ids = torch.cat([b * L + torch.randperm(L, device=dev, generator=g)[:n_per] for b in range(B)]).to(torch.int32) if n_per else \
        torch.empty(0, device=dev, dtype=torch.int32)
    cls, z = ops.TokenTaps.apply(latent, ids)
    rows = (ids + torch.div(ids, L, rounding_mode='floor') + 1).long()
    assert torch.equal(cls, latent.detach()[:, 0])
    assert torch.equal(z, latent.detach().reshape(B * T, D)[rows])
    dcls, dz = torch.randn_like(cls), torch.randn_like(z)
    (cls * dcls).sum().add((z * dz).sum()).backward()
    ref = torch.zeros(B * T, D, device=dev)
    ref.view(B, T, D)[:, 0] = dcls
    ref[rows] = dz
    assert torch.equal(latent.grad.reshape(B * T, D), ref)
    # only one of the two readers has a gradient
    from ofb_amd import hip
    for dc, dzz in ((dcls, None), (None, dz)):
        if dzz is None or ids.numel():
            g2 = torch.full((B, T, D), float('nan'), device=dev)
            hip.token_taps_bwd(dc, dzz, ids if ids.numel() else None, int(ids.numel()), B, T, D, g2)
            ref2 = torch.zeros(B * T, D, device=dev)
            if dc is not None:
                ref2.view(B, T, D)[:, 0] = dc
            if dzz is not None:
                ref2[rows] = dzz
            assert torch.equal(g2.reshape(B * T, D), ref2)


def test_droppath_scales_match_the_timm_expression():
    from ofb_amd import hip
    dev = torch.device('cuda')
    g = torch.Generator(device=dev).manual_seed(11)
    u = torch.rand(22, 128, device=dev, generator=g)
    keep = (1.0 - torch.linspace(0.0, 0.1, 12, device=dev)[1:].repeat_interleave(2)).unsqueeze(1).contiguous()
    out = torch.empty_like(u)
    hip.droppath_scales(u, keep, out, 22, 128)
    assert torch.equal(out, torch.floor(keep + u) / keep)


def test_adamw_reuses_its_device_table_only_while_every_address_stays():
    """optim.AdamW keeps the device-side tensor table of a launch while the parameter, gradient and moment addresses are what they
    were: steps with in-place gradients reuse it, a new gradient tensor or replaced moments rebuild it; the walk equals
    torch.optim.AdamW's either way"""
    from ofb_amd.optim import AdamW
    torch.manual_seed(3)
    shapes = [(37, 5), (64,), (3, 7, 2)]
    ps = [torch.nn.Parameter(torch.randn(s, device='cuda')) for s in shapes]
    ref = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    opt = AdamW(ps, None, lr=1e-2, weight_decay=0.05)
    ropt = torch.optim.AdamW(ref, lr=1e-2, weight_decay=0.05)
    grads = [torch.empty_like(p) for p in ps]
    for step in range(6):
        if step == 3:
            grads = [torch.empty_like(p) for p in ps]         # gradients at new addresses (the old ones are still alive)
        for p, r, g in zip(ps, ref, grads):
            g.copy_(torch.randn_like(p))
            p.grad, r.grad = g, g.clone()
        if step == 5:                                         # moments replaced (what compress() does through AdamW.update)
            for p in ps:
                opt.state[p]['exp_avg'] = opt.state[p]['exp_avg'].clone()
        hits = opt.table_hits
        opt.step(); ropt.step()
        assert (opt.table_hits > hits) == (step in (1, 2, 4)), f'step {step}: table reuse'
        for p, r in zip(ps, ref):
            assert (p - r).abs().max().item() <= 2e-6 * r.abs().max().item(), f'step {step}'


def test_model_ema_fast_path_sees_replaced_and_re_pointed_parameters():
    """ModelEma.update re-validates the pairs of its last full walk instead of walking both state dicts: a parameter that was replaced
    (compress / adopt_state) or re-pointed (p.data = ...) must still be averaged from where it lives NOW"""
    import ofb_amd
    from ofb_amd.utils import ModelEma
    torch.manual_seed(0)
    m = ofb_amd.VisionTransformer(embed_dim=64, depth=2, num_heads=2, num_classes=5).cuda()
    ema = ModelEma(m, decay=0.5)
    want = {k: v.clone() for k, v in ema.ema.state_dict().items()}

    def step_and_check(tag):
        ema.update(m)
        for k, v in m.state_dict().items():
            want[k] = want[k] * 0.5 + (1. - 0.5) * v
        for k, v in ema.ema.state_dict().items():
            assert torch.equal(v, want[k]), (tag, k)

    step_and_check('full walk')
    assert ema._fast is not None and ema._fast[2] == ModelEma._REWALK
    with torch.no_grad():
        for p in m.parameters():
            p.add_(0.25)
    step_and_check('fast')
    assert ema._fast[2] == ModelEma._REWALK - 1               # the fast path ran
    m.head.weight = torch.nn.Parameter(m.head.weight.detach() * 2.0)          # replaced object
    step_and_check('replaced parameter')
    m.head.bias.data = m.head.bias.data + 1.0                                   # same object, new storage
    step_and_check('re-pointed parameter')


def test_run_backward_keeps_the_pass_on_the_calling_thread():
    """engine.run_backward: the Python backward functions of a step run on the thread that asked for the pass (torch's default hands a
    CUDA graph to a per-device worker thread); gradients are the same either way"""
    import threading
    from ofb_amd import engine
    seen = []

    class Probe(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x * 2.0

        @staticmethod
        def backward(ctx, g):
            seen.append(threading.get_ident())
            return g * 2.0

    x = torch.randn(8, device='cuda', requires_grad=True)
    engine.run_backward(Probe.apply(x).sum())
    g1 = x.grad.clone()
    x.grad = None
    Probe.apply(x).sum().backward()
    assert seen[0] == threading.get_ident() and (engine._BACKWARD_ON_CALLER is False or seen[1] != seen[0])
    assert torch.equal(g1, x.grad) and torch.equal(g1, torch.full_like(g1, 2.0))
