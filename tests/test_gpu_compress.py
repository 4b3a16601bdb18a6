"""compress() life cycle on the HIP-backed model against the reference's own run (tests/golden/mini_c.npz):
step -> crafted alphas -> compress -> step -> crafted alphas -> compress (finishes) -> step -> eval -> fuse -> eval.
Checks, per stage, the cell switches, flags, every parameter shape / value, the three optimizers' parameter lists and
AdamW moments, then losses / FLOPs / gradients / updated parameters of the following search step."""
import types

import pytest
import torch

from oracle import ofb_oracle as O
from tests.compress_util import Lifecycle, check_snapshot, check_step
from tests.test_gpu_model import build_product

pytestmark = pytest.mark.gpu


def _state_view(m, cfg):
    st = types.SimpleNamespace(switch={}, finished={}, execute={}, heads={})
    for mod, name in zip(m.searchable_modules, O.module_names(cfg)):
        st.switch[name] = mod.switch_cell
        st.finished[name], st.execute[name] = mod.finish_search, mod.execute_prune
        if hasattr(mod, 'head_num'):
            st.heads[name] = mod.head_num
    return st


def _opt_view(m, opts):
    name_of = {id(p): k for k, p in m.named_parameters()}
    names, state = {}, {}
    for tag, o in opts.items():
        for gi in (0, 1):
            if o is not None and gi < len(o.param_groups):
                got = [name_of[id(p)] for p in o.param_groups[gi]['params']]
                assert got == list(o.param_names[gi]), (tag, gi)
                names[f'{tag}.{gi}'] = got
                for p in o.param_groups[gi]['params']:
                    s = o.state.get(p)
                    if s:
                        assert s['exp_avg'].shape == p.shape and s['exp_avg'].is_contiguous()
                        state[name_of[id(p)]] = (s['step'], s['exp_avg'], s['exp_avg_sq'])
            elif gi == 0 or tag != 'a':
                names[f'{tag}.{gi}'] = []
    return dict(names=names, state=state)


def test_compress_lifecycle_matches_reference():
    from ofb_amd.engine import build_optimizers
    from ofb_amd.losses import OFBSearchLOSS, DistillationLoss, LabelSmoothingCrossEntropy
    lc = Lifecycle(torch.float32)
    z, cfg = lc.z, lc.cfg
    inputs = dict(patch_noise=lc.pnoise, droppath_u=lc.dnoise)
    m = build_product(cfg, lc.st, inputs)
    opt_p, opt_a, opt_d = build_optimizers(m, lr=lc.lr, weight_decay=1e-3)
    opts = {'p': opt_p, 'd': opt_d, 'a': opt_a}
    crit = OFBSearchLOSS(DistillationLoss(LabelSmoothingCrossEntropy(0.1), None, 'none', 0.5, 1.0), torch.device('cuda'),
                         attn_w=0.5, mlp_w=0.5, patch_w=0.0, embedding_w=0.5, flops_w=5.0)
    imgs, labels = lc.imgs.cuda(), lc.labels.cuda()

    def step(finish):
        m.train()
        for p in m.parameters():
            p.grad = None
        logits, (dec, _) = m(imgs)
        loss = crit(imgs, logits, labels, m, 'arch', 1.0, finish)
        base, arch = loss if isinstance(loss, tuple) else (loss, torch.zeros((), device='cuda'))
        total = base + arch + (base / dec).detach() * dec
        total.backward()
        tot, sea = m.get_flops()
        grads = {k: (None if p.grad is None else p.grad.detach().clone()) for k, p in m.named_parameters()}
        for o in opts.values():
            if o is not None:
                o.step()
        torch.cuda.synchronize()
        out = dict(logits=logits, decoder_loss=dec.detach(), base=base.detach(), arch=arch.detach(), loss_total=total.detach(),
                   flops_total=tot, flops_searched=sea)
        return out, grads

    out, grads = step(False)
    print('s0 worst grad rel err', check_step(z, 's0', out, grads, dict(m.named_parameters()), tol=1e-4, grad_tol=3e-3))
    for stage in (1, 2):
        for name, a in lc.crafted(stage).items():
            mod = dict(zip(O.module_names(cfg), m.searchable_modules))[name]
            assert tuple(mod.alpha.shape) == tuple(a.shape), name
            mod.alpha.data.copy_(a)
        fin, ex, opts['p'], opts['d'], opts['a'] = m.compress(lc.thresh, opts['p'], opts['d'], opts['a'])
        assert [int(fin), int(ex)] == z[f'c{stage}.model_flags'].tolist()
        check_snapshot(z, f'c{stage}', cfg, dict(m.named_parameters()), _state_view(m, cfg), _opt_view(m, opts))
        if fin:
            opts['a'] = None
        out, grads = step(bool(fin))
        print(f's{stage} worst grad rel err', check_step(z, f's{stage}', out, grads, dict(m.named_parameters()), tol=1e-3, grad_tol=3e-3))
    assert fin and m.finish_search
    m.eval()
    with torch.no_grad():
        lg = m(imgs)[0]
        assert float((lg.cpu() - torch.from_numpy(z['eval.logits_prefuse'])).abs().max()) < 2e-3
        m.fuse()
        lg2 = m(imgs)[0]
        assert float((lg2.cpu() - torch.from_numpy(z['eval.logits_fused'])).abs().max()) < 2e-3
        assert float((lg2 - lg).abs().max()) < 1e-4


def test_index_select_kernel():
    from ofb_amd import hip
    torch.manual_seed(0)
    t = torch.randn(5, 37, 12, device='cuda')
    for dim, n in ((0, 3), (1, 20), (2, 7), (-1, 12)):
        idx = torch.randperm(t.shape[dim])[:n]
        got = hip.index_select(t, idx, dim)
        assert got.is_contiguous() and torch.equal(got, t.index_select(dim % 3, idx.cuda()))
    with pytest.raises(hip.OfbError):
        hip.index_select(t, torch.tensor([0, 37]), 1)
    with pytest.raises(hip.OfbError):
        hip.index_select(t, torch.tensor([0, 37], device='cuda'), 1)


def _load_reference_fused_checkpoint():
    import gzip
    import io
    import os
    from ofb_amd import utils
    from tests.golden_util import GOLDEN_DIR
    utils.install_reference_aliases()
    with gzip.open(os.path.join(GOLDEN_DIR, 'mini_c_model_fused.pth.gz'), 'rb') as f:
        return torch.load(io.BytesIO(f.read()), map_location='cpu', weights_only=False)


def test_reference_whole_object_checkpoint_runs_on_hip():
    """a `model_fused.pth` written by the reference (class paths models.*) unpickles into this package's classes and its
    eval forward on the HIP kernels reproduces the reference's logits."""
    import ofb_amd
    lc = Lifecycle(torch.float32)
    m = _load_reference_fused_checkpoint()
    assert isinstance(m, ofb_amd.MIMVisionTransformer) and m.finish_search and m.fused
    m.cuda().eval()
    with torch.no_grad():
        lg = m(lc.imgs.cuda())[0]
    assert float((lg.cpu() - torch.from_numpy(lc.z['eval.logits_fused'])).abs().max()) < 2e-3
    tot, sea = m.get_flops()
    assert abs(float(sea) - float(lc.z['s2.flops_searched'])) < 1e-6 and abs(tot - float(lc.z['s2.flops_total'])) < 1e-6


def test_intersect_loads_searched_model_into_finetune_vit():
    """finetune.py:182-249: the plain ViT adopts the cut shapes / head counts of the searched model; with the gates
    already folded (fused) both models are the same function."""
    import ofb_amd
    from ofb_amd import utils
    lc = Lifecycle(torch.float32)
    src = _load_reference_fused_checkpoint().cuda().eval()
    ft = ofb_amd.VisionTransformer(embed_dim=128, depth=3, num_heads=4, num_classes=10, drop_path_rate=0.1).cuda()
    utils.intersect(ft, src)
    assert [b.attn.num_heads for b in ft.blocks] == [2, 4, 2]
    assert ft.blocks[1].mlp.fc1.out_features == 128 and ft.norm.normalized_shape[0] == 100 and ft.head.in_features == 100
    ft.eval()
    with torch.no_grad():
        lg = ft(lc.imgs.cuda())
    assert float((lg.cpu() - torch.from_numpy(lc.z['eval.logits_fused'])).abs().max()) < 2e-3
    ft2 = ofb_amd.VisionTransformer(embed_dim=128, depth=3, num_heads=4, num_classes=7).cuda()
    utils.intersect(ft2, src, exclude=['head'])
    assert tuple(ft2.head.weight.shape) == (7, 100)
    assert tuple(ft2(lc.imgs.cuda()).shape) == (2, 7)


def test_model_ema_fused_update_and_reshape():
    from ofb_amd.utils import ModelEma
    from tests.golden_util import load_case
    z, cfg, st, inputs, lr = load_case('micro_a')
    m = build_product(cfg, st, inputs)
    ema = ModelEma(m, decay=0.99)
    assert not any(p.requires_grad for p in ema.ema.parameters())
    before = {k: v.clone() for k, v in ema.ema.state_dict().items()}
    torch.manual_seed(1)
    with torch.no_grad():
        for p in m.parameters():
            p.add_(torch.randn_like(p) * 0.01)
    ema.update(m)
    ema.update(m)                                                     # cached table path
    msd = m.state_dict()
    for k, v in ema.ema.state_dict().items():
        e1 = before[k] * 0.99 + (1. - 0.99) * msd[k]
        e2 = e1 * 0.99 + (1. - 0.99) * msd[k]
        assert torch.equal(v, e2), k                                  # bit-identical to the reference's expression
    # a compress() changes shapes: the EMA copy adopts the new tensors (utils.py:442-447)
    mods = dict(zip(O.module_names(cfg), m.searchable_modules))
    a = torch.full((1, 7), -6.0)
    a[0, 2] = 0.0
    mods['blocks.0.mlp'].alpha.data.copy_(a)
    m.compress(0.2)
    ema.update(m)
    esd = ema.ema.state_dict()
    assert tuple(esd['blocks.0.mlp.fc1.weight'].shape) == tuple(m.blocks[0].mlp.fc1.weight.shape) == (128, 64)
    assert torch.equal(esd['blocks.0.mlp.fc1.weight'], m.blocks[0].mlp.fc1.weight.data)
    assert ema.ema.blocks[0].mlp.fc1.out_features == 128
    ema.update(m)


def test_patch_cell_compress_lifecycle_matches_reference():
    """patch-number search driven through compress() (reference vision_transformer.py:789-820; tests/golden/micro_pc.npz from the
    reference's own run): two patch cells die, then one is left (finished, alpha_patch frozen).  After the first patch-cell cut
    the FLOPs model counts the probability-weighted patch number (:768) and alpha_patch receives its gradient."""
    from ofb_amd.engine import build_optimizers
    from ofb_amd.losses import OFBSearchLOSS, DistillationLoss, LabelSmoothingCrossEntropy
    from tests.patch_compress_util import load, check_step as pc_step, check_snapshot as pc_snap
    z, cfg, c = load()
    st = O.SearchState(w_p=c['w_p'])
    inputs = dict(patch_noise=c['pnoise'], droppath_u=torch.zeros(2 * cfg.depth, c['batch']))
    m = build_product(cfg, st, inputs)
    opt_p, opt_a, opt_d = build_optimizers(m, lr=c['lr'], weight_decay=1e-3)
    opts = {'p': opt_p, 'd': opt_d, 'a': opt_a}
    crit = OFBSearchLOSS(DistillationLoss(LabelSmoothingCrossEntropy(0.1), None, 'none', 0.5, 1.0), torch.device('cuda'),
                         attn_w=0.5, mlp_w=0.5, patch_w=0.5, embedding_w=0.5, flops_w=5.0)
    imgs, labels = c['imgs'].cuda(), c['labels'].cuda()

    def step():
        m.train()
        for p in m.parameters():
            p.grad = None
        logits, (dec, _) = m(imgs)
        base, arch = crit(imgs, logits, labels, m, 'arch', 1.0, False)
        total = base + arch + (base / dec).detach() * dec
        total.backward()
        lp = m.get_sparsity_loss(torch.device('cuda'))[2]
        grads = {k: (None if p.grad is None else p.grad.detach().clone()) for k, p in m.named_parameters()}
        for o in opts.values():
            if o is not None:
                o.step()
        torch.cuda.synchronize()
        return dict(logits=logits, decoder_loss=dec.detach(), base=base.detach(), arch=arch.detach(), loss_total=total.detach(),
                    loss_patch=lp.detach()), grads

    out, grads = step()
    pc_step(z, 's0', out, grads, m.alpha_patch, tol=1e-4)
    names = O.module_names(cfg)
    for stage in (1, 2):
        m.alpha_patch.data.copy_(torch.from_numpy(z[f'craft{stage}.alpha_patch']))
        fin, ex, opts['p'], opts['d'], opts['a'] = m.compress(c['thresh'], opts['p'], opts['d'], opts['a'])
        pc_snap(z, f'c{stage}', fin, ex, m.switch_cell_patch, m.alpha_patch, m.alpha_patch.requires_grad, m.weighted_mask,
                {n: mod.switch_cell for n, mod in zip(names, m.searchable_modules)}, {k: tuple(p.shape) for k, p in m.named_parameters()})
        out, grads = step()
        pc_step(z, f's{stage}', out, grads, m.alpha_patch, tol=1e-4)
    assert int(m.switch_cell_patch.sum()) == 1 and not m.alpha_patch.requires_grad
