"""Pin the CPU oracle against golden vectors produced by the imported reference.

Runs on CPU (-m "not gpu").  Tolerances: the reference computes in fp32, the oracle in fp32 or fp64;
differences are pure fp32 rounding/summation-order effects.
"""
import numpy as np
import pytest
import torch

from oracle import ofb_oracle as O
from tests.golden_util import load_case, sample, rel_err, GOLDEN_DIR


def run_oracle(tag, dtype):
    z, cfg, st, inp, lr = load_case(tag)
    from tests.golden_util import CASES
    p = {k: v.requires_grad_(True) for k, v in O.formula_params(cfg, dtype).items()}
    patch_w = CASES[tag].get('patch_w', 0.0)
    p['alpha_patch'].requires_grad_(patch_w != 0.0)              # correct_require_grad(w_patch) freezes it otherwise
    out = O.search_step_loss(cfg, p, st, inp['imgs'].to(dtype), inp['labels'], inp['patch_noise'].to(dtype),
                             inp['droppath_u'].to(dtype), w=(0.5, 0.5, patch_w, 0.5, 5.0))
    out['loss_total'].backward()
    return z, cfg, st, p, out, lr


@pytest.mark.parametrize('tag,dtype', [('micro_a', torch.float64), ('micro_a', torch.float32), ('micro_b', torch.float64),
                                       ('tiny_a', torch.float64), ('small_a', torch.float32), ('micro_h', torch.float64),
                                       ('micro_c', torch.float64), ('micro_p', torch.float64)])
def test_search_step_matches_reference(tag, dtype):
    z, cfg, st, p, out, lr = run_oracle(tag, dtype)
    tol = 2e-5 if dtype == torch.float64 else 1e-4
    out['loss_patch'] = O.sparsity_losses(cfg, p, st, out['gates'])[2]
    for k in ['base', 'arch', 'loss_attn', 'loss_mlp', 'loss_embed', 'decoder_loss', 'flops_total', 'flops_searched', 'loss_total'] + \
            (['loss_patch'] if 'loss_patch' in z.files else []):
        got, exp = float(out[k]), float(z[k])
        assert abs(got - exp) <= tol * max(1.0, abs(exp)), (k, got, exp)
    assert rel_err(out['logits'].detach(), z['logits']) < tol * 5
    for name in O.module_names(cfg):
        g, wr, wm, _ = out['gates'][name]
        for nm, t in (('g', g), ('wr', wr), ('wm', wm)):
            assert rel_err(t.detach(), z[f'gate.{name}.{nm}']) < 1e-6, (name, nm)
    # gradients: full for alpha/score (and small tensors), strided samples + norms for the rest
    worst = 0.0
    for k, v in p.items():
        if f'gnorm.{k}' not in z.files:
            assert v.grad is None or float(v.grad.abs().max()) == 0.0 or k == 'alpha_patch', k
            continue
        gn = float(z[f'gnorm.{k}'])
        assert abs(float(v.grad.double().norm()) - gn) <= 20 * tol * max(gn, 1e-6), (k, float(v.grad.norm()), gn)
        if f'grad.{k}' in z.files:
            e = rel_err(v.grad, z[f'grad.{k}'])
        else:
            e = rel_err(sample(v.grad), z[f'gsamp.{k}'])
        worst = max(worst, e)
        assert e < 50 * tol, (k, e)
    print(f'{tag} {dtype}: worst grad rel err {worst:.2e}')


@pytest.mark.parametrize('tag', ['micro_a', 'micro_b'])
def test_adamw_step_matches_reference(tag):
    z, cfg, st, p, out, lr = run_oracle(tag, torch.float64)
    hyper = {'nodecay': (0.9, 0.0), 'decay': (0.9, 1e-3), 'decoder_nodecay': (0.9, 0.0), 'decoder_decay': (0.9, 1e-3),
             'arch': (0.5, 1e-3)}
    listed = {}
    for line in z['groups']:
        g, names = str(line).split(':')
        for n in names.split(','):
            if n:
                listed[n] = g
    for k, v in p.items():
        if not v.requires_grad:
            continue
        grp = O.optimizer_group(k, tuple(v.shape))
        assert listed[k] == grp, (k, grp, listed[k])           # grouping rule matches search.py:486-508
        b1, wd = hyper[grp]
        new, _, _ = O.adamw_step(v.detach(), v.grad, torch.zeros_like(v), torch.zeros_like(v), 1, lr, b1, 0.999, 1e-8, wd)
        # Adam turns gradients that are pure rounding noise (e.g. the key bias, whose true gradient is 0 by
        # softmax shift-invariance) into +-lr steps, so compare only where the gradient is well above fp32 noise.
        if f'after.{k}' in z.files:
            got, exp, gsel = new.reshape(-1), torch.from_numpy(z[f'after.{k}']).reshape(-1), v.grad.reshape(-1)
        else:
            got, exp, gsel = sample(new), torch.from_numpy(z[f'asamp.{k}']), sample(v.grad)
        ok = gsel.abs() > 1e-5 * float(v.grad.abs().max()) + 1e-9
        assert float(ok.float().mean()) > 0.5 or 'qkv.bias' in k, k
        assert float((got[ok] - exp[ok].double()).abs().max()) < 2e-6, k


def test_norm_targets_matches_reference():
    from oracle import fill
    z = np.load(f'{GOLDEN_DIR}/norm_targets.npz')
    imgs = torch.from_numpy(fill.images(1, tag='nt'))
    imgs[0, 2, :40, :] = 0.25
    t = O.norm_targets(imgs, 47)
    got = t[0][:, z['rows'], :]
    exp = torch.from_numpy(z['out'])
    # flat regions divide fp32 rounding noise by sqrt(1e-6): compare where the reference variance is sane
    err = (got - exp).abs()
    assert float(err[:2].max()) < 2e-4
    assert float(err.median()) < 1e-5


def test_keep_mask_counts():
    from oracle import fill
    n = torch.from_numpy(fill.patch_noise(4))
    for keep in (0.95, 0.75):
        m = O.keep_mask_from_noise(n, int(196 * keep))
        assert m.sum(1).tolist() == [196 - int(196 * keep)] * 4


def test_chunked_oracle_step_equals_the_one_shot_step():
    """tests/fullsize_util.chunked_oracle_step (the fp64 oracle over a batch it cannot hold at once: the bs-128 parity test of
    tests/test_gpu_fullsize.py) is the same function of the parameters as O.search_step_loss: every gradient to fp64 rounding."""
    from oracle import fill
    from tests.fullsize_util import chunked_oracle_step
    cfg = O.Config(**O.MICRO, drop_path_rate=0.1)
    st = O.SearchState(w_p=0.8, keep_ratio=0.85)
    B = 5
    imgs, labels = torch.from_numpy(fill.images(B)).double(), torch.from_numpy(fill.labels(B, cfg.num_classes))
    noise, u = torch.from_numpy(fill.patch_noise(B, cfg.num_patches)).double(), torch.from_numpy(fill.droppath_noise(2 * cfg.depth, B)).double()

    def params():
        p = {k: v.requires_grad_(True) for k, v in O.formula_params(cfg, torch.float64).items()}
        p['alpha_patch'].requires_grad_(False)
        return p

    p1 = params()
    ref = O.search_step_loss(cfg, p1, st, imgs, labels, noise, u)
    ref['loss_total'].backward()
    p2 = params()
    got = chunked_oracle_step(cfg, p2, st, imgs, labels, noise, u, chunk=2)
    for k in ('base', 'arch', 'decoder_loss', 'loss_total'):
        assert abs(float(got[k]) - float(ref[k].detach())) < 1e-12 * max(1.0, abs(float(ref[k].detach()))), k
    assert float((got['logits'] - ref['logits'].detach()).abs().max()) < 1e-12
    for k in p1:
        if p1[k].grad is None:
            assert p2[k].grad is None, k
            continue
        assert float((p1[k].grad - p2[k].grad).abs().max()) <= 1e-11 * max(1e-30, float(p1[k].grad.abs().max())), k
