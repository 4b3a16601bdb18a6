"""Shared driver / checkers for the compress() life-cycle fixture (tests/golden/mini_c.npz): used by the CPU test that
pins the oracle and by the GPU test that checks the HIP-backed model against the same reference outputs."""
import os

import numpy as np
import torch

from oracle import fill
from oracle import ofb_oracle as O
from tests.golden_util import GOLDEN_DIR, rel_err, sample

MINI = dict(embed_dim=128, depth=3, num_heads=4, num_classes=10)


class Lifecycle:
    def __init__(self, dtype):
        self.z = z = np.load(os.path.join(GOLDEN_DIR, 'mini_c.npz'))
        batch, w_p, keep, dp, self.lr, self.thresh = [float(v) for v in z['meta']]
        batch = int(batch)
        self.cfg = O.Config(**MINI, drop_path_rate=dp)
        self.st = O.SearchState(w_p=w_p, keep_ratio=keep)
        self.st.frozen.add('alpha_patch')
        self.p = O.formula_params(self.cfg, dtype)
        self.opt = O.OptimState(self.p, frozen=self.st.frozen)
        self.imgs = torch.from_numpy(fill.images(batch)).to(dtype)
        self.labels = torch.from_numpy(fill.labels(batch, self.cfg.num_classes))
        self.pnoise = torch.from_numpy(fill.patch_noise(batch, self.cfg.num_patches)).to(dtype)
        self.dnoise = torch.from_numpy(fill.droppath_noise(2 * self.cfg.depth, batch)).to(dtype)

    def crafted(self, stage):
        pre = f'craft{stage}.'
        return {k[len(pre):]: torch.from_numpy(self.z[k]) for k in self.z.files if k.startswith(pre)}

    def craft(self, stage):
        for name, a in self.crafted(stage).items():
            assert tuple(self.p[name + '.alpha'].shape) == tuple(a.shape), name
            self.p[name + '.alpha'] = a.to(self.p[name + '.alpha'].dtype)

    def opt_view(self):
        o = self.opt
        return dict(names={'p.0': o.groups['p0'], 'p.1': o.groups['p1'], 'd.0': o.groups['d0'], 'd.1': o.groups['d1'],
                           'a.0': o.groups['a0']},
                    state={k: (v['step'], v['m'], v['v']) for k, v in o.state.items()})


def _small(k):
    return 'alpha' in k or 'score' in k


def check_step(z, pre, out, grads, params_after, tol, grad_tol=None):
    """losses / logits / gradients / post-AdamW parameters of one search step against fixture stage `pre`."""
    grad_tol = grad_tol or 50 * tol
    for k in ['base', 'arch', 'decoder_loss', 'loss_total', 'flops_total', 'flops_searched']:
        got, exp = float(torch.as_tensor(out[k]).detach()), float(z[f'{pre}.{k}'])
        assert abs(got - exp) <= tol * max(1.0, abs(exp)), (pre, k, got, exp)
    assert rel_err(out['logits'].detach().cpu(), z[f'{pre}.logits']) < 5 * tol, pre
    worst = 0.0
    for k, g in grads.items():
        if f'{pre}.gnorm.{k}' not in z.files:
            assert g is None or float(g.abs().max()) == 0.0, (pre, k)
            continue
        g = g.detach().cpu()
        gn = float(z[f'{pre}.gnorm.{k}'])
        assert abs(float(g.double().norm()) - gn) <= 20 * tol * max(gn, 1e-6), (pre, k, float(g.norm()), gn)
        e = rel_err(g, z[f'{pre}.grad.{k}']) if _small(k) else rel_err(sample(g), z[f'{pre}.gsamp.{k}'])
        worst = max(worst, e)
        assert e < grad_tol, (pre, k, e)
    for k, v in params_after.items():
        v = v.detach().cpu()
        got = v.reshape(-1) if _small(k) else sample(v)
        exp = torch.from_numpy(z[f'{pre}.after.{k}' if _small(k) else f'{pre}.asamp.{k}']).reshape(-1)
        assert got.shape == exp.shape, (pre, k, got.shape, exp.shape)
        g = grads.get(k)
        if g is None:
            assert float((got.double() - exp.double()).abs().max()) < 1e-6, (pre, k)
            continue
        g = g.detach().cpu()
        gsel = g.reshape(-1) if _small(k) else sample(g)
        # Adam turns rounding-noise gradients into +-lr moves: compare where the gradient is well above fp32 noise
        ok = gsel.abs() > 1e-4 * float(g.abs().max()) + 1e-9
        if k.endswith('qkv.bias') or not bool(ok.any()):
            continue
        assert float((got[ok].double() - exp[ok].double()).abs().max()) < 3e-5 + 100 * tol * 1e-3, (pre, k)
    return worst


def check_snapshot(z, pre, cfg, params, st, opt):
    """state right after a compress(): cell switches, flags, shapes, alpha/score values, weights, optimizer lists and
    moments.  `params`: name -> tensor; `st`: object with .switch/.finished/.execute/.heads dicts; `opt`: dict(names, state)."""
    for name in O.module_names(cfg):
        sw = st.switch.get(name)
        exp = z[f'{pre}.switch.{name}']
        assert sw is not None and np.array_equal(np.asarray(sw.cpu()).astype(bool), exp), (pre, name)
        fin, ex, hn = z[f'{pre}.flags.{name}'].tolist()
        assert bool(st.finished.get(name, False)) == bool(fin) and bool(st.execute.get(name, False)) == bool(ex), (pre, name)
        if hn >= 0:
            assert st.heads.get(name) == hn, (pre, name, st.heads.get(name), hn)
    for k, v in params.items():
        v = v.detach().cpu()
        assert list(v.shape) == z[f'{pre}.shape.{k}'].tolist(), (pre, k, tuple(v.shape))
        if _small(k):
            assert float((v.double() - torch.from_numpy(z[f'{pre}.val.{k}']).double()).abs().max()) < 2e-5, (pre, k)
        elif not k.endswith('qkv.bias'):
            assert float((sample(v).double() - torch.from_numpy(z[f'{pre}.vsamp.{k}']).double()).abs().max()) < 1e-4, (pre, k)
    for g, names in opt['names'].items():
        key = f'{pre}.optnames.{g}'
        exp = [str(s) for s in z[key]] if key in z.files else []
        assert list(names) == exp, (pre, g, [n for n in names if n not in exp], [n for n in exp if n not in names])
    for k, (step, m, v) in opt['state'].items():
        assert int(step) == int(z[f'{pre}.optstep.{k}']), (pre, k)
        m, v = m.detach().cpu(), v.detach().cpu()
        gm, gv = (m, v) if _small(k) else (sample(m), sample(v))
        em, ev = z[f'{pre}.m.{k}'], z[f'{pre}.v.{k}']
        assert tuple(gm.reshape(-1).shape) == tuple(em.reshape(-1).shape), (pre, k)
        if float(np.abs(em).max()) == 0.0:
            assert float(gm.abs().max()) == 0.0 and float(gv.abs().max()) == 0.0, (pre, k)
        else:
            assert rel_err(gm, em) < 2e-3 and rel_err(gv, ev) < 2e-3, (pre, k, rel_err(gm, em), rel_err(gv, ev))
    n_state = sum(1 for f in z.files if f.startswith(f'{pre}.optstep.'))
    assert n_state == len(opt['state']), (pre, n_state, len(opt['state']))
