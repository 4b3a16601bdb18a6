"""Pin the oracle's restatement of the epoch protocol (oracle.ofb_oracle.search_epoch: schedules, gradient accumulation, three
optimizers with per-step learning rates, meters, compress trigger) against the reference's own run of engine.search_one_epoch
(tests/golden/mini_epoch.npz from tests/golden/make_golden_epoch.py).  CPU only."""
import numpy as np
import torch

from oracle import ofb_oracle as O
from tests import epoch_util as E
from tests.golden_util import rel_err, sample


def test_search_epoch_matches_reference():
    z = E.load()
    cfg = O.Config(**E.MINI, drop_path_rate=0.0)
    st = O.SearchState()
    st.frozen.add('alpha_patch')
    p = O.formula_params(cfg, torch.float64)
    opt = O.OptimState(p, frozen=st.frozen)

    def hook(i):
        stage = E.CRAFT_AT.get(i)
        if stage:
            for name, a in E.crafted(z, stage).items():
                assert tuple(p[name + '.alpha'].shape) == tuple(a.shape), name
                p[name + '.alpha'] = a.double()

    stats, fin, pruned, per_it = O.search_epoch(
        cfg, p, st, opt, E.N_ITER, lambda i: (lambda b: (b[0].double(), b[1]))(E.batch_of(i, cfg.num_classes)),
        lambda i: E.noise_of(i, cfg.num_patches).double(), epoch=0, accum_iter=E.ACCUM, warmup_epochs=E.WARMUP_EPOCHS, lr=E.LR0,
        lr_sched=E.lr_at, hook=hook)
    assert [int(fin), int(pruned)] == z['flags'].tolist()
    for k in ('base', 'arch', 'dec', 'keep_ratio', 'w_p', 'finish'):
        got, exp = np.array(per_it[k]), z[f'it.{k}']
        if k == 'w_p':
            got, exp = got[exp >= 0], exp[exp >= 0]          # (-1 in the fixture: no module left searching)
        assert np.allclose(got, exp, rtol=2e-4, atol=1e-9), (k, got, exp)
    E.check_stats(z, stats, 2e-4)
    # everything the epoch left behind: shapes, switches, alphas / scores, sampled weights, optimizer parameter lists
    for k, v in p.items():
        assert tuple(v.shape) == tuple(z[f'shape.{k}']), k
        assert bool(z[f'rg.{k}']) == (k not in st.frozen), k
        if 'alpha' in k or 'score' in k:
            assert rel_err(v, z[f'val.{k}']) < 2e-4, k
        elif not k.endswith('qkv.bias'):                   # (the k third of a qkv bias has a zero gradient: Adam turns its rounding noise into +-lr moves)
            assert float((sample(v).double() - torch.from_numpy(z[f'vsamp.{k}']).double()).abs().max()) < 1e-4, k
    for name in O.module_names(cfg):
        assert np.array_equal(st.cell_mask(name, p[name + '.alpha']).numpy(), z[f'switch.{name}']), name
        fl = z[f'flags.{name}']
        assert bool(fl[0]) == st.finished.get(name, False), name
    for tag, g in (('p.0', 'p0'), ('p.1', 'p1'), ('d.0', 'd0'), ('d.1', 'd1')):
        assert opt.groups[g] == list(z[f'optnames.{tag}']), tag
