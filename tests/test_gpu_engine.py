"""BASELINE config 1 plumbing: DeiT-Tiny OFB search, 2-class synthetic subset, bs 8, driven by the epoch engine
(reference engine.search_one_epoch protocol: progressive masking ratio, w_p warm-up, 3 optimizers, lr schedulers),
plus the finetune engine on a plain ViT."""
import types

import pytest
import torch

pytestmark = pytest.mark.gpu


class _Sched:
    def __init__(self):
        self.calls = []

    def step_update(self, it):
        self.calls.append(it)


def _loader(n_iter, bs, ncls, seed=0):
    g = torch.Generator().manual_seed(seed)
    return [(torch.randn(bs, 3, 224, 224, generator=g), torch.randint(0, ncls, (bs,), generator=g)) for _ in range(n_iter)]


def test_search_one_epoch_deit_tiny():
    import ofb_amd
    from ofb_amd import engine
    from ofb_amd.losses import OFBSearchLOSS, DistillationLoss, LabelSmoothingCrossEntropy
    torch.manual_seed(0)
    dev = torch.device('cuda')
    m = ofb_amd.create_model('deit_tiny_patch16_224_mim', method='search', num_classes=2, drop_path_rate=0.1, patch_search=False,
                             mask_ratio=1.0).to(dev)
    m.correct_require_grad(0.5, 0.5, 0, 0.5)
    opt_p, opt_a, opt_d = engine.build_optimizers(m, 2.5e-4 * 8 / 256)
    crit = OFBSearchLOSS(DistillationLoss(LabelSmoothingCrossEntropy(0.1), None, 'none', 0.5, 1.0), dev, 0.5, 0.5, 0.0, 0.5, 5.0)
    args = types.SimpleNamespace(accum_iter=2, warmup_epochs=20, epochs=100)
    before = {k: v.detach().clone() for k, v in m.named_parameters()}
    scheds = [_Sched(), _Sched(), _Sched()]
    stats, finish, pruned, *_ = engine.search_one_epoch(m, crit, 1.0, _loader(6, 8, 2), opt_p, opt_d, opt_a, scheds[0], scheds[1],
                                                        scheds[2], dev, epoch=3, args=args, print_freq=3)
    torch.cuda.synchronize()
    assert not finish and not pruned
    assert all(torch.isfinite(torch.tensor(v)) for v in stats.values()), stats
    assert scheds[0].calls == [3 * 6 + 1, 3 * 6 + 3, 3 * 6 + 5]                    # one scheduler step per accumulation boundary
    # schedules followed the reference formulas at t = epoch + it/len (engine.py:102-117)
    t_last = 3 + 4 / 6
    assert abs(m.blocks[0].attn.w_p - (0.99 - 0.89 * t_last / 20)) < 1e-9
    assert abs(m.patch_ratio_list[0] - (0.95 - 0.2 * t_last / 20)) < 1e-9
    changed = [k for k, v in m.named_parameters() if v.requires_grad and not torch.equal(v.detach(), before[k])]
    assert len(changed) == sum(1 for _, v in m.named_parameters() if v.requires_grad)       # every trainable tensor stepped
    assert torch.equal(m.alpha_patch.detach(), before['alpha_patch'])


def test_train_one_epoch_finetune_vit():
    import ofb_amd
    from ofb_amd import engine
    from ofb_amd.optim import AdamW
    from ofb_amd.losses import DistillationLoss, LabelSmoothingCrossEntropy
    torch.manual_seed(0)
    dev = torch.device('cuda')
    m = ofb_amd.VisionTransformer(embed_dim=192, depth=3, num_heads=3, num_classes=5, drop_path_rate=0.1).to(dev)
    torch.nn.init.normal_(m.head.weight, std=0.02)
    opt = AdamW(m.parameters(), None, lr=1e-3, weight_decay=0.05)
    crit = DistillationLoss(LabelSmoothingCrossEntropy(0.1), None, 'none', 0.5, 1.0)
    args = types.SimpleNamespace(accum_iter=1)
    data = _loader(1, 8, 5) * 8                                            # the same batch 8 times: the loss must go down
    sched = _Sched()
    first = engine.train_one_epoch(m, crit, data[:1], opt, sched, dev, 0, args=args, set_training_mode=False)
    last = engine.train_one_epoch(m, crit, data, opt, sched, dev, 1, args=args, set_training_mode=False)
    assert last['loss'] < first['loss'], (first, last)
    ev = engine.evaluate(data[:2], m, dev)
    assert 0.0 <= ev['acc1'] <= ev['acc5'] <= 100.0 and ev['loss'] > 0


def test_train_one_epoch_with_mixup_soft_targets():
    """finetune.py's default recipe: Mixup(0.8, 1.0) + SoftTargetCrossEntropy (finetune.py:310,390-393) through the engine; the
    first step's loss is checked against the CPU oracle's Mixup + soft-target CE on the same logits."""
    import numpy as np
    import ofb_amd
    from ofb_amd import engine
    from ofb_amd.optim import AdamW
    from ofb_amd.losses import DistillationLoss
    from oracle import data_oracle as DO
    torch.manual_seed(0)
    dev = torch.device('cuda')
    m = ofb_amd.VisionTransformer(embed_dim=192, depth=2, num_heads=3, num_classes=5, drop_path_rate=0.0).to(dev)
    torch.nn.init.normal_(m.head.weight, std=0.02)
    data = _loader(1, 8, 5)
    # oracle side: mix on the CPU, run the HIP model on the mixed batch, soft CE in fp64
    np.random.seed(7)
    xr, tr = DO.Mixup(0.8, 1.0, num_classes=5)(data[0][0].clone(), data[0][1])
    m.eval()
    with torch.no_grad():
        ref = float(DO.soft_target_cross_entropy(m(xr.to(dev)).double().cpu(), tr.double()))
    opt = AdamW(m.parameters(), None, lr=1e-3, weight_decay=0.05)
    crit = DistillationLoss(ofb_amd.SoftTargetCrossEntropy(), None, 'none', 0.5, 1.0)
    args = types.SimpleNamespace(accum_iter=1)
    np.random.seed(7)
    first = engine.train_one_epoch(m, crit, data, opt, _Sched(), dev, 0, mixup_fn=ofb_amd.Mixup(0.8, 1.0, num_classes=5), args=args,
                                   set_training_mode=False)
    assert abs(first['loss'] - ref) <= 1e-4 * abs(ref), (first, ref)          # north_star tolerance 1e-3; measured ~1e-6
    np.random.seed(7)
    fixed = [(data[0][0].clone(), data[0][1])] * 8
    mix = ofb_amd.Mixup(0.8, 1.0, prob=0.0, num_classes=5)                     # lam = 1: pure label smoothing targets, a fixed objective
    a = engine.train_one_epoch(m, crit, fixed[:1], opt, _Sched(), dev, 1, mixup_fn=mix, args=args, set_training_mode=False)
    b = engine.train_one_epoch(m, crit, fixed, opt, _Sched(), dev, 2, mixup_fn=mix, args=args, set_training_mode=False)
    assert b['loss'] < a['loss'], (a, b)


def test_search_epoch_fed_by_device_loader():
    """the whole input side in front of the engine: decoded uint8 images -> DeviceTransform (crop / flip / RandAugment / normalize /
    erase on the GPU, prefetched on a side stream by DeviceLoader) -> search_one_epoch."""
    import random
    import numpy as np
    import ofb_amd
    from ofb_amd import engine
    from ofb_amd.losses import OFBSearchLOSS, DistillationLoss, LabelSmoothingCrossEntropy
    torch.manual_seed(0); random.seed(0); np.random.seed(0)
    dev = torch.device('cuda')
    rng = np.random.default_rng(1)
    batches = [([rng.integers(0, 256, size=(int(rng.integers(240, 320)), int(rng.integers(240, 320)), 3), dtype=np.uint8) for _ in range(4)],
                rng.integers(0, 2, size=4).tolist()) for _ in range(3)]
    loader = ofb_amd.DeviceLoader(batches, ofb_amd.DeviceTransform(224, True, 'bicubic', auto_augment='rand-m9-mstd0.5-inc1', re_prob=0.25))
    loader.__len__ = lambda: 3
    m = ofb_amd.create_model('deit_tiny_patch16_224_mim', method='search', num_classes=2, drop_path_rate=0.1, patch_search=False,
                             mask_ratio=1.0).to(dev)
    m.correct_require_grad(0.5, 0.5, 0, 0.5)
    opt_p, opt_a, opt_d = engine.build_optimizers(m, 2.5e-4 * 4 / 256)
    crit = OFBSearchLOSS(DistillationLoss(LabelSmoothingCrossEntropy(0.1), None, 'none', 0.5, 1.0), dev, 0.5, 0.5, 0.0, 0.5, 5.0)
    args = types.SimpleNamespace(accum_iter=1, warmup_epochs=20, epochs=100)

    class _Sized:
        def __init__(self, it, n):
            self.it, self.n = it, n

        def __iter__(self):
            return iter(self.it)

        def __len__(self):
            return self.n

    stats, *_ = engine.search_one_epoch(m, crit, 1.0, _Sized(loader, 3), opt_p, opt_d, opt_a, _Sched(), _Sched(), _Sched(), dev, epoch=0,
                                        args=args, print_freq=1)
    torch.cuda.synchronize()
    assert all(torch.isfinite(torch.tensor(v)) for v in stats.values()), stats


def test_graphed_step_reproduces_eager_steps():
    """engine.GraphedStep: the whole search step (forward, losses, backward, the three AdamW steps) captured into a hipGraph and
    replayed must walk the parameters exactly like eager steps do (forced mask / DropPath noise: no random draws; AdamW's learning
    rate and bias corrections come from device memory inside the graph and are refreshed per replay)."""
    import torch
    from ofb_amd import engine
    from ofb_amd.losses import OFBSearchLOSS, DistillationLoss, LabelSmoothingCrossEntropy
    from tests.golden_util import load_case
    from tests.test_gpu_model import build_product
    z, cfg, st, inputs, lr = load_case('micro_a')
    prev = torch.cuda.current_stream()
    torch.cuda.set_stream(torch.cuda.Stream())               # the job lives on one non-default stream (see GraphedStep.capture)
    try:
        results = []
        for graphed in (False, False, True):
            m = build_product(cfg, st, inputs)
            opts = engine.build_optimizers(m, 1e-3)
            crit = OFBSearchLOSS(DistillationLoss(LabelSmoothingCrossEntropy(0.1), None, 'none', 0.5, 1.0), torch.device('cuda'),
                                 attn_w=0.5, mlp_w=0.5, patch_w=0.0, embedding_w=0.5, flops_w=5.0)
            imgs, labels = inputs['imgs'].cuda(), inputs['labels'].cuda()
            step = lambda: engine.search_step(m, crit, imgs, labels, 1.0, opts)
            n_replay = 3
            if graphed:
                gs = engine.GraphedStep(step, opts)
                gs.capture(warm_steps=2)                     # two eager steps on the capture stream, then the capture
                for g in opts[0].param_groups:
                    g['lr'] = 2e-3                           # a scheduler change between replays must reach the graph
                for _ in range(n_replay):
                    out = gs()
            else:
                for i in range(2 + n_replay):
                    if i == 2:
                        for g in opts[0].param_groups:
                            g['lr'] = 2e-3
                    out = step()
            torch.cuda.synchronize()
            results.append(({k: v.detach().clone() for k, v in m.state_dict().items()}, float(out[3])))
        (sd_e, loss_e), (sd_e2, loss_e2), (sd_g, loss_g) = results
        rel = lambda a, b: {k: float((a[k].double() - b[k].double()).abs().max() / (a[k].double().abs().max() + 1e-12)) for k in a}
        r_ee, r_eg = rel(sd_e, sd_e2), rel(sd_e, sd_g)
        k_ee, k_eg = max(r_ee, key=r_ee.get), max(r_eg, key=r_eg.get)
        print(f'eager vs eager: worst {r_ee[k_ee]:.2e} ({k_ee}); graphed vs eager: loss {loss_g:.6f} / {loss_e:.6f}, worst {r_eg[k_eg]:.2e} ({k_eg})')
        print('   differing:', {k: f'{v:.1e}' for k, v in r_eg.items() if v > 1e-7})
        assert abs(loss_e - loss_g) <= 1e-5 * max(1.0, abs(loss_e))
        assert r_eg[k_eg] <= max(1e-5, 2 * r_ee[k_ee])
    finally:
        torch.cuda.set_stream(prev)


# ------------------------------------------------------------------------------------------------------------------
# the epoch protocol against the REFERENCE's own run of engine.search_one_epoch / train_one_epoch (tests/golden/mini_epoch.npz,
# micro_train_epoch.npz, made by tests/golden/make_golden_epoch.py; the oracle is pinned to the same fixtures in
# tests/test_oracle_epoch.py): schedules, gradient accumulation, the compress trigger, the three optimizers with per-step learning
# rates and the returned statistics (every key of the reference's MetricLogger)
# ------------------------------------------------------------------------------------------------------------------
def test_search_one_epoch_matches_reference_run():
    import numpy as np
    from oracle import ofb_oracle as O
    from ofb_amd import engine
    from ofb_amd.losses import OFBSearchLOSS, DistillationLoss, LabelSmoothingCrossEntropy
    from tests import epoch_util as E
    from tests.golden_util import sample
    from tests.test_gpu_model import build_product
    z = E.load()
    cfg = O.Config(**E.MINI, drop_path_rate=0.0)
    dev = torch.device('cuda')
    inputs = dict(patch_noise=E.noise_of(0, cfg.num_patches), droppath_u=torch.zeros(2 * cfg.depth, E.BATCH))
    m = build_product(cfg, O.SearchState(), inputs)
    by_name = dict(zip(O.module_names(cfg), m.searchable_modules))
    opt_p, opt_a, opt_d = engine.build_optimizers(m, lr=E.LR0['p'], lr_arch=E.LR0['a'], lr_decoder=E.LR0['d'], weight_decay=1e-3)

    class Sched:
        def __init__(self, opt, which):
            self.opt, self.which, self.calls = opt, which, []

        def step_update(self, gstep):
            self.calls.append(gstep)
            for g in self.opt.param_groups:
                g['lr'] = E.lr_at(self.which, gstep)

    sch = {'p': Sched(opt_p, 'p'), 'a': Sched(opt_a, 'a'), 'd': Sched(opt_d, 'd')}
    per_it = dict(base=[], arch=[], keep_ratio=[], w_p=[])
    crit0 = OFBSearchLOSS(DistillationLoss(LabelSmoothingCrossEntropy(0.1), None, 'none', 0.5, 1.0), dev, attn_w=0.5, mlp_w=0.5,
                          patch_w=0.0, embedding_w=0.5, flops_w=5.0)

    def crit(samples, outputs, targets, mdl, phase, target_flops, finish):
        loss = crit0(samples, outputs, targets, mdl, phase, target_flops, finish)
        base, arch = loss if isinstance(loss, tuple) else (loss, None)
        per_it['base'].append(base.detach())
        per_it['arch'].append(arch.detach() if arch is not None else torch.zeros((), device=dev))
        per_it['keep_ratio'].append(float(m.patch_ratio_list[0]))
        live = [x.w_p for x in m.searchable_modules if not x.finish_search]
        per_it['w_p'].append(float(live[0]) if live else -1.0)
        return loss

    class Loader:                                            # the loader of the fixture: crafts the alphas, selects the noise
        def __len__(self):
            return E.N_ITER

        def __iter__(self):
            for i in range(E.N_ITER):
                stage = E.CRAFT_AT.get(i)
                if stage:
                    for name, a in E.crafted(z, stage).items():
                        assert tuple(by_name[name].alpha.shape) == tuple(a.shape), name
                        by_name[name].alpha.data.copy_(a)
                m._forced = dict(patch_noise=E.noise_of(i, cfg.num_patches).to(dev), droppath_u=torch.zeros(2 * cfg.depth, E.BATCH, device=dev))
                yield E.batch_of(i, cfg.num_classes)

    args = types.SimpleNamespace(accum_iter=E.ACCUM, warmup_epochs=E.WARMUP_EPOCHS, epochs=E.EPOCHS)
    stats, fin, pruned, o_p, o_d, o_a = engine.search_one_epoch(m, crit, 1.0, Loader(), opt_p, opt_d, opt_a, sch['p'], sch['a'], sch['d'], dev,
                                                                epoch=0, args=args, print_freq=2)
    torch.cuda.synchronize()
    assert [int(fin), int(pruned)] == z['flags'].tolist() and o_a is None
    for w in 'pad':
        assert sch[w].calls == z[f'sched_calls.{w}'].tolist(), (w, sch[w].calls)
    got = {k: np.array([float(v) for v in per_it[k]]) for k in per_it}
    for k in ('base', 'arch', 'keep_ratio', 'w_p'):
        exp = z[f'it.{k}']
        ok = exp >= 0 if k == 'w_p' else np.ones_like(exp, bool)
        assert np.allclose(got[k][ok], exp[ok], rtol=1e-3, atol=1e-7), (k, got[k], exp)
    E.check_stats(z, stats, 1e-3)                           # north_star tolerance; the same keys as the reference's dict
    assert abs(o_p.param_groups[0]['lr'] - z['lr_final'][0]) < 1e-12 and abs(o_d.param_groups[0]['lr'] - z['lr_final'][1]) < 1e-12
    name_of = {id(p): k for k, p in m.named_parameters()}
    for tag, o in (('p', o_p), ('d', o_d)):
        for gi, grp in enumerate(o.param_groups):
            assert [name_of[id(p)] for p in grp['params']] == list(z[f'optnames.{tag}.{gi}']), (tag, gi)
    for name, mod in by_name.items():
        assert np.array_equal(mod.switch_cell.cpu().numpy().astype(bool), z[f'switch.{name}']), name
        fl = z[f'flags.{name}'].tolist()
        assert [int(mod.finish_search), int(mod.execute_prune)] == fl[:2], name
    for k, p in m.named_parameters():
        v = p.detach().cpu()
        assert list(v.shape) == z[f'shape.{k}'].tolist() and bool(p.requires_grad) == bool(z[f'rg.{k}']), k
        if 'alpha' in k or 'score' in k:
            assert float((v - torch.from_numpy(z[f'val.{k}'])).abs().max()) < 1e-3 * max(1.0, float(np.abs(z[f'val.{k}']).max())), k
        elif not k.endswith('qkv.bias'):                    # (zero-gradient third: Adam moves it by +-lr on rounding noise)
            assert float((sample(v).double() - torch.from_numpy(z[f'vsamp.{k}']).double()).abs().max()) < 2e-4, k


def test_train_one_epoch_matches_reference_run():
    import numpy as np
    import ofb_amd
    from ofb_amd import engine
    from ofb_amd.optim import AdamW
    from ofb_amd.losses import DistillationLoss, LabelSmoothingCrossEntropy
    from oracle import fill
    from oracle import ofb_oracle as O
    from tests import epoch_util as E
    from tests.golden_util import sample
    z = E.load('micro_train_epoch')
    batch, n_iter = int(z['meta'][0]), int(z['meta'][1])
    dev = torch.device('cuda')
    m = ofb_amd.VisionTransformer(embed_dim=E.FT['embed_dim'], depth=E.FT['depth'], num_heads=E.FT['num_heads'],
                                  num_classes=E.FT['num_classes'], drop_path_rate=0.0)
    m.load_state_dict({k: torch.from_numpy(fill.param_value(k, tuple(v.shape))) for k, v in m.state_dict().items()})
    m.to(dev)
    names, params = {0: [], 1: []}, {0: [], 1: []}
    for k, p in m.named_parameters():
        gi = 0 if O.optimizer_group(k, tuple(p.shape)) == 'nodecay' else 1
        names[gi].append(k)
        params[gi].append(p)
    opt = AdamW([{'params': params[0], 'weight_decay': 0.}, {'params': params[1], 'weight_decay': 1e-3}], names, lr=1e-3)

    class Sched:
        calls = []

        def step_update(self, gstep):
            self.calls.append(gstep)
            for g in opt.param_groups:
                g['lr'] = E.ft_lr_at(gstep)

    data = [E.batch_of(i, E.FT['num_classes'], batch) for i in range(n_iter)]
    crit = DistillationLoss(LabelSmoothingCrossEntropy(0.1), None, 'none', 0.5, 1.0)
    sched = Sched()
    stats = engine.train_one_epoch(m, crit, data, opt, sched, dev, epoch=1, args=types.SimpleNamespace(accum_iter=1), print_freq=1)
    torch.cuda.synchronize()
    assert sched.calls == z['sched_calls'].tolist()
    E.check_stats(z, stats, 1e-3)
    for k, p in m.named_parameters():
        if not k.endswith('qkv.bias'):
            assert float((sample(p.detach().cpu()).double() - torch.from_numpy(z[f'vsamp.{k}']).double()).abs().max()) < 2e-4, k
