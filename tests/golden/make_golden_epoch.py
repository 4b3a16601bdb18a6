"""Golden vectors of the EPOCH engines: the reference's own `engine.search_one_epoch` / `engine.train_one_epoch`
(/root/reference/engine.py:75-219, :18-72) run unmodified, on CPU, in the build container.

Run:  python tests/golden/make_golden_epoch.py            (writes tests/golden/mini_epoch.npz, micro_train_epoch.npz)

search epoch: MINI model (embed 128, depth 3, 4 heads, 10 classes), batch 2, SIX iterations, accum_iter 2 -> three accumulation
windows, a compress() after each of them while the search is live (engine.py:201: every len // 3 // accum = 1 windows).  The data
loader is an input of the engine: ours crafts the alphas (the closed-form overrides of make_golden.craft_alphas) when it hands out
batches 0 and 2, so that the first compress() cuts, the second one finishes the search and iterations 4, 5 run the finished model
(criterion returns the base loss only, the architecture optimizer is gone).  Patch-mask noise is closed-form per iteration (a patched
torch.rand, as in make_golden.py), drop_path 0.  The three lr "schedulers" are closed-form functions of the global step, different
per optimizer.  Stored: the returned stats dict (MetricLogger.global_avg of every meter: engine.py:216-218), the flags, per-iteration
losses / keep ratio / w_p (observed through the criterion and the model, not the engine), the optimizers' parameter-name lists and
final learning rates, all alphas / scores / switches / shapes and sampled weights after the epoch.
finetune epoch: a plain micro ViT, batch 3, three iterations, accum_iter 1, one AdamW.
"""
import contextlib
import io
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G                                    # noqa: E402  (sets up sys.path for the reference and the shims)
from oracle import fill                                   # noqa: E402
from oracle import ofb_oracle as O                        # noqa: E402

import engine as RENG                                      # noqa: E402  reference engine.py
import losses as RLOSS                                     # noqa: E402
import optim as ROPT                                       # noqa: E402
import models.vision_transformer as RVT                    # noqa: E402
import models.layers as RL                                 # noqa: E402
from timm.loss import LabelSmoothingCrossEntropy           # noqa: E402  (shim)

N_ITER, ACCUM, BATCH = 6, 2, 2
LR0 = {'p': 1.0e-3, 'a': 2.0e-3, 'd': 5.0e-4}


def lr_at(which, gstep):
    """the lr "schedule" of the fixture: closed-form in the global step handed to step_update (engine.py:173-179)"""
    return LR0[which] * (1.0 + {'p': 0.10, 'a': 0.05, 'd': 0.20}[which] * (gstep + 1))


class Sched:
    def __init__(self, opt, which):
        self.opt, self.which, self.calls = opt, which, []

    def step_update(self, gstep):
        self.calls.append(gstep)
        for g in self.opt.param_groups:
            g['lr'] = lr_at(self.which, gstep)


def batch_of(i, ncls, batch=BATCH):
    imgs = torch.from_numpy(fill.images(batch, tag=f'epoch_imgs{i}'))
    labels = torch.from_numpy((fill.labels(batch, ncls) + i) % ncls)
    return imgs, labels


class Loader:
    """hands out the closed-form batches; `hook(i)` runs before batch i leaves (alpha crafting, noise selection)"""
    def __init__(self, n, ncls, hook, batch=BATCH):
        self.n, self.ncls, self.hook, self.batch = n, ncls, hook, batch

    def __len__(self):
        return self.n

    def __iter__(self):
        for i in range(self.n):
            self.hook(i)
            yield batch_of(i, self.ncls, self.batch)


def run_search_epoch(tag='mini_epoch', w_p0=0.99, thresh_note=0.2):
    cfg = O.Config(**G.MINI, drop_path_rate=0.0)
    model = G.build_reference(cfg, 0.0)
    names = O.module_names(cfg)
    by_name = dict(zip(names, model.searchable_modules))
    groups = {'nodecay': [], 'decay': [], 'decoder_nodecay': [], 'decoder_decay': [], 'arch': []}
    gnames = {k: [] for k in groups}
    for k, p in model.named_parameters():
        if p.requires_grad:
            grp = O.optimizer_group(k, tuple(p.shape))
            groups[grp].append(p)
            gnames[grp].append(k)
    opt_p = ROPT.AdamW([{'params': groups['nodecay'], 'weight_decay': 0.}, {'params': groups['decay'], 'weight_decay': 1e-3}],
                       {0: gnames['nodecay'], 1: gnames['decay']}, lr=LR0['p'], eps=1e-8, betas=(0.9, 0.999))
    opt_d = ROPT.AdamW([{'params': groups['decoder_nodecay'], 'weight_decay': 0.}, {'params': groups['decoder_decay'], 'weight_decay': 1e-3}],
                       {0: gnames['decoder_nodecay'], 1: gnames['decoder_decay']}, lr=LR0['d'], eps=1e-8, betas=(0.9, 0.999))
    opt_a = ROPT.AdamW(groups['arch'], {0: gnames['arch']}, lr=LR0['a'], eps=1e-8, betas=(0.5, 0.999), weight_decay=1e-3)
    sch = {'p': Sched(opt_p, 'p'), 'a': Sched(opt_a, 'a'), 'd': Sched(opt_d, 'd')}
    crit0 = RLOSS.OFBSearchLOSS(RLOSS.DistillationLoss(LabelSmoothingCrossEntropy(0.1), None, 'none', 0.5, 1.0),
                                 torch.device('cpu'), attn_w=0.5, mlp_w=0.5, patch_w=0.0, embedding_w=0.5, flops_w=5.0)
    out = dict(meta=np.array([BATCH, N_ITER, ACCUM, 0.2], np.float64))
    per_it = {k: [] for k in ('base', 'arch', 'dec', 'keep_ratio', 'w_p', 'finish')}
    state = {'it': -1}

    def hook(i):
        state['it'] = i
        stage = {0: 1, 2: 2}.get(i)
        if stage:
            for name, a in G.craft_alphas(stage).items():
                mod = by_name[name]
                assert tuple(mod.alpha.shape) == a.shape, (name, mod.alpha.shape, a.shape)
                mod.alpha.data.copy_(torch.from_numpy(a))
                out[f'craft{stage}.{name}'] = a

    def fake_rand(*shape, **kw):
        shape = shape[0] if len(shape) == 1 and isinstance(shape[0], (tuple, list)) else shape
        if tuple(shape) == (BATCH, cfg.num_patches):
            return torch.from_numpy(fill.patch_noise(BATCH, cfg.num_patches, tag=f'epoch_noise{state["it"]}'))
        raise RuntimeError(f'unexpected rand shape {shape}')

    wrapped = G._Wrap(model)
    seen = {}

    class Crit:                                               # observes, never alters: what the criterion saw and returned
        def __call__(self, samples, outputs, targets, mdl, phase, target_flops, finish_search):
            loss = crit0(samples, outputs, targets, mdl, phase, target_flops, finish_search)
            base, arch = loss if isinstance(loss, tuple) else (loss, torch.zeros(()))
            per_it['base'].append(float(base.detach()))
            per_it['arch'].append(float(arch.detach()))
            per_it['finish'].append(float(bool(finish_search)))
            per_it['keep_ratio'].append(float(model.patch_ratio_list[0]))
            live = [m.w_p for m in model.searchable_modules if not m.finish_search]
            per_it['w_p'].append(float(live[0]) if live else -1.0)
            per_it['dec'].append(seen['dec'])
            return loss

    real_fwd = model.forward

    def fwd(x):
        res = real_fwd(x)
        d = res[1][0]
        seen['dec'] = float(d.detach()) if not isinstance(d, float) else 0.0
        return res

    model.forward = fwd
    args = types.SimpleNamespace(accum_iter=ACCUM, warmup_epochs=2, epochs=10)
    real_rand, real_sync = torch.rand, torch.cuda.synchronize
    torch.rand = fake_rand
    torch.cuda.synchronize = lambda *a, **k: None
    try:
        with contextlib.redirect_stdout(io.StringIO()) as log:
            stats, fin, pruned, o_p, o_d, o_a = RENG.search_one_epoch(
                wrapped, Crit(), 1.0, Loader(N_ITER, cfg.num_classes, hook), opt_p, opt_d, opt_a, sch['p'], sch['a'], sch['d'],
                torch.device('cpu'), epoch=0, max_norm=None, model_ema=None, mixup_fn=None, set_training_mode=True, use_amp=False,
                finish_search=False, args=args, progressive=True, max_ratio=0.95, min_ratio=0.75)
    finally:
        torch.rand, torch.cuda.synchronize = real_rand, real_sync
        model.forward = real_fwd
    assert fin and pruned and o_a is None and o_p is opt_p and o_d is opt_d
    for k, v in stats.items():
        out[f'stats.{k}'] = np.float64(v)
    out['stats_keys'] = np.array(sorted(stats))
    out['flags'] = np.array([int(fin), int(pruned)], np.int64)
    for k, v in per_it.items():
        out[f'it.{k}'] = np.array(v, np.float64)
    for w in 'pad':
        out[f'sched_calls.{w}'] = np.array(sch[w].calls, np.int64)
    out['lr_final'] = np.array([opt_p.param_groups[0]['lr'], opt_d.param_groups[0]['lr']], np.float64)
    pid = {id(p): k for k, p in model.named_parameters()}
    for on, o in (('p', opt_p), ('d', opt_d)):
        for gi, grp in enumerate(o.param_groups):
            got = [pid[id(p)] for p in grp['params']]
            assert got == list(o.param_names[gi]), (on, gi)
            out[f'optnames.{on}.{gi}'] = np.array(got)
    for name, mod in by_name.items():
        out[f'switch.{name}'] = mod.switch_cell.numpy().copy()
        out[f'flags.{name}'] = np.array([mod.finish_search, mod.execute_prune, getattr(mod, 'head_num', -1)], np.int64)
    for k, p in model.named_parameters():
        out[f'shape.{k}'] = np.array(p.shape, np.int64)
        out[f'rg.{k}'] = np.array(p.requires_grad)
        if 'alpha' in k or 'score' in k:
            out[f'val.{k}'] = p.detach().numpy().copy()
        else:
            out[f'vsamp.{k}'] = G.sample(p)
    path = os.path.join(HERE, f'{tag}.npz')
    np.savez_compressed(path, **out)
    print(f'{tag}: stats ' + ' '.join(f'{k}={v:.6f}' for k, v in sorted(stats.items())))
    print('   per-iteration base', np.round(out['it.base'], 5).tolist(), 'finish', out['it.finish'].tolist(),
          'keep', np.round(out['it.keep_ratio'], 4).tolist())
    print(f'   -> {os.path.getsize(path) / 1024:.0f} KiB')


# ----------------------------------------------------------------------------------------------------------------
# finetune epoch (engine.py:18-72) on a plain micro ViT
# ----------------------------------------------------------------------------------------------------------------
FT = dict(embed_dim=64, depth=2, num_heads=2, num_classes=10)


def ft_lr_at(gstep):
    return 1.0e-3 * (1.0 + 0.25 * (gstep + 1))


def run_train_epoch(tag='micro_train_epoch', n_iter=3, batch=3):
    from functools import partial
    RL.ModuleInjection.method = 'full'
    m = RVT.VisionTransformer(patch_size=16, embed_dim=FT['embed_dim'], depth=FT['depth'], num_heads=FT['num_heads'], mlp_ratio=4,
                              qkv_bias=True, norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), num_classes=FT['num_classes'],
                              drop_path_rate=0.0)
    sd = {k: torch.from_numpy(fill.param_value(k, tuple(v.shape))) for k, v in m.state_dict().items()}
    m.load_state_dict(sd, strict=True)
    names = {0: [], 1: []}
    params = {0: [], 1: []}
    for k, p in m.named_parameters():
        gi = 0 if O.optimizer_group(k, tuple(p.shape)) == 'nodecay' else 1
        names[gi].append(k)
        params[gi].append(p)
    opt = ROPT.AdamW([{'params': params[0], 'weight_decay': 0.}, {'params': params[1], 'weight_decay': 1e-3}], names, lr=1e-3, eps=1e-8,
                     betas=(0.9, 0.999))

    class FtSched:
        calls = []

        def step_update(self, gstep):
            self.calls.append(gstep)
            for g in opt.param_groups:
                g['lr'] = ft_lr_at(gstep)

    crit = RLOSS.DistillationLoss(LabelSmoothingCrossEntropy(0.1), None, 'none', 0.5, 1.0)
    args = types.SimpleNamespace(accum_iter=1)
    sched = FtSched()
    real_sync = torch.cuda.synchronize
    torch.cuda.synchronize = lambda *a, **k: None
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            stats = RENG.train_one_epoch(m, crit, Loader(n_iter, FT['num_classes'], lambda i: None, batch), opt, sched, torch.device('cpu'),
                                         epoch=1, loss_scaler=None, max_norm=None, model_ema=None, mixup_fn=None, set_training_mode=True,
                                         use_amp=False, args=args)
    finally:
        torch.cuda.synchronize = real_sync
    out = dict(meta=np.array([batch, n_iter], np.float64), stats_keys=np.array(sorted(stats)), sched_calls=np.array(sched.calls, np.int64))
    for k, v in stats.items():
        out[f'stats.{k}'] = np.float64(v)
    for k, p in m.named_parameters():
        out[f'vsamp.{k}'] = G.sample(p)
    path = os.path.join(HERE, f'{tag}.npz')
    np.savez_compressed(path, **out)
    print(f'{tag}: stats ' + ' '.join(f'{k}={v:.6f}' for k, v in sorted(stats.items())), f'-> {os.path.getsize(path) / 1024:.0f} KiB')


if __name__ == '__main__':
    torch.manual_seed(0)
    torch.set_num_threads(8)
    run_search_epoch()
    run_train_epoch()
