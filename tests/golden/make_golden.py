"""Generate golden vectors by importing the REFERENCE (/root/reference) in the build container.

Run:  python tests/golden/make_golden.py            (writes tests/golden/*.npz)

The reference never travels to the GPU box; only these small fixtures (expected OUTPUTS) do.
Inputs and parameters are closed-form (oracle/fill.py) so both sides regenerate them.
Randomness inside the reference (patch-mask noise `torch.rand(N, L)`, vision_transformer.py:597,
and DropPath keep draws) is replaced by the same closed-form noise via a patched torch.rand.
"""
import os
import sys
import types
import io
import contextlib

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [os.path.join(ROOT, 'oracle', '_shims'), '/root/reference', ROOT]
sys.modules['torch._six'] = types.SimpleNamespace(inf=float('inf'))

from oracle import fill                                   # noqa: E402
from oracle import ofb_oracle as O                        # noqa: E402  (only for Config / name helpers)

import models.vision_transformer as RVT                    # noqa: E402  reference
import models.layers as RL                                 # noqa: E402
import losses as RLOSS                                     # noqa: E402
import optim as ROPT                                       # noqa: E402
from timm.loss import LabelSmoothingCrossEntropy           # noqa: E402  (shim)
from timm.models.layers import DropPath                    # noqa: E402  (shim)


class _Wrap(torch.nn.Module):                              # engine/loss call model.module.*
    def __init__(self, m):
        super().__init__()
        self.module = m

    def forward(self, x):
        return self.module(x)


def build_reference(cfg: O.Config, drop_path: float):
    RL.ModuleInjection.method = 'search'
    RL.ModuleInjection.searchable_modules = []
    from functools import partial
    m = RVT.MIMVisionTransformer(
        patch_size=cfg.patch_size, embed_dim=cfg.embed_dim, depth=cfg.depth, num_heads=cfg.num_heads, mlp_ratio=4,
        qkv_bias=True, norm_layer=partial(RL.LayerNorm, eps=1e-6), embed_layer=RL.PatchEmbed, mae=True,
        head_search=cfg.attn_space == 'head', channel_search=cfg.attn_space == 'channel', num_classes=cfg.num_classes,
        drop_path_rate=drop_path, attn_search=True, mlp_search=True, embed_search=True, patch_search=cfg.patch_search, mask_ratio=1.0)
    m.searchable_modules = [x for x in m.modules() if hasattr(x, 'alpha')]
    sd = {k: torch.from_numpy(np.ones(v.shape, np.float32) if (k == 'alpha_patch' and tuple(v.shape) == (1, 1))
                              else fill.param_value(k, tuple(v.shape)))
          for k, v in m.state_dict().items()}
    m.load_state_dict(sd, strict=True)
    m.correct_require_grad(0.5, 0.5, 0.5 if cfg.patch_search else 0, 0.5)
    return m


def sample(t: torch.Tensor, n=256):
    """<= ~n strided samples per tensor (stride = numel // n); tests use the same rule."""
    flat = t.detach().reshape(-1)
    return flat[::max(1, flat.numel() // n)].numpy().copy()


def run_case(tag, cfg_kw, batch, w_p, keep_ratio, drop_path, switches, lr=1e-3, full_grads=False):
    cfg = O.Config(**cfg_kw, drop_path_rate=drop_path)
    model = build_reference(cfg, drop_path)
    names = O.module_names(cfg)
    for mod, name in zip(model.searchable_modules, names):
        mod.w_p = w_p
        if name in switches:
            mod.switch_cell = torch.from_numpy(switches[name])
    if cfg.patch_search:
        # the keep ratio is the first LIVE patch cell's (vision_transformer.py:593); the ratio list stays the constructor's
        if 'patch' in switches:
            model.switch_cell_patch = torch.from_numpy(switches['patch'])
        live = [r for i, r in enumerate(model.patch_ratio_list) if bool(model.switch_cell_patch[0, i])]
        assert abs(live[0] - keep_ratio) < 1e-12, (live, keep_ratio)
    else:
        model.patch_ratio_list = [keep_ratio]
    patch_w = 0.5 if cfg.patch_search else 0.0
    model.train()

    imgs = torch.from_numpy(fill.images(batch))
    labels = torch.from_numpy(fill.labels(batch, cfg.num_classes))
    pnoise = torch.from_numpy(fill.patch_noise(batch, cfg.num_patches))
    dnoise = torch.from_numpy(fill.droppath_noise(2 * cfg.depth, batch))
    calls = {'dp': 0}

    def fake_rand(*shape, **kw):
        shape = shape[0] if len(shape) == 1 and isinstance(shape[0], (tuple, list)) else shape
        if tuple(shape) == (batch, cfg.num_patches):
            return pnoise.clone()
        if tuple(shape) == (batch, 1, 1):
            r = dnoise[calls['dp']].view(batch, 1, 1).clone()
            calls['dp'] += 1
            return r
        raise RuntimeError(f'unexpected rand shape {shape}')

    real_rand = torch.rand
    torch.rand = fake_rand
    DropPath.rand = staticmethod(fake_rand)
    try:
        crit = RLOSS.OFBSearchLOSS(RLOSS.DistillationLoss(LabelSmoothingCrossEntropy(0.1), None, 'none', 0.5, 1.0),
                                    torch.device('cpu'), attn_w=0.5, mlp_w=0.5, patch_w=patch_w, embedding_w=0.5, flops_w=5.0)
        wrapped = _Wrap(model)
        with contextlib.redirect_stdout(io.StringIO()):
            logits, (dec_loss, score_loss) = wrapped(imgs)
            base, arch = crit(imgs, logits, labels, wrapped, 'arch', 1.0, False)   # losses.py:80
            l_attn, l_mlp, l_patch, l_emb = model.get_sparsity_loss(torch.device('cpu'))
            f_tot, f_sea = model.get_flops()
        assert score_loss is None
        total = base + arch                                                        # engine.py:134-144
        total = total + (base / dec_loss).data.clone() * dec_loss
        total.backward()
    finally:
        torch.rand = real_rand
        DropPath.rand = staticmethod(real_rand)

    out = dict(logits=logits.detach().numpy(), decoder_loss=dec_loss.item(), base=base.item(), arch=arch.item(),
               loss_attn=l_attn.item(), loss_mlp=l_mlp.item(), loss_embed=l_emb.item(), loss_patch=l_patch.item(),
               flops_total=float(f_tot), flops_searched=f_sea.item(), loss_total=total.item(),
               meta=np.array([batch, w_p, keep_ratio, drop_path, lr], np.float64))
    for mod, name in zip(model.searchable_modules, names):
        wr, prob = mod.get_weight()
        # restricted attention spaces keep score as (H, 1) / (1, d) while the staircase is stored broadcast to (H, d)
        shp = wr.shape if wr.numel() != mod.score.numel() else mod.score.shape
        out[f'gate.{name}.wr'] = wr.detach().reshape(shp).numpy()
        out[f'gate.{name}.wm'] = mod.weighted_mask.detach().reshape(shp).numpy()
        g = (1 - w_p) * wr.detach().reshape(shp) + w_p * mod.score.detach().sigmoid()
        out[f'gate.{name}.g'] = g.numpy()
    for k, p in model.named_parameters():
        if p.grad is None:
            continue
        g = p.grad
        out[f'gnorm.{k}'] = np.float64(g.double().norm().item())
        if 'alpha' in k or 'score' in k or (full_grads and g.numel() <= 5000):
            out[f'grad.{k}'] = g.numpy().copy()
        else:
            out[f'gsamp.{k}'] = sample(g)

    # one optimizer step with the three reference AdamW instances (search.py:486-559 grouping)
    groups = {'nodecay': [], 'decay': [], 'decoder_nodecay': [], 'decoder_decay': [], 'arch': []}
    gnames = {k: [] for k in groups}
    for k, p in model.named_parameters():
        if not p.requires_grad:
            continue
        grp = O.optimizer_group(k, tuple(p.shape))
        groups[grp].append(p)
        gnames[grp].append(k)
    opt_p = ROPT.AdamW([{'params': groups['nodecay'], 'weight_decay': 0.}, {'params': groups['decay'], 'weight_decay': 1e-3}],
                       {0: gnames['nodecay'], 1: gnames['decay']}, lr=lr, eps=1e-8, betas=(0.9, 0.999))
    opt_d = ROPT.AdamW([{'params': groups['decoder_nodecay'], 'weight_decay': 0.}, {'params': groups['decoder_decay'], 'weight_decay': 1e-3}],
                       {0: gnames['decoder_nodecay'], 1: gnames['decoder_decay']}, lr=lr, eps=1e-8, betas=(0.9, 0.999))
    opt_a = ROPT.AdamW(groups['arch'], {0: gnames['arch']}, lr=lr, eps=1e-8, betas=(0.5, 0.999), weight_decay=1e-3)
    for o in (opt_p, opt_a, opt_d):
        o.step()
    for k, p in model.named_parameters():
        if 'alpha' in k or 'score' in k:
            out[f'after.{k}'] = p.detach().numpy().copy()
        else:
            out[f'asamp.{k}'] = sample(p)
    out['groups'] = np.array([f'{g}:{",".join(v)}' for g, v in gnames.items()])
    for name, sw in switches.items():
        out[f'switch.{name}'] = sw
    path = os.path.join(HERE, f'{tag}.npz')
    np.savez_compressed(path, **out)
    print(f'{tag}: base={out["base"]:.6f} arch={out["arch"]:.6f} dec={out["decoder_loss"]:.6f} '
          f'flops={out["flops_searched"]:.5f}/{out["flops_total"]:.5f} -> {os.path.getsize(path) / 1024:.0f} KiB')


# ----------------------------------------------------------------------------------------------------------------
# compress() life cycle: step -> crafted alphas -> compress -> step -> crafted alphas -> compress (finishes)
# -> step (finish_search) -> eval -> fuse -> eval.  Everything the reference mutates is recorded per stage.
# ----------------------------------------------------------------------------------------------------------------
MINI = dict(embed_dim=128, depth=3, num_heads=4, num_classes=10)


def craft_alphas(stage):
    """closed-form alpha overrides (inputs of the case; stored in the fixture as craft<stage>.<name>)."""
    lo = -6.0

    def kept(r, c, seed):
        # non-uniform values for surviving cells (exactly uniform probabilities put tan(pi/2 - 0) into the sparsity loss)
        i, j = np.meshgrid(np.arange(r), np.arange(c), indexing='ij')
        return (0.3 * np.sin(1.7 * (i * c + j) + seed)).astype(np.float32)

    if stage == 1:
        e = kept(1, 17, 0.1); e[0, 14:] = lo                                   # embed: last 3 options die -> D 128 -> 116
        a0 = np.full((2, 7), lo, np.float32); a0[0, 4] = 0.5                   # attn0: one cell survives -> finished (2 heads x 24)
        a1 = kept(2, 7, 0.2); a1[:, 6] = lo                                    # attn1: last column dies -> 4 heads x 28
        a2 = kept(2, 7, 0.3); a2[1, :] = lo                                    # attn2: last row dies -> 2 heads x 32
        m0 = np.full((1, 7), lo, np.float32); m0[0, 3] = 0.25                  # mlp0: finished, hidden 320
        m1 = kept(1, 7, 0.4); m1[0, 5:] = lo                                   # mlp1: last two die -> hidden 384
        m2 = kept(1, 7, 0.5); m2[0, 2] = lo                                    # mlp2: a middle cell dies, no shape change
        return {'patch_embed': e, 'blocks.0.attn': a0, 'blocks.1.attn': a1, 'blocks.2.attn': a2,
                'blocks.0.mlp': m0, 'blocks.1.mlp': m1, 'blocks.2.mlp': m2}
    e = np.full((1, 14), lo, np.float32); e[0, 9] = 0.0                        # embed finishes at option 9 -> D 100
    a1 = np.full((2, 6), lo, np.float32); a1[1, 2] = 0.0                       # attn1: 4 heads x 16
    a2 = np.full((1, 7), lo, np.float32); a2[0, 5] = 0.0                       # attn2: 2 heads x 28
    m1 = np.full((1, 5), lo, np.float32); m1[0, 0] = 0.0                       # mlp1: hidden 128
    m2 = np.full((1, 7), lo, np.float32); m2[0, 6] = 0.0                       # mlp2: keeps all 512 (cell 2 already off)
    return {'patch_embed': e, 'blocks.1.attn': a1, 'blocks.2.attn': a2, 'blocks.1.mlp': m1, 'blocks.2.mlp': m2}


def run_compress_case(tag='mini_c', batch=2, w_p=0.7, keep_ratio=0.9, drop_path=0.1, lr=1e-3, thresh=0.2):
    cfg = O.Config(**MINI, drop_path_rate=drop_path)
    model = build_reference(cfg, drop_path)
    names = O.module_names(cfg)
    by_name = dict(zip(names, model.searchable_modules))
    for mod in model.searchable_modules:
        mod.w_p = w_p
    model.patch_ratio_list = [keep_ratio]
    imgs = torch.from_numpy(fill.images(batch))
    labels = torch.from_numpy(fill.labels(batch, cfg.num_classes))
    pnoise = torch.from_numpy(fill.patch_noise(batch, cfg.num_patches))
    dnoise = torch.from_numpy(fill.droppath_noise(2 * cfg.depth, batch))
    calls = {'dp': 0}

    def fake_rand(*shape, **kw):
        shape = shape[0] if len(shape) == 1 and isinstance(shape[0], (tuple, list)) else shape
        if tuple(shape) == (batch, cfg.num_patches):
            return pnoise.clone()
        if tuple(shape) == (batch, 1, 1):
            r = dnoise[calls['dp']].view(batch, 1, 1).clone()
            calls['dp'] += 1
            return r
        raise RuntimeError(f'unexpected rand shape {shape}')

    groups = {'nodecay': [], 'decay': [], 'decoder_nodecay': [], 'decoder_decay': [], 'arch': []}
    gnames = {k: [] for k in groups}
    for k, p in model.named_parameters():
        if p.requires_grad:
            grp = O.optimizer_group(k, tuple(p.shape))
            groups[grp].append(p)
            gnames[grp].append(k)
    opt_p = ROPT.AdamW([{'params': groups['nodecay'], 'weight_decay': 0.}, {'params': groups['decay'], 'weight_decay': 1e-3}],
                       {0: gnames['nodecay'], 1: gnames['decay']}, lr=lr, eps=1e-8, betas=(0.9, 0.999))
    opt_d = ROPT.AdamW([{'params': groups['decoder_nodecay'], 'weight_decay': 0.}, {'params': groups['decoder_decay'], 'weight_decay': 1e-3}],
                       {0: gnames['decoder_nodecay'], 1: gnames['decoder_decay']}, lr=lr, eps=1e-8, betas=(0.9, 0.999))
    opt_a = ROPT.AdamW(groups['arch'], {0: gnames['arch']}, lr=lr, eps=1e-8, betas=(0.5, 0.999), weight_decay=1e-3)
    opts = {'p': opt_p, 'd': opt_d, 'a': opt_a}
    crit = RLOSS.OFBSearchLOSS(RLOSS.DistillationLoss(LabelSmoothingCrossEntropy(0.1), None, 'none', 0.5, 1.0),
                                torch.device('cpu'), attn_w=0.5, mlp_w=0.5, patch_w=0.0, embedding_w=0.5, flops_w=5.0)
    wrapped = _Wrap(model)
    out = dict(meta=np.array([batch, w_p, keep_ratio, drop_path, lr, thresh], np.float64))
    real_rand, real_sync = torch.rand, torch.cuda.synchronize
    torch.cuda.synchronize = lambda *a, **k: None            # the reference synchronises the (absent) device inside compress()

    def step(pre, finish):
        """one search step incl. the three optimizer updates; records losses / grads / updated params under `pre`."""
        model.train()
        for p in model.parameters():
            p.grad = None
        calls['dp'] = 0
        torch.rand = fake_rand
        DropPath.rand = staticmethod(fake_rand)
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                logits, (dec_loss, _) = wrapped(imgs)
                loss = crit(imgs, logits, labels, wrapped, 'arch', 1.0, finish)
                f_tot, f_sea = model.get_flops()
            if isinstance(loss, tuple):
                base, arch = loss
                total = base + arch
            else:
                base, arch, total = loss, torch.zeros(()), loss
            total = total + (base / dec_loss).data.clone() * dec_loss
            total.backward()
        finally:
            torch.rand = real_rand
            DropPath.rand = staticmethod(real_rand)
        out.update({f'{pre}.logits': logits.detach().numpy(), f'{pre}.decoder_loss': dec_loss.item(), f'{pre}.base': base.item(),
                    f'{pre}.arch': float(arch.detach()), f'{pre}.loss_total': total.item(), f'{pre}.flops_total': float(f_tot),
                    f'{pre}.flops_searched': float(f_sea)})
        for k, p in model.named_parameters():
            if p.grad is None:
                continue
            out[f'{pre}.gnorm.{k}'] = np.float64(p.grad.double().norm().item())
            if 'alpha' in k or 'score' in k:
                out[f'{pre}.grad.{k}'] = p.grad.numpy().copy()
            else:
                out[f'{pre}.gsamp.{k}'] = sample(p.grad)
        for o in opts.values():
            if o is not None:
                o.step()
        for k, p in model.named_parameters():
            out[f'{pre}.{"after" if ("alpha" in k or "score" in k) else "asamp"}.{k}'] = \
                p.detach().numpy().copy() if ('alpha' in k or 'score' in k) else sample(p)

    def snapshot(pre):
        """model + optimizer state right after a compress() call."""
        for name, mod in by_name.items():
            out[f'{pre}.switch.{name}'] = mod.switch_cell.numpy().copy()
            out[f'{pre}.flags.{name}'] = np.array([mod.finish_search, mod.execute_prune, getattr(mod, 'head_num', -1)], np.int64)
        pid = {}
        for k, p in model.named_parameters():
            pid[id(p)] = k
            out[f'{pre}.shape.{k}'] = np.array(p.shape, np.int64)
            out[f'{pre}.rg.{k}'] = np.array(p.requires_grad)
            if 'alpha' in k or 'score' in k:
                out[f'{pre}.val.{k}'] = p.detach().numpy().copy()
            else:
                out[f'{pre}.vsamp.{k}'] = sample(p)
        for on, o in opts.items():
            if o is None:
                continue
            for gi, grp in enumerate(o.param_groups):
                got = [pid[id(p)] for p in grp['params']]
                assert got == list(o.param_names[gi]), (on, gi)
                out[f'{pre}.optnames.{on}.{gi}'] = np.array(got)
                for p in grp['params']:
                    st = o.state.get(p)
                    if st:
                        k = pid[id(p)]
                        out[f'{pre}.optstep.{k}'] = np.int64(st['step'])
                        assert st['exp_avg'].shape == p.shape and st['exp_avg_sq'].shape == p.shape, k
                        small = 'alpha' in k or 'score' in k
                        out[f'{pre}.m.{k}'] = st['exp_avg'].numpy().copy() if small else sample(st['exp_avg'])
                        out[f'{pre}.v.{k}'] = st['exp_avg_sq'].numpy().copy() if small else sample(st['exp_avg_sq'])

    try:
        step('s0', False)
        for stage in (1, 2):
            for name, a in craft_alphas(stage).items():
                mod = by_name[name]
                assert tuple(mod.alpha.shape) == a.shape, (name, mod.alpha.shape, a.shape)
                mod.alpha.data.copy_(torch.from_numpy(a))
                out[f'craft{stage}.{name}'] = a
            with contextlib.redirect_stdout(io.StringIO()):
                fin, ex, opts['p'], opts['d'], opts['a'] = model.compress(thresh, opts['p'], opts['d'], opts['a'])
            out[f'c{stage}.model_flags'] = np.array([fin, ex], np.int64)
            snapshot(f'c{stage}')
            if fin:
                opts['a'] = None                                         # engine.py:207-209
            step(f's{stage}', bool(fin))
        assert fin, 'stage 2 must finish the search'
        model.eval()
        with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
            out['eval.logits_prefuse'] = wrapped(imgs)[0].numpy()
            model.fuse()
            out['eval.logits_fused'] = wrapped(imgs)[0].numpy()
        for k, p in model.named_parameters():
            out[f'fused.vsamp.{k}'] = sample(p)
        # whole-object checkpoint exactly as search.py:775-781 writes `model_fused.pth` (class paths models.layers.* /
        # models.vision_transformer.* inside): the compatibility fixture for ofb_amd.utils.install_reference_aliases
        import gzip
        buf = io.BytesIO()
        torch.save(model, buf)
        with gzip.open(os.path.join(HERE, 'mini_c_model_fused.pth.gz'), 'wb', compresslevel=9) as f:
            f.write(buf.getvalue())
    finally:
        torch.cuda.synchronize = real_sync
    path = os.path.join(HERE, f'{tag}.npz')
    np.savez_compressed(path, **out)
    shapes = {k[len('c2.shape.'):]: tuple(v) for k, v in out.items() if k.startswith('c2.shape.') and 'weight' in k and 'norm' not in k}
    print(f'{tag}: ' + ' '.join(f's{i}: total={out[f"s{i}.loss_total"]:.5f} flops={out[f"s{i}.flops_searched"]:.5f}' for i in range(3)))
    print('   final shapes:', shapes)
    print(f'   -> {os.path.getsize(path) / 1024:.0f} KiB')


# ----------------------------------------------------------------------------------------------------------------
# patch-cell compress() (vision_transformer.py:789-820): the patch-number search driven through compress.
# step -> crafted alpha_patch -> compress (two cells die) -> step -> crafted alpha_patch -> compress (one cell left: the patch
# search finishes, alpha_patch frozen) -> step.  The reference calls reduce_tensor(alpha_patch) without the try/except its module
# compress()es have (Appendix C note 4): in this one-process run it is replaced by the identity it is for a world of one.
# ----------------------------------------------------------------------------------------------------------------
def craft_alpha_patch(stage):
    if stage == 1:
        return np.array([[0.30, -6.0, 0.10, -6.0, 0.20]], np.float32)          # cells 1 and 3 die (p <= 0.2 / 5)
    return np.array([[-6.0, 0.0, 0.50, 0.0, -6.0]], np.float32)                # of the live {0, 2, 4} only cell 2 survives -> finished


def run_patch_compress_case(tag='micro_pc', batch=2, w_p=0.8, drop_path=0.0, lr=1e-3, thresh=0.2):
    cfg = O.Config(**dict(O.MICRO, patch_search=True), drop_path_rate=drop_path)
    model = build_reference(cfg, drop_path)
    names = O.module_names(cfg)
    by_name = dict(zip(names, model.searchable_modules))
    for mod in model.searchable_modules:
        mod.w_p = w_p
    imgs = torch.from_numpy(fill.images(batch))
    labels = torch.from_numpy(fill.labels(batch, cfg.num_classes))
    pnoise = torch.from_numpy(fill.patch_noise(batch, cfg.num_patches))

    def fake_rand(*shape, **kw):
        shape = shape[0] if len(shape) == 1 and isinstance(shape[0], (tuple, list)) else shape
        if tuple(shape) == (batch, cfg.num_patches):
            return pnoise.clone()
        raise RuntimeError(f'unexpected rand shape {shape}')

    groups = {'nodecay': [], 'decay': [], 'decoder_nodecay': [], 'decoder_decay': [], 'arch': []}
    gnames = {k: [] for k in groups}
    for k, p in model.named_parameters():
        if p.requires_grad:
            grp = O.optimizer_group(k, tuple(p.shape))
            groups[grp].append(p)
            gnames[grp].append(k)
    opt_p = ROPT.AdamW([{'params': groups['nodecay'], 'weight_decay': 0.}, {'params': groups['decay'], 'weight_decay': 1e-3}],
                       {0: gnames['nodecay'], 1: gnames['decay']}, lr=lr, eps=1e-8, betas=(0.9, 0.999))
    opt_d = ROPT.AdamW([{'params': groups['decoder_nodecay'], 'weight_decay': 0.}, {'params': groups['decoder_decay'], 'weight_decay': 1e-3}],
                       {0: gnames['decoder_nodecay'], 1: gnames['decoder_decay']}, lr=lr, eps=1e-8, betas=(0.9, 0.999))
    opt_a = ROPT.AdamW(groups['arch'], {0: gnames['arch']}, lr=lr, eps=1e-8, betas=(0.5, 0.999), weight_decay=1e-3)
    opts = {'p': opt_p, 'd': opt_d, 'a': opt_a}
    crit = RLOSS.OFBSearchLOSS(RLOSS.DistillationLoss(LabelSmoothingCrossEntropy(0.1), None, 'none', 0.5, 1.0),
                                torch.device('cpu'), attn_w=0.5, mlp_w=0.5, patch_w=0.5, embedding_w=0.5, flops_w=5.0)
    wrapped = _Wrap(model)
    out = dict(meta=np.array([batch, w_p, drop_path, lr, thresh], np.float64))
    real_rand, real_sync, real_reduce = torch.rand, torch.cuda.synchronize, RVT.reduce_tensor
    torch.cuda.synchronize = lambda *a, **k: None
    RVT.reduce_tensor = lambda t: t.clone()                   # world of one: the average over ranks is the tensor itself

    def step(pre):
        model.train()
        for p in model.parameters():
            p.grad = None
        torch.rand = fake_rand
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                logits, (dec_loss, _) = wrapped(imgs)
                base, arch = crit(imgs, logits, labels, wrapped, 'arch', 1.0, False)
                l_attn, l_mlp, l_patch, l_emb = model.get_sparsity_loss(torch.device('cpu'))
            total = base + arch + (base / dec_loss).data.clone() * dec_loss
            total.backward()
        finally:
            torch.rand = real_rand
        out.update({f'{pre}.logits': logits.detach().numpy(), f'{pre}.decoder_loss': dec_loss.item(), f'{pre}.base': base.item(),
                    f'{pre}.arch': arch.item(), f'{pre}.loss_total': total.item(), f'{pre}.loss_patch': float(l_patch)})
        for k, p in model.named_parameters():
            if p.grad is None:
                continue
            out[f'{pre}.gnorm.{k}'] = np.float64(p.grad.double().norm().item())
            if 'alpha' in k or 'score' in k:
                out[f'{pre}.grad.{k}'] = p.grad.numpy().copy()
        for o in opts.values():
            o.step()
        out[f'{pre}.after.alpha_patch'] = model.alpha_patch.detach().numpy().copy()

    def snapshot(pre, fin, ex):
        out[f'{pre}.model_flags'] = np.array([fin, ex], np.int64)
        out[f'{pre}.switch_patch'] = model.switch_cell_patch.numpy().copy()
        out[f'{pre}.alpha_patch'] = model.alpha_patch.detach().numpy().copy()
        out[f'{pre}.alpha_patch_rg'] = np.array(bool(model.alpha_patch.requires_grad))
        out[f'{pre}.weighted_mask_patch'] = model.weighted_mask.detach().numpy().copy()
        for name, mod in by_name.items():
            out[f'{pre}.switch.{name}'] = mod.switch_cell.numpy().copy()
            out[f'{pre}.flags.{name}'] = np.array([mod.finish_search, mod.execute_prune], np.int64)
        for k, p in model.named_parameters():
            out[f'{pre}.shape.{k}'] = np.array(p.shape, np.int64)

    try:
        step('s0')
        for stage in (1, 2):
            a = craft_alpha_patch(stage)
            model.alpha_patch.data.copy_(torch.from_numpy(a))
            out[f'craft{stage}.alpha_patch'] = a
            with contextlib.redirect_stdout(io.StringIO()):
                fin, ex, opts['p'], opts['d'], opts['a'] = model.compress(thresh, opts['p'], opts['d'], opts['a'])
            snapshot(f'c{stage}', fin, ex)
            step(f's{stage}')
    finally:
        torch.cuda.synchronize = real_sync
        RVT.reduce_tensor = real_reduce
    assert int(model.switch_cell_patch.sum()) == 1 and not model.alpha_patch.requires_grad
    path = os.path.join(HERE, f'{tag}.npz')
    np.savez_compressed(path, **out)
    print(f'{tag}: ' + ' '.join(f's{i}: total={out[f"s{i}.loss_total"]:.5f} patch={out[f"s{i}.loss_patch"]:.5f}' for i in range(3)),
          '| switches', out['c1.switch_patch'].astype(int).tolist(), out['c2.switch_patch'].astype(int).tolist(),
          f'-> {os.path.getsize(path) / 1024:.0f} KiB')


def kernel_goldens():
    """Piece-level vectors straight from reference functions."""
    imgs = torch.from_numpy(fill.images(1, tag='nt'))
    # flat border region + a constant plane exercise the clamp / count_include_pad=False paths
    imgs[0, 2, :40, :] = 0.25
    t = RVT.norm_targets(imgs, 47)
    ys = np.arange(0, 224, 9)
    np.savez_compressed(os.path.join(HERE, 'norm_targets.npz'), rows=ys, out=t[0][:, ys, :].numpy())
    print('norm_targets: done')


if __name__ == '__main__':
    torch.manual_seed(0)
    torch.set_num_threads(8)
    if len(sys.argv) > 1 and sys.argv[1] == 'compress':
        run_compress_case()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'patch_compress':  # only the patch-cell life cycle (round 3)
        run_patch_compress_case()
        sys.exit(0)
    micro = O.MICRO
    if len(sys.argv) > 1 and sys.argv[1] == 'ctor':          # only the constructor-surface cases (round 2)
        micro4 = dict(embed_dim=64, depth=2, num_heads=4, num_classes=10)
        run_case('micro_h', dict(micro4, attn_space='head'), batch=2, w_p=0.6, keep_ratio=0.9, drop_path=0.0, switches={}, full_grads=True)
        run_case('micro_c', dict(micro4, attn_space='channel'), batch=2, w_p=0.6, keep_ratio=0.9, drop_path=0.0, switches={}, full_grads=True)
        sw_p = {'patch': np.array([[False, True, True, False, True]])}
        run_case('micro_p', dict(micro, patch_search=True), batch=2, w_p=0.8, keep_ratio=0.625, drop_path=0.0, switches=sw_p, full_grads=True)
        sys.exit(0)
    sw_b = {
        'patch_embed': np.array([[0, 0] + [1] * 15], bool),
        'blocks.0.attn': np.array([[0, 1, 1, 1, 1, 1, 1]], bool),
        'blocks.1.mlp': np.array([[0, 1, 1, 0, 1, 1, 1]], bool),
    }
    run_case('micro_a', micro, batch=2, w_p=0.99, keep_ratio=0.95, drop_path=0.0, switches={}, full_grads=True)
    run_case('micro_b', micro, batch=3, w_p=0.545, keep_ratio=0.85, drop_path=0.1, switches=sw_b)
    micro4 = dict(embed_dim=64, depth=2, num_heads=4, num_classes=10)
    # constructor surface beyond the default workflow: head-only / channel-only attention spaces (layers.py:424-448) and the
    # patch-number search with a live patch term in the architecture loss (vision_transformer.py:470-477, base_model.py:39-51)
    run_case('micro_h', dict(micro4, attn_space='head'), batch=2, w_p=0.6, keep_ratio=0.9, drop_path=0.0, switches={}, full_grads=True)
    run_case('micro_c', dict(micro4, attn_space='channel'), batch=2, w_p=0.6, keep_ratio=0.9, drop_path=0.0, switches={}, full_grads=True)
    sw_p = {'patch': np.array([[False, True, True, False, True]])}
    run_case('micro_p', dict(micro, patch_search=True), batch=2, w_p=0.8, keep_ratio=0.625, drop_path=0.0, switches=sw_p, full_grads=True)
    run_case('tiny_a', dict(O.DEIT_TINY, num_classes=2), batch=2, w_p=0.99, keep_ratio=0.95, drop_path=0.1, switches={})
    sw_s = {'blocks.3.attn': np.array([[1, 1, 1, 1, 1, 1, 0], [1, 1, 0, 1, 1, 1, 1], [0, 1, 1, 1, 1, 1, 1]], bool)}
    run_case('small_a', dict(O.DEIT_SMALL, num_classes=1000), batch=2, w_p=0.7, keep_ratio=0.9, drop_path=0.1, switches=sw_s)
    kernel_goldens()
    run_compress_case()
    run_patch_compress_case()
