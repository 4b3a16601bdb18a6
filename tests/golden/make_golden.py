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
        head_search=False, channel_search=False, num_classes=cfg.num_classes, drop_path_rate=drop_path,
        attn_search=True, mlp_search=True, embed_search=True, patch_search=False, mask_ratio=1.0)
    m.searchable_modules = [x for x in m.modules() if hasattr(x, 'alpha')]
    sd = {k: torch.from_numpy(np.ones(v.shape, np.float32) if k == 'alpha_patch' else fill.param_value(k, tuple(v.shape)))
          for k, v in m.state_dict().items()}
    m.load_state_dict(sd, strict=True)
    m.correct_require_grad(0.5, 0.5, 0, 0.5)
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
    model.patch_ratio_list = [keep_ratio]
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
                                    torch.device('cpu'), attn_w=0.5, mlp_w=0.5, patch_w=0.0, embedding_w=0.5, flops_w=5.0)
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
        out[f'gate.{name}.wr'] = wr.detach().reshape(mod.score.shape).numpy()
        out[f'gate.{name}.wm'] = mod.weighted_mask.detach().reshape(mod.score.shape).numpy()
        g = (1 - w_p) * wr.detach().reshape(mod.score.shape) + w_p * mod.score.detach().sigmoid()
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
    micro = O.MICRO
    sw_b = {
        'patch_embed': np.array([[0, 0] + [1] * 15], bool),
        'blocks.0.attn': np.array([[0, 1, 1, 1, 1, 1, 1]], bool),
        'blocks.1.mlp': np.array([[0, 1, 1, 0, 1, 1, 1]], bool),
    }
    run_case('micro_a', micro, batch=2, w_p=0.99, keep_ratio=0.95, drop_path=0.0, switches={}, full_grads=True)
    run_case('micro_b', micro, batch=3, w_p=0.545, keep_ratio=0.85, drop_path=0.1, switches=sw_b)
    run_case('tiny_a', dict(O.DEIT_TINY, num_classes=2), batch=2, w_p=0.99, keep_ratio=0.95, drop_path=0.1, switches={})
    sw_s = {'blocks.3.attn': np.array([[1, 1, 1, 1, 1, 1, 0], [1, 1, 0, 1, 1, 1, 1], [0, 1, 1, 1, 1, 1, 1]], bool)}
    run_case('small_a', dict(O.DEIT_SMALL, num_classes=1000), batch=2, w_p=0.7, keep_ratio=0.9, drop_path=0.1, switches=sw_s)
    kernel_goldens()
