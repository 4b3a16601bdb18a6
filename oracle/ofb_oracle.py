"""CPU oracle: a closed-form restatement of the Once-for-Both search-training hot path.

TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's
`cpu_baseline` leg may import this module; the product package (once-for-both_amd/)
never does and fails loudly when its HIP library is missing.

Parity pinning: the reference has no tests, so this oracle is pinned against
golden vectors produced by importing the reference itself in the build container
(tests/golden/make_golden.py -> tests/golden/*.npz; checked by tests/test_oracle_golden.py).

Everything is written functionally over a flat {name: tensor} parameter dict that uses the
reference's state_dict names, in plain torch ops on the CPU, fp64 by default.
File:line citations are into /root/reference.
"""
import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F

from . import fill


# ----------------------------------------------------------------------------------------
# configuration / search spaces (Appendix A-0)
# ----------------------------------------------------------------------------------------
@dataclass
class Config:
    embed_dim: int = 384
    depth: int = 12
    num_heads: int = 6
    num_classes: int = 1000
    mlp_ratio: float = 4.0
    img_size: int = 224
    patch_size: int = 16
    in_chans: int = 3
    ln_eps: float = 1e-6            # partial(LayerNorm, eps=1e-6), models/model.py:94
    drop_path_rate: float = 0.0

    @property
    def head_dim(self):
        return self.embed_dim // self.num_heads

    @property
    def hidden(self):
        return int(self.embed_dim * self.mlp_ratio)

    @property
    def num_patches(self):
        return (self.img_size // self.patch_size) ** 2

    # models/layers.py:450-454 (joint head x channel space)
    def attn_heads(self) -> List[int]:
        return list(range(2, self.num_heads + 1, 2))

    def attn_channels(self) -> List[int]:
        d = self.head_dim
        return [int(d * (i / d)) for i in range(d // 4, d + 1, max(d // 8, 1))]

    # models/layers.py:813-816
    def mlp_channels(self) -> List[int]:
        h = self.hidden
        return [int((i / h) * h) for i in range(h // 4, h + 1, h // 8)]

    # models/layers.py:143-146
    def embed_channels(self) -> List[int]:
        D = self.embed_dim
        return [int((i / D) * D) for i in range(D // 2, D + 1, min(D // 32, 12))]

    def dpr(self) -> List[float]:
        # torch.linspace(0, drop_path_rate, depth)  (vision_transformer.py:442)
        return [x.item() for x in torch.linspace(0, self.drop_path_rate, self.depth)]


DEIT_TINY = dict(embed_dim=192, depth=12, num_heads=3)
DEIT_SMALL = dict(embed_dim=384, depth=12, num_heads=6)
DEIT_BASE = dict(embed_dim=768, depth=12, num_heads=12)
MICRO = dict(embed_dim=64, depth=2, num_heads=2, num_classes=10)


def param_shapes(cfg: Config) -> Dict[str, tuple]:
    """state_dict names/shapes of the search model (probed)."""
    D, H, hid, P = cfg.embed_dim, cfg.num_heads, cfg.hidden, cfg.patch_size
    s = {
        'cls_token': (1, 1, D), 'pos_embed': (1, cfg.num_patches + 1, D), 'alpha_patch': (1, 1),
        'mask_token': (1, 1, D),
        'patch_embed.alpha': (1, len(cfg.embed_channels())), 'patch_embed.score': (1, D),
        'patch_embed.proj.weight': (D, cfg.in_chans, P, P), 'patch_embed.proj.bias': (D,),
    }
    for i in range(cfg.depth):
        b = f'blocks.{i}.'
        s.update({
            b + 'norm1.weight': (D,), b + 'norm1.bias': (D,),
            b + 'attn.alpha': (len(cfg.attn_heads()), len(cfg.attn_channels())),
            b + 'attn.score': (H, cfg.head_dim),
            b + 'attn.qkv.weight': (3 * D, D), b + 'attn.qkv.bias': (3 * D,),
            b + 'attn.proj.weight': (D, D), b + 'attn.proj.bias': (D,),
            b + 'norm2.weight': (D,), b + 'norm2.bias': (D,),
            b + 'mlp.alpha': (1, len(cfg.mlp_channels())), b + 'mlp.score': (1, hid),
            b + 'mlp.fc1.weight': (hid, D), b + 'mlp.fc1.bias': (hid,),
            b + 'mlp.fc2.weight': (D, hid), b + 'mlp.fc2.bias': (D,),
        })
    s.update({'norm.weight': (D,), 'norm.bias': (D,), 'head.weight': (cfg.num_classes, D),
              'head.bias': (cfg.num_classes,),
              'decoder.0.weight': (P * P * cfg.in_chans, D, 1, 1), 'decoder.0.bias': (P * P * cfg.in_chans,)})
    return s


def formula_params(cfg: Config, dtype=torch.float64) -> Dict[str, torch.Tensor]:
    out = {}
    for k, shp in param_shapes(cfg).items():
        v = np.ones(shp, np.float32) if k == 'alpha_patch' else fill.param_value(k, shp)
        out[k] = torch.from_numpy(v).to(dtype)
    return out


def module_names(cfg: Config) -> List[str]:
    """searchable_modules order = model.modules() order filtered by hasattr(alpha) (model.py:95)."""
    names = ['patch_embed']
    for i in range(cfg.depth):
        names += [f'blocks.{i}.attn', f'blocks.{i}.mlp']
    return names


@dataclass
class SearchState:
    """Non-parameter module state of the reference (plain attributes, SURVEY 5 checkpoint row)."""
    w_p: float = 0.99                              # update_w, layers.py:169-171
    keep_ratio: float = 0.95                       # adjust_masking_ratio, vision_transformer.py:521
    switch: Dict[str, torch.Tensor] = field(default_factory=dict)   # name -> bool (A0,A1); default all on

    def cell_mask(self, name, alpha):
        sw = self.switch.get(name)
        return torch.ones_like(alpha, dtype=torch.bool) if sw is None else sw.to(torch.bool)


# ----------------------------------------------------------------------------------------
# bi-mask gate (Appendix A-1 / A-2;  layers.py:179-191, 494-509, 847-858)
# ----------------------------------------------------------------------------------------
def masked_softmax(alpha, on):
    a = torch.where(on, alpha, torch.full_like(alpha, -float('inf')))
    return torch.softmax(a.reshape(-1), 0).reshape_as(alpha)


def desc_rank(v, dim):
    """position of each entry in a stable descending sort along `dim` (argsort(argsort(desc)))."""
    order = torch.argsort(v, dim=dim, descending=True, stable=True)
    return torch.argsort(order, dim=dim, stable=True)


def bimask_gate(alpha, on, score, head_thr, chan_thr, w_p):
    """score (H,C); alpha/on (A0,A1); head_thr[A0], chan_thr[A1].
    Returns gate g, restored staircase wr, staircase wm (all (H,C)) and cell probabilities p."""
    H, C = score.shape
    p = masked_softmax(alpha, on)
    hidx = torch.arange(H).view(1, H, 1, 1)
    cidx = torch.arange(C).view(1, 1, 1, C)
    ht = torch.as_tensor(head_thr).view(-1, 1, 1, 1)
    ct = torch.as_tensor(chan_thr).view(1, 1, -1, 1)
    cells = ((hidx < ht) & (cidx < ct)).to(score.dtype)                       # (A0,H,A1,C) staircase masks
    wm = (p.view(p.shape[0], 1, p.shape[1], 1) * on.view(p.shape[0], 1, p.shape[1], 1) * cells).sum((0, 2))
    sig = torch.sigmoid(score)
    rank_c = desc_rank(score, 1)                                               # layers.py:499-500
    rank_h = desc_rank(sig.sum(1), 0)                                          # layers.py:502-504
    wr = wm[rank_h][torch.arange(H).unsqueeze(1), rank_c]                      # gather rows then channels (:505-506)
    g = (1 - w_p) * wr + w_p * sig                                             # layers.py:507
    return g, wr, wm, p


# ----------------------------------------------------------------------------------------
# dense pieces
# ----------------------------------------------------------------------------------------
def layer_norm(x, w, b, eps):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def gelu_erf(x):
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def gated_attention(x, wqkv, bqkv, wproj, bproj, g, heads, scale):
    """layers.py:488-517.  g (H,d) scales q,k,v channels; scale is the frozen 64^-0.5-style constant (D-2)."""
    B, N, _ = x.shape
    qkv = (x @ wqkv.t() + bqkv).reshape(B, N, 3, heads, -1).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * g.unsqueeze(1), qkv[1] * g.unsqueeze(1), qkv[2] * g.unsqueeze(1)
    attn = torch.softmax((q @ k.transpose(-2, -1)) * scale, dim=-1)
    ctx = (attn @ v).transpose(1, 2).reshape(B, N, -1)
    return ctx @ wproj.t() + bproj


def gated_mlp(x, w1, b1, w2, b2, g):
    """layers.py:843-865: gate applied to fc1 output before GELU(erf)."""
    return gelu_erf((x @ w1.t() + b1) * g) @ w2.t() + b2


def box_sums_47(img64, k=47):
    """zero-padded kxk box sums via 2-D prefix sums in fp64."""
    r = k // 2
    Hh, Ww = img64.shape[-2:]
    pad = F.pad(img64, (r + 1, r, r + 1, r))
    c = pad.cumsum(-1).cumsum(-2)
    return c[..., k:k + Hh, k:k + Ww] - c[..., :Hh, k:k + Ww] - c[..., k:k + Hh, :Ww] + c[..., :Hh, :Ww]


def norm_targets(imgs, k=47):
    """vision_transformer.py:121-141: (x-mu)/sqrt(max(var*cnt/(cnt-1),0)+1e-6), count_include_pad=False."""
    x = imgs.to(torch.float64)
    cnt = box_sums_47(torch.ones_like(x), k)
    mean = box_sums_47(x, k) / cnt
    sq_mean = box_sums_47(x * x, k) / cnt
    var = torch.clamp((sq_mean - mean ** 2) * (cnt / (cnt - 1)), min=0.0)
    return ((x - mean) / torch.sqrt(var + 1e-6)).to(imgs.dtype)


def keep_mask_from_noise(noise, len_keep):
    """vision_transformer.py:597-607: rank of each patch in ascending noise; first len_keep kept (0), rest removed (1)."""
    rank = torch.argsort(torch.argsort(noise, dim=1, stable=True), dim=1, stable=True)
    return (rank >= len_keep).to(noise.dtype)


def label_smoothing_ce(logits, labels, smoothing=0.1):
    """timm LabelSmoothingCrossEntropy (third-party, unpinned; standard definition)."""
    logp = torch.log_softmax(logits, -1)
    nll = -logp.gather(1, labels.view(-1, 1)).squeeze(1)
    return ((1 - smoothing) * nll + smoothing * (-logp.mean(-1))).mean()


# ----------------------------------------------------------------------------------------
# whole search-model forward (vision_transformer.py:614-745) + losses
# ----------------------------------------------------------------------------------------
def gates_for(cfg: Config, p: Dict[str, torch.Tensor], st: SearchState):
    """All 25 (g, wr, wm, prob) tuples keyed by searchable-module name."""
    out = {}
    D, H, d, hid = cfg.embed_dim, cfg.num_heads, cfg.head_dim, cfg.hidden
    a = p['patch_embed.alpha']
    out['patch_embed'] = bimask_gate(a, st.cell_mask('patch_embed', a), p['patch_embed.score'], [1],
                                     cfg.embed_channels(), st.w_p)
    for i in range(cfg.depth):
        n = f'blocks.{i}.attn'
        a = p[n + '.alpha']
        out[n] = bimask_gate(a, st.cell_mask(n, a), p[n + '.score'], cfg.attn_heads(), cfg.attn_channels(), st.w_p)
        n = f'blocks.{i}.mlp'
        a = p[n + '.alpha']
        out[n] = bimask_gate(a, st.cell_mask(n, a), p[n + '.score'], [1], cfg.mlp_channels(), st.w_p)
    return out


def search_forward(cfg: Config, p: Dict[str, torch.Tensor], st: SearchState, imgs, patch_noise=None,
                   droppath_u=None, training=True, scale=None):
    """Returns dict(logits, decoder_loss, mask, latent, x_rec, gates).  droppath_u: (2*depth, B) uniforms,
    consumed in call order attn0, mlp0, attn1, ...; keep = floor(keep_prob + u) (timm DropPath)."""
    D, H, Pz, L = cfg.embed_dim, cfg.num_heads, cfg.patch_size, cfg.num_patches
    B = imgs.shape[0]
    scale = (D // H) ** -0.5 if scale is None else scale
    gates = gates_for(cfg, p, st)
    g_e, wr_e = gates['patch_embed'][0], gates['patch_embed'][1]                # (1,D)

    # patch embed conv16/16 as a GEMM over patchified pixels (layers.py:177) + gate (:191)
    gh = cfg.img_size // Pz
    patches = imgs.reshape(B, cfg.in_chans, gh, Pz, gh, Pz).permute(0, 2, 4, 1, 3, 5).reshape(B, L, -1)
    x = (patches @ p['patch_embed.proj.weight'].reshape(D, -1).t() + p['patch_embed.proj.bias']) * g_e
    x = x + p['pos_embed'][:, 1:] * g_e                                         # vision_transformer.py:628
    mask = None
    if training:
        len_keep = int(L * st.keep_ratio)                                       # :593
        if len_keep != L:
            mask = keep_mask_from_noise(patch_noise, len_keep)                  # :597-607
            x = x * (1 - mask).unsqueeze(-1) + mask.unsqueeze(-1) * p['mask_token'] * g_e   # :608,637
    cls = ((p['cls_token'] + p['pos_embed'][:, :1]) * g_e).expand(B, -1, -1)    # :646
    x = torch.cat([cls, x], 1)
    dpr = cfg.dpr()
    call = 0

    def drop_path(y, rate):
        nonlocal call
        if rate == 0.0 or not training:
            return y
        keep = torch.floor((1 - rate) + droppath_u[call].to(y.dtype)).view(B, 1, 1)
        call += 1
        return y / (1 - rate) * keep

    # MAEBlock.forward (vision_transformer.py:189-205).  While the embed search is running some staircase
    # entries lie strictly inside (0,1) and the block takes its "reserved channel" branch, in which the
    # LayerNorm output REPLACES the residual stream:  x <- LN1(x); x <- x + dp(attn(x)); x <- LN2(x); x <- x + dp(mlp(x)).
    # (All entries are > 0 while the last embed cell is on, so "reserved" is every channel, D-4.)
    # Once every entry is 0/1 the block is the usual pre-LN residual block (:203-204).
    replace_stream = bool(((wr_e > 0) & (wr_e < 1)).any())
    for i in range(cfg.depth):
        b = f'blocks.{i}.'
        h1 = layer_norm(x, p[b + 'norm1.weight'], p[b + 'norm1.bias'], cfg.ln_eps)
        a = gated_attention(h1, p[b + 'attn.qkv.weight'], p[b + 'attn.qkv.bias'], p[b + 'attn.proj.weight'],
                            p[b + 'attn.proj.bias'], gates[b + 'attn'][0], H, scale)
        x = (h1 if replace_stream else x) + drop_path(a, dpr[i])
        h2 = layer_norm(x, p[b + 'norm2.weight'], p[b + 'norm2.bias'], cfg.ln_eps)
        m = gated_mlp(h2, p[b + 'mlp.fc1.weight'], p[b + 'mlp.fc1.bias'], p[b + 'mlp.fc2.weight'],
                      p[b + 'mlp.fc2.bias'], gates[b + 'mlp'][0])
        x = (h2 if replace_stream else x) + drop_path(m, dpr[i])
    assert bool((wr_e > 0).all()), "reserved/dropped channel split is only an identity while all staircase entries > 0"
    latent = layer_norm(x, p['norm.weight'], p['norm.bias'], cfg.ln_eps)        # :663-668

    out = dict(latent=latent, mask=mask, gates=gates)
    if mask is not None:                                                        # PMIM branch :719-729
        z = latent[:, 1:]
        rec = z @ p['decoder.0.weight'].reshape(-1, D).t() + p['decoder.0.bias']          # 1x1 conv, (B,L,P*P*3)
        # PixelShuffle(P): channel o = c*P*P + i*P + j -> pixel (c, P*py+i, P*px+j)
        x_rec = rec.reshape(B, gh, gh, cfg.in_chans, Pz, Pz).permute(0, 3, 1, 4, 2, 5).reshape(B, cfg.in_chans, cfg.img_size, cfg.img_size)
        Mpix = mask.view(B, gh, gh).repeat_interleave(Pz, 1).repeat_interleave(Pz, 2).unsqueeze(1)
        t = norm_targets(imgs, 47)
        out['x_rec'] = x_rec
        out['targets'] = t
        out['decoder_loss'] = ((t - x_rec).abs() * Mpix).sum() / (Mpix.sum() + 1e-5) / cfg.in_chans
    else:
        out['decoder_loss'] = 0.0
    out['logits'] = latent[:, 0] @ p['head.weight'].t() + p['head.bias']
    return out


def sparsity_losses(cfg: Config, p, st: SearchState, gates, entropy=True, var=True, norm=True):
    """base_model.py:37-86 (patch term is 0: one patch cell).  Returns (attn, mlp, patch, embed)."""
    z = p['cls_token'].new_zeros(())
    acc = {'attn': z.clone(), 'mlp': z.clone(), 'embed': z.clone()}
    for name in module_names(cfg):
        alpha = p[name + '.alpha']
        on = st.cell_mask(name, alpha)
        n = int(on.sum())
        if n == 1:
            continue
        pr = torch.softmax(alpha[on], -1)
        loss = -(pr * pr.log()).sum() if entropy else z.clone()
        if var:
            sigma = ((pr - pr.mean()) ** 2).sum() / (1.0 - 1.0 / n)
            loss = loss + torch.tan(math.pi / 2 - math.pi * sigma) / n
        kind = 'attn' if name.endswith('attn') else ('embed' if name == 'patch_embed' else 'mlp')
        if norm:
            loss = loss + torch.sigmoid(p[name + '.score']).sum() * (4e-4 if kind == 'attn' else 1e-4)
        acc[kind] = acc[kind] + loss
    return acc['attn'], acc['mlp'], z.clone(), acc['embed']


def flops_G(cfg: Config, gates):
    """(total, searched) MACs/1e9 (vision_transformer.py:759-783, :207-220; layers.py:345-360,747-766,1032-1044)."""
    N = cfg.num_patches
    n = N                                   # active_patches: model has no weighted_mask attr -> num_patch
    D, H, d, hid, P2 = cfg.embed_dim, cfg.num_heads, cfg.head_dim, cfg.hidden, cfg.patch_size ** 2
    e = gates['patch_embed'][2].sum()
    total = N * D * 3 * P2
    searched = N * e * 3 * P2
    for i in range(cfg.depth):
        sd = gates[f'blocks.{i}.attn'][2].sum()
        hh = gates[f'blocks.{i}.mlp'][2].sum()
        total += 2 * D * N
        searched = searched + 2 * D * n                                           # norm1.normalized_shape[0] = D
        total += N * (H * d * 3 * H * d) + 3 * N * H * d + H * N * d * N + H * N * N + 5 * H * N * N + H * N * N * d \
            + N * (H * d * H * d) + N * H * d
        searched = searched + n * (e * 3 * sd) + 3 * n * sd + n * n * sd + H * n * n + 5 * H * n * n + n * n * sd \
            + n * (sd * e) + n * e
        total += (2 * D * hid + D + hid) * N
        searched = searched + (e * hh + hh * e + e + hh) * n
    total += D * cfg.num_classes
    searched = searched + e * cfg.num_classes
    return total / 1e9, searched / 1e9


def search_step_loss(cfg: Config, p, st: SearchState, imgs, labels, patch_noise, droppath_u=None, target_flops=1.0,
                     w=(0.5, 0.5, 0.0, 0.5, 5.0)):
    """engine.py:131-144 + losses.py:80-106.  Returns dict with loss_total and every component."""
    out = search_forward(cfg, p, st, imgs, patch_noise, droppath_u, training=True)
    base = label_smoothing_ce(out['logits'], labels)
    l_attn, l_mlp, l_patch, l_emb = sparsity_losses(cfg, p, st, out['gates'])
    tot_f, sea_f = flops_G(cfg, out['gates'])
    l_flops = ((sea_f - target_flops) / tot_f) ** 2
    arch = w[0] * l_attn + w[1] * l_mlp + w[2] * l_patch + w[3] * l_emb + w[4] * l_flops
    total = base + arch
    dec = out['decoder_loss']
    if not isinstance(dec, float):
        total = total + (base / dec).detach() * dec
    out.update(base=base, arch=arch, loss_attn=l_attn, loss_mlp=l_mlp, loss_embed=l_emb, loss_flops=l_flops,
               flops_total=tot_f, flops_searched=sea_f, loss_total=total)
    return out


def adamw_step(param, grad, m, v, step, lr, beta1, beta2, eps, wd):
    """optim.py:56-120 (decoupled decay first, then Adam with bias correction)."""
    param = param * (1 - lr * wd)
    m = m * beta1 + grad * (1 - beta1)
    v = v * beta2 + grad * grad * (1 - beta2)
    denom = v.sqrt() / math.sqrt(1 - beta2 ** step) + eps
    return param - (lr / (1 - beta1 ** step)) * m / denom, m, v


def optimizer_group(name: str, shape) -> str:
    """search.py:486-508: 'nodecay' | 'decay' | 'decoder_nodecay' | 'decoder_decay' | 'arch'."""
    skip = ['pos_embed', 'cls_token', 'dist_token', 'scale_weight', 'mask_token', 'score']
    if len(shape) == 1 or name.endswith('.bias') or any(s in name for s in skip):
        return 'decoder_nodecay' if 'decoder' in name else 'nodecay'
    if 'alpha' in name:
        return 'arch'
    return 'decoder_decay' if 'decoder' in name else 'decay'


# plain (pruned, un-gated) ViT forward used by the finetune path (vision_transformer.py:332-358)
def vit_forward(p, imgs, depth, heads: List[int], scale, eps=1e-6, patch=16):
    B, C, Hh, _ = imgs.shape
    gh = Hh // patch
    D = p['patch_embed.proj.weight'].shape[0]
    patches = imgs.reshape(B, C, gh, patch, gh, patch).permute(0, 2, 4, 1, 3, 5).reshape(B, gh * gh, -1)
    x = patches @ p['patch_embed.proj.weight'].reshape(D, -1).t() + p['patch_embed.proj.bias']
    x = torch.cat([p['cls_token'].expand(B, -1, -1), x], 1) + p['pos_embed']
    for i in range(depth):
        b = f'blocks.{i}.'
        h1 = layer_norm(x, p[b + 'norm1.weight'], p[b + 'norm1.bias'], eps)
        ones = torch.ones(heads[i], p[b + 'attn.qkv.weight'].shape[0] // 3 // heads[i], dtype=x.dtype)
        x = x + gated_attention(h1, p[b + 'attn.qkv.weight'], p[b + 'attn.qkv.bias'], p[b + 'attn.proj.weight'],
                                p[b + 'attn.proj.bias'], ones, heads[i], scale)
        h2 = layer_norm(x, p[b + 'norm2.weight'], p[b + 'norm2.bias'], eps)
        hid = p[b + 'mlp.fc1.weight'].shape[0]
        x = x + gated_mlp(h2, p[b + 'mlp.fc1.weight'], p[b + 'mlp.fc1.bias'], p[b + 'mlp.fc2.weight'],
                          p[b + 'mlp.fc2.bias'], torch.ones(1, hid, dtype=x.dtype))
    x = layer_norm(x, p['norm.weight'], p['norm.bias'], eps)
    return x[:, 0] @ p['head.weight'].t() + p['head.bias']
