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
    attn_space: str = 'joint'       # 'joint' (default) | 'head' (--head_search) | 'channel' (--channel_search): layers.py:424-467
    patch_search: bool = False      # alpha_patch cells over linspace(.5, 1, 5) keep ratios (vision_transformer.py:470-477)

    def patch_ratios(self) -> List[float]:
        return np.linspace(0.5, 1.0, 5).tolist()

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
        'cls_token': (1, 1, D), 'pos_embed': (1, cfg.num_patches + 1, D), 'alpha_patch': (1, 5) if cfg.patch_search else (1, 1),
        'mask_token': (1, 1, D),
        'patch_embed.alpha': (1, len(cfg.embed_channels())), 'patch_embed.score': (1, D),
        'patch_embed.proj.weight': (D, cfg.in_chans, P, P), 'patch_embed.proj.bias': (D,),
    }
    for i in range(cfg.depth):
        b = f'blocks.{i}.'
        s.update({
            b + 'norm1.weight': (D,), b + 'norm1.bias': (D,),
            b + 'attn.alpha': (len(cfg.attn_heads()) if cfg.attn_space != 'channel' else 1,
                               len(cfg.attn_channels()) if cfg.attn_space != 'head' else 1),
            b + 'attn.score': (H if cfg.attn_space != 'channel' else 1, cfg.head_dim if cfg.attn_space != 'head' else 1),
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
        v = np.ones(shp, np.float32) if (k == 'alpha_patch' and shp == (1, 1)) else fill.param_value(k, shp)
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
    finished: Dict[str, bool] = field(default_factory=dict)         # module.finish_search (set by compress)
    execute: Dict[str, bool] = field(default_factory=dict)          # module.execute_prune of the last compress
    heads: Dict[str, int] = field(default_factory=dict)             # attention head_num after a compress
    frozen: set = field(default_factory=set)                        # parameter names whose requires_grad went False
    fused: bool = False
    patch_weighted_mask: Optional[torch.Tensor] = None             # model.weighted_mask, set by a patch-cell compress() (:811-813)

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
    g = torch.ones((), dtype=x.dtype) if g is None else g.unsqueeze(1)          # fused module: gate already in the weights
    q, k, v = qkv[0] * g, qkv[1] * g, qkv[2] * g
    attn = torch.softmax((q @ k.transpose(-2, -1)) * scale, dim=-1)
    ctx = (attn @ v).transpose(1, 2).reshape(B, N, -1)
    return ctx @ wproj.t() + bproj


def gated_mlp(x, w1, b1, w2, b2, g):
    """layers.py:843-865: gate applied to fc1 output before GELU(erf)."""
    h = x @ w1.t() + b1
    return gelu_erf(h if g is None else h * g) @ w2.t() + b2


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
def search_space(cfg: Config, name: str):
    """(head thresholds, channel thresholds) of a searchable module in ORIGINAL units; compress() only ever
    drops trailing options, so a module whose alpha shrank to (a0, a1) uses the first a0 / a1 entries."""
    if name == 'patch_embed':
        return [1], cfg.embed_channels()
    if name.endswith('.attn'):
        # head-only: the channel axis of alpha / score has one entry; channel-only: the head axis (layers.py:424-448)
        return (cfg.attn_heads() if cfg.attn_space != 'channel' else [1]), (cfg.attn_channels() if cfg.attn_space != 'head' else [1])
    return [1], cfg.mlp_channels()


def gates_for(cfg: Config, p: Dict[str, torch.Tensor], st: SearchState):
    """(g, wr, wm, prob) per searchable module.  A finished module's `score` IS its gate (layers.py:192-195,
    516-528, 859-862) and its staircase is all ones over the kept channels; a fused one has no gate (g = None)."""
    out = {}
    for n in module_names(cfg):
        sc = p[n + '.score']
        if st.finished.get(n, False):
            ones = torch.ones_like(sc)
            out[n] = (None if st.fused else sc, ones, ones, None)
            continue
        a = p[n + '.alpha']
        ht, ct = search_space(cfg, n)
        g, wr, wm, pr = bimask_gate(a, st.cell_mask(n, a), sc, ht[:a.shape[0]], ct[:a.shape[1]], st.w_p)
        if n.endswith('.attn') and cfg.attn_space != 'joint':
            # the reference keeps the staircase broadcast to (H, 1, d) (mask is ones along the axis that is not searched) and
            # q, k, v are scaled by the broadcast gate (layers.py:496-509): FLOPs use the broadcast sum
            H, d = st.heads.get(n, cfg.num_heads), cfg.head_dim
            g, wr, wm = g.expand(H, d), wr.expand(H, d), wm.expand(H, d)
        out[n] = (g, wr, wm, pr)
    return out


def search_forward(cfg: Config, p: Dict[str, torch.Tensor], st: SearchState, imgs, patch_noise=None,
                   droppath_u=None, training=True, scale=None):
    """Returns dict(logits, decoder_loss, mask, latent, x_rec, gates).  droppath_u: (2*depth, B) uniforms,
    consumed in call order attn0, mlp0, attn1, ...; keep = floor(keep_prob + u) (timm DropPath)."""
    Pz, L = cfg.patch_size, cfg.num_patches
    D = p['patch_embed.proj.weight'].shape[0]                                   # shrinks when compress() cuts the embedding
    B = imgs.shape[0]
    scale = cfg.head_dim ** -0.5 if scale is None else scale                    # qk_scale is pinned at construction (layers.py:418)
    gates = gates_for(cfg, p, st)
    g_e, wr_e = gates['patch_embed'][0], gates['patch_embed'][1]                # (1,D)
    if g_e is None:
        g_e = torch.ones((), dtype=imgs.dtype)                                  # fused: gate folded into weights and tokens

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
        H = st.heads.get(b + 'attn', cfg.num_heads)
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
    """base_model.py:37-86.  Returns (attn, mlp, patch, embed)."""
    z = p['cls_token'].new_zeros(())
    acc = {'attn': z.clone(), 'mlp': z.clone(), 'embed': z.clone()}
    l_patch = z.clone()
    ap = p['alpha_patch']
    on_p = st.cell_mask('patch', ap)
    if int(on_p.sum()) != 1:                                     # :39-51: entropy + tan term (no 1/n, no score term), always on
        pr = torch.softmax(ap[on_p], -1)
        sigma = ((pr - pr.mean()) ** 2).sum() / (1.0 - 1.0 / int(on_p.sum()))
        l_patch = -(pr * pr.log()).sum() + torch.tan(math.pi / 2 - math.pi * sigma)
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
    return acc['attn'], acc['mlp'], l_patch, acc['embed']


def active_patches(cfg: Config, p, st: SearchState):
    """`active_patches` of the FLOPs model (vision_transformer.py:768): the sum of the probability-weighted keep mask that a
    patch-cell compress() leaves behind (:811-813), differentiable w.r.t. alpha_patch; None (= num_patches) before any."""
    if st is None or getattr(st, 'patch_weighted_mask', None) is None or p is None:
        return None
    a = p['alpha_patch']
    on = st.cell_mask('patch', a)
    pr = masked_softmax(a, on)
    L = cfg.num_patches
    return sum(pr[0, j] * float(int(L * r)) for j, r in enumerate(cfg.patch_ratios()) if bool(on[0, j]))


def flops_G(cfg: Config, gates, st: SearchState = None, p=None):
    """(total, searched) MACs/1e9 (vision_transformer.py:759-783, :207-220; layers.py:345-360,747-766,1032-1044).
    After compress() the LayerNorm term uses the cut embedding width (norm1.normalized_shape[0]) and the softmax /
    q@k terms the surviving head count (head_num); `total` keeps the original architecture."""
    N = cfg.num_patches
    n = active_patches(cfg, p, st)          # model.weighted_mask exists only after a patch-cell compress(), else num_patch
    n = N if n is None else n
    D, H, d, hid, P2 = cfg.embed_dim, cfg.num_heads, cfg.head_dim, cfg.hidden, cfg.patch_size ** 2
    e = gates['patch_embed'][2].sum()
    D_act = gates['patch_embed'][2].shape[-1]
    total = N * D * 3 * P2
    searched = N * e * 3 * P2
    for i in range(cfg.depth):
        sd = gates[f'blocks.{i}.attn'][2].sum()
        hh = gates[f'blocks.{i}.mlp'][2].sum()
        aH = st.heads.get(f'blocks.{i}.attn', H) if st is not None else H
        total += 2 * D * N
        searched = searched + 2 * D_act * n                                       # norm1.normalized_shape[0]
        total += N * (H * d * 3 * H * d) + 3 * N * H * d + H * N * d * N + H * N * N + 5 * H * N * N + H * N * N * d \
            + N * (H * d * H * d) + N * H * d
        searched = searched + n * (e * 3 * sd) + 3 * n * sd + n * n * sd + aH * n * n + 5 * aH * n * n + n * n * sd \
            + n * (sd * e) + n * e
        total += (2 * D * hid + D + hid) * N
        searched = searched + (e * hh + hh * e + e + hh) * n
    total += D * cfg.num_classes
    searched = searched + e * cfg.num_classes
    return total / 1e9, searched / 1e9


def search_step_loss(cfg: Config, p, st: SearchState, imgs, labels, patch_noise, droppath_u=None, target_flops=1.0,
                     w=(0.5, 0.5, 0.0, 0.5, 5.0), finish_search=False):
    """engine.py:131-144 + losses.py:80-106.  Returns dict with loss_total and every component.
    finish_search=True: the criterion returns the base loss only (losses.py:104-106)."""
    out = search_forward(cfg, p, st, imgs, patch_noise, droppath_u, training=True)
    base = label_smoothing_ce(out['logits'], labels)
    l_attn, l_mlp, l_patch, l_emb = sparsity_losses(cfg, p, st, out['gates'])
    tot_f, sea_f = flops_G(cfg, out['gates'], st, p)
    l_flops = ((sea_f - target_flops) / tot_f) ** 2
    arch = w[0] * l_attn + w[1] * l_mlp + w[2] * l_patch + w[3] * l_emb + w[4] * l_flops
    if finish_search:
        arch = arch.detach() * 0
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


# ----------------------------------------------------------------------------------------
# compress(): alpha-threshold pruning + physical slicing + optimizer-state surgery
# (vision_transformer.py:785-950, layers.py:218-338 / 559-728 / 883-1025, optim.py:122-182)
# ----------------------------------------------------------------------------------------
class OptimState:
    """Per-parameter AdamW state of the three reference optimizers, keyed by parameter NAME.
    groups: {'p0': [...], 'p1': [...], 'd0': [...], 'd1': [...], 'a0': [...]} = param_names of optimizer_params
    (no-decay / decay), optimizer_decoder (no-decay / decay) and optimizer_archs."""
    HYPER = {'p0': (0.9, 0.0), 'p1': (0.9, 1e-3), 'd0': (0.9, 0.0), 'd1': (0.9, 1e-3), 'a0': (0.5, 1e-3)}
    GROUP_OF = {'nodecay': 'p0', 'decay': 'p1', 'decoder_nodecay': 'd0', 'decoder_decay': 'd1', 'arch': 'a0'}

    def __init__(self, p: Dict[str, torch.Tensor], frozen=()):
        self.groups = {g: [] for g in self.HYPER}
        self.state = {}
        for k, v in p.items():
            if k in frozen:
                continue
            self.groups[self.GROUP_OF[optimizer_group(k, tuple(v.shape))]].append(k)

    def group_of(self, name):
        return next(g for g, names in self.groups.items() if name in names)

    def step(self, p, grads, lr):
        """one AdamW step on every listed parameter that received a gradient (optim.py:71-73 skips grad None).
        lr: one float for the three optimizers, or {'p': .., 'd': .., 'a': ..} - an optimizer that is absent from the dict does not
        step (engine.py:207-209: optimizer_arch is None once the search has finished)."""
        for g, names in self.groups.items():
            b1, wd = self.HYPER[g]
            if isinstance(lr, dict) and g[0] not in lr:
                continue
            glr = lr[g[0]] if isinstance(lr, dict) else lr
            for k in names:
                if grads.get(k) is None:
                    continue
                st = self.state.setdefault(k, dict(step=0, m=torch.zeros_like(p[k]), v=torch.zeros_like(p[k])))
                st['step'] += 1
                p[k], st['m'], st['v'] = adamw_step(p[k], grads[k], st['m'], st['v'], st['step'], glr, b1, 0.999, 1e-8, wd)

    # optim.py:122-182 -------------------------------------------------------------------
    def take(self, name, index, dim):
        """keep `index` along `dim` of both moments (index: 1-D -> index_select, same-rank -> gather)."""
        st = self.state[name]
        for key in ('m', 'v'):
            t = st[key]
            st[key] = t.index_select(dim % t.dim(), index) if index.dim() == 1 else torch.gather(t, dim % t.dim(), index)

    def reinit(self, name, like):
        assert name in self.state                       # the reference pops the old entry (KeyError if never stepped)
        self.state[name] = dict(step=0, m=torch.zeros_like(like), v=torch.zeros_like(like))

    def drop(self, name):
        self.groups[self.group_of(name)].remove(name)
        self.state.pop(name)


def _slice_param(p, opt, name, index, dim):
    """replace p[name] by its `index` slice along `dim` and cut the optimizer moments the same way."""
    t = p[name]
    p[name] = t.index_select(dim % t.dim(), index)
    if opt is not None and name in opt.state:
        opt.take(name, index, dim)


def compress_module(cfg: Config, p, st: SearchState, opt: OptimState, name: str, thresh: float):
    """One searchable module's compress().  Returns (keep_index or None) - the embedding module's kept output channels.

    1. cells whose softmax probability among the live cells is <= thresh / n_live are switched off and their alpha
       zeroed; the alpha optimizer state restarts (layers.py:230-247).
    2. one live cell left -> the module is finished: channels/heads are cut to that cell's size by score rank, `score`
       becomes the frozen-shape gate w_p*sigmoid(score)+(1-w_p) (a trainable parameter with fresh optimizer state),
       alpha leaves the arch optimizer (:258-293).
    3. else if the largest option(s) died -> alpha / mask / switch lose their trailing rows/columns and the weights are
       cut to the largest surviving option, optimizer moments following (:294-334)."""
    alpha = p[name + '.alpha']
    on = st.cell_mask(name, alpha)
    if int(on.sum()) == 1:
        st.finished[name], st.execute[name] = True, False
        st.frozen.add(name + '.alpha')
        return None
    thr = thresh / int(on.sum())
    pr = masked_softmax(alpha.detach(), on)
    if float(pr[on].min()) > thr:
        st.execute[name] = False
        return None
    st.execute[name] = True
    on = pr > thr
    alpha = torch.where(on, alpha.detach(), torch.zeros_like(alpha))
    p[name + '.alpha'], st.switch[name] = alpha, on
    if opt is not None:
        opt.reinit(name + '.alpha', alpha)
    heads_thr, chan_thr = search_space(cfg, name)
    score = p[name + '.score'].detach()
    is_attn = name.endswith('.attn')
    idx = torch.nonzero(on)
    finished = int(on.sum()) == 1
    if finished:
        i_max, j_max = int(idx[0, 0]), int(idx[0, 1])
    elif is_attn and (int(on[:, -1].sum()) == 0 or int(on[-1, :].sum()) == 0):
        i_max, j_max = int(idx[:, 0].max()), int(idx[:, 1].max())
    elif (not is_attn) and int(on[0, -1]) == 0:
        i_max, j_max = 0, int(idx[-1, 1])
    else:
        return None                                                            # cells died inside the staircase only
    if finished:
        st.finished[name] = True
        st.frozen.add(name + '.alpha')
        if opt is not None:
            opt.drop(name + '.alpha')
    else:
        ar0, ar1 = torch.arange(i_max + 1), torch.arange(j_max + 1)
        _slice_param(p, opt, name + '.alpha', ar0, 0)
        _slice_param(p, opt, name + '.alpha', ar1, 1)
        st.switch[name] = on[:i_max + 1, :j_max + 1]
    n_chan = chan_thr[j_max]
    chan_index = torch.argsort(score, dim=1, descending=True)[:, :n_chan]      # per current head: kept channels, best first
    if is_attn:
        H_cur, d_cur = score.shape
        n_head = heads_thr[i_max]
        head_index = torch.argsort(torch.sigmoid(score).sum(-1), dim=0, descending=True)[:n_head]
        chan_index = chan_index[head_index]                                     # (n_head, n_chan)
        st.heads[name] = n_head
        new_score = torch.gather(score[head_index], 1, chan_index)
        rows = (head_index.view(-1, 1) * d_cur + chan_index).reshape(-1)        # positions inside one of q / k / v
        qkv_rows = torch.cat([rows + t * H_cur * d_cur for t in range(3)])
        if finished:
            new_score = st.w_p * torch.sigmoid(new_score) + (1 - st.w_p) * torch.ones_like(new_score)
        p[name + '.score'] = new_score
        if opt is not None:
            if finished:
                opt.reinit(name + '.score', new_score)
            else:
                opt.take(name + '.score', head_index, 0)
                opt.take(name + '.score', chan_index, 1)
        _slice_param(p, opt, name + '.qkv.weight', qkv_rows, 0)
        _slice_param(p, opt, name + '.qkv.bias', qkv_rows, 0)
        _slice_param(p, opt, name + '.proj.weight', rows, 1)
        return None
    keep = chan_index.reshape(-1)
    new_score = score[:, keep]
    if finished:
        new_score = st.w_p * torch.sigmoid(new_score) + (1 - st.w_p) * torch.ones_like(new_score)
    p[name + '.score'] = new_score
    if opt is not None:
        if finished:
            opt.reinit(name + '.score', new_score)
        else:
            opt.take(name + '.score', keep, 1)
    if name == 'patch_embed':
        _slice_param(p, opt, 'patch_embed.proj.weight', keep, 0)
        _slice_param(p, opt, 'patch_embed.proj.bias', keep, 0)
        return keep
    _slice_param(p, opt, name + '.fc1.weight', keep, 0)
    _slice_param(p, opt, name + '.fc1.bias', keep, 0)
    _slice_param(p, opt, name + '.fc2.weight', keep, 1)
    return None


def compress_patch_cells(cfg: Config, p, st: SearchState, thresh: float):
    """Patch-number part of MIMVisionTransformer.compress (vision_transformer.py:789-820).  Returns (finished, executed).
    Live patch cells whose softmax probability is <= thresh / n_live are switched off and their alpha zeroed; alpha_patch is
    replaced by a NEW tensor that no optimizer knows (the reference builds a fresh nn.Parameter and never re-registers it, so it
    stops moving); one live cell left -> the patch search is finished and alpha_patch frozen.  `st.patch_weighted_mask` receives
    the probability-weighted sum of the live cells' keep masks (:811-813)."""
    a = p['alpha_patch']
    on = st.cell_mask('patch', a)
    if int(on.sum()) == 1:
        st.frozen.add('alpha_patch')
        return True, False
    thr = thresh / int(on.sum())
    pr = masked_softmax(a.detach(), on)
    if float(pr[on].min()) > thr:
        return False, False
    on = pr > thr
    a = torch.where(on, a.detach(), torch.zeros_like(a))
    p['alpha_patch'], st.switch['patch'] = a, on
    pr = masked_softmax(a, on)
    L = cfg.num_patches
    wm = torch.zeros(1, L, 1, dtype=a.dtype)
    for j, r in enumerate(cfg.patch_ratios()):
        if bool(on[0, j]):
            m = torch.zeros(1, L, 1, dtype=a.dtype)
            m[:, :int(L * r)] = 1
            wm = wm + pr[0, j] * m
    st.patch_weighted_mask = wm
    finished = int(on.sum()) == 1
    if finished:
        st.frozen.add('alpha_patch')
    return finished, True


def patch_keep_ratio(cfg: Config, st: SearchState):
    """keep ratio the forward uses under the patch-number search: the FIRST live cell's (vision_transformer.py:593)"""
    on = st.switch.get('patch')
    ratios = cfg.patch_ratios()
    return ratios[0] if on is None else next(r for j, r in enumerate(ratios) if bool(on.reshape(-1)[j]))


def compress_model(cfg: Config, p, st: SearchState, opt: OptimState = None, thresh=0.2):
    """MIMVisionTransformer.compress (vision_transformer.py:785-950).  Returns (finish_search, execute_prune)."""
    finish_patch, execute_patch = compress_patch_cells(cfg, p, st, thresh) if cfg.patch_search else (True, False)
    if execute_patch and opt is not None:                                       # the replaced alpha_patch is unknown to the optimizer
        for names in opt.groups.values():
            if 'alpha_patch' in names:
                names.remove('alpha_patch')
        opt.state.pop('alpha_patch', None)
    keep = compress_module(cfg, p, st, opt, 'patch_embed', thresh)
    finish, execute = st.finished.get('patch_embed', False), st.execute.get('patch_embed', False)
    finish, execute = finish and finish_patch, execute or execute_patch
    if keep is not None:                                                        # every consumer of the embedding width
        for k in ['mask_token', 'cls_token', 'pos_embed', 'norm.weight', 'norm.bias']:
            _slice_param(p, opt, k, keep, -1)
        for i in range(cfg.depth):
            for k in ['norm1.weight', 'norm1.bias', 'norm2.weight', 'norm2.bias']:
                _slice_param(p, opt, f'blocks.{i}.{k}', keep, 0)
        _slice_param(p, opt, 'head.weight', keep, 1)
        _slice_param(p, opt, 'decoder.0.weight', keep, 1)
    for name in module_names(cfg)[1:]:
        if (not st.finished.get(name, False)) or st.execute.get(name, False):
            compress_module(cfg, p, st, opt, name, thresh)
        if keep is not None:                                                    # compress_patchembed, layers.py:698-712 / 994-1008
            a, b = ('qkv', 'proj') if name.endswith('.attn') else ('fc1', 'fc2')
            _slice_param(p, opt, f'{name}.{a}.weight', keep, 1)
            _slice_param(p, opt, f'{name}.{b}.weight', keep, 0)
            _slice_param(p, opt, f'{name}.{b}.bias', keep, 0)
        finish &= st.finished.get(name, False)
        execute |= st.execute.get(name, False)
    return finish, execute


def search_epoch(cfg: Config, p, st: SearchState, opt: OptimState, n_iter, batch_of, noise_of, *, epoch, accum_iter, warmup_epochs,
                 lr, lr_sched, hook=None, target_flops=1.0, w=(0.5, 0.5, 0.0, 0.5, 5.0), thresh=0.2, progressive=True,
                 max_ratio=0.95, min_ratio=0.75, finish_search=False):
    """engine.search_one_epoch (engine.py:75-219) on the functional oracle: per accumulation window the masking-ratio and w_p
    schedules at t = it / len + epoch (:101-115; vision_transformer.py:521-523, layers.py:169-171), per micro-step the search loss
    / accum_iter accumulated into the gradients (:152,169), at the window's end the three AdamW steps and the lr schedulers'
    step_update(epoch * len + it) (:170-184), the meters of MetricLogger (:186-196; global_avg = mean over the updates a meter
    received, utils.py:62-64) and, while the search is live, compress() every len // 3 // accum windows (:201-209).
    batch_of(it) -> (imgs, labels); noise_of(it) -> patch-mask noise; lr: {'p','a','d'} initial rates; lr_sched(which, gstep) -> rate;
    hook(it) runs before batch `it` is drawn (the data loader's side effects).
    Returns (stats, finish_search, execute_pruned, per-iteration dict)."""
    lr = dict(lr)
    has_arch = not finish_search
    acc, meters, per_it = {}, {}, {k: [] for k in ('base', 'arch', 'dec', 'keep_ratio', 'w_p', 'finish')}
    execute_pruned = False

    def meter(k, v):
        meters.setdefault(k, []).append(float(v))

    for it in range(n_iter):
        if hook is not None:
            hook(it)
        imgs, labels = batch_of(it)
        if it % accum_iter == 0:
            t = it / n_iter + epoch
            if progressive and t <= warmup_epochs:
                st.keep_ratio = max_ratio - (max_ratio - min_ratio) * t / warmup_epochs
            if t <= warmup_epochs:
                st.w_p = (0.1 - 0.99) / warmup_epochs * t + 0.99             # every module that has not finished its search
        leaves = {k: v.detach().clone().requires_grad_(k not in st.frozen) for k, v in p.items()}
        out = search_step_loss(cfg, leaves, st, imgs, labels, noise_of(it), None, target_flops, w, finish_search)
        (out['loss_total'] / accum_iter).backward()
        for k, v in leaves.items():
            if v.grad is not None:
                acc[k] = v.grad if k not in acc else acc[k] + v.grad
        per_it['base'].append(float(out['base'].detach())); per_it['arch'].append(float(out['arch'].detach()))
        per_it['dec'].append(float(out['decoder_loss'].detach())); per_it['keep_ratio'].append(st.keep_ratio)
        per_it['w_p'].append(st.w_p); per_it['finish'].append(float(finish_search))
        boundary = (it + 1) % accum_iter == 0
        if boundary:
            opt.step(p, acc, {k: v for k, v in lr.items() if k != 'a' or has_arch})
            acc = {}
            gstep = epoch * n_iter + it
            for which in ('p', 'a', 'd'):
                if which != 'a' or has_arch:
                    lr[which] = lr_sched(which, gstep)
        meter('loss_param', out['base'].detach()); meter('loss_total', out['loss_total'].detach()); meter('lr_param', lr['p'])
        if has_arch:
            meter('loss_arch', out['arch'].detach()); meter('lr_arch', lr['a'])
        meter('loss_decoder', out['decoder_loss'].detach()); meter('lr_decoder', lr['d'])
        if not finish_search and boundary and ((it + 1) // accum_iter) % (n_iter // 3 // accum_iter) == 0:
            fin, ex = compress_model(cfg, p, st, opt, thresh)
            execute_pruned |= bool(ex)
            if fin:
                finish_search, has_arch = True, False
    stats = {k: sum(v) / len(v) for k, v in meters.items()}
    return stats, finish_search, execute_pruned, per_it


def fuse_params(cfg: Config, p, st: SearchState):
    """MIMVisionTransformer.fuse (vision_transformer.py:747-757) + module fuse()s: fold every frozen gate into the
    weights / tokens it scales."""
    we = p['patch_embed.score'].reshape(-1)
    for k in ['mask_token', 'cls_token', 'pos_embed']:
        p[k] = p[k] * we
    p['patch_embed.proj.weight'] = p['patch_embed.proj.weight'] * we.view(-1, 1, 1, 1)
    p['patch_embed.proj.bias'] = p['patch_embed.proj.bias'] * we
    for i in range(cfg.depth):
        sc = p[f'blocks.{i}.attn.score'].reshape(-1).repeat(3)
        p[f'blocks.{i}.attn.qkv.weight'] = p[f'blocks.{i}.attn.qkv.weight'] * sc.unsqueeze(-1)
        p[f'blocks.{i}.attn.qkv.bias'] = p[f'blocks.{i}.attn.qkv.bias'] * sc
        sc = p[f'blocks.{i}.mlp.score'].reshape(-1)
        p[f'blocks.{i}.mlp.fc1.weight'] = p[f'blocks.{i}.mlp.fc1.weight'] * sc.unsqueeze(-1)
        p[f'blocks.{i}.mlp.fc1.bias'] = p[f'blocks.{i}.mlp.fc1.bias'] * sc
    st.fused = True


# ----------------------------------------------------------------------------------------
# evaluation meters (engine.py:222-290) and the FLOPs / parameter bookkeeping beside the loss
# (vision_transformer.py:144-170,360-377; layers.py:345-360,396-414,735-766,792-801,1032-1044; base_model.py:104-109)
# ----------------------------------------------------------------------------------------
def evaluate_meters(batches):
    """engine.py:234-257 / :271-290 for a list of (logits, labels) batches: `loss` is the mean over BATCHES of
    CrossEntropyLoss()(logits, labels) (MetricLogger.update(loss=...) counts one per call, :246 / :283), `acc1` / `acc5` are the
    sample-weighted means of timm's accuracy() percentages (meters updated with n = batch size, :247-248)."""
    loss_sum, hits1, hits5, n = 0.0, 0, 0, 0
    for logits, labels in batches:
        logp = torch.log_softmax(logits.double(), -1)
        loss_sum += float(-logp.gather(1, labels.view(-1, 1)).mean())
        top = torch.argsort(logits, dim=1, descending=True, stable=True)[:, :min(5, logits.shape[1])]
        hit = top == labels.view(-1, 1)
        hits1, hits5, n = hits1 + int(hit[:, 0].sum()), hits5 + int(hit.sum()), n + labels.numel()
    return {'loss': loss_sum / len(batches), 'acc1': 100.0 * hits1 / n, 'acc5': 100.0 * hits5 / n}


def plain_vit_flops(embed, blocks, num_patches=196, patch=16, num_classes=1000):
    """VisionTransformer.get_flops() (vision_transformer.py:360-377) of a plain / pruned ViT.  blocks: [(heads, head_dim, hidden)];
    per block Block.get_flops (:162-170) = LayerNorm 2 D N + Attention.get_flops (layers.py:404-414) + Mlp.get_flops (:799-801)."""
    N, D = num_patches, embed
    total = N * D * 3 * patch ** 2 + D * num_classes
    for H, d, hid in blocks:
        total += 2 * D * N
        total += N * (D * 3 * H * d) + 3 * N * H * d + H * N * d * N + H * N * N + 5 * H * N * N + H * N * N * d + N * (H * d * D) + N * D
        total += (D * hid + hid * D + hid + D) * N
    return total


def module_counts(cfg: Config, gates, name, num_patches, active_patches, heads=None):
    """(params (total, active), flops (total, active)) of one searchable module from the staircases `gates[...][2]` of the current
    forward: MAEPatchEmbed layers.py:345-360 (it also returns its active width), MAESparseAttention :735-766, MAESparseMlp :1032-1044.
    The embed staircase of the SAME forward is what the blocks receive as weighted_mask_embed (vision_transformer.py:617-624)."""
    D, H, d, hid, P2 = cfg.embed_dim, cfg.num_heads, cfg.head_dim, cfg.hidden, cfg.patch_size ** 2
    N, n = num_patches, active_patches
    e = gates['patch_embed'][2].sum()
    if name == 'patch_embed':
        tp, ap = 3 * D * P2 + D + D * 2, 3 * e * P2 + e + e * 2
        return (tp, ap, e), ((tp - 2 * D) * N + (4 * D + 1) * N, (ap - 2 * e) * N + (4 * e + 1) * N)
    wm = gates[name][2].sum()
    if name.endswith('attn'):
        aH = H if heads is None else heads
        tp, ap = D * D * 3 + D * 3 + D * D + D, e * wm * 3 + wm * 3 + wm * e + e
        tf = N * (H * d * 3 * H * d) + 3 * N * H * d + H * N * d * N + H * N * N + 5 * H * N * N + H * N * N * d + N * (H * d * H * d) + N * H * d
        af = n * (e * 3 * wm) + 3 * n * wm + n * n * wm + aH * n * n + 5 * aH * n * n + n * n * wm + n * (wm * e) + n * e
        return (tp, ap), (tf, af)
    tp, ap = 2 * D * hid + D + hid, e * wm + wm * e + e + wm
    return (tp, ap), (tp * N, ap * n)
