"""Host-side mirror of the reference's `models/vision_transformer.py` (+ `models/base_model.py`) for the OFB
search path: MIMVisionTransformer / MAEBlock (search model) and VisionTransformer / Block (pruned-subnet
finetune model), driving the fused HIP ops.  Class / attribute / state_dict names follow the reference.
"""
import math
from functools import partial

import torch
import torch.nn as nn

from . import hip, ops
from .layers import (Attention, DropPath, drop_path_scale, LayerNorm, MAEPatchEmbed, MAESparseAttention, MAESparseMlp, Mlp, ModuleInjection,
                     PatchEmbed, reduce_tensor, trunc_normal_)


def norm_targets(targets, patch_size):
    """reference vision_transformer.py:121-141 (box size must be 47, the only value the reference uses)."""
    assert patch_size % 2 == 1
    return ops.norm_targets(targets, patch_size)


def _init_vit_weights(m, n='', head_bias=0.):
    """DeiT-style init (reference :953-984, non-jax branch)."""
    if isinstance(m, nn.Linear):
        if n.startswith('head'):
            nn.init.zeros_(m.weight)
            nn.init.constant_(m.bias, head_bias)
        else:
            trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.zeros_(m.bias)
    elif isinstance(m, (nn.LayerNorm, LayerNorm)):
        nn.init.zeros_(m.bias)
        nn.init.ones_(m.weight)


class MAEBaseModel(nn.Module):
    """reference models/base_model.py: loss / bookkeeping API over `searchable_modules`."""

    def __init__(self):
        super().__init__()
        self.searchable_modules = []

    def give_alphas(self):
        attn, mlp, embed = [], [], []
        patch = self.alpha_patch.detach().cpu().reshape(-1).tolist()
        for m in self.searchable_modules:
            a = m.alpha.detach().cpu().reshape(-1).tolist()
            (attn if hasattr(m, 'num_heads') else embed if hasattr(m, 'embed_ratio_list') else mlp).append(a)
        return attn, mlp, patch, embed

    def get_flops(self):
        raise NotImplementedError

    # Generic forms of the two architecture losses over the per-module API (get_alpha / get_weight / get_flops), for subclasses that
    # only provide that API.  MIMVisionTransformer overrides both: there all 25 modules' terms come out of ONE gate kernel.
    def get_flops_loss(self, target_flops):
        """reference base_model.py:31-35 (without its per-call print, SURVEY D-10)."""
        total, searched = self.get_flops()
        return (((searched - target_flops) / total) ** 2).mean()

    @staticmethod
    def _one_hot_terms(alpha, switch, entropy, var, per_cell):
        """entropy + tan(pi/2 - pi * sigma / sigma_target) of softmax(alpha[live]) (base_model.py:41-50, 62-72)."""
        n = int(switch.sum())
        pr = torch.softmax(alpha[switch], dim=-1)
        loss = -(pr * pr.log()).sum() if entropy else alpha.new_zeros(())
        if var:
            sigma = ((pr - pr.mean()) ** 2).sum() / (1.0 - 1.0 / n)
            loss = loss + torch.tan(math.pi / 2 - math.pi * sigma) / (n if per_cell else 1)
        return loss

    def get_sparsity_loss(self, device, entropy=True, var=True, norm=True):
        """reference base_model.py:37-86: (attention, MLP, patch, embedding) adaptive one-hot terms."""
        zero = torch.zeros((), device=device)
        sw_p = self.switch_cell_patch.to(device)
        loss_patch = self._one_hot_terms(self.alpha_patch, sw_p, True, True, False) if int(sw_p.sum()) != 1 else zero.clone()
        sums = {'attn': zero.clone(), 'mlp': zero.clone(), 'embed': zero.clone()}
        for m in self.searchable_modules:
            alpha, switch = m.get_alpha()
            if int(switch.sum()) == 1:
                continue
            loss = self._one_hot_terms(alpha, switch, entropy, var, True)
            is_attn = hasattr(m, 'num_heads')
            if norm:
                loss = loss + m.get_weight()[1].sum() * (4e-4 if is_attn else 1e-4)
            key = 'attn' if is_attn else ('embed' if hasattr(m, 'embed_ratio_list') else 'mlp')
            sums[key] = sums[key] + loss
        return sums['attn'], sums['mlp'], loss_patch, sums['embed']

    def correct_require_grad(self, w_head, w_mlp, w_patch, w_embedding):
        for m in self.searchable_modules:
            is_attn, is_embed = hasattr(m, 'num_heads'), hasattr(m, 'embed_ratio_list')
            if (is_attn and w_head == 0) or (is_embed and w_embedding == 0) or (not is_attn and not is_embed and w_mlp == 0):
                m.alpha.requires_grad = False
        if w_patch == 0:
            self.alpha_patch.requires_grad = False

    def get_params(self):
        """reference base_model.py:104-109: (trainable parameters, the same with every searchable module counted at its active size)."""
        total = sum(p.numel() for p in self.parameters() if p.requires_grad)
        searched = total
        for m in self.searchable_modules:
            cnt = m.get_params_count()
            searched = searched - cnt[0] + cnt[1]
        return total, searched.item() if isinstance(searched, torch.Tensor) else searched


class Block(nn.Module):
    """plain pre-LN block of the finetune model (reference :144-170); residual adds and their gradients are
    fused into the branch GEMMs / LayerNorm backward."""

    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, qk_scale=None, drop=0., attn_drop=0., drop_path=0.,
                 act_layer=nn.GELU, norm_layer=LayerNorm):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop, proj_drop=drop)
        self.drop_path = DropPath(drop_path) if drop_path > 0. else nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)

    def _row_scale(self, x, u=None):
        return drop_path_scale(self.drop_path, x.shape[0], x.device, u, tokens=x.shape[1])

    def forward(self, x):
        y, xr = ops.layer_norm_fork(x, self.norm1.weight, self.norm1.bias, self.norm1.eps)
        x = self.attn._branch(y, xr, None, self._row_scale(x), self.attn.num_heads)
        y, xr = ops.layer_norm_fork(x, self.norm2.weight, self.norm2.bias, self.norm2.eps)
        return self.mlp._branch(y, xr, None, self._row_scale(x))

    def get_flops(self, num_patches):
        """reference vision_transformer.py:162-170."""
        return 2 * self.norm1.normalized_shape[0] * num_patches + self.attn.get_flops(num_patches) + self.mlp.get_flops(num_patches)


class MAEBlock(nn.Module):
    """reference :173-220.  While the embed search is live the reference normalises the residual stream itself
    (x <- LN1(x); x <- x + dp(attn(x)); x <- LN2(x); x <- x + dp(mlp(x)), :193-201); afterwards it is the usual
    pre-LN block (:203-204)."""

    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, qk_scale=None, drop=0., attn_drop=0., drop_path=0.,
                 act_layer=nn.GELU, norm_layer=LayerNorm, head_search=False, channel_search=False, attn_search=True,
                 mlp_search=True):
        super().__init__()
        self.in_feature = dim
        self.norm1 = norm_layer(dim)
        attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop, proj_drop=drop)
        self.attn = ModuleInjection.make_searchable_maeattn(attn, head_search, channel_search, attn_search)
        self.drop_path = DropPath(drop_path) if drop_path > 0. else nn.Identity()
        self.norm2 = norm_layer(dim)
        mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)
        self.mlp = ModuleInjection.make_searchable_maemlp(mlp, mlp_search)

    def __getstate__(self):
        """checkpoint / deepcopy view: the embed staircase handed through the blocks is an autograd output of the last forward."""
        d = self.__dict__.copy()
        if isinstance(d.get('weighted_mask_embed'), torch.Tensor):
            d['weighted_mask_embed'] = d['weighted_mask_embed'].detach()
        return d

    def run(self, x, replace_stream, g_attn, g_mlp, rs_attn, rs_mlp):
        heads = self.attn.active_heads() if hasattr(self.attn, 'active_heads') else self.attn.num_heads
        if replace_stream:
            y = ops.layer_norm(x, self.norm1.weight, self.norm1.bias, self.norm1.eps)
            x = self.attn._branch(y, None, g_attn, rs_attn, heads)
            y = ops.layer_norm(x, self.norm2.weight, self.norm2.bias, self.norm2.eps)
            return self.mlp._branch(y, None, g_mlp, rs_mlp)
        y, xr = ops.layer_norm_fork(x, self.norm1.weight, self.norm1.bias, self.norm1.eps)
        x = self.attn._branch(y, xr, g_attn, rs_attn, heads)
        y, xr = ops.layer_norm_fork(x, self.norm2.weight, self.norm2.bias, self.norm2.eps)
        return self.mlp._branch(y, xr, g_mlp, rs_mlp)

    def forward(self, input):
        x, weighted_mask_embed, weighted_embed = input
        self.weighted_mask_embed = weighted_mask_embed
        replace = weighted_mask_embed is not None and bool(((weighted_mask_embed < 1) & (weighted_mask_embed > 0)).any())
        rs = lambda: drop_path_scale(self.drop_path, x.shape[0], x.device, tokens=x.shape[1])
        g_a = self.attn.current_gate() if hasattr(self.attn, 'current_gate') else None
        g_m = self.mlp.current_gate() if hasattr(self.mlp, 'current_gate') else None
        return self.run(x, replace, g_a, g_m, rs(), rs()), weighted_mask_embed

    def get_flops(self, num_patches, active_patches):
        """reference vision_transformer.py:207-220: (total, searched) MACs of this block."""
        flops = 2 * self.in_feature * num_patches
        searched = 2 * self.norm1.normalized_shape[0] * active_patches
        for m in (self.attn, self.mlp):
            if hasattr(m, 'finish_search'):
                f, sf = m.get_flops(num_patches, active_patches)
            else:
                f = sf = m.get_flops(num_patches)
            flops, searched = flops + f, searched + sf
        return flops, searched


class VisionTransformer(nn.Module):
    """pruned-subnet / plain ViT of the finetune path (reference :222-377)."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, embed_dim=768, depth=12, num_heads=12,
                 mlp_ratio=4., qkv_bias=True, qk_scale=None, representation_size=None, distilled=False, drop_rate=0.,
                 attn_drop_rate=0., drop_path_rate=0., embed_layer=PatchEmbed, norm_layer=None, act_layer=None, weight_init=''):
        super().__init__()
        if distilled or representation_size:
            raise NotImplementedError('distilled / representation_size variants are not on the OFB path')
        self.num_classes = num_classes
        self.num_features = self.embed_dim = embed_dim
        self.num_tokens = 1
        norm_layer = norm_layer or partial(LayerNorm, eps=1e-6)
        self.patch_embed = embed_layer(img_size=img_size, patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim)
        num_patches = self.patch_embed.num_patches
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.dist_token = None
        self.pos_embed = nn.Parameter(torch.zeros(1, num_patches + 1, embed_dim))
        self.pos_drop = nn.Dropout(p=drop_rate)
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, depth)]
        self.blocks = nn.Sequential(*[
            Block(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale, drop=drop_rate,
                  attn_drop=attn_drop_rate, drop_path=dpr[i], norm_layer=norm_layer, act_layer=act_layer or nn.GELU)
            for i in range(depth)])
        self.norm = norm_layer(embed_dim)
        self.pre_logits = nn.Identity()
        self.head = nn.Linear(self.num_features, num_classes) if num_classes > 0 else nn.Identity()
        self.head_dist = None
        trunc_normal_(self.pos_embed, std=.02)
        trunc_normal_(self.cls_token, std=.02)
        self.apply(_init_vit_weights)

    def _init_weights(self, m):
        _init_vit_weights(m)

    @torch.jit.ignore
    def no_weight_decay(self):
        return ['pos_embed', 'cls_token', 'dist_token']

    def get_classifier(self):
        return self.head

    def reset_classifier(self, num_classes, global_pool=''):
        """reference vision_transformer.py:315-319 (no distillation head on this path)."""
        self.num_classes = num_classes
        self.head = nn.Linear(self.embed_dim, num_classes) if num_classes > 0 else nn.Identity()

    def get_flops(self):
        """reference vision_transformer.py:360-377: MACs of the (pruned) subnet from its CURRENT shapes; finetune.py:426 logs it."""
        P, N, D = self.patch_embed.patch_size[0], self.patch_embed.num_patches, self.patch_embed.proj.out_channels
        blocks = sum(b.get_flops(N) for b in self.blocks)
        return N * D * 3 * P ** 2 + blocks + D * self.num_classes

    def forward_features(self, x):
        hip.begin_forward()                                  # planes of the weights are re-made by the first GEMM of this pass
        pe = self.patch_embed
        x = ops.PatchEmbedTokens.apply(x, pe.proj.weight, pe.proj.bias, None, self.pos_embed, self.cls_token, None, None,
                                       pe.patch_size[0])
        x = self.blocks(x)
        x = ops.layer_norm(x, self.norm.weight, self.norm.bias, self.norm.eps)
        return x[:, 0].contiguous()

    def forward(self, x):
        x = self.forward_features(x)
        return ops.Linear.apply(x, self.head.weight, self.head.bias)


class MIMVisionTransformer(MAEBaseModel):
    """The OFB search model (reference :380-950): bi-mask gated DeiT + PMIM reconstruction branch."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, embed_dim=768, depth=12, num_heads=12,
                 mlp_ratio=4., qkv_bias=True, qk_scale=None, representation_size=None, distilled=False, drop_rate=0.,
                 attn_drop_rate=0., drop_path_rate=0., embed_layer=PatchEmbed, norm_layer=None, act_layer=None, weight_init='',
                 head_search=False, channel_search=False, attn_search=True, mlp_search=True, embed_search=True,
                 patch_search=True, mae=True, norm_pix_loss=False, mask_ratio=1.0):
        super().__init__()
        if distilled or representation_size:
            raise NotImplementedError('distilled / representation_size variants are not on the OFB path')
        self.num_classes = num_classes
        self.num_features = self.embed_dim = embed_dim
        self.num_tokens = 1
        norm_layer = norm_layer or partial(LayerNorm, eps=1e-6)
        self.patch_size, self.in_chans = patch_size, in_chans
        self.finish_search = self.execute_prune = self.fused = False

        pe = embed_layer(img_size=img_size, patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim)
        self.patch_embed = ModuleInjection.make_searchable_patchembed(pe, embed_search)
        self.num_patches = self.patch_embed.num_patches
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.dist_token = None
        self.pos_embed = nn.Parameter(torch.zeros(1, self.num_patches + 1, embed_dim))
        self.pos_drop = nn.Dropout(p=drop_rate)
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, depth)]
        self.blocks = nn.ModuleList([
            MAEBlock(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale, drop=drop_rate,
                     attn_drop=attn_drop_rate, drop_path=dpr[i], norm_layer=norm_layer, act_layer=act_layer or nn.GELU,
                     head_search=head_search, channel_search=channel_search, attn_search=attn_search, mlp_search=mlp_search)
            for i in range(depth)])
        self.norm = norm_layer(embed_dim)
        self.pre_logits = nn.Identity()
        self.head = nn.Linear(self.num_features, num_classes) if num_classes > 0 else nn.Identity()
        self.head_dist = None

        # patch-number search space (reference :470-485): five keep ratios with a cell parameter each, or a single cell.  The cells
        # act on the forward only through the FIRST live ratio (patch_masking, :593) and on the loss through the patch term of the
        # adaptive one-hot loss (base_model.py:39-51); compress() prunes them like any other cell row (:789-820).
        self.mae = mae
        if patch_search:
            import numpy as np
            self.patch_ratio_list = np.linspace(0.5, 1.0, 5).tolist()
            self.alpha_patch = nn.Parameter(torch.rand(1, len(self.patch_ratio_list)))
        else:
            self.patch_ratio_list = [mask_ratio]
            self.alpha_patch = nn.Parameter(torch.tensor([[1.]]))
        self.switch_cell_patch = self.alpha_patch > 0
        self.patch_search_mask = torch.zeros(len(self.patch_ratio_list), 1, self.num_patches, 1)
        for i, r in enumerate(self.patch_ratio_list):
            self.patch_search_mask[i, :, :int(self.num_patches * r), :] = 1
        if self.mae:
            self.mask_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
            self.decoder = nn.Sequential(nn.Conv2d(self.num_features, patch_size ** 2 * 3, kernel_size=1), nn.PixelShuffle(patch_size))
            self.norm_pix_loss = norm_pix_loss
        else:
            self.mask_token = None

        trunc_normal_(self.pos_embed, std=.02)
        trunc_normal_(self.cls_token, std=.02)
        if self.mae:
            trunc_normal_(self.mask_token, std=.02)
        self.apply(_init_vit_weights)
        w = self.patch_embed.proj.weight.data
        nn.init.xavier_uniform_(w.view([w.shape[0], -1]))
        self._hidden0 = int(embed_dim * mlp_ratio)           # original MLP width (the FLOPs model's `total`)
        self._flops_maps = None
        self._gate_flags = (1, 1, 1)
        self._forced = None         # parity tests: dict(patch_noise=(B,L), droppath_u=(2*depth,B))
        self._gate_out = None
        self._side_stream = None

    # ---- checkpoints: whole-object pickles (search.py:671-740) and deepcopy (ModelEma) --------------
    _TRANSIENT = ('_gate_out', '_side_stream', '_flops_maps', '_forced', '_masked_ids', '_dp_keep', '_active_patch_cache')

    def __getstate__(self):
        d = self.__dict__.copy()
        for k in self._TRANSIENT:
            d[k] = None
        return d

    def __setstate__(self, state):
        """also accepts the attribute set of a model pickled by the reference (utils.install_reference_aliases)."""
        super().__setstate__(state)
        for k in self._TRANSIENT:
            self.__dict__.setdefault(k, None)
        self.__dict__.setdefault('_gate_flags', (1, 1, 1))
        if '_hidden0' not in self.__dict__:
            mlp = self.blocks[0].mlp
            self._hidden0 = getattr(mlp, 'hidden_features', mlp.fc1.out_features)

    # ---- small reference API -------------------------------------------------------------------
    def adjust_masking_ratio(self, epoch, warmup_epochs, total_epochs, min_ratio=0.75, max_ratio=0.95, method='linear'):
        if epoch <= warmup_epochs:
            self.patch_ratio_list = [max_ratio - (max_ratio - min_ratio) * epoch / warmup_epochs]

    @torch.jit.ignore
    def no_weight_decay(self):
        return ['pos_embed', 'cls_token', 'dist_token', 'scale_weight', 'mask_token', 'score']

    def freeze_decoder(self):
        if self.mask_token is not None:
            self.mask_token.requires_grad = False
        for name, p in self.named_parameters():
            if 'decoder' in name:
                p.requires_grad = False

    def reset_mask_ratio(self, mask_ratio):
        self.patch_ratio_list = [mask_ratio]

    def get_classifier(self):
        return self.head

    def reset_classifier(self, num_classes, global_pool=''):
        """reference vision_transformer.py:566-570."""
        self.num_classes = num_classes
        self.head = nn.Linear(self.embed_dim, num_classes) if num_classes > 0 else nn.Identity()

    def _init_weights(self, m):
        _init_vit_weights(m)

    def patch_masking(self, x):
        """reference vision_transformer.py:586-612 as a stand-alone call: (x with the removed patches zeroed, mask) - the fused forward
        folds the same mask into the token assembly kernel instead (`patch_masking_mask`)."""
        mask = self.patch_masking_mask(x.shape[0], x.device)
        return (x, None) if mask is None else (x * (1.0 - mask).unsqueeze(-1), mask)

    # ---- gates (all searchable modules, one launch) ---------------------------------------------
    def _live_modules(self):
        return [m for m in self.searchable_modules if not m.finish_search]

    def _compute_gates(self):
        live = self._live_modules()
        if not live:
            self._gate_out = None
            return
        plan, params = [], []
        for m in live:
            plan.append(m.gate_plan())
            params += [m.alpha, m.score]
        outs = ops.BiMaskGates.apply(plan, self._gate_flags, *params)
        n = len(live)
        for i, m in enumerate(live):
            m._set_gate_outputs(outs[i], outs[n + i], outs[2 * n + i])
        # q | k | v share an attention module's gate (layers.py:507-509): the tiled copies the qkv GEMM's column scale needs are made
        # for ALL blocks here (stack + one expanding copy) instead of one repeat launch per block; each gate carries its row along
        att = [m for m in live if hasattr(m, 'num_heads') and m._g.dim() == 2 and m._g.shape == (m.num_heads, m.head_dim)]
        if len(att) > 1 and len({m._g.numel() for m in att}) == 1:
            hd = att[0]._g.numel()
            g3_all = torch.stack([m._g.detach().reshape(-1) for m in att]).unsqueeze(1).expand(-1, 3, -1).reshape(len(att), 3 * hd)
            for i, m in enumerate(att):
                m._g._ofb_g3 = g3_all[i]
        wsum = outs[3 * n]
        scales = [m.wsum_scale() if hasattr(m, 'wsum_scale') else 1.0 for m in live]
        if any(sc != 1.0 for sc in scales):                   # head-only / channel-only attention: the FLOPs model sums the broadcast staircase
            wsum = wsum * torch.tensor(scales, device=wsum.device, dtype=wsum.dtype)
        self._gate_out = dict(live=live, wsum=wsum, spars=outs[3 * n + 1], per_module=outs[3 * n + 2])

    def _module_gate(self, m):
        """gate tensor the fused block should apply for module m (None = no gate)."""
        if not hasattr(m, 'finish_search'):
            return None
        if not m.finish_search:
            return m._g
        return None if m.fused else m.score.to(self.pos_embed.device)

    # ---- forward -------------------------------------------------------------------------------
    def patch_masking_mask(self, B, device):
        """0 keep / 1 remove mask (reference :586-612) or None when every patch is kept."""
        L = self.num_patches
        sw = self.switch_cell_patch.reshape(-1).tolist()
        # reference :593: ratios of the live cells (index i is read from switch_cell_patch[:, i], so a ratio list that
        # adjust_masking_ratio shortened to one entry pairs with cell 0); the FIRST live one sets the keep count
        len_keeps = [int(L * r) for i, r in enumerate(self.patch_ratio_list) if i < len(sw) and sw[i]]
        if len_keeps == [L]:
            return None
        len_keep = len_keeps[0]
        forced = self._forced
        noise = forced['patch_noise'] if forced and 'patch_noise' in forced else torch.rand(B, L, device=device)
        mask = torch.empty(B, L, device=device)
        self._masked_ids = torch.empty(B * (L - len_keep), device=device, dtype=torch.int32)
        hip.patch_mask(noise.contiguous(), mask, B, L, len_keep, self._masked_ids)
        return mask

    def forward_features(self, x):
        B = x.shape[0]
        dev = x.device
        hip.begin_forward()                                  # planes of the weights are re-made by the first GEMM of this pass
        self._compute_gates()
        pe = self.patch_embed
        g_e = self._module_gate(pe)
        mask = self.patch_masking_mask(B, dev) if self.training else None
        x = ops.PatchEmbedTokens.apply(x, pe.proj.weight, pe.proj.bias, g_e, self.pos_embed, self.cls_token,
                                       self.mask_token if mask is not None else None, mask, self.patch_size)
        # the reference tests the embed staircase for entries strictly inside (0,1) on the device every block
        # (:193); that is equivalent to "more than one embed cell is still on", which is host state.
        replace = (not pe.finish_search) and int(pe.switch_cell.sum()) > 1 if hasattr(pe, 'switch_cell') else False
        # what the reference hands through the blocks as `weighted_mask_embedding` (:617-624) and every module keeps for its FLOPs /
        # parameter model (layers.py:735-766,1032-1040): the restored embed staircase while that search is live, else the kept mask
        if hasattr(pe, 'switch_cell'):
            wme = pe._wr_view() if not pe.finish_search else getattr(pe, 'weighted_mask', None)
            for blk in self.blocks:
                blk.__dict__['weighted_mask_embed'] = blk.attn.__dict__['weighted_mask_embed'] = blk.mlp.__dict__['weighted_mask_embed'] = wme
        depth = len(self.blocks)
        rates = [float(getattr(b.drop_path, 'drop_prob', 0.0)) for b in self.blocks]
        u = None
        if self.training and any(r > 0 for r in rates):
            forced = self._forced
            u = forced['droppath_u'] if forced and 'droppath_u' in forced else torch.rand(2 * depth, B, device=dev)
        call = 0
        scales = None
        if u is not None:
            # timm DropPath keep / keep_prob factors of all residual branches in one shot: the uniforms are consumed in call order
            # (attn_i, mlp_i) by the blocks whose rate is > 0; one factor per SAMPLE, the kernels index it by token // tokens
            live = [i for i in range(depth) if rates[i] > 0]
            key = (tuple(rates), str(dev))
            if getattr(self, '_dp_keep', None) is None or self._dp_keep[0] != key:
                self._dp_keep = (key, torch.tensor([1.0 - rates[i] for i in live for _ in range(2)], device=dev).unsqueeze(1))
            keep = self._dp_keep[1]
            un = u[:2 * len(live)]
            if un.is_cuda and un.dtype == torch.float32 and un.is_contiguous():
                scales = torch.empty_like(un)
                hip.droppath_scales(un, keep, scales, un.shape[0], un.shape[1])        # floor(keep + u) / keep in one launch
            else:
                scales = torch.floor(keep + un) / keep
        for i, blk in enumerate(self.blocks):
            rs = [None, None]
            if scales is not None and rates[i] > 0:
                rs = [scales[call], scales[call + 1]]
                call += 2
            x = blk.run(x, replace, self._module_gate(blk.attn), self._module_gate(blk.mlp), rs[0], rs[1])
        x = ops.layer_norm(x, self.norm.weight, self.norm.bias, self.norm.eps)
        return x, mask, None, None

    def forward(self, imgs):
        latent, mask, _, _ = self.forward_features(imgs)
        B, T, D = latent.shape
        cls_rows = None
        if self.mae and mask is not None:
            L, P, Cc = T - 1, self.patch_size, self.in_chans
            # only masked patches reach the loss (M = 0 elsewhere, vision_transformer.py:724-729): decode just those rows.
            # token row of global patch id p = b*L + l is  b*(L+1) + 1 + l = p + p // L + 1
            ids = self._masked_ids
            cls_rows, z = ops.TokenTaps.apply(latent, ids)
            dec = self.decoder[0]
            rec = ops.Linear.apply(z, dec.weight.view(dec.weight.shape[0], -1), dec.bias)      # 1x1 conv, patch layout
            # norm_targets(imgs, 47) for the masked patches' pixels only: one fused kernel over their 62 x 62 windows
            if imgs.is_cuda and ids.numel() > 0 and P <= 16:
                targets = ops.norm_targets_masked(imgs, ids, L, P, 47)
            else:
                targets = norm_targets(imgs, 47)
            decoder_loss = ops.PmimLoss.apply(rec, targets, mask, ids, B, L, P, Cc)
        else:
            decoder_loss = 0.
        x = ops.Linear.apply(cls_rows if cls_rows is not None else latent[:, 0].contiguous(), self.head.weight, self.head.bias)
        return x, (decoder_loss, None)

    # ---- losses (reference base_model.py:31-86) -------------------------------------------------
    def _flops_slots(self):
        """the FLOPs model's module slots {embed, attn_0, mlp_0, ...}: which are live (index into the gate kernel's wsum)
        and the constant staircase sum of the ones compress() has finished (all ones over the kept channels).  Device
        arrays are cached until the next compress()."""
        if self._flops_maps is None:
            mods = [self.patch_embed] + [m for b in self.blocks for m in (b.attn, b.mlp)]
            live_pos = {id(m): i for i, m in enumerate(self._live_modules())}
            slot, const, heads = [], [], []
            for m in mods:
                searchable = hasattr(m, 'finish_search')
                j = live_pos.get(id(m), -1) if searchable else -1
                slot.append(j)
                const.append(0.0 if j >= 0 else float(m.score.numel() if searchable else self._plain_width(m)))
            for b in self.blocks:
                heads.append(b.attn.active_heads() if hasattr(b.attn, 'active_heads') else b.attn.num_heads)
            dev = self.pos_embed.device
            self._flops_maps = dict(slot=torch.tensor(slot, dtype=torch.int32, device=dev), const=torch.tensor(const, device=dev),
                                    heads=torch.tensor(heads, dtype=torch.int32, device=dev), n_live=len(live_pos))
        return self._flops_maps

    @staticmethod
    def _plain_width(m):
        """width a never-searched module contributes (its full q/k/v, hidden or embedding width)."""
        if hasattr(m, 'qkv'):
            return m.qkv.out_features // 3
        return m.fc1.out_features if hasattr(m, 'fc1') else m.proj.out_channels

    def _flops_cfg(self, target):
        cfg = hip.FlopsCfg()
        a0 = self.blocks[0].attn
        cfg.num_patches, cfg.embed_dim, cfg.num_heads, cfg.head_dim = self.num_patches, self.embed_dim, a0.num_heads, a0.head_dim
        cfg.hidden, cfg.patch_area = self._hidden0, self.patch_size ** 2
        cfg.num_classes, cfg.depth, cfg.target = self.num_classes, len(self.blocks), float(target)
        maps = self._flops_slots()
        cfg.ln_dim = self.norm.normalized_shape[0]
        cfg.active_heads, cfg.live_slot, cfg.wconst = maps['heads'].data_ptr(), maps['slot'].data_ptr(), maps['const'].data_ptr()
        cfg.n_live = maps['n_live']
        return cfg

    def _flops_eval(self, target):
        """(loss, out3) of the FLOPs model; differentiable w.r.t. the live staircase sums."""
        if self._gate_out is None:
            self._compute_gates()
        cfg = self._flops_cfg(target)
        n_act = self._active_patches()
        if cfg.n_live == 0:                                   # search finished: every slot is a constant
            out4 = torch.empty(4, device=self.pos_embed.device)
            if n_act is not None:
                n_act = n_act.detach().float().contiguous()
                cfg.active_patches = n_act.data_ptr()
            hip.flops_loss(None, cfg, out4, None)
            return out4[0].clone(), out4
        return ops.FlopsLoss.apply(self._gate_out['wsum'], cfg, n_act)

    def _active_patches(self):
        """`active_patches` of the FLOPs model (reference :768): self.weighted_mask.sum() once a patch-cell compress() has
        produced the probability-weighted keep mask (:811-813), else None (= num_patches).  While more than one patch cell
        is live it is a function of alpha_patch (the reference keeps the graph it built inside compress())."""
        if getattr(self, 'weighted_mask', None) is None:
            return None
        # switch_cell_patch / patch_search_mask are host-side state that changes in compress() only: the live-cell indices and their
        # patch counts go to the device ONCE per compress (no boolean-mask indexing on device tensors = no host sync per step, legal
        # inside a GraphedStep capture).  Deliberate deviation: the reference freezes weighted_mask at compress time (:811-813);
        # here the term follows alpha_patch while more than one cell is live.
        dev = self.alpha_patch.device
        # (object identity AND version counters: an in-place edit of the switch or a rebuilt / edited patch_search_mask refreshes the cache)
        key = (id(self.switch_cell_patch), self.switch_cell_patch._version, id(self.patch_search_mask), self.patch_search_mask._version, str(dev))
        cache = getattr(self, '_active_patch_cache', None)
        if cache is None or cache[0] != key:
            sw = self.switch_cell_patch.reshape(-1).to('cpu', torch.bool)
            idx = torch.nonzero(sw).reshape(-1)
            counts = self.patch_search_mask.sum((1, 2, 3)).reshape(-1).to('cpu', torch.float32)[idx]
            cache = (key, int(idx.numel()), idx.to(dev), counts.to(dev), (self.switch_cell_patch, self.patch_search_mask))   # (keeps the keyed objects alive)
            self._active_patch_cache = cache
        _, n_live, idx, counts, _ = cache
        if n_live == 1:
            return counts.sum()
        pr = torch.softmax(self.alpha_patch.reshape(-1).index_select(0, idx), 0)
        return (pr * counts).sum()

    def get_flops(self):
        _, out3 = self._flops_eval(0.0)
        total = self._total_flops()
        return total / 1e9, out3[2]

    def _total_flops(self):
        N, D, H, d = self.num_patches, self.embed_dim, self.blocks[0].attn.num_heads, self.blocks[0].attn.head_dim
        hid, P2 = self._hidden0, self.patch_size ** 2
        per_block = 2 * D * N + N * (H * d * 3 * H * d) + 3 * N * H * d + H * N * d * N + H * N * N + 5 * H * N * N \
            + H * N * N * d + N * (H * d * H * d) + N * H * d + (2 * D * hid + D + hid) * N
        return N * D * 3 * P2 + len(self.blocks) * per_block + D * self.num_classes

    def get_flops_loss(self, target_flops):
        loss, _ = self._flops_eval(target_flops)
        return loss

    def get_sparsity_loss(self, device, entropy=True, var=True, norm=True):
        flags = (int(entropy), int(var), int(norm))
        if self._gate_out is None or flags != self._gate_flags:
            self._gate_flags = flags
            self._compute_gates()
        zero = torch.zeros((), device=device)
        if self._gate_out is None:
            return zero, zero.clone(), zero.clone(), zero.clone()
        sp = self._gate_out['spars']
        return sp[0], sp[1], self._patch_cell_loss(zero), sp[2]

    def _patch_cell_loss(self, zero):
        """patch term of the adaptive one-hot loss (reference base_model.py:39-51): entropy + tan term over the live patch cells
        (no 1/n, no score term).  Five scalars: a handful of ATen ops, only when patch_search is on."""
        n = int(self.switch_cell_patch.sum())               # host-side state (it changes in compress() only)
        if n == 1:
            return zero
        sw = self.switch_cell_patch.to(self.alpha_patch.device)
        pr = torch.softmax(self.alpha_patch[sw], dim=-1)
        sigma = ((pr - pr.mean()) ** 2).sum() / (1.0 - 1.0 / n)
        return -(pr * pr.log()).sum() + torch.tan(math.pi / 2 - math.pi * sigma)

    def _compress_patch_cells(self, thresh):
        """reference :789-820: prune patch cells whose probability fell to <= thresh / n_live.  Returns (finished, executed)."""
        from .layers import plan_cell_pruning
        from .dp import average_scalars
        sw = self.switch_cell_patch.detach().to('cpu', torch.bool)
        if int(sw.sum()) == 1:
            self.alpha_patch.requires_grad = False
            return True, False
        a = average_scalars([self.alpha_patch.data])[0].detach().to('cpu', torch.float32)
        thr = thresh / int(sw.sum())
        prob = plan_cell_pruning(a, sw, thr)
        if prob is None:
            return False, False
        sw = prob > thr
        old = self.alpha_patch
        self.switch_cell_patch = sw
        self.alpha_patch = nn.Parameter(torch.where(sw, a, torch.zeros_like(a)).to(old.device), requires_grad=old.requires_grad)
        alpha = torch.softmax(torch.where(sw, a, torch.full_like(a, float('-inf'))).reshape(-1), 0).reshape_as(a)
        self.weighted_mask = sum(alpha[0, j] * self.patch_search_mask[j] for j in range(alpha.shape[1]) if bool(sw[0, j]))
        finished = int(sw.sum()) == 1
        if finished:
            self.alpha_patch.requires_grad = False
        return finished, True

    def _cut_embedding(self, keep, optimizer_params, optimizer_decoder):
        """every consumer of the embedding width outside the searchable modules follows a cut of the patch embedding
        (reference vision_transformer.py:838-907): tokens, all LayerNorms, classifier input, PMIM decoder input."""
        def swap(owner, attr, dim, opt, name, group, opt_dim):
            old = getattr(owner, attr)
            new = nn.Parameter(hip.index_select(old.data, keep, dim), requires_grad=old.requires_grad)
            setattr(owner, attr, new)
            if opt is not None and old.requires_grad:
                opt.update(old, new, name, group, keep, opt_dim)

        width = keep.numel()
        for attr in (['mask_token'] if self.mask_token is not None else []) + ['cls_token', 'pos_embed']:
            swap(self, attr, -1, optimizer_params, attr, 0, -1)
        norms = [(self.norm, 'norm')]
        for i, blk in enumerate(self.blocks):
            norms += [(blk.norm1, f'blocks.{i}.norm1'), (blk.norm2, f'blocks.{i}.norm2')]
        for ln, name in norms:
            ln.normalized_shape[0] = width
            swap(ln, 'weight', 0, optimizer_params, f'{name}.weight', 0, -1)
            swap(ln, 'bias', 0, optimizer_params, f'{name}.bias', 0, -1)
        if isinstance(self.head, nn.Linear):
            self.head.in_features = width
            swap(self.head, 'weight', 1, optimizer_params, 'head.weight', 1, -1)
        if self.mae:
            self.decoder[0].in_channels = width
            swap(self.decoder[0], 'weight', 1, optimizer_decoder, 'decoder.0.weight', 1, 1)

    def compress(self, thresh=0.2, optimizer_params=None, optimizer_decoder=None, optimizer_archs=None):
        """reference vision_transformer.py:785-950: prune search cells whose probability fell to <= thresh / n_live, cut
        the weights of modules whose largest options died (or that are down to one cell) and keep the three optimizers
        consistent.  Returns (finish_search, execute_prune, optimizer_params, optimizer_decoder, optimizer_archs).

        Rank-averaged alphas (collective C3) come from ONE fused all-reduce over every live module instead of one per
        module; decisions are taken on the host from that single copy, so all ranks cut identically."""
        from .dp import average_scalars
        hip.bump_weight_epoch()                                  # weights are about to be cut: drop their P-format copies
        finish_patch, execute_patch = self._compress_patch_cells(thresh)
        if not self.searchable_modules:
            self.searchable_modules = [m for m in self.modules() if hasattr(m, 'alpha')]
        names = {id(m): n for n, m in self.named_modules()}
        live = [m for m in self.searchable_modules if int(m.switch_cell.sum()) != 1]
        averaged = dict(zip((id(m) for m in live), average_scalars([m.alpha.data for m in live]))) if live else {}

        finish_embed = execute_embed = False
        keep = None
        for m in self.searchable_modules:
            if hasattr(m, 'embed_ratio_list'):
                keep, optimizer_params, optimizer_decoder, optimizer_archs = m.compress(
                    thresh, optimizer_params, optimizer_decoder, optimizer_archs, 'patch_embed', averaged.get(id(m)))
                finish_embed, execute_embed = m.finish_search, m.execute_prune
                if keep is not None:
                    self._cut_embedding(keep, optimizer_params, optimizer_decoder)
                break
        self.finish_search = finish_patch and finish_embed
        self.execute_prune = execute_patch or execute_embed
        for m in self.searchable_modules:
            if hasattr(m, 'embed_ratio_list'):
                continue
            name = names[id(m)]
            if (not m.finish_search) or m.execute_prune:
                optimizer_params, optimizer_decoder, optimizer_archs = m.compress(
                    thresh, optimizer_params, optimizer_decoder, optimizer_archs, name, averaged.get(id(m)))
            if keep is not None:
                optimizer_params, optimizer_decoder, optimizer_archs = m.compress_patchembed(
                    keep, optimizer_params, optimizer_decoder, optimizer_archs, name)
            self.finish_search &= m.finish_search
            self.execute_prune |= m.execute_prune
        self._gate_out = None                                    # shapes / cells changed: gates and FLOPs maps are stale
        self._flops_maps = None
        return self.finish_search, self.execute_prune, optimizer_params, optimizer_decoder, optimizer_archs

    def fuse(self):
        """reference vision_transformer.py:747-757: fold every frozen gate into the weights / tokens it scales."""
        assert self.finish_search == True
        self.fused = True
        hip.bump_weight_epoch()
        we = self.patch_embed.score.data.clone().unsqueeze(-2)            # (1, 1, D)
        if self.mask_token is not None:
            self.mask_token = nn.Parameter(self.mask_token.data * we)
        self.cls_token = nn.Parameter(self.cls_token.data * we)
        self.pos_embed = nn.Parameter(self.pos_embed.data * we)
        for m in self.searchable_modules:
            m.fuse()


VisionTransformerSearched = MIMVisionTransformer      # north_star alias
