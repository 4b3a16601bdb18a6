"""Host-side mirror of the reference's `models/layers.py` module API on top of the HIP ops.

Same class names, constructor arguments, attribute names and state_dict keys as the reference
 so reference-style search.py / finetune.py drivers can be pointed at this package.
nn.Linear / nn.Conv2d are used purely as parameter containers; their torch forwards are never
called.  No CPU fallback: forwards need device tensors and the HIP library.
"""
import math

import torch
import torch.nn as nn

from . import hip, ops


def to_2tuple(x):
    return tuple(x) if isinstance(x, (tuple, list)) else (x, x)


def trunc_normal_(t, std=1.0):
    return nn.init.trunc_normal_(t, mean=0.0, std=std, a=-2.0, b=2.0)


def reduce_tensor(tensor):
    """average over ranks (reference models/layers.py:9-14)."""
    rt = tensor.clone()
    torch.distributed.all_reduce(rt, op=torch.distributed.ReduceOp.SUM)
    rt /= torch.distributed.get_world_size()
    return rt


class DropPath(nn.Module):
    """Stochastic depth (timm definition): per-sample Bernoulli(keep)/keep scaling of a residual branch.
    The fused blocks only ask it for the per-sample scale vector."""

    def __init__(self, drop_prob=0.0):
        super().__init__()
        self.drop_prob = float(drop_prob)

    def row_scale(self, batch, device, u=None, tokens=1):
        """per-sample keep/keep_prob factor, expanded to one entry per token for the fused branch kernels."""
        if self.drop_prob == 0.0 or not self.training:
            return None
        keep = 1.0 - self.drop_prob
        u = torch.rand(batch, device=device) if u is None else u
        s = torch.floor(keep + u) / keep
        return s.repeat_interleave(tokens) if tokens > 1 else s

    def forward(self, x):
        s = self.row_scale(x.shape[0], x.device)
        return x if s is None else x * s.view(-1, *([1] * (x.dim() - 1)))


def drop_path_scale(dp, batch, device, u=None, tokens=1):
    """row_scale for any stochastic-depth module that carries `drop_prob` (ours, or timm's DropPath inside a checkpoint
    unpickled where timm is installed); None for nn.Identity."""
    if not hasattr(dp, 'drop_prob'):
        return None
    return DropPath.row_scale(dp, batch, device, u, tokens)


class LayerNorm(nn.Module):
    """reference models/layers.py:17-102: F.layer_norm over the last dim; `normalized_shape` is a mutable list so
    compress() can shrink it."""

    def __init__(self, normalized_shape, eps=1e-5, elementwise_affine=True):
        super().__init__()
        if isinstance(normalized_shape, int):
            normalized_shape = [normalized_shape]
        self.normalized_shape = list(normalized_shape)
        self.eps = eps
        self.elementwise_affine = elementwise_affine
        if not elementwise_affine:
            raise NotImplementedError('the OFB path only uses affine LayerNorm')
        self.weight = nn.Parameter(torch.ones(*normalized_shape))
        self.bias = nn.Parameter(torch.zeros(*normalized_shape))

    def reset_parameters(self):
        nn.init.ones_(self.weight)
        nn.init.zeros_(self.bias)

    def forward(self, input):
        return ops.layer_norm(input, self.weight, self.bias, self.eps)

    def extra_repr(self):
        return f'{self.normalized_shape}, eps={self.eps}'


class PatchEmbed(nn.Module):
    """2D image -> patch tokens (reference models/layers.py:105-128)."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, norm_layer=None):
        super().__init__()
        self.img_size, self.patch_size = to_2tuple(img_size), to_2tuple(patch_size)
        self.grid_size = (self.img_size[0] // self.patch_size[0], self.img_size[1] // self.patch_size[1])
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.norm_layer = norm_layer
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=self.patch_size, stride=self.patch_size)   # parameter container
        self.norm = norm_layer(embed_dim) if norm_layer else nn.Identity()

    def _check(self, x):
        assert x.shape[2] == self.img_size[0] and x.shape[3] == self.img_size[1], \
            f"Input image size ({x.shape[2]}*{x.shape[3]}) doesn't match model ({self.img_size[0]}*{self.img_size[1]})."

    def conv_tokens(self, x, gate=None):
        """(B, L, D) conv output, optionally gated; built from the token-assembly op with zero pos/cls."""
        self._check(x)
        D = self.proj.out_channels
        zeros_pos = torch.zeros(self.num_patches + 1, D, device=x.device)
        tok = ops.PatchEmbedTokens.apply(x, self.proj.weight, self.proj.bias, gate, zeros_pos, zeros_pos[0], None, None,
                                         self.patch_size[0])
        return tok[:, 1:]

    def forward(self, x):
        return self.norm(self.conv_tokens(x).contiguous())


def _staircase_state(n_rows, n_cols, head_list, chan_list, dim_rows, dim_cols):
    """reference `mask` tensors: (A0, H, A1, d) for attention, (A1, dim) for 1-D modules."""
    if n_rows is None:
        m = torch.zeros(len(chan_list), dim_cols)
        for j, c in enumerate(chan_list):
            m[j, :c] = 1
        return m
    m = torch.zeros(len(head_list), dim_rows, len(chan_list), dim_cols)
    for i, h in enumerate(head_list):
        for j, c in enumerate(chan_list):
            m[i, :h, j, :c] = 1
    return m


class _Searchable:
    """Shared bi-mask state of the three searchable module kinds (plain attributes, exactly as the reference keeps
    them: not buffers, so they are pickled with the module but absent from state_dict)."""

    def _init_search_state(self):
        self.finish_search = False
        self.execute_prune = False
        self.fused = False
        self.w_p = 0.99

    def __getstate__(self):
        """checkpoint / deepcopy view: per-forward gate tensors (autograd outputs) are caches, not state."""
        d = self.__dict__.copy()
        d['_g'] = d['_wr'] = None
        for k in ('weighted_mask', 'weighted_mask_embed'):
            if isinstance(d.get(k), torch.Tensor):
                d[k] = d[k].detach()
        return d

    def update_w(self, cur_epoch, warmup_epochs, max=0.99, min=0.1):
        if cur_epoch <= warmup_epochs:                      # reference layers.py:169-171
            self.w_p = (min - max) / warmup_epochs * cur_epoch + max

    def get_alpha(self):
        return self.alpha, self.switch_cell.to(self.alpha.device)

    def decompress(self):
        """reference layers.py:340-343, 730-733, 1027-1030: re-open a finished module's search."""
        self.execute_prune = False
        self.alpha.requires_grad = True
        self.finish_search = False

    def _wm_sum(self):
        """sum of this module's staircase (`self.weighted_mask.sum()` of the reference's FLOPs / parameter model); the staircase is
        an output of the gate kernel of the last forward - a module that has not run yet computes its gate alone."""
        if getattr(self, 'weighted_mask', None) is None:
            self._single_gate()
        return self.weighted_mask.sum()

    def _embed_sum(self, default):
        wme = getattr(self, 'weighted_mask_embed', None)
        return default if wme is None else wme.sum()

    def gate_plan(self):
        """static + current description of this module's gate for ops.BiMaskGates."""
        raise NotImplementedError

    def _single_gate(self):
        """gate of this module alone (used when the module is called outside the fused model forward)."""
        plan = [self.gate_plan()]
        outs = ops.BiMaskGates.apply(plan, (1, 1, 1), self.alpha, self.score)
        self._set_gate_outputs(outs[0], outs[1], outs[2])
        return outs[0]

    def _set_gate_outputs(self, g, wr, wm):
        # plain (unregistered) attributes, written ~75 times per step: straight into __dict__, which is where nn.Module.__setattr__
        # puts them after its Parameter / buffer / sub-module checks
        d = self.__dict__
        d['_g'], d['_wr'], d['weighted_mask'] = g, wr, self._shape_wm(wm)

    def get_weight(self):
        """(weight_restore, prob_score) of the CURRENT forward (reference get_weight re-derives the same values)."""
        if getattr(self, '_wr', None) is None:
            self._single_gate()
        return self._wr_view(), self._prob_view()

    # ---- compress(): host decisions (reference layers.py:218-338 / 559-696 / 883-992) ----------------------
    def _averaged_alpha(self):
        a = self.alpha.data
        if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            a = reduce_tensor(a)
        return a

    def _prune_cells(self, thresh, optimizer_archs, prefix, alpha_reduced=None):
        """Cell-level part of compress().  Cells whose probability among the live cells is <= thresh / n_live die: their
        switch goes off, their alpha is zeroed and alpha's optimizer state restarts.  Returns None when the module's
        shapes stay as they are, else (finished, i_max, j_max): the module must be cut to option (i_max, j_max) of
        its search space - the single surviving cell (finished) or the largest surviving head / channel option."""
        sw = self.switch_cell.detach().to('cpu', torch.bool)
        if int(sw.sum()) == 1:
            self.finish_search, self.execute_prune = True, False
            self.alpha.requires_grad = False
            return None
        a_dev = self._averaged_alpha() if alpha_reduced is None else alpha_reduced
        a = a_dev.detach().to('cpu', torch.float32)
        thr = thresh / int(sw.sum())
        prob = plan_cell_pruning(a, sw, thr)
        if prob is None:
            self.execute_prune = False
            return None
        self.execute_prune = True
        sw = prob > thr
        old_alpha = self.alpha
        self.alpha = nn.Parameter(torch.where(sw, a, torch.zeros_like(a)).to(old_alpha.device), requires_grad=old_alpha.requires_grad)
        self.switch_cell = sw
        if optimizer_archs is not None and old_alpha.requires_grad:
            optimizer_archs.update(old_alpha, self.alpha, f'{prefix}.alpha', 0, torch.arange(self.alpha.shape[-1]), dim=-1,
                                   initialize=True)
        self._wr = self._g = None
        on = torch.nonzero(sw)
        if int(sw.sum()) == 1:
            self.finish_search = True
            if optimizer_archs is not None and self.alpha.requires_grad:
                self.alpha.requires_grad = False
                optimizer_archs.update(self.alpha, self.alpha, f'{prefix}.alpha', 0, None, dim=-1)
            self.alpha.requires_grad = False
            return True, int(on[0, 0]), int(on[0, 1])
        if int(sw[:, -1].sum()) == 0 or int(sw[-1, :].sum()) == 0:
            i_max, j_max = int(on[:, 0].max()), int(on[:, 1].max())
            old_alpha = self.alpha
            self.alpha = nn.Parameter(old_alpha.data[:i_max + 1, :j_max + 1].clone(), requires_grad=old_alpha.requires_grad)
            self.switch_cell = sw[:i_max + 1, :j_max + 1].clone()
            if optimizer_archs is not None and old_alpha.requires_grad:
                optimizer_archs.update(old_alpha, self.alpha, f'{prefix}.alpha', 0,
                                       [torch.arange(i_max + 1), torch.arange(j_max + 1)], dim=[0, -1])
            return False, i_max, j_max
        return None

    def _swap(self, owner, attr, new_data, optimizer, name, group_idx, keep_idx, dim, initialize=False):
        """replace parameter `owner.attr` by `new_data` (same requires_grad) and let the optimizer follow."""
        old = getattr(owner, attr)
        new = nn.Parameter(new_data, requires_grad=old.requires_grad)
        setattr(owner, attr, new)
        if optimizer is not None and old.requires_grad:
            optimizer.update(old, new, name, group_idx, keep_idx, dim, initialize=initialize)
        return new


def plan_cell_pruning(alpha, switch, thr):
    """Pure host rule of compress() (reference layers.py:230-239): softmax over the live cells of `alpha`; returns the
    probabilities (0 for dead cells) when at least one live cell is <= thr, else None (nothing to prune)."""
    masked = torch.where(switch, alpha, torch.full_like(alpha, float('-inf')))
    prob = torch.softmax(masked.reshape(-1), 0).reshape_as(alpha)
    return prob if float(prob[switch].min()) <= thr else None


def rank_cut(score, n_chan, n_head=None):
    """Which channels (and heads) survive a cut, best first (reference layers.py:268-269, 613-619): per head the
    `n_chan` highest scores; heads ranked by their summed sigmoid(score).  Host tensors; returns (head_index or None,
    chan_index [heads_kept][n_chan])."""
    score = score.detach().to('cpu', torch.float32)
    chan = torch.argsort(score, dim=1, descending=True)[:, :n_chan]
    if n_head is None:
        return None, chan
    heads = torch.argsort(score.sigmoid().sum(-1), dim=0, descending=True)[:n_head] if score.shape[0] != 1 else torch.arange(n_head)
    return heads, chan[heads]


class MAEPatchEmbed(PatchEmbed, _Searchable):
    """reference models/layers.py:131-365 (search state + embed gate)."""

    def __init__(self, patchmodule, embed_search=True):
        super().__init__(patchmodule.img_size[0], patchmodule.patch_size[0], patchmodule.proj.in_channels,
                         patchmodule.proj.out_channels, patchmodule.norm_layer)
        self._init_search_state()
        D = patchmodule.proj.out_channels
        self.embed_dim = D
        if embed_search:
            self.embed_ratio_list = [i / D for i in range(D // 2, D + 1, min(D // 32, 12))]
            self.alpha = nn.Parameter(torch.rand(1, len(self.embed_ratio_list)))
            self.switch_cell = self.alpha > 0
            self.mask = _staircase_state(None, None, None, self._chan_thr(), None, D)
            self.score = nn.Parameter(torch.rand(1, D))
            trunc_normal_(self.score, std=.2)
        else:
            self.embed_ratio_list = [1.0]
            self.alpha = nn.Parameter(torch.tensor([1.]))
            self.switch_cell = self.alpha > 0
            self.weighted_mask = self.mask = torch.ones(1, D)
            self.score = torch.ones(1, D)
            self.finish_search = True

    def _chan_thr(self):
        return [int(r * self.embed_dim) for r in self.embed_ratio_list]

    def gate_plan(self):
        A1 = self.alpha.shape[-1]                      # compress() drops trailing options
        return dict(H=1, C=self.score.shape[-1], A0=1, A1=A1, kind=2, head_thr=[1],
                    chan_thr=self._chan_thr()[:A1], norm_coef=1e-4, w_p=float(self.w_p),
                    on=[int(v) for v in self.switch_cell.reshape(-1).tolist()])

    def _shape_wm(self, wm):
        return wm.view(1, -1)

    def _wr_view(self):
        return self._wr.view(1, -1)

    def _prob_view(self):
        return self.score.sigmoid()

    def forward(self, x):
        if not self.finish_search:
            g = self._single_gate()
        elif not self.fused:
            g = self.score
        else:
            g = None
        return self.norm(self.conv_tokens(x, g).contiguous())

    def compress(self, thresh, optimizer_params, optimizer_decoder, optimizer_archs, prefix='', alpha_reduced=None):
        """reference layers.py:218-338.  Returns (keep_index | None, optimizer_params, optimizer_decoder, optimizer_archs);
        keep_index (host int64 tensor) lists the surviving embedding channels, best score first."""
        cut = self._prune_cells(thresh, optimizer_archs, prefix, alpha_reduced)
        if cut is None:
            return None, optimizer_params, optimizer_decoder, optimizer_archs
        finished, _, j = cut
        width = self._chan_thr()[j]
        _, chan = rank_cut(self.score, width)
        keep = chan.reshape(-1)
        dev = self.proj.weight.device
        sc = self.score.detach().to('cpu', torch.float32)[:, keep]
        if finished:                                   # the gate freezes into a trainable per-channel scale (:272)
            sc = self.w_p * sc.sigmoid() + (1 - self.w_p) * torch.ones_like(sc)
        else:
            self.mask = self.mask[:j + 1, :width]
        self.weighted_mask = torch.ones(1, width, device=dev)          # recomputed by the next forward while live
        self.proj.out_channels = width
        self._swap(self, 'score', sc.to(dev), optimizer_params, f'{prefix}.score', 0, keep, -1, initialize=finished)
        self._swap(self.proj, 'weight', hip.index_select(self.proj.weight.data, keep, 0), optimizer_params, f'{prefix}.proj.weight', 1, keep, 0)
        self._swap(self.proj, 'bias', hip.index_select(self.proj.bias.data, keep, 0), optimizer_params, f'{prefix}.proj.bias', 0, keep, -1)
        if self.norm_layer:
            self.norm.normalized_shape[0] = width
            self._swap(self.norm, 'weight', hip.index_select(self.norm.weight.data, keep, 0), optimizer_params, f'{prefix}.norm.weight', 0, keep, -1)
            self._swap(self.norm, 'bias', hip.index_select(self.norm.bias.data, keep, 0), optimizer_params, f'{prefix}.norm.bias', 0, keep, -1)
        return keep, optimizer_params, optimizer_decoder, optimizer_archs

    def fuse(self):
        """fold the frozen gate into the conv weights (reference layers.py:202-206); one-off, host-driven."""
        self.fused = True
        self.score.requires_grad = False
        sc = self.score.data.reshape(-1)
        self.proj.weight = nn.Parameter(self.proj.weight.data * sc.view(-1, 1, 1, 1))
        self.proj.bias = nn.Parameter(self.proj.bias.data * sc)

    def get_params_count(self):
        """reference layers.py:345-352: (total, active, active width); `active*` are tensors that depend on alpha."""
        dim1, dim2 = self.proj.in_channels, self.embed_dim
        k = self.proj.kernel_size[0] * self.proj.kernel_size[1]
        act = self._wm_sum()
        return dim1 * dim2 * k + dim2 + dim2 * 2, dim1 * act * k + act + act * 2, act

    def get_flops(self, num_patches):
        """reference layers.py:354-360."""
        total_params, active_params, act = self.get_params_count()
        total = (total_params - self.embed_dim * 2) * num_patches + (4 * self.embed_dim + 1) * num_patches
        active = (active_params - act * 2) * num_patches + (4 * act + 1) * num_patches
        return total, active

    @staticmethod
    def from_patchembed(patchmodule, embed_search=True):
        return MAEPatchEmbed(patchmodule, embed_search)


class Attention(nn.Module):
    """plain multi-head attention (reference models/layers.py:368-414); head_dim inferred from qkv width so
    pruned checkpoints work (finetune path)."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0., proj_drop=0., num_patches=197):
        super().__init__()
        self.num_heads = num_heads
        self.num_patches = num_patches
        self.head_dim = dim // num_heads
        self.qk_scale = qk_scale
        self.scale = qk_scale or self.head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        if attn_drop or proj_drop:
            raise NotImplementedError('attention/projection dropout is 0 on the OFB path')

    def _branch(self, x, resid, gate, rowscale, heads):
        return ops.attn_branch(x, resid, self.qkv.weight, self.qkv.bias, self.proj.weight, self.proj.bias, gate, rowscale,
                               heads, float(self.scale))

    def forward(self, x):
        zero = torch.zeros_like(x)
        return self._branch(x, zero, None, None, self.num_heads)

    def get_params_count(self):
        """reference layers.py:396-402 (current, possibly pruned, shapes)."""
        total = self.qkv.in_features * self.qkv.out_features + self.qkv.out_features
        return total + self.proj.in_features * self.proj.out_features + self.proj.out_features

    def get_flops(self, num_patches):
        """reference layers.py:404-414: MACs of one pass over `num_patches` tokens."""
        H, N = self.num_heads, num_patches
        d = self.qkv.out_features // H // 3
        e = self.proj.out_features
        flops = N * (e * (3 * H * d)) + 3 * N * H * d           # qkv
        flops += H * N * d * N + H * N * N                       # q @ k
        flops += 5 * H * N * N                                   # softmax
        flops += H * N * N * d                                   # attn @ v
        return flops + N * (H * d * e) + N * e                   # proj


class MAESparseAttention(Attention, _Searchable):
    """reference models/layers.py:416-771: joint head x channel (default), head-only or channel-only bi-mask."""

    def __init__(self, attn_module, head_search=False, channel_search=False, attn_search=True):
        super().__init__(attn_module.qkv.in_features, attn_module.num_heads, True, attn_module.scale,
                         attn_module.attn_drop.p, attn_module.proj_drop.p)
        self._init_search_state()
        H, d = self.num_heads, self.head_dim
        if attn_search:
            heads = list(range(2, H + 1, 2))
            ratios = [i / d for i in range(d // 4, d + 1, max(d // 8, 1))]
            # restricted spaces (reference layers.py:424-448): head-only keeps ONE channel option (all of d) and a per-head
            # score (H, 1); channel-only keeps ONE head option (all heads) and a per-channel score (1, d) shared by the
            # heads.  The gate kernel sees them as (H, 1) / (1, d) modules; the gate is broadcast to (H, d) where q, k, v are
            # scaled (layers.py:496-509 keeps weighted_mask as the broadcast (H, 1, d)).
            self.space = 'head' if head_search else ('channel' if channel_search else 'joint')
            if head_search:
                self.head_num_list = heads
                self.alpha = nn.Parameter(torch.rand(len(heads), 1))
                self.score = nn.Parameter(torch.rand(H, 1))
                self.mask = _staircase_state(len(heads), 1, heads, [d], H, d)
            elif channel_search:
                self.qkv_channel_ratio_list = ratios
                self.alpha = nn.Parameter(torch.rand(1, len(ratios)))
                self.score = nn.Parameter(torch.rand(1, d))
                self.mask = _staircase_state(1, len(ratios), [H], self._chan_thr(), H, d)
            else:
                self.head_num_list, self.qkv_channel_ratio_list = heads, ratios
                self.alpha = nn.Parameter(torch.rand(len(heads), len(ratios)))
                self.score = nn.Parameter(torch.rand(H, d))
                self.mask = _staircase_state(len(heads), len(ratios), heads, self._chan_thr(), H, d)
            self.switch_cell = self.alpha > 0
            trunc_normal_(self.score, std=.2)
        else:
            self.head_num_list, self.qkv_channel_ratio_list = [H], [1.0]
            self.alpha = nn.Parameter(torch.ones(1, 1))
            self.switch_cell = self.alpha > 0
            self.weighted_mask = self.mask = torch.ones(1, H, 1, d)
            self.finish_search = True
            self.score = torch.ones(H, d)
        self.in_features = self.qkv.in_features

    def _chan_thr(self):
        return [int(self.head_dim * r) for r in self.qkv_channel_ratio_list]

    def gate_plan(self):
        H, d = self.score.shape                        # (H, d) joint, (H, 1) head-only, (1, d) channel-only
        A0, A1 = self.alpha.shape                      # compress() drops trailing head / channel options
        space = getattr(self, 'space', 'joint')
        head_thr = list(self.head_num_list)[:A0] if space != 'channel' else [1]
        chan_thr = self._chan_thr()[:A1] if space != 'head' else [1]
        return dict(H=H, C=d, A0=A0, A1=A1, kind=0, head_thr=head_thr, chan_thr=chan_thr, norm_coef=4e-4, w_p=float(self.w_p),
                    on=[int(v) for v in self.switch_cell.reshape(-1).tolist()])

    def wsum_scale(self):
        """the FLOPs model sums the BROADCAST staircase (reference layers.py:747-766 on weighted_mask (H, 1, d)): the gate
        kernel's sum over (H, 1) / (1, d) counts each entry once, so it is scaled by the broadcast extent"""
        space = getattr(self, 'space', 'joint')
        return float(self.head_dim) if space == 'head' else (float(self.active_heads()) if space == 'channel' else 1.0)

    def _bcast(self, t):
        return t if getattr(self, 'space', 'joint') == 'joint' else t.expand(self.active_heads(), self.head_dim)

    def _shape_wm(self, wm):
        wm = self._bcast(wm)
        return wm.reshape(wm.shape[0], 1, wm.shape[1])         # (H, 1, d) as the reference stores it

    def _wr_view(self):
        return self._bcast(self._wr)

    def _prob_view(self):
        return self.score.sigmoid()

    def active_heads(self):
        return self.head_num if hasattr(self, 'head_num') else self.num_heads

    def current_gate(self):
        if not self.finish_search:
            return self._single_gate()
        return None if self.fused else self.score.to(self.qkv.weight.device)

    def forward(self, x, mask_embed=None, weighted_embed=None):
        self.weighted_mask_embed = mask_embed
        return self._branch(x, torch.zeros_like(x), self.current_gate(), None, self.active_heads())

    def compress(self, thresh, optimizer_params, optimizer_decoder, optimizer_archs, prefix='', alpha_reduced=None):
        """reference layers.py:559-696: cut heads (ranked by summed sigmoid(score)) and per-head q/k/v channels (ranked by
        score) down to the surviving option; proj loses the matching input columns."""
        if getattr(self, 'space', 'joint') != 'joint' and int(self.switch_cell.sum()) != 1:
            # the reference's own cut is inconsistent for these spaces (layers.py:612-617: channel indices are gathered from a
            # (H, 1) score, i.e. one channel per head survives while qkv.out_features is set for d channels; with a (1, d) score
            # the head gather indexes past its single row): there is no behaviour to reproduce
            raise NotImplementedError('compress() of a head-only / channel-only attention space is undefined in the reference')
        cut = self._prune_cells(thresh, optimizer_archs, prefix, alpha_reduced)
        if cut is None:
            return optimizer_params, optimizer_decoder, optimizer_archs
        finished, i, j = cut
        H_cur, d_cur = self.score.shape
        n_head, n_chan = self.head_num_list[i], self._chan_thr()[j]
        heads, chan = rank_cut(self.score, n_chan, n_head)
        dev = self.qkv.weight.device
        sc = torch.gather(self.score.detach().to('cpu', torch.float32)[heads], 1, chan)
        if finished:
            sc = self.w_p * sc.sigmoid() + (1 - self.w_p) * torch.ones_like(sc)
        else:
            self.mask = self.mask[:i + 1, :n_head, :j + 1, :n_chan]
        self.weighted_mask = torch.ones(n_head, 1, n_chan, device=dev)
        self.head_num = n_head
        self.scale = self.qk_scale or n_chan ** -0.5
        rows = (heads.view(-1, 1) * d_cur + chan).reshape(-1)                   # positions inside one of q / k / v
        qkv_rows = torch.cat([rows + t * H_cur * d_cur for t in range(3)])
        self.qkv.out_features, self.proj.in_features = 3 * n_head * n_chan, n_head * n_chan
        self._swap(self, 'score', sc.to(dev), optimizer_params, f'{prefix}.score', 0, [heads, chan], [0, -1], initialize=finished)
        self._swap(self.qkv, 'weight', hip.index_select(self.qkv.weight.data, qkv_rows, 0), optimizer_params, f'{prefix}.qkv.weight', 1, qkv_rows, 0)
        if self.qkv.bias is not None:
            self._swap(self.qkv, 'bias', hip.index_select(self.qkv.bias.data, qkv_rows, 0), optimizer_params, f'{prefix}.qkv.bias', 0, qkv_rows, -1)
        self._swap(self.proj, 'weight', hip.index_select(self.proj.weight.data, rows, 1), optimizer_params, f'{prefix}.proj.weight', 1, rows, -1)
        return optimizer_params, optimizer_decoder, optimizer_archs

    def compress_patchembed(self, info, optimizer_params, optimizer_decoder, optimizer_archs, prefix=''):
        """the embedding width shrank (reference layers.py:698-728): qkv loses input columns, proj output rows.
        info: kept-channel index tensor, or a keep ratio / count (first channels)."""
        keep = _embed_keep(info, self.in_features)
        self.qkv.in_features = self.proj.out_features = keep.numel()
        self._swap(self.qkv, 'weight', hip.index_select(self.qkv.weight.data, keep, 1), optimizer_params, f'{prefix}.qkv.weight', 1, keep, -1)
        self._swap(self.proj, 'weight', hip.index_select(self.proj.weight.data, keep, 0), optimizer_params, f'{prefix}.proj.weight', 1, keep, 0)
        if self.proj.bias is not None:
            self._swap(self.proj, 'bias', hip.index_select(self.proj.bias.data, keep, 0), optimizer_params, f'{prefix}.proj.bias', 0, keep, -1)
        return optimizer_params, optimizer_decoder, optimizer_archs

    def fuse(self):
        """fold the frozen gate into qkv (rows, score tiled x3; reference layers.py:539-543)."""
        self.fused = True
        self.score.requires_grad = False
        sc = self.score.data.reshape(-1).repeat(3)
        self.qkv.weight = nn.Parameter(self.qkv.weight.data * sc.unsqueeze(-1))
        if self.qkv.bias is not None:
            self.qkv.bias = nn.Parameter(self.qkv.bias.data * sc)

    def get_params_count(self):
        """reference layers.py:735-745: (total, active)."""
        dim, act_dim = self.in_features, self.qkv.in_features
        wme = getattr(self, 'weighted_mask_embed', None)
        act_e = wme.sum() if wme is not None and bool(((wme < 1) & (wme > 0)).any()) else act_dim
        sd = self._wm_sum()
        total = dim * dim * 3 + dim * 3 + dim * dim + dim
        return total, act_e * sd * 3 + sd * 3 + sd * act_e + act_e

    def get_flops(self, num_patches, active_patches):
        """reference layers.py:747-766: (total, active) MACs; the active part is differentiable in alpha through the staircase sums."""
        H, aH, N, n, d = self.num_heads, self.active_heads(), num_patches, active_patches, self.head_dim
        sd, e = self._wm_sum(), self._embed_sum(self.qkv.in_features)
        total = N * (H * d * (3 * H * d)) + 3 * N * H * d + H * N * d * N + H * N * N + 5 * H * N * N + H * N * N * d \
            + N * (H * d * (H * d)) + N * H * d
        active = n * (e * (3 * sd)) + 3 * n * sd + n * n * sd + aH * n * n + 5 * aH * n * n + n * n * sd + n * (sd * e) + n * e
        return total, active

    @staticmethod
    def from_attn(attn_module, head_search=False, channel_search=False, attn_search=True):
        return MAESparseAttention(attn_module, head_search, channel_search, attn_search)


class Mlp(nn.Module):
    """reference models/layers.py:774-801."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features)
        self.drop = nn.Dropout(drop)
        if drop:
            raise NotImplementedError('MLP dropout is 0 on the OFB path')

    def _branch(self, x, resid, gate, rowscale):
        return ops.mlp_branch(x, resid, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias, gate, rowscale)

    def forward(self, x):
        return self._branch(x, torch.zeros_like(x), None, None)

    def get_params_count(self):
        """reference layers.py:792-797."""
        d1, d2, d3 = self.fc1.in_features, self.fc1.out_features, self.fc2.out_features
        return d1 * d2 + d2 * d3 + d2 + d3

    def get_flops(self, num_patches):
        """reference layers.py:799-801."""
        return self.get_params_count() * num_patches


class MAESparseMlp(Mlp, _Searchable):
    """reference models/layers.py:804-1049: hidden-channel bi-mask applied to fc1's output before GELU."""

    def __init__(self, mlp_module, mlp_search=True):
        super().__init__(mlp_module.fc1.in_features, mlp_module.fc1.out_features, mlp_module.fc2.out_features,
                         act_layer=nn.GELU, drop=mlp_module.drop.p)
        self._init_search_state()
        hid = self.fc1.out_features
        self.hidden_features = hid
        if mlp_search:
            self.hidden_ratio_list = [i / hid for i in range(hid // 4, hid + 1, hid // 8)]
            self.alpha = nn.Parameter(torch.rand(1, len(self.hidden_ratio_list)))
            self.switch_cell = self.alpha > 0
            self.mask = _staircase_state(None, None, None, self._chan_thr(), None, hid)
            self.score = nn.Parameter(torch.rand(1, hid))
            trunc_normal_(self.score, std=.2)
        else:
            self.hidden_ratio_list = [1.0]
            self.alpha = nn.Parameter(torch.ones(1, 1))
            self.switch_cell = self.alpha > 0
            self.weighted_mask = self.mask = torch.ones(1, hid)
            self.finish_search = True
            self.score = torch.ones(1, hid)
        self.in_features = self.fc1.in_features

    def _chan_thr(self):
        return [int(r * self.hidden_features) for r in self.hidden_ratio_list]

    def gate_plan(self):
        A1 = self.alpha.shape[-1]
        return dict(H=1, C=self.score.shape[-1], A0=1, A1=A1, kind=1, head_thr=[1],
                    chan_thr=self._chan_thr()[:A1], norm_coef=1e-4, w_p=float(self.w_p),
                    on=[int(v) for v in self.switch_cell.reshape(-1).tolist()])

    def _shape_wm(self, wm):
        return wm.view(1, -1)

    def _wr_view(self):
        return self._wr.view(1, -1)

    def _prob_view(self):
        return self.score.sigmoid()

    def current_gate(self):
        if not self.finish_search:
            return self._single_gate()
        return None if self.fused else self.score.to(self.fc1.weight.device)

    def forward(self, x, mask_embed=None, weighted_embed=None):
        self.weighted_mask_embed = mask_embed
        return self._branch(x, torch.zeros_like(x), self.current_gate(), None)

    def compress(self, thresh, optimizer_params, optimizer_decoder, optimizer_archs, prefix='', alpha_reduced=None):
        """reference layers.py:883-992: keep the hidden channels with the highest scores; fc2 loses the matching columns."""
        cut = self._prune_cells(thresh, optimizer_archs, prefix, alpha_reduced)
        if cut is None:
            return optimizer_params, optimizer_decoder, optimizer_archs
        finished, _, j = cut
        width = self._chan_thr()[j]
        _, chan = rank_cut(self.score, width)
        keep = chan.reshape(-1)
        dev = self.fc1.weight.device
        sc = self.score.detach().to('cpu', torch.float32)[:, keep]
        if finished:
            sc = self.w_p * sc.sigmoid() + (1 - self.w_p) * torch.ones_like(sc)
        else:
            self.mask = self.mask[:j + 1, :width]
        self.weighted_mask = torch.ones(1, width, device=dev)
        self.fc1.out_features = self.fc2.in_features = width
        self._swap(self, 'score', sc.to(dev), optimizer_params, f'{prefix}.score', 0, keep, -1, initialize=finished)
        self._swap(self.fc1, 'weight', hip.index_select(self.fc1.weight.data, keep, 0), optimizer_params, f'{prefix}.fc1.weight', 1, keep, 0)
        self._swap(self.fc1, 'bias', hip.index_select(self.fc1.bias.data, keep, 0), optimizer_params, f'{prefix}.fc1.bias', 0, keep, -1)
        self._swap(self.fc2, 'weight', hip.index_select(self.fc2.weight.data, keep, 1), optimizer_params, f'{prefix}.fc2.weight', 1, keep, -1)
        return optimizer_params, optimizer_decoder, optimizer_archs

    def compress_patchembed(self, info, optimizer_params, optimizer_decoder, optimizer_archs, prefix=''):
        """the embedding width shrank (reference layers.py:994-1025): fc1 loses input columns, fc2 output rows."""
        keep = _embed_keep(info, self.in_features)
        self.fc1.in_features = self.fc2.out_features = keep.numel()
        self._swap(self.fc1, 'weight', hip.index_select(self.fc1.weight.data, keep, 1), optimizer_params, f'{prefix}.fc1.weight', 1, keep, -1)
        self._swap(self.fc2, 'weight', hip.index_select(self.fc2.weight.data, keep, 0), optimizer_params, f'{prefix}.fc2.weight', 1, keep, 0)
        if self.fc2.bias is not None:
            self._swap(self.fc2, 'bias', hip.index_select(self.fc2.bias.data, keep, 0), optimizer_params, f'{prefix}.fc2.bias', 0, keep, -1)
        return optimizer_params, optimizer_decoder, optimizer_archs

    def fuse(self):
        """fold the frozen gate into fc1 (reference layers.py:867-871)."""
        self.fused = True
        self.score.requires_grad = False
        sc = self.score.data.reshape(-1)
        self.fc1.weight = nn.Parameter(self.fc1.weight.data * sc.unsqueeze(-1))
        self.fc1.bias = nn.Parameter(self.fc1.bias.data * sc)

    def get_params_count(self):
        """reference layers.py:1032-1040: (total, active)."""
        dim1, dim2 = self.in_features, self.hidden_features
        act2, act_e = self._wm_sum(), self._embed_sum(self.fc1.in_features)
        return 2 * (dim1 * dim2) + dim1 + dim2, act_e * act2 + act2 * act_e + act_e + act2

    def get_flops(self, num_patches, active_patches):
        """reference layers.py:1042-1044."""
        total, active = self.get_params_count()
        return total * num_patches, active * active_patches

    @staticmethod
    def from_mlp(mlp_module, mlp_search=True):
        return MAESparseMlp(mlp_module, mlp_search)


def _embed_keep(info, in_features):
    """compress_patchembed's `info`: an index tensor, or a keep ratio (float) / count (int) meaning the first channels."""
    if isinstance(info, torch.Tensor):
        return info.detach().to('cpu', torch.int64).reshape(-1)
    return torch.arange(int(in_features * info) if isinstance(info, float) else int(info))


class ModuleInjection:
    """construction-time class swap (reference models/layers.py:1052-1081)."""
    method = 'full'
    searchable_modules = []

    @staticmethod
    def make_searchable_patchembed(patchmodule, embed_search=True):
        if ModuleInjection.method == 'full':
            return patchmodule
        m = MAEPatchEmbed.from_patchembed(patchmodule, embed_search)
        if embed_search:
            ModuleInjection.searchable_modules.append(m)
        return m

    @staticmethod
    def make_searchable_maeattn(attn_module, head_search=False, channel_search=False, attn_search=True):
        if ModuleInjection.method == 'full':
            return attn_module
        m = MAESparseAttention.from_attn(attn_module, head_search, channel_search, attn_search)
        if attn_search:
            ModuleInjection.searchable_modules.append(m)
        return m

    @staticmethod
    def make_searchable_maemlp(mlp_module, mlp_search=True):
        if ModuleInjection.method == 'full':
            return mlp_module
        m = MAESparseMlp.from_mlp(mlp_module, mlp_search)
        if mlp_search:
            ModuleInjection.searchable_modules.append(m)
        return m


# north_star aliases (the names in BASELINE.json do not exist in the reference)
SearchableAttention = MAESparseAttention
SearchableMlp = MAESparseMlp
