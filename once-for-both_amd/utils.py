"""Host-side pieces around the search / finetune workflow that live in the reference's utils.py / finetune.py:
weight EMA on one fused kernel, loading a searched (compressed) model into the plain finetune ViT, and the alias table
that lets whole-object checkpoints written by the reference (`torch.save(model)`, search.py:671-740) unpickle into this
package's classes."""
import sys
import types
from copy import deepcopy

import torch
import torch.nn as nn

from . import hip


class ModelEma:
    """reference utils.py:333-447.  `update()` is ONE multi-tensor launch over the whole state_dict instead of three
    elementwise ops per tensor; entries whose shape changed under compress() adopt the model's tensor (the reference's
    `intersect`, utils.py:366-410)."""

    def __init__(self, model, decay=0.9999, device='', resume=''):
        self.ema = deepcopy(model)
        self.ema.eval()
        self.decay = decay
        self.device = device
        if device:
            self.ema.to(device=device)
        self.ema_has_module = hasattr(self.ema, 'module')
        if resume:
            self._load_checkpoint(resume)
        for p in self.ema.parameters():
            p.requires_grad_(False)
        self._table = None                       # (device table, keep-alive, n, max_numel, signature)

    def _load_checkpoint(self, checkpoint_path):
        ckpt = torch.load(checkpoint_path, map_location='cpu', weights_only=False)
        assert isinstance(ckpt, dict)
        if 'state_dict_ema' in ckpt:
            sd = {}
            for k, v in ckpt['state_dict_ema'].items():
                sd[('module.' + k) if self.ema_has_module and not k.startswith('module') else k] = v
            self.ema.load_state_dict(sd)

    def _build_table(self, pairs):
        tab = (hip.EmaTensor * len(pairs))()
        maxn = 0
        for i, (e, m) in enumerate(pairs):
            tab[i].ema, tab[i].src, tab[i].n = e.data_ptr(), m.data_ptr(), e.numel()
            maxn = max(maxn, e.numel())
        dev_tab, host = hip.upload_structs(tab, pairs[0][0].device)
        return dev_tab, host, len(pairs), maxn

    @torch.no_grad()
    def update(self, model):
        needs_module = hasattr(model, 'module') and not self.ema_has_module
        msd = model.state_dict()
        pairs, changed = [], {}
        for k, ema_v in self.ema.state_dict().items():
            mk = 'module.' + k if needs_module else k
            model_v = msd[mk].detach()
            if model_v.shape != ema_v.shape:
                changed[k] = model_v
            elif ema_v.dtype == torch.float32:
                if not (ema_v.is_contiguous() and model_v.is_contiguous() and model_v.device == ema_v.device):
                    raise hip.OfbError(f'ModelEma.update: {k} must be contiguous and on the EMA device')
                pairs.append((ema_v, model_v))
            else:
                ema_v.copy_(model_v)
        if pairs:
            sig = tuple((e.data_ptr(), m.data_ptr()) for e, m in pairs)
            if self._table is None or self._table[-1] != sig:          # pointers are stable between compress() calls
                self._table = (*self._build_table(pairs), sig)
            dev_tab, _, n, maxn, _ = self._table
            hip.ema_update(dev_tab, n, maxn, self.decay)
            hip.bump_weight_epoch()              # the EMA weights changed under their P-format copies
        if changed:
            self.intersect(changed)

    def intersect(self, state):
        """adopt re-shaped tensors from the model (after compress()) and fix the owning layer's size attributes."""
        adopt_state(self.ema, state)
        for p in self.ema.parameters():
            p.requires_grad_(False)
        self._table = None


def _owner(root, dotted):
    parts = dotted.split('.')
    mod = root
    for p in parts[:-1]:
        mod = mod[int(p)] if p.isdigit() else getattr(mod, p)
    return mod, parts[-1]


def adopt_state(model, state):
    """state: name -> tensor with the NEW shape.  Replaces the parameters and updates in/out feature counts, channel
    counts and LayerNorm.normalized_shape like finetune.intersect / ModelEma.intersect do (finetune.py:208-226)."""
    for k, v in state.items():
        owner, attr = _owner(model, k)
        old = getattr(owner, attr)
        new = nn.Parameter(v.detach().clone().to(old.device), requires_grad=old.requires_grad)
        setattr(owner, attr, new)
        if attr == 'weight':
            if hasattr(owner, 'out_channels'):
                owner.out_channels, owner.in_channels = new.shape[0], new.shape[1]
            if hasattr(owner, 'out_features'):
                owner.out_features, owner.in_features = new.shape[0], new.shape[1]
            if hasattr(owner, 'normalized_shape'):
                owner.normalized_shape[0] = new.shape[-1]


def intersect(model, pretrained_model, exclude=None):
    """reference finetune.py:182-249: load a searched model into the plain finetune `VisionTransformer`.
    An unfinished search is closed with compress(1.0) first (every module collapses to its most probable cell), the cut
    weights / LayerNorms / tokens are adopted, attention blocks take the surviving head count and qk_scale, and - with
    `exclude=['head']` - the classifier is re-created for a new label set."""
    for m in pretrained_model.modules():
        if hasattr(m, 'finish_search') and m is not pretrained_model and not m.finish_search:
            pretrained_model.compress(1.0)
            break
    state = pretrained_model.state_dict()
    own = dict(model.named_parameters())
    take = {}
    for k, v in state.items():
        if k not in own:
            continue                                   # alpha / score / mask_token / decoder: search-only tensors
        if exclude and any(e in k for e in exclude):
            continue
        take[k] = v
    adopt_state(model, take)
    if exclude and any('head' in e for e in exclude):
        model.head = nn.Linear(state['head.weight'].shape[1], model.head.weight.shape[0]).to(state['head.weight'].device)
    src = dict(pretrained_model.named_modules())
    for name, layer in model.named_modules():
        if hasattr(layer, 'num_heads') and name in src:
            s = src[name]
            layer.num_heads = s.head_num if hasattr(s, 'head_num') else s.num_heads
            layer.qk_scale = s.qk_scale                # `scale` itself is left alone, as in the reference (SURVEY D-2)
    return model


# ---- whole-object checkpoint compatibility -------------------------------------------------------------------
_ALIASES = {'models.layers': 'layers', 'models.vision_transformer': 'vision_transformer', 'models.base_model': 'vision_transformer',
            'models.model': 'model'}


def install_reference_aliases():
    """make `models.layers.*` / `models.vision_transformer.*` (the class paths inside the reference's pickled checkpoints)
    resolve to this package, plus minimal homes for the timm classes those pickles mention, so that
    `torch.load(path, weights_only=False)` of a reference `best.pth` / `model_fused.pth` yields this package's modules."""
    import importlib
    pkg = __name__.rsplit('.', 1)[0]
    root = sys.modules.setdefault('models', types.ModuleType('models'))
    for ref_name, ours in _ALIASES.items():
        mod = importlib.import_module(f'{pkg}.{ours}')
        sys.modules[ref_name] = mod
        setattr(root, ref_name.split('.', 1)[1], mod)
    try:
        import timm.models.layers.drop                  # noqa: F401  (a real timm wins when installed)
    except Exception:
        from .layers import DropPath
        for name in ('timm', 'timm.models', 'timm.models.layers', 'timm.models.layers.drop'):
            sys.modules.setdefault(name, types.ModuleType(name))
        sys.modules['timm.models.layers.drop'].DropPath = DropPath
        sys.modules['timm.models.layers'].DropPath = DropPath
