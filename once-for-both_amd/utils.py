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
        self._fast, self._fast_model, self._has_other = None, None, False     # update(): pairs of the last full walk

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

    _REWALK = 64                                 # fast updates between two full walks of the state dicts

    def _walk(self, model):
        """the full walk (reference utils.py ModelEma.update: both state dicts, key by key): returns the float32 pairs, copies the
        other dtypes, adopts re-shaped tensors; remembers WHERE every pair lives (owning dict + name on both sides) for _fast_pairs"""
        needs_module = hasattr(model, 'module') and not self.ema_has_module
        msd = model.state_dict(keep_vars=True)
        pairs, changed, homes = [], {}, []
        where_m, where_e = _tensor_homes(model), _tensor_homes(self.ema)
        for k, ema_v in self.ema.state_dict(keep_vars=True).items():
            mk = 'module.' + k if needs_module else k
            model_v = msd[mk]
            if model_v.shape != ema_v.shape:
                changed[k] = model_v.detach()
            elif ema_v.dtype == torch.float32:
                if not (ema_v.is_contiguous() and model_v.is_contiguous() and model_v.device == ema_v.device):
                    raise hip.OfbError(f'ModelEma.update: {k} must be contiguous and on the EMA device')
                pairs.append((ema_v, model_v))
                homes.append((where_e.get(id(ema_v)), where_m.get(id(model_v))))
            else:
                ema_v.detach().copy_(model_v.detach())
                self._has_other = True                           # (copied by the full walk only: such a model never takes the fast path)
        return pairs, changed, homes

    def _fast_pairs(self):
        """the pairs of the last full walk if every one of them still sits where it sat (same tensor objects under the same names in
        the same modules' _parameters / _buffers: compress() and adopt_state() REPLACE parameters, which this sees), else None"""
        f = self._fast
        if f is None or f[2] <= 0:
            return None
        pairs, homes, _ = f
        for (e, m), (he, hm) in zip(pairs, homes):
            if he is None or hm is None or he[0].get(he[1]) is not e or hm[0].get(hm[1]) is not m:
                return None
        f[2] -= 1
        return pairs

    @torch.no_grad()
    def update(self, model):
        # Two state_dict() walks per update cost the host ~1 ms per finetune micro-step: between two full walks (every _REWALK updates,
        # and whenever a pair moved) the pairs of the last walk are re-validated in place instead.  Models with non-float32 state
        # (integer buffers, copied rather than averaged) always take the full walk.
        pairs = self._fast_pairs() if self._fast is not None and self._fast_model is model else None
        changed = {}
        if pairs is None:
            self._has_other = False
            pairs, changed, homes = self._walk(model)
            ok = not changed and not self._has_other and all(he is not None and hm is not None for he, hm in homes)
            self._fast, self._fast_model = ([pairs, homes, self._REWALK] if ok else None), model
        if pairs:
            sig = tuple([e.data_ptr() for e, _ in pairs] + [m.data_ptr() for _, m in pairs])
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
        self._fast = None


def _tensor_homes(root):
    """id(tensor) -> (the _parameters / _buffers dict that holds it, its name there) for every parameter and buffer under root"""
    homes = {}
    for mod in root.modules():
        for name, t in mod._parameters.items():
            if t is not None:
                homes[id(t)] = (mod._parameters, name)
        for name, t in mod._buffers.items():
            if t is not None:
                homes[id(t)] = (mod._buffers, name)
    return homes


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


# ---- process-group contract of the drivers (reference utils.py:177-244; SURVEY 5: one process per GPU, env:// rendezvous) ---------
def is_dist_avail_and_initialized():
    return torch.distributed.is_available() and torch.distributed.is_initialized()


def get_world_size():
    return torch.distributed.get_world_size() if is_dist_avail_and_initialized() else 1


def get_rank():
    return torch.distributed.get_rank() if is_dist_avail_and_initialized() else 0


def is_main_process():
    return get_rank() == 0


def save_on_master(*args, **kwargs):
    """`torch.save` on rank 0 only (search.py:674,713,734,784; finetune.py:450,470)."""
    if is_main_process():
        torch.save(*args, **kwargs)


def setup_for_distributed(is_master):
    """silence `print` on the other ranks unless called with force=True (reference utils.py:177-188)."""
    import builtins
    plain = builtins.print

    def rank0_print(*args, **kwargs):
        force = kwargs.pop('force', False)
        if is_master or force:
            plain(*args, **kwargs)

    builtins.print = rank0_print


def init_distributed_mode(args):
    """reference utils.py:221-244: RANK / WORLD_SIZE / LOCAL_RANK from the launcher (or SLURM_PROCID), one process per GPU, backend
    'nccl' (= RCCL over xGMI on this machine), `args.dist_url` rendezvous, a barrier, rank-0-only printing.  Fills in
    args.rank / world_size / gpu / distributed / dist_backend."""
    import os
    if isinstance(getattr(args, 'gpu', None), str):      # the reference exports the --gpu list before anything touches the device
        os.environ['CUDA_VISIBLE_DEVICES'] = args.gpu
    if 'RANK' in os.environ and 'WORLD_SIZE' in os.environ:
        args.rank, args.world_size, args.gpu = int(os.environ['RANK']), int(os.environ['WORLD_SIZE']), int(os.environ['LOCAL_RANK'])
    elif 'SLURM_PROCID' in os.environ:
        args.rank = int(os.environ['SLURM_PROCID'])
        args.gpu = args.rank % torch.cuda.device_count()
    else:
        print('Not using distributed mode')
        args.distributed = False
        return
    args.distributed = True
    torch.cuda.set_device(args.gpu)
    args.dist_backend = 'nccl'
    print('| distributed init (rank {}): {}'.format(args.rank, args.dist_url), flush=True)
    torch.distributed.init_process_group(backend=args.dist_backend, init_method=args.dist_url, world_size=args.world_size,
                                         rank=args.rank)
    torch.distributed.barrier()
    setup_for_distributed(args.rank == 0)


def _load_checkpoint_for_ema(model_ema, checkpoint):
    """reference utils.py:167-174: hand an already-loaded checkpoint object to ModelEma._load_checkpoint (search.py:367)."""
    import io
    buf = io.BytesIO()
    torch.save(checkpoint, buf)
    buf.seek(0)
    model_ema._load_checkpoint(buf)


def get_grad_norm_(parameters, norm_type: float = 2.0):
    """reference utils.py:317-330: norm of all gradients (a metric; one small reduction per tensor)."""
    if isinstance(parameters, torch.Tensor):
        parameters = [parameters]
    grads = [p.grad.detach() for p in parameters if p.grad is not None]
    if not grads:
        return torch.tensor(0.)
    if norm_type == float('inf'):
        return max(g.abs().max() for g in grads)
    return torch.norm(torch.stack([torch.norm(g, float(norm_type)) for g in grads]), float(norm_type))


class NativeScalerWithGradNormCount:
    """reference utils.py:282-314.  The drivers construct it and checkpoint its state (search.py:560,681; finetune.py:385,456) but
    the fp32 epoch engines never scale with it (SURVEY D-8).  Kept with the same call contract on a scale of 1 (this path has no
    reduced-precision autocast): backward, optional clipping, optimizer steps; `state_dict` has GradScaler's keys so that
    checkpoints written by either side load in the other."""
    state_dict_key = 'amp_scaler'

    def __init__(self):
        self._state = {'scale': 1.0, 'growth_factor': 2.0, 'backoff_factor': 0.5, 'growth_interval': 2000, '_growth_tracker': 0}

    def __call__(self, loss, optimizer, clip_grad=None, parameters=None, create_graph=False, update_grad=True):
        loss.backward(create_graph=create_graph)
        if not update_grad:
            return None
        if clip_grad is not None:
            assert parameters is not None
            norm = torch.nn.utils.clip_grad_norm_(parameters, clip_grad)
        else:
            norm = get_grad_norm_(parameters)
        for opt in (optimizer if isinstance(optimizer, list) else [optimizer]):
            opt.step()
        return norm

    def state_dict(self):
        return dict(self._state)

    def load_state_dict(self, state_dict):
        self._state.update(state_dict)
