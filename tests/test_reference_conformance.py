"""Driver-facing API conformance against the IMPORTED reference (build container only: skipped where /root/reference is absent).

`tests/reference_signatures.py` (a subprocess) imports the reference with the oracle's shims and reports (i) the signature of every
own method of the classes the drivers program against, (ii) the signatures of the engine / utils / factory functions they import,
(iii) every call expression in search.py / finetune.py / engine.py / losses.py / models/base_model.py.  This test asserts that the
repo's mirror accepts the same positional and keyword arguments: a maintainer's import swap cannot meet a
TypeError or a missing method.  What is deliberately absent is listed in NOT_MIRRORED with the reason.
"""
import inspect
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
pytestmark = pytest.mark.skipif(not os.path.isdir('/root/reference'), reason='the reference only exists in the build container')

# reference class -> attribute path inside ofb_amd
CLASS_MAP = {
    'MIMVisionTransformer': 'vision_transformer.MIMVisionTransformer', 'VisionTransformer': 'vision_transformer.VisionTransformer',
    'MAEBlock': 'vision_transformer.MAEBlock', 'Block': 'vision_transformer.Block', 'MAEBaseModel': 'vision_transformer.MAEBaseModel',
    'MAESparseAttention': 'layers.MAESparseAttention', 'MAESparseMlp': 'layers.MAESparseMlp', 'MAEPatchEmbed': 'layers.MAEPatchEmbed',
    'Attention': 'layers.Attention', 'Mlp': 'layers.Mlp', 'PatchEmbed': 'layers.PatchEmbed', 'LayerNorm': 'layers.LayerNorm',
    'ModuleInjection': 'layers.ModuleInjection', 'AdamW': 'optim.AdamW', 'ModelEma': 'utils.ModelEma',
    'OFBSearchLOSS': 'losses.OFBSearchLOSS', 'DistillationLoss': 'losses.DistillationLoss',
}
FUNC_MAP = {
    'engine.evaluate': 'engine.evaluate', 'engine.evaluate_finetune': 'engine.evaluate_finetune',
    'engine.search_one_epoch': 'engine.search_one_epoch', 'engine.train_one_epoch': 'engine.train_one_epoch',
    'utils.init_distributed_mode': 'utils.init_distributed_mode', 'utils.get_rank': 'utils.get_rank',
    'utils.get_world_size': 'utils.get_world_size', 'utils.is_main_process': 'utils.is_main_process',
    'utils.save_on_master': 'utils.save_on_master', 'utils.is_dist_avail_and_initialized': 'utils.is_dist_avail_and_initialized',
    'utils.setup_for_distributed': 'utils.setup_for_distributed', 'utils._load_checkpoint_for_ema': 'utils._load_checkpoint_for_ema',
    'models.vision_transformer.norm_targets': 'vision_transformer.norm_targets', 'models.layers.reduce_tensor': 'layers.reduce_tensor',
    'models.model.deit_small_patch16_224_mim': 'model.deit_small_patch16_224_mim',
    'models.model.deit_base_patch16_224_mim': 'model.deit_base_patch16_224_mim',
    'models.model.deit_small_patch16_224_finetune': 'model.deit_small_patch16_224_finetune',
    'models.model.deit_base_patch16_224_finetune': 'model.deit_base_patch16_224_finetune',
}
# (reference class, method) the mirror deliberately does not carry
NOT_MIRRORED = {
    ('MIMVisionTransformer', 'forward_decoder'): 'dead code: reads self.decoder_embed / decoder_blocks, which the constructor never creates (SURVEY 2 row 2)',
    ('MIMVisionTransformer', 'forward_loss'): 'dead code: only reachable from forward_decoder users; never called',
    ('MIMVisionTransformer', 'patchify'): 'dead code: only forward_loss calls it',
}
# free names called in the drivers -> what the swap binds them to
DRIVER_CALLS = {
    'evaluate': 'engine.evaluate', 'evaluate_finetune': 'engine.evaluate_finetune', 'search_one_epoch': 'engine.search_one_epoch',
    'train_one_epoch': 'engine.train_one_epoch', 'AdamW': 'optim.AdamW', 'ModelEma': 'utils.ModelEma',
    'OFBSearchLOSS': 'losses.OFBSearchLOSS', 'DistillationLoss': 'losses.DistillationLoss', 'create_model': 'model.create_model',
    'LabelSmoothingCrossEntropy': 'losses.LabelSmoothingCrossEntropy', 'SoftTargetCrossEntropy': 'data.SoftTargetCrossEntropy',
    'Mixup': 'data.Mixup', 'RASampler': 'data.RASampler', 'NativeScaler': 'utils.NativeScalerWithGradNormCount',
    'create_scheduler': 'lr_sched.create_scheduler', 'lrd.param_groups_lrd': 'lr_decay.param_groups_lrd',
    'utils._load_checkpoint_for_ema': 'utils._load_checkpoint_for_ema', 'utils.get_rank': 'utils.get_rank',
    'utils.get_world_size': 'utils.get_world_size', 'utils.init_distributed_mode': 'utils.init_distributed_mode',
    'utils.is_main_process': 'utils.is_main_process', 'utils.save_on_master': 'utils.save_on_master',
}
# receivers in the drivers that are the search model / the plain model / a searchable module / an optimizer / the EMA / a scheduler
RECEIVERS = {
    'model': ('MIMVisionTransformer', 'VisionTransformer'), 'model.module': ('MIMVisionTransformer', 'VisionTransformer'),
    'model_without_ddp': ('MIMVisionTransformer', 'VisionTransformer'), 'best_model': ('MIMVisionTransformer',),
    'pretrained_model': ('MIMVisionTransformer',), 'self': ('MIMVisionTransformer',),
    'm': ('MAESparseAttention', 'MAESparseMlp', 'MAEPatchEmbed'), 'l_block': ('MAESparseAttention', 'MAESparseMlp', 'MAEPatchEmbed'),
    'optimizer': ('AdamW',), 'optimizer_param': ('AdamW',), 'optimizer_arch': ('AdamW',), 'optimizer_decoder': ('AdamW',),
    'model_ema': ('ModelEma',),
    'lr_schedule': ('_Sched',), 'lr_scheduler_param': ('_Sched',), 'lr_scheduler_arch': ('_Sched',), 'lr_scheduler_decoder': ('_Sched',),
    'loss_scaler': ('_Scaler',),
}


@pytest.fixture(scope='module')
def ref():
    out = subprocess.run([sys.executable, os.path.join(HERE, 'reference_signatures.py')], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads(out.stdout)


@pytest.fixture(scope='module')
def pkg():
    import ofb_amd
    return ofb_amd


def resolve(pkg, path):
    obj = pkg
    for part in path.split('.'):
        obj = getattr(obj, part)
    return obj


def compatible(ref_params, fn, skip_self):
    """every way of calling the reference's function is a valid call of `fn`: same positional names in the same order, a default
    wherever the reference has one; `fn` may add parameters only WITH defaults (or *args / **kwargs)."""
    ours = list(inspect.signature(fn).parameters.values())
    refp = [p for p in ref_params]
    if skip_self and refp and refp[0][0] in ('self', 'cls'):
        refp = refp[1:]
        if ours and ours[0].name in ('self', 'cls'):
            ours = ours[1:]
    has_var_kw = any(p.kind is p.VAR_KEYWORD for p in ours)
    has_var_pos = any(p.kind is p.VAR_POSITIONAL for p in ours)
    ours_named = [p for p in ours if p.kind in (p.POSITIONAL_ONLY, p.POSITIONAL_OR_KEYWORD, p.KEYWORD_ONLY)]
    problems = []
    pos = 0
    for name, kind, has_default in refp:
        if kind in ('VAR_POSITIONAL', 'VAR_KEYWORD'):
            if kind == 'VAR_KEYWORD' and not has_var_kw:
                problems.append(f'reference takes **{name}')
            continue
        mine = next((p for p in ours_named if p.name == name), None)
        if mine is None:
            if not has_var_kw:
                problems.append(f'missing parameter {name!r}')
            continue
        if kind == 'POSITIONAL_OR_KEYWORD':
            if mine.kind is mine.KEYWORD_ONLY or (ours_named.index(mine) != pos and not has_var_pos):
                problems.append(f'{name!r} is positional #{pos} in the reference, #{ours_named.index(mine)} here')
            pos += 1
        if has_default and mine.default is inspect.Parameter.empty:
            problems.append(f'{name!r} has a default in the reference, none here')
    ref_names = {p[0] for p in refp}
    for p in ours_named:
        if p.name not in ref_names and p.default is inspect.Parameter.empty:
            problems.append(f'extra required parameter {p.name!r}')
    return problems


def test_every_reference_method_is_mirrored_with_a_compatible_signature(ref, pkg):
    bad = []
    for cname, methods in ref['classes'].items():
        cls = resolve(pkg, CLASS_MAP[cname])
        for mname, params in methods.items():
            if (cname, mname) in NOT_MIRRORED:
                continue
            if not hasattr(cls, mname):
                bad.append(f'{cname}.{mname}: absent')
                continue
            raw = inspect.getattr_static(cls, mname)
            fn = getattr(cls, mname)
            is_static = isinstance(raw, staticmethod)
            for prob in compatible(params, fn, skip_self=not is_static):
                bad.append(f'{cname}.{mname}: {prob}')
    assert not bad, '\n'.join(bad)
    for key in NOT_MIRRORED:                             # the allow-list must not go stale
        assert key[1] in ref['classes'][key[0]], f'{key} is no longer a reference method'


def test_every_reference_function_the_drivers_import_is_mirrored(ref, pkg):
    bad = []
    for rname, params in ref['functions'].items():
        fn = resolve(pkg, FUNC_MAP[rname])
        bad += [f'{rname}: {p}' for p in compatible(params, inspect.unwrap(fn), skip_self=False)]
    assert not bad, '\n'.join(bad)


class _Sched:
    def step_update(self, num_updates, metric=None):
        pass

    def step(self, epoch, metric=None):
        pass


def _bind(fn, nargs, kwargs, drop_self):
    sig = inspect.signature(fn)
    args = [None] * (nargs + (1 if drop_self else 0))
    sig.bind(*args, **{k: None for k in kwargs})


def test_every_driver_call_site_binds(ref, pkg):
    """AST-collected calls of search.py / finetune.py / engine.py / losses.py / base_model.py: free functions the import swap
    rebinds, and methods invoked on the model / a searchable module / an optimizer / the EMA / a scheduler."""
    classes = {k: resolve(pkg, v) for k, v in CLASS_MAP.items()}
    classes['_Sched'] = pkg.lr_sched.CosineLRSchedulerwithLayerDecay
    classes['_Scaler'] = pkg.utils.NativeScalerWithGradNormCount
    bad, checked = [], 0
    for file, line, callee, nargs, kwargs, star in ref['calls']:
        if star:
            continue
        where = f'{file}:{line} {callee}({nargs} positional, {kwargs})'
        if callee in DRIVER_CALLS:
            fn = resolve(pkg, DRIVER_CALLS[callee])
            target = fn.__init__ if inspect.isclass(fn) else fn
            try:
                _bind(target, nargs, kwargs, drop_self=inspect.isclass(fn))
                checked += 1
            except TypeError as e:
                bad.append(f'{where}: {e}')
            continue
        recv, _, meth = callee.rpartition('.')
        if recv not in RECEIVERS:
            continue
        import torch
        generic = set(dir(torch.nn.Module)) | set(dir(torch.optim.Optimizer))
        if meth in generic and meth not in ('forward',):
            continue
        cands = [c for c in RECEIVERS[recv] if c.startswith('_') or meth in ref['classes'].get(c, {})]
        if not cands:
            continue                                    # not a method of the mirrored classes (tensor attribute chains etc.)
        for c in cands:
            cls = classes[c]
            if not hasattr(cls, meth):
                bad.append(f'{where}: {c} has no {meth}')
                continue
            try:
                _bind(getattr(cls, meth), nargs, kwargs, drop_self=not isinstance(inspect.getattr_static(cls, meth), staticmethod))
                checked += 1
            except TypeError as e:
                bad.append(f'{where} on {c}: {e}')
    assert not bad, '\n'.join(bad)
    assert checked > 60, checked                         # the walk really found the drivers' call sites


def test_module_wrapper_exposes_what_the_drivers_reach_through_it(pkg):
    """search.py:617-620,644-645,743,754 / finetune.py:421-426: `model = DistributedDataParallel(model, device_ids=[gpu],
    find_unused_parameters=True)`, `model.module.<method>`, `model_without_ddp = model.module`."""
    sig = inspect.signature(pkg.dp.DistributedDataParallel.__init__)
    sig.bind(None, None, device_ids=[0], find_unused_parameters=True)
    for name in ('forward', 'no_sync'):
        assert hasattr(pkg.dp.DistributedDataParallel, name)
