"""Signature probe of the REFERENCE's driver-facing API (runs only in the build container, where /root/reference exists; started as
a subprocess by tests/test_reference_conformance.py so that the reference's top-level module names - `engine`, `utils`, `losses`,
`optim`, `models` - never enter the test process).

Prints ONE JSON object:
  classes   {reference class: {own method: [[param, kind, has_default], ...]}}   (methods nn.Module does not already have)
  functions {"engine.evaluate": [...], "utils.init_distributed_mode": [...], ...}
  calls     every call expression in the drivers (search.py, finetune.py) and in the files through which they re-enter the model
            (engine.py, losses.py, models/base_model.py): [file, line, dotted callee, n positional, [keywords], has *args/**kw]
Nothing is executed beyond the imports: signatures and syntax trees only.
"""
import ast
import inspect
import json
import os
import sys
import types

REF = '/root/reference'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def params_of(fn):
    out = []
    for p in inspect.signature(fn).parameters.values():
        out.append([p.name, p.kind.name, p.default is not inspect.Parameter.empty])
    return out


def own_methods(cls, stop):
    out = {}
    for k in cls.__mro__:
        if k in stop or k.__module__.startswith('torch'):
            continue
        for n, v in vars(k).items():
            if n.startswith('__') and n not in ('__init__',):
                continue
            f = v.__func__ if isinstance(v, (staticmethod, classmethod)) else v
            if inspect.isfunction(f) and n not in out:
                out[n] = params_of(f)
    return out


def dotted(node):
    if isinstance(node, ast.Name):
        return node.id
    if isinstance(node, ast.Attribute):
        base = dotted(node.value)
        return None if base is None else base + '.' + node.attr
    return None


def calls_in(path):
    tree = ast.parse(open(path).read(), path)
    out = []
    for node in ast.walk(tree):
        if isinstance(node, ast.Call):
            name = dotted(node.func)
            if name is None:
                continue
            star = any(isinstance(a, ast.Starred) for a in node.args) or any(k.arg is None for k in node.keywords)
            out.append([os.path.relpath(path, REF), node.lineno, name, len([a for a in node.args if not isinstance(a, ast.Starred)]),
                        [k.arg for k in node.keywords if k.arg is not None], star])
    return out


def main():
    sys.path[:0] = [os.path.join(ROOT, 'oracle', '_shims'), REF]
    sys.modules['torch._six'] = types.SimpleNamespace(inf=float('inf'))
    import torch.nn as nn
    import engine
    import losses
    import optim
    import utils
    import models.layers as RL
    import models.vision_transformer as RVT
    import models.base_model as RB
    import models.model as RM

    stop = (nn.Module, object)
    classes = {}
    for c in (RVT.MIMVisionTransformer, RVT.VisionTransformer, RVT.MAEBlock, RVT.Block, RL.MAESparseAttention, RL.MAESparseMlp,
              RL.MAEPatchEmbed, RL.Attention, RL.Mlp, RL.PatchEmbed, RL.LayerNorm, RL.ModuleInjection, RB.MAEBaseModel, optim.AdamW,
              utils.ModelEma, losses.OFBSearchLOSS, losses.DistillationLoss):
        classes[c.__name__] = own_methods(c, stop)
    functions = {}
    for mod, names in ((engine, ('evaluate', 'evaluate_finetune', 'search_one_epoch', 'train_one_epoch')),
                       (utils, ('init_distributed_mode', 'get_rank', 'get_world_size', 'is_main_process', 'save_on_master',
                                'is_dist_avail_and_initialized', 'setup_for_distributed', '_load_checkpoint_for_ema')),
                       (RVT, ('norm_targets',)), (RL, ('reduce_tensor',)),
                       (RM, ('deit_small_patch16_224_mim', 'deit_base_patch16_224_mim', 'deit_small_patch16_224_finetune',
                             'deit_base_patch16_224_finetune'))):
        for n in names:
            f = getattr(mod, n)
            functions[f'{mod.__name__}.{n}'] = params_of(inspect.unwrap(f))
    calls = []
    for rel in ('search.py', 'finetune.py', 'engine.py', 'losses.py', 'models/base_model.py'):
        calls += calls_in(os.path.join(REF, rel))
    json.dump(dict(classes=classes, functions=functions, calls=calls), sys.stdout)


if __name__ == '__main__':
    main()
