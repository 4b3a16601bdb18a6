"""Golden vectors of the EVALUATION engines and of the FLOPs / parameter bookkeeping: the reference's own `engine.evaluate`
(/root/reference/engine.py:222-257), `engine.evaluate_finetune` (:260-290), `VisionTransformer.get_flops` (models/vision_transformer.py:360-377)
and the per-module `get_flops / get_params_count` + `MAEBaseModel.get_params` (models/layers.py:345-360,735-766,1032-1044;
models/base_model.py:104-109), run unmodified on CPU in the build container.

Run:  python tests/golden/make_golden_eval.py            (writes tests/golden/mini_eval.npz, micro_eval_finetune.npz, flops_counts.npz)

evaluate: the MINI search model (live search, eval mode), three batches of sizes 3 / 2 / 3 - unequal on purpose: the reference's loss
meter averages the per-batch means (n = 1 per update), the accuracy meters are sample-weighted.  Labels are inputs of the case and are
stored: per sample the model's own rank-0 / rank-2 / last-ranked class in turn, so that top-1 and top-5 are neither 0 nor 100.
evaluate_finetune: the same on a plain micro ViT.  FLOPs: the configs[4] subnet shapes of bench.py on a plain reference DeiT-S
(`get_flops()` reads shape attributes only), and every count of the MINI search model after one eval forward.
"""
import contextlib
import io
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G                                    # noqa: E402  (sets up sys.path for the reference and the shims)
from oracle import fill                                   # noqa: E402
from oracle import ofb_oracle as O                        # noqa: E402

import engine as RENG                                      # noqa: E402  reference engine.py
import models.vision_transformer as RVT                    # noqa: E402
import models.layers as RL                                 # noqa: E402

SIZES = (3, 2, 3)
FT = dict(embed_dim=64, depth=2, num_heads=2, num_classes=10)


def eval_images(i, batch):
    return torch.from_numpy(fill.images(batch, tag=f'eval_imgs{i}'))


class Loader:
    def __init__(self, labels):
        self.labels = labels

    def __len__(self):
        return len(self.labels)

    def __iter__(self):
        for i, lab in enumerate(self.labels):
            yield eval_images(i, len(lab)), torch.from_numpy(lab)


def craft_labels(logits_of):
    """labels from the model's own ranking: sample j takes its rank-0 class, its rank-2 class or its last-ranked class (j mod 3)."""
    labels, j = [], 0
    for i, b in enumerate(SIZES):
        order = torch.argsort(logits_of(eval_images(i, b)), dim=1, descending=True)
        lab = []
        for r in range(b):
            lab.append(int(order[r, (0, 2, -1)[j % 3]]))
            j += 1
        labels.append(np.array(lab, np.int64))
    return labels


def run_evaluate(tag='mini_eval'):
    cfg = O.Config(**G.MINI, drop_path_rate=0.0)
    model = G.build_reference(cfg, 0.0)
    for mod in model.searchable_modules:
        mod.w_p = 0.8
    model.eval()
    with torch.no_grad():
        labels = craft_labels(lambda x: model(x)[0])
    with contextlib.redirect_stdout(io.StringIO()):
        stats = RENG.evaluate(Loader(labels), model, torch.device('cpu'), use_amp=False)
    out = dict(meta=np.array([0.8, len(SIZES)], np.float64), sizes=np.array(SIZES, np.int64), stats_keys=np.array(list(stats)))
    for i, lab in enumerate(labels):
        out[f'labels.{i}'] = lab
    for k, v in stats.items():
        out[f'stats.{k}'] = np.float64(v)
    # FLOPs / parameter bookkeeping of the same model after that forward (weighted masks of the eval forward)
    with torch.no_grad():
        total, searched = model.get_flops()
        out['flops.model'] = np.array([float(total), float(searched)], np.float64)
        out['params.model'] = np.array([float(v) for v in model.get_params()], np.float64)
        N = model.patch_embed.num_patches
        for i, blk in enumerate(model.blocks):
            out[f'flops.blocks.{i}'] = np.array([float(v) for v in blk.get_flops(N, N - 20)], np.float64)
        for name, mod in zip(O.module_names(cfg), model.searchable_modules):
            out[f'params.{name}'] = np.array([float(v) for v in mod.get_params_count()], np.float64)
            fl = mod.get_flops(N) if hasattr(mod, 'embed_ratio_list') else mod.get_flops(N, N - 20)
            out[f'flops.{name}'] = np.array([float(v) for v in fl], np.float64)
    path = os.path.join(HERE, f'{tag}.npz')
    np.savez_compressed(path, **out)
    print(f'{tag}:', {k: round(v, 6) for k, v in stats.items()}, 'flops', out['flops.model'].tolist(), 'params', out['params.model'].tolist())


def run_evaluate_finetune(tag='micro_eval_finetune'):
    from functools import partial
    RL.ModuleInjection.method = 'full'
    m = RVT.VisionTransformer(patch_size=16, embed_dim=FT['embed_dim'], depth=FT['depth'], num_heads=FT['num_heads'], mlp_ratio=4,
                              qkv_bias=True, norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), num_classes=FT['num_classes'],
                              drop_path_rate=0.0)
    m.load_state_dict({k: torch.from_numpy(fill.param_value(k, tuple(v.shape))) for k, v in m.state_dict().items()}, strict=True)
    m.eval()
    with torch.no_grad():
        labels = craft_labels(lambda x: m(x))
    with contextlib.redirect_stdout(io.StringIO()):
        stats = RENG.evaluate_finetune(Loader(labels), m, torch.device('cpu'), use_amp=False)
    out = dict(sizes=np.array(SIZES, np.int64), stats_keys=np.array(list(stats)), flops=np.float64(m.get_flops()))
    for i, lab in enumerate(labels):
        out[f'labels.{i}'] = lab
    for k, v in stats.items():
        out[f'stats.{k}'] = np.float64(v)
    path = os.path.join(HERE, f'{tag}.npz')
    np.savez_compressed(path, **out)
    print(f'{tag}:', {k: round(v, 6) for k, v in stats.items()}, 'flops', float(out['flops']))


def run_subnet_flops(tag='flops_counts'):
    """`VisionTransformer.get_flops()` of a plain reference DeiT-S whose layers carry the configs[4] subnet shapes (what
    finetune.intersect leaves behind, finetune.py:182-249: re-shaped Linear / LayerNorm layers, `num_heads` = surviving heads), next
    to the un-pruned DeiT-S / DeiT-B figures."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    import bench
    from functools import partial
    RL.ModuleInjection.method = 'full'
    out = {}

    def plain(D, H, ncls=1000):
        return RVT.VisionTransformer(patch_size=16, embed_dim=D, depth=12, num_heads=H, mlp_ratio=4, qkv_bias=True,
                                     norm_layer=partial(RL.LayerNorm, eps=1e-6), num_classes=ncls)

    out['deit_small'] = np.float64(plain(384, 6).get_flops())
    out['deit_base'] = np.float64(plain(768, 12).get_flops())
    m = plain(384, 6)
    D = bench.FT_EMBED
    m.patch_embed.proj = torch.nn.Conv2d(3, D, 16, 16)
    m.head = torch.nn.Linear(D, 1000)
    for blk, (h, dh, hid) in zip(m.blocks, bench.FT_BLOCKS):
        blk.norm1.normalized_shape[0] = blk.norm2.normalized_shape[0] = D
        blk.attn.qkv, blk.attn.proj, blk.attn.num_heads = torch.nn.Linear(D, 3 * h * dh), torch.nn.Linear(h * dh, D), h
        blk.mlp.fc1, blk.mlp.fc2 = torch.nn.Linear(D, hid), torch.nn.Linear(hid, D)
    out['subnet'] = np.float64(m.get_flops())
    out['subnet_embed'] = np.int64(D)
    out['subnet_blocks'] = np.array(bench.FT_BLOCKS, np.int64)
    path = os.path.join(HERE, f'{tag}.npz')
    np.savez_compressed(path, **out)
    print(f'{tag}:', {k: float(v) for k, v in out.items() if np.ndim(v) == 0})


if __name__ == '__main__':
    torch.manual_seed(0)
    torch.set_num_threads(8)
    run_evaluate()
    run_evaluate_finetune()
    run_subnet_flops()
