"""time the IMPORTED reference (/root/reference, CPU, fp32) and the oracle on the same synthetic search step in
the build container (this script cannot run on the GPU box: the reference does not travel).  Forward + OFBSearchLOSS + backward of
one micro-step, as bench.py's cpu_baseline leg times the oracle there.

    python scripts/time_reference_cpu.py          -> profiles/r01_reference_cpu_timing.json
"""
import contextlib
import io
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import make_golden as G                                   # sets up the shims and imports the reference  # noqa: E402
from oracle import ofb_oracle as O                        # noqa: E402


def time_reference(cfg_kw, batch, ncls, reps):
    cfg = O.Config(**cfg_kw, num_classes=ncls, drop_path_rate=0.1)
    model = G.build_reference(cfg, 0.1)
    for mod in model.searchable_modules:
        mod.w_p = 0.99
    model.patch_ratio_list = [0.95]
    model.train()
    g = torch.Generator().manual_seed(1234)
    imgs = torch.randn(batch, 3, 224, 224, generator=g)
    labels = torch.randint(0, ncls, (batch,), generator=g)
    crit = G.RLOSS.OFBSearchLOSS(G.RLOSS.DistillationLoss(G.LabelSmoothingCrossEntropy(0.1), None, 'none', 0.5, 1.0),
                                 torch.device('cpu'), attn_w=0.5, mlp_w=0.5, patch_w=0.0, embedding_w=0.5, flops_w=5.0)
    wrapped = G._Wrap(model)

    def step():
        model.zero_grad(set_to_none=True)
        with contextlib.redirect_stdout(io.StringIO()):
            logits, (dec, _) = wrapped(imgs)
            base, arch = crit(imgs, logits, labels, wrapped, 'arch', 1.0, False)
        total = base + arch + (base / dec).data.clone() * dec
        total.backward()

    step()
    t0 = time.time()
    for _ in range(reps):
        step()
    return (time.time() - t0) / reps


def time_oracle(cfg_kw, batch, ncls, reps):
    cfg = O.Config(**cfg_kw, num_classes=ncls, drop_path_rate=0.1)
    p = {k: v.requires_grad_(True) for k, v in O.formula_params(cfg, torch.float32).items()}
    p['alpha_patch'].requires_grad_(False)
    st = O.SearchState(w_p=0.99, keep_ratio=0.95)
    g = torch.Generator().manual_seed(1234)
    imgs = torch.randn(batch, 3, 224, 224, generator=g)
    labels = torch.randint(0, ncls, (batch,), generator=g)

    def step():
        for v in p.values():
            v.grad = None
        out = O.search_step_loss(cfg, p, st, imgs, labels, torch.rand(batch, 196, generator=g), torch.rand(2 * cfg.depth, batch, generator=g))
        out['loss_total'].backward()

    step()
    t0 = time.time()
    for _ in range(reps):
        step()
    return (time.time() - t0) / reps


if __name__ == '__main__':
    torch.manual_seed(0)
    res = dict(host=dict(threads=torch.get_num_threads(), note='build container, 8 vCPU'), cases=[])
    for name, kw, batch, ncls, reps in [('configs[0]: DeiT-T OFB search step, 2 classes, bs 8', O.DEIT_TINY, 8, 2, 3),
                                       ('DeiT-S OFB search step, bs 8 (the sample bench.py times on the GPU host)', O.DEIT_SMALL, 8, 1000, 2)]:
        tr = time_reference(kw, batch, ncls, reps)
        to = time_oracle(kw, batch, ncls, reps)
        res['cases'].append(dict(case=name, reference_s_per_step=round(tr, 3), reference_images_per_s=round(batch / tr, 2),
                                 oracle_s_per_step=round(to, 3), oracle_images_per_s=round(batch / to, 2)))
        print(res['cases'][-1], flush=True)
    with open(os.path.join(ROOT, 'profiles', 'r01_reference_cpu_timing.json'), 'w') as f:
        json.dump(res, f, indent=1)
