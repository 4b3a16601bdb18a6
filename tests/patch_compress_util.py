"""Shared checker for the patch-cell compress() fixture (tests/golden/micro_pc.npz, reference vision_transformer.py:789-820):
used by the CPU test that pins the oracle and by the GPU test that runs the HIP-backed model through the same life cycle."""
import os

import numpy as np
import torch

from oracle import fill
from oracle import ofb_oracle as O
from tests.golden_util import GOLDEN_DIR, rel_err


def load():
    z = np.load(os.path.join(GOLDEN_DIR, 'micro_pc.npz'))
    batch, w_p, dp, lr, thresh = [float(v) for v in z['meta']]
    batch = int(batch)
    cfg = O.Config(**dict(O.MICRO, patch_search=True), drop_path_rate=dp)
    imgs = torch.from_numpy(fill.images(batch))
    labels = torch.from_numpy(fill.labels(batch, cfg.num_classes))
    pnoise = torch.from_numpy(fill.patch_noise(batch, cfg.num_patches))
    return z, cfg, dict(batch=batch, w_p=w_p, lr=lr, thresh=thresh, imgs=imgs, labels=labels, pnoise=pnoise)


def check_step(z, pre, out, grads, alpha_patch_after, tol):
    for k in ['base', 'arch', 'decoder_loss', 'loss_total', 'loss_patch']:
        got, exp = float(torch.as_tensor(out[k]).detach()), float(z[f'{pre}.{k}'])
        assert abs(got - exp) <= tol * max(1.0, abs(exp)), (pre, k, got, exp)
    assert rel_err(out['logits'].detach().cpu(), z[f'{pre}.logits']) < 5 * tol, pre
    for k, g in grads.items():
        if f'{pre}.gnorm.{k}' not in z.files:
            assert g is None or float(g.abs().max()) == 0.0, (pre, k)
            continue
        gn = float(z[f'{pre}.gnorm.{k}'])
        assert g is not None, (pre, k)
        assert abs(float(g.detach().double().norm()) - gn) <= 50 * tol * max(gn, 1e-6), (pre, k, float(g.norm()), gn)
        if f'{pre}.grad.{k}' in z.files and gn > 1e-9:
            assert rel_err(g.detach().cpu(), z[f'{pre}.grad.{k}']) < 100 * tol, (pre, k)
    got = alpha_patch_after.detach().cpu().double()
    exp = torch.from_numpy(z[f'{pre}.after.alpha_patch']).double()
    assert float((got - exp).abs().max()) < 2e-5, (pre, got, exp)


def check_snapshot(z, pre, fin, ex, switch_patch, alpha_patch, alpha_rg, wm_patch, module_switches, shapes):
    assert [int(fin), int(ex)] == z[f'{pre}.model_flags'].tolist(), (pre, fin, ex)
    assert np.array_equal(np.asarray(switch_patch.cpu()).astype(bool), z[f'{pre}.switch_patch']), pre
    assert float((alpha_patch.detach().cpu().double() - torch.from_numpy(z[f'{pre}.alpha_patch']).double()).abs().max()) < 1e-6, pre
    assert bool(alpha_rg) == bool(z[f'{pre}.alpha_patch_rg']), pre
    assert rel_err(wm_patch.detach().cpu().reshape(-1), z[f'{pre}.weighted_mask_patch'].reshape(-1)) < 1e-6, pre
    for name, sw in module_switches.items():
        assert np.array_equal(np.asarray(sw.cpu()).astype(bool), z[f'{pre}.switch.{name}']), (pre, name)
    for k, shp in shapes.items():
        assert list(shp) == z[f'{pre}.shape.{k}'].tolist(), (pre, k)
