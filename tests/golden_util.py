"""Helpers shared by the oracle-vs-golden (CPU) and HIP-vs-oracle/golden (GPU) tests."""
import os

import numpy as np
import torch

from oracle import fill
from oracle import ofb_oracle as O

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')

CASES = {
    'micro_a': dict(cfg=O.MICRO),
    'micro_b': dict(cfg=O.MICRO),
    'tiny_a': dict(cfg=dict(O.DEIT_TINY, num_classes=2)),
    'small_a': dict(cfg=dict(O.DEIT_SMALL, num_classes=1000)),
    # constructor surface beyond the default workflow (round 2): head-only / channel-only attention spaces, patch-number search
    'micro_h': dict(cfg=dict(embed_dim=64, depth=2, num_heads=4, num_classes=10, attn_space='head')),
    'micro_c': dict(cfg=dict(embed_dim=64, depth=2, num_heads=4, num_classes=10, attn_space='channel')),
    'micro_p': dict(cfg=dict(O.MICRO, patch_search=True), patch_w=0.5),
}


def load_case(tag):
    z = np.load(os.path.join(GOLDEN_DIR, f'{tag}.npz'))
    batch, w_p, keep, dp, lr = z['meta']
    cfg = O.Config(**CASES[tag]['cfg'], drop_path_rate=float(dp))
    st = O.SearchState(w_p=float(w_p), keep_ratio=float(keep))
    for k in z.files:
        if k.startswith('switch.'):
            st.switch[k[len('switch.'):]] = torch.from_numpy(z[k])
    batch = int(batch)
    inputs = dict(
        imgs=torch.from_numpy(fill.images(batch)), labels=torch.from_numpy(fill.labels(batch, cfg.num_classes)),
        patch_noise=torch.from_numpy(fill.patch_noise(batch, cfg.num_patches)),
        droppath_u=torch.from_numpy(fill.droppath_noise(2 * cfg.depth, batch)))
    return z, cfg, st, inputs, float(lr)


def sample(t, n=256):
    flat = t.detach().reshape(-1)
    return flat[::max(1, flat.numel() // n)]


def rel_err(a, b):
    a = torch.as_tensor(np.asarray(a), dtype=torch.float64).reshape(-1)
    b = torch.as_tensor(np.asarray(b), dtype=torch.float64).reshape(-1)
    return float((a - b).norm() / (b.norm() + 1e-30))
