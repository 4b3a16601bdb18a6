"""Inputs of the epoch-engine fixtures (tests/golden/mini_epoch.npz, micro_train_epoch.npz; generator: tests/golden/
make_golden_epoch.py): the closed-form batches, noise, lr schedules and alpha crafting that the generator fed to the reference's
engine.search_one_epoch / train_one_epoch - shared by the CPU test that pins the oracle and the GPU test of ofb_amd.engine."""
import os

import numpy as np
import torch

from oracle import fill
from tests.golden_util import GOLDEN_DIR

MINI = dict(embed_dim=128, depth=3, num_heads=4, num_classes=10)
FT = dict(embed_dim=64, depth=2, num_heads=2, num_classes=10)
N_ITER, ACCUM, BATCH = 6, 2, 2
LR0 = {'p': 1.0e-3, 'a': 2.0e-3, 'd': 5.0e-4}
WARMUP_EPOCHS, EPOCHS = 2, 10


def lr_at(which, gstep):
    return LR0[which] * (1.0 + {'p': 0.10, 'a': 0.05, 'd': 0.20}[which] * (gstep + 1))


def ft_lr_at(gstep):
    return 1.0e-3 * (1.0 + 0.25 * (gstep + 1))


def batch_of(i, ncls, batch=BATCH):
    return torch.from_numpy(fill.images(batch, tag=f'epoch_imgs{i}')), torch.from_numpy((fill.labels(batch, ncls) + i) % ncls)


def noise_of(i, n_patches=196, batch=BATCH):
    return torch.from_numpy(fill.patch_noise(batch, n_patches, tag=f'epoch_noise{i}'))


def load(tag='mini_epoch'):
    return np.load(os.path.join(GOLDEN_DIR, f'{tag}.npz'))


def crafted(z, stage):
    pre = f'craft{stage}.'
    return {k[len(pre):]: torch.from_numpy(z[k]) for k in z.files if k.startswith(pre)}


CRAFT_AT = {0: 1, 2: 2}          # batch index -> crafting stage (the loader edits the alphas before it hands the batch out)


def check_stats(z, stats, tol):
    assert sorted(stats) == list(z['stats_keys']), (sorted(stats), list(z['stats_keys']))
    for k in stats:
        got, exp = float(stats[k]), float(z[f'stats.{k}'])
        assert abs(got - exp) <= tol * max(1.0, abs(exp)), (k, got, exp)
