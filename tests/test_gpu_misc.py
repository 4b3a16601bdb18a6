"""norm_targets / PMIM loss / CE / patch mask kernels vs oracle definitions and the reference golden."""
import numpy as np
import pytest
import torch

from oracle import fill
from oracle import ofb_oracle as O
from tests.golden_util import GOLDEN_DIR

pytestmark = pytest.mark.gpu


def test_norm_targets_golden_and_oracle():
    from ofb_amd import ops
    z = np.load(f'{GOLDEN_DIR}/norm_targets.npz')
    imgs = torch.from_numpy(fill.images(1, tag='nt'))
    imgs[0, 2, :40, :] = 0.25
    t = ops.norm_targets(imgs.cuda()).cpu()
    exp = torch.from_numpy(z['out'])
    err = (t[0][:, z['rows'], :] - exp).abs()
    print(f'norm_targets vs reference: max {err[:2].max():.2e} (textured planes), median {err.median():.2e}')
    assert float(err[:2].max()) < 2e-4 and float(err.median()) < 1e-5
    g = torch.Generator().manual_seed(3)
    x = torch.randn(3, 3, 224, 224, generator=g)
    got = ops.norm_targets(x.cuda()).cpu()
    assert float((got - O.norm_targets(x)).abs().max()) < 1e-4


def test_patch_mask_and_ce():
    from ofb_amd import hip, ops
    n = torch.from_numpy(fill.patch_noise(5))
    for keep in (0.95, 0.75, 0.5):
        lk = int(196 * keep)
        mask = torch.empty(5, 196, device='cuda')
        ids = torch.empty(5 * (196 - lk), device='cuda', dtype=torch.int32)
        hip.patch_mask(n.cuda(), mask, 5, 196, lk, ids)
        ref = O.keep_mask_from_noise(n, lk)
        assert torch.equal(mask.cpu(), ref)
        assert sorted(ids.cpu().tolist()) == torch.nonzero(ref.reshape(-1)).reshape(-1).tolist()
    g = torch.Generator().manual_seed(4)
    logits = (torch.randn(37, 1000, generator=g) * 3).requires_grad_(True)
    labels = torch.randint(0, 1000, (37,), generator=g)
    ref = O.label_smoothing_ce(logits.double(), labels)
    ref.backward()
    ld = logits.detach().cuda().requires_grad_(True)
    loss = ops.LabelSmoothingCE.apply(ld, labels.cuda(), 0.1)
    (loss * 1.7).backward()
    assert abs(float(loss) - float(ref)) < 1e-5
    assert float((ld.grad.cpu() / 1.7 - logits.grad).abs().max()) < 1e-7
