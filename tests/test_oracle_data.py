"""CPU checks of the input-side oracle and host logic: RASampler against the reference's own index lists, the resampler
restatement against Pillow, Mixup parameter draws of the product against the oracle's (same np.random stream)."""
import os
import random

import numpy as np
import pytest
import torch

from oracle import data_oracle as DO

HERE = os.path.dirname(os.path.abspath(__file__))


class _DS:
    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n


def test_ra_sampler_matches_reference_golden():
    import ofb_amd
    z = np.load(os.path.join(HERE, 'golden', 'ra_sampler.npz'))
    assert len(z.files) == 15
    for key in z.files:
        n, w, e, r = (int(p[1:]) for p in key.split('_'))
        exp = z[key].tolist()
        assert DO.ra_sampler_indices(n, w, r, e) == exp, key                      # oracle restatement
        s = ofb_amd.RASampler(_DS(n), num_replicas=w, rank=r, shuffle=True)       # product host logic
        s.set_epoch(e)
        assert list(iter(s)) == exp and len(s) == len(exp), key


@pytest.mark.parametrize('cubic', [False, True])
def test_resize_oracle_matches_pillow(cubic):
    from PIL import Image
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, size=(157, 211, 3), dtype=np.uint8)
    # smooth content as well as noise: low-pass one copy
    img2 = np.asarray(Image.fromarray(img).resize((211, 157), Image.BILINEAR, box=(10, 10, 60, 50)))
    worst = 0
    for src in (img, img2):
        for box in [(0, 0, 157, 211), (13, 40, 96, 128), (50, 60, 20, 31), (3, 5, 150, 200)]:
            top, left, h, w = box
            ref = np.asarray(Image.fromarray(src).resize((64, 64), Image.BICUBIC if cubic else Image.BILINEAR,
                                                         box=(left, top, left + w, top + h)))
            got = DO.pil_like_resize(src, box, 64, cubic)
            d = np.abs(ref.astype(int) - got.astype(int))
            worst = max(worst, int(d.max()))
            assert d.max() <= 1 and (d > 0).mean() < 0.02, (box, d.max(), (d > 0).mean())
    assert worst <= 1


def test_mixup_plan_follows_oracle_draws():
    """The product draws its parameters on the host exactly like the library (same np.random consumption)."""
    import ofb_amd
    for mode in ('batch', 'pair', 'elem'):
        for seed in range(6):
            shape = (8, 3, 32, 32)
            np.random.seed(seed)
            x = torch.arange(8 * 3 * 32 * 32, dtype=torch.float32).reshape(shape).clone()
            t = torch.arange(8) % 5
            ox, ot = DO.Mixup(0.8, 1.0, mode=mode, num_classes=5)(x, t)
            state_after = np.random.get_state()[1][:8].copy()
            np.random.seed(seed)
            rec = ofb_amd.Mixup(0.8, 1.0, mode=mode, num_classes=5).plan(shape)
            assert (np.random.get_state()[1][:8] == state_after).all(), (mode, seed)
            # the targets of the oracle imply each sample's lam: check the plan's
            on = 1 - 0.1 + 0.1 / 5
            off = 0.1 / 5
            for b, (lam, cm, box) in enumerate(rec):
                y1, y2 = int(t[b]), int(t[7 - b])
                exp = torch.full((5,), off) * 1.0
                e1 = torch.full((5,), off); e1[y1] = on
                e2 = torch.full((5,), off); e2[y2] = on
                exp = e1 * lam + e2 * (1. - lam)
                assert torch.allclose(ot[b], exp, atol=1e-6), (mode, seed, b)


def test_random_resized_crop_params_follow_oracle():
    import ofb_amd
    for seed in range(5):
        random.seed(seed)
        a = [DO.random_resized_crop_params(h, w) for h, w in [(375, 500), (64, 900), (900, 64), (224, 224)]]
        random.seed(seed)
        b = [ofb_amd.data.random_resized_crop_params(h, w) for h, w in [(375, 500), (64, 900), (900, 64), (224, 224)]]
        assert a == b


def test_data_entry_points_reject_cpu_tensors():
    import ofb_amd
    from ofb_amd import hip
    with pytest.raises(hip.OfbError):
        ofb_amd.Mixup(0.8, 1.0, num_classes=5)(torch.zeros(4, 3, 8, 8), torch.zeros(4, dtype=torch.int64))
    with pytest.raises(hip.OfbError):
        ofb_amd.SoftTargetCrossEntropy()(torch.zeros(4, 5), torch.zeros(4, 5))


def test_random_erasing_plan_follows_oracle_and_philox_reference():
    import ofb_amd
    random.seed(3)
    a = [DO.random_erasing_plan(224, 224) for _ in range(40)]
    random.seed(3)
    er = ofb_amd.RandomErasing(0.25)
    b = [er.plan_one(224, 224) for _ in range(40)]
    assert a == b and 5 <= sum(1 for p in a if p[2]) <= 20
    # Philox4x32-10 known-answer vectors (Random123 kat_vectors): counter 0 / key 0, and all ones
    assert [int(v) for v in DO._philox4x32_10([0, 0, 0, 0], [0, 0])] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert [int(v) for v in DO._philox4x32_10([0xffffffff] * 4, [0xffffffff] * 2)] == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]


def test_randaugment_plan_consumes_draws_like_the_pillow_restatement():
    """host logic: op choice (np.random.choice) and magnitude / sign / skip draws (`random`) leave both generators in the same state as
    the timm restatement that runs the ops on Pillow, and the aa string of the reference parses."""
    import ofb_amd
    imgs = np.zeros((5, 3, 32, 32), np.uint8)
    random.seed(8); np.random.seed(8)
    DO.rand_augment(imgs)
    s_py, s_np = random.getstate(), np.random.get_state()[1][:6].copy()
    random.seed(8); np.random.seed(8)
    plan = ofb_amd.RandAugment().plan(5, 32, 32)
    assert random.getstate() == s_py and (np.random.get_state()[1][:6] == s_np).all()
    assert len(plan) == 2 and all(len(layer) == 5 for layer in plan)
    with pytest.raises(ofb_amd.hip.OfbError):
        ofb_amd.RandAugment()(torch.zeros(2, 3, 8, 8, dtype=torch.uint8))          # CPU tensor: no fallback
    with pytest.raises(NotImplementedError):
        ofb_amd.DeviceTransform(224, True, auto_augment='original', device='cpu')
    tf = ofb_amd.DeviceTransform(224, True, auto_augment='rand-m9-mstd0.5-inc1', device='cpu')
    assert tf.randaug.magnitude == 9 and tf.randaug.magnitude_std == 0.5 and tf.randaug.fill == (124, 116, 104)
