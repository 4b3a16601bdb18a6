"""Input side of the training step: sampler, GPU crop/resize/flip/normalize, Mixup/CutMix, soft targets.

Mirrors what the reference takes from `samplers.py`, `datasets.build_transform` and the timm fork (`Mixup`,
`SoftTargetCrossEntropy`, `RandomResizedCropAndInterpolation`); the arithmetic runs in csrc/data.hip.  Host code only draws
the random parameters (same generators and draw order as the reference stack: `np.random` for Mixup, Python `random` for the
crop) and uploads one small parameter table per batch.
"""
import math
import random

import numpy as np
import os
import torch

from . import hip

IMAGENET_DEFAULT_MEAN = (0.485, 0.456, 0.406)
IMAGENET_DEFAULT_STD = (0.229, 0.224, 0.225)


# -----------------------------------------------------------------------------------------------------------------
# samplers.py:8-59
# -----------------------------------------------------------------------------------------------------------------
class RASampler(torch.utils.data.Sampler):
    """Repeated-augmentation sampler (reference samplers.py:8-59): epoch-seeded permutation, every index repeated 3x,
    strided by rank, truncated to len // 256 * 256 / world samples.  Host-only index logic."""

    def __init__(self, dataset, num_replicas=None, rank=None, shuffle=True):
        import torch.distributed as dist
        if num_replicas is None:
            if not dist.is_available() or not dist.is_initialized():
                raise RuntimeError('Requires distributed package to be available')
            num_replicas = dist.get_world_size()
        if rank is None:
            if not dist.is_available() or not dist.is_initialized():
                raise RuntimeError('Requires distributed package to be available')
            rank = dist.get_rank()
        self.dataset, self.num_replicas, self.rank, self.epoch = dataset, num_replicas, rank, 0
        self.num_samples = int(math.ceil(len(self.dataset) * 3.0 / self.num_replicas))
        self.total_size = self.num_samples * self.num_replicas
        self.num_selected_samples = int(math.floor(len(self.dataset) // 256 * 256 / self.num_replicas))
        self.shuffle = shuffle

    def __iter__(self):
        g = torch.Generator()
        g.manual_seed(self.epoch)
        n = len(self.dataset)
        idx = torch.randperm(n, generator=g) if self.shuffle else torch.arange(n)
        idx = torch.repeat_interleave(idx, 3)
        if self.total_size > idx.numel():
            idx = torch.cat([idx, idx[:self.total_size - idx.numel()]])
        assert idx.numel() == self.total_size
        idx = idx[self.rank:self.total_size:self.num_replicas]
        assert idx.numel() == self.num_samples
        return iter(idx[:self.num_selected_samples].tolist())

    def __len__(self):
        return self.num_selected_samples

    def set_epoch(self, epoch):
        self.epoch = epoch


# -----------------------------------------------------------------------------------------------------------------
# timm Mixup (search.py:481-484, 651-655; finetune.py:310) on the resident batch
# -----------------------------------------------------------------------------------------------------------------
def rand_bbox(img_shape, lam, margin=0., count=None):
    ratio = np.sqrt(1 - lam)
    img_h, img_w = img_shape[-2:]
    cut_h, cut_w = int(img_h * ratio), int(img_w * ratio)
    margin_y, margin_x = int(margin * cut_h), int(margin * cut_w)
    cy = np.random.randint(0 + margin_y, img_h - margin_y, size=count)
    cx = np.random.randint(0 + margin_x, img_w - margin_x, size=count)
    yl = np.clip(cy - cut_h // 2, 0, img_h)
    yh = np.clip(cy + cut_h // 2, 0, img_h)
    xl = np.clip(cx - cut_w // 2, 0, img_w)
    xh = np.clip(cx + cut_w // 2, 0, img_w)
    return yl, yh, xl, xh


def rand_bbox_minmax(img_shape, minmax, count=None):
    assert len(minmax) == 2
    img_h, img_w = img_shape[-2:]
    cut_h = np.random.randint(int(img_h * minmax[0]), int(img_h * minmax[1]), size=count)
    cut_w = np.random.randint(int(img_w * minmax[0]), int(img_w * minmax[1]), size=count)
    yl = np.random.randint(0, img_h - cut_h, size=count)
    xl = np.random.randint(0, img_w - cut_w, size=count)
    return yl, yl + cut_h, xl, xl + cut_w


def cutmix_bbox_and_lam(img_shape, lam, ratio_minmax=None, correct_lam=True, count=None):
    if ratio_minmax is not None:
        yl, yu, xl, xu = rand_bbox_minmax(img_shape, ratio_minmax, count=count)
    else:
        yl, yu, xl, xu = rand_bbox(img_shape, lam, count=count)
    if correct_lam or ratio_minmax is not None:
        bbox_area = (yu - yl) * (xu - xl)
        lam = 1. - bbox_area / float(img_shape[-2] * img_shape[-1])
    return (yl, yu, xl, xu), lam


class Mixup:
    """timm `Mixup` call contract: `x, soft_target = mixup_fn(x, target)`; x (device, f32, NCHW) is mixed IN PLACE by one
    kernel, the soft targets come from a second one.  Random draws follow timm's order on the global `np.random` state."""

    def __init__(self, mixup_alpha=1., cutmix_alpha=0., cutmix_minmax=None, prob=1.0, switch_prob=0.5, mode='batch',
                 correct_lam=True, label_smoothing=0.1, num_classes=1000):
        self.mixup_alpha, self.cutmix_alpha, self.cutmix_minmax = mixup_alpha, cutmix_alpha, cutmix_minmax
        if self.cutmix_minmax is not None:
            assert len(self.cutmix_minmax) == 2
            self.cutmix_alpha = 1.0
        self.mix_prob, self.switch_prob = prob, switch_prob
        self.label_smoothing, self.num_classes = label_smoothing, num_classes
        self.mode, self.correct_lam, self.mixup_enabled = mode, correct_lam, True

    def _params_per_elem(self, batch_size):
        lam = np.ones(batch_size, dtype=np.float32)
        use_cutmix = np.zeros(batch_size, dtype=bool)
        if self.mixup_enabled:
            if self.mixup_alpha > 0. and self.cutmix_alpha > 0.:
                use_cutmix = np.random.rand(batch_size) < self.switch_prob
                lam_mix = np.where(use_cutmix, np.random.beta(self.cutmix_alpha, self.cutmix_alpha, size=batch_size),
                                   np.random.beta(self.mixup_alpha, self.mixup_alpha, size=batch_size))
            elif self.mixup_alpha > 0.:
                lam_mix = np.random.beta(self.mixup_alpha, self.mixup_alpha, size=batch_size)
            elif self.cutmix_alpha > 0.:
                use_cutmix = np.ones(batch_size, dtype=bool)
                lam_mix = np.random.beta(self.cutmix_alpha, self.cutmix_alpha, size=batch_size)
            else:
                assert False, 'One of mixup_alpha > 0., cutmix_alpha > 0., cutmix_minmax not None should be true.'
            lam = np.where(np.random.rand(batch_size) < self.mix_prob, lam_mix.astype(np.float32), lam)
        return lam, use_cutmix

    def _params_per_batch(self):
        lam, use_cutmix = 1., False
        if self.mixup_enabled and np.random.rand() < self.mix_prob:
            if self.mixup_alpha > 0. and self.cutmix_alpha > 0.:
                use_cutmix = np.random.rand() < self.switch_prob
                lam_mix = np.random.beta(self.cutmix_alpha, self.cutmix_alpha) if use_cutmix else \
                    np.random.beta(self.mixup_alpha, self.mixup_alpha)
            elif self.mixup_alpha > 0.:
                lam_mix = np.random.beta(self.mixup_alpha, self.mixup_alpha)
            elif self.cutmix_alpha > 0.:
                use_cutmix = True
                lam_mix = np.random.beta(self.cutmix_alpha, self.cutmix_alpha)
            else:
                assert False, 'One of mixup_alpha > 0., cutmix_alpha > 0., cutmix_minmax not None should be true.'
            lam = float(lam_mix)
        return lam, use_cutmix

    def plan(self, shape):
        """Draws this batch's parameters: returns a list of per-sample records
        (lam, use_cutmix, (yl, yh, xl, xh)); lam is the value that also weights the targets."""
        B = shape[0]
        box0 = (0, 0, 0, 0)
        if self.mode == 'batch':
            lam, use_cutmix = self._params_per_batch()
            if lam == 1.:
                return [(1.0, False, box0)] * B
            if use_cutmix:
                (yl, yh, xl, xh), lam = cutmix_bbox_and_lam(shape, lam, ratio_minmax=self.cutmix_minmax, correct_lam=self.correct_lam)
                return [(float(lam), True, (int(yl), int(yh), int(xl), int(xh)))] * B
            return [(float(lam), False, box0)] * B
        n = B if self.mode == 'elem' else B // 2
        lam_batch, use_cutmix = self._params_per_elem(n)
        rec = []
        for i in range(n):
            lam = lam_batch[i]
            if lam != 1. and use_cutmix[i]:
                (yl, yh, xl, xh), lam = cutmix_bbox_and_lam(shape, lam, ratio_minmax=self.cutmix_minmax, correct_lam=self.correct_lam)
                lam_batch[i] = lam
                rec.append((float(lam_batch[i]), True, (int(yl), int(yh), int(xl), int(xh))))
            else:
                rec.append((float(lam), False, box0))
        if self.mode == 'pair':
            rec = rec + rec[::-1]
        return rec

    def __call__(self, x, target):
        assert len(x) % 2 == 0, 'Batch size should be even when using this'
        if self.mode not in ('batch', 'pair', 'elem'):
            raise ValueError(self.mode)
        if x.dtype != torch.float32 or not x.is_contiguous() or x.dim() != 4:
            raise hip.OfbError('Mixup expects a contiguous float32 NCHW device batch')
        B, Cc, H, W = x.shape
        rec = self.plan(x.shape)
        tab = (hip.MixParam * B)()
        for b, (lam, cm, (yl, yh, xl, xh)) in enumerate(rec):
            t = tab[b]
            t.lam, t.one_minus_lam, t.use_cutmix = lam, 1. - lam, int(cm)
            t.yl, t.yh, t.xl, t.xh = yl, yh, xl, xh
        dev_tab, host = hip.upload_structs(tab, x.device)
        if any(r[0] != 1. for r in rec):
            hip.mixup_batch(x, dev_tab, B, Cc, H, W)
        off_value = self.label_smoothing / self.num_classes
        on_value = 1. - self.label_smoothing + off_value
        soft = torch.empty(B, self.num_classes, device=x.device, dtype=torch.float32)
        hip.mixup_targets(target.contiguous(), dev_tab, soft, B, self.num_classes, on_value, off_value)
        self._keep = (dev_tab, host)
        return x, soft


class _SoftCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target):
        logits = logits if logits.is_contiguous() else logits.contiguous()
        target = target if target.is_contiguous() else target.contiguous()
        Bn, Cn = logits.shape
        row = torch.empty(Bn, device=logits.device, dtype=torch.float32)
        loss = torch.empty(1, device=logits.device, dtype=torch.float32)
        grad = torch.empty_like(logits)
        hip.soft_cross_entropy(logits, target, row, loss, grad, Bn, Cn)
        ctx.save_for_backward(grad)
        return loss[0]

    @staticmethod
    def backward(ctx, up):
        (grad,) = ctx.saved_tensors
        out = torch.empty_like(grad)
        hip.scale_by_scalar(grad, up.contiguous().reshape(1), out, grad.numel())
        return out, None


class SoftTargetCrossEntropy(torch.nn.Module):
    """timm SoftTargetCrossEntropy (search.py:581-583): mean_b sum_c -target * log_softmax(x); fused forward + gradient."""

    def forward(self, x, target):
        return _SoftCE.apply(x.float(), target.float())


# -----------------------------------------------------------------------------------------------------------------
# build_transform (datasets.py:127-163) on the GPU, from decoded uint8 HWC images
# -----------------------------------------------------------------------------------------------------------------
def random_resized_crop_params(height, width, scale=(0.08, 1.0), ratio=(3. / 4., 4. / 3.)):
    """timm RandomResizedCropAndInterpolation.get_params (Python `random` draws): returns (top, left, h, w)."""
    area = height * width
    for _ in range(10):
        target_area = random.uniform(*scale) * area
        log_ratio = (math.log(ratio[0]), math.log(ratio[1]))
        aspect_ratio = math.exp(random.uniform(*log_ratio))
        w = int(round(math.sqrt(target_area * aspect_ratio)))
        h = int(round(math.sqrt(target_area / aspect_ratio)))
        if w <= width and h <= height:
            i = random.randint(0, height - h)
            j = random.randint(0, width - w)
            return i, j, h, w
    in_ratio = width / height
    if in_ratio < min(ratio):
        w = width
        h = int(round(w / min(ratio)))
    elif in_ratio > max(ratio):
        h = height
        w = int(round(h * max(ratio)))
    else:
        w, h = width, height
    return (height - h) // 2, (width - w) // 2, h, w


def center_crop_params(height, width, input_size=224):
    """Resize(int(256/224*input_size)) + CenterCrop(input_size) (datasets.py:147-153) as ONE box in source pixels: the short
    side maps to `size`, the crop takes input_size/size of it around the centre."""
    size = int((256 / 224) * input_size)
    short = min(height, width)
    side = short * input_size / size
    top, left = (height - side) / 2., (width - side) / 2.
    return int(round(top)), int(round(left)), int(round(side)), int(round(side))


class RandomErasing:
    """timm RandomErasing, mode 'pixel', on the normalized device batch (datasets.py:133-141: re_prob 0.25, re_count 1).  The
    rectangle of every sample is drawn on the host with Python `random` in the library's order; the N(0, 1) fill is generated
    inside the kernel (Philox4x32-10 + Box-Muller, keyed by `seed` and a per-call counter)."""

    def __init__(self, probability=0.25, min_area=0.02, max_area=1 / 3, min_aspect=0.3, max_aspect=None, min_count=1, max_count=None,
                 seed=0):
        self.probability, self.min_area, self.max_area = probability, min_area, max_area
        max_aspect = max_aspect or 1 / min_aspect
        self.log_aspect_ratio = (math.log(min_aspect), math.log(max_aspect))
        self.min_count, self.max_count = min_count, max_count or min_count
        self.seed, self.calls = seed, 0
        if self.min_count != 1 or self.max_count != 1:
            raise NotImplementedError('one rectangle per image (the reference default --recount 1)')

    def plan_one(self, img_h, img_w):
        if random.random() > self.probability:
            return (0, 0, 0, 0)
        area = img_h * img_w
        for _ in range(10):
            target_area = random.uniform(self.min_area, self.max_area) * area
            aspect_ratio = math.exp(random.uniform(*self.log_aspect_ratio))
            h = int(round(math.sqrt(target_area * aspect_ratio)))
            w = int(round(math.sqrt(target_area / aspect_ratio)))
            if w < img_w and h < img_h:
                top = random.randint(0, img_h - h)
                left = random.randint(0, img_w - w)
                return (top, left, h, w)
        return (0, 0, 0, 0)

    def __call__(self, x, plan=None):
        if x.dtype != torch.float32 or not x.is_contiguous() or x.dim() != 4:
            raise hip.OfbError('RandomErasing expects a contiguous float32 NCHW device batch')
        B, Cc, H, W = x.shape
        plan = [self.plan_one(H, W) for _ in range(B)] if plan is None else plan
        if not any(p[2] for p in plan):
            return x
        tab = (hip.EraseParam * B)()
        for b, (top, left, h, w) in enumerate(plan):
            tab[b].top, tab[b].left, tab[b].h, tab[b].w = top, left, h, w
        dev_tab, host = hip.upload_structs(tab, x.device)
        call_seed = (self.seed << 20) + self.calls
        self.calls += 1
        hip.random_erase(x, dev_tab, B, Cc, H, W, call_seed)
        self._keep = (dev_tab, host)
        return x


# -----------------------------------------------------------------------------------------------------------------
# RandAugment 'rand-m9-mstd0.5-inc1' (timm auto_augment.rand_augment_transform as configured by datasets.build_transform:
# --aa default, search.py:123) - the op choice / magnitude draws on the host in timm's order, the pixels on the GPU
# -----------------------------------------------------------------------------------------------------------------
_MAX_LEVEL = 10.
RAND_INCREASING_TRANSFORMS = ['AutoContrast', 'Equalize', 'Invert', 'Rotate', 'PosterizeIncreasing', 'SolarizeIncreasing', 'SolarizeAdd',
                              'ColorIncreasing', 'ContrastIncreasing', 'BrightnessIncreasing', 'SharpnessIncreasing', 'ShearX', 'ShearY',
                              'TranslateXRel', 'TranslateYRel']
_OP_CODE = dict(AutoContrast=1, Equalize=2, Invert=3, PosterizeIncreasing=4, SolarizeIncreasing=5, SolarizeAdd=6, ColorIncreasing=7,
                ContrastIncreasing=8, BrightnessIncreasing=9, SharpnessIncreasing=10)
_PIL_RESAMPLE = dict(nearest=0, bilinear=2, bicubic=3)


def _randomly_negate(v):
    return -v if random.random() > 0.5 else v


def rotate_matrix(degrees, w, h):
    """Pillow Image.rotate(angle) -> the AFFINE matrix it hands to transform() (expand = False, centre = image centre)."""
    angle = degrees % 360.0
    angle = -math.radians(angle)
    matrix = [round(math.cos(angle), 15), round(math.sin(angle), 15), 0.0, round(-math.sin(angle), 15), round(math.cos(angle), 15), 0.0]

    def transform(x, y, m):
        a, b, c, d, e, f = m
        return a * x + b * y + c, d * x + e * y + f

    cx, cy = w / 2.0, h / 2.0
    matrix[2], matrix[5] = transform(-cx, -cy, matrix)
    matrix[2] += cx
    matrix[5] += cy
    return matrix


class RandAugment:
    """`rand_augment_transform('rand-m{m}-mstd{s}-inc1', hparams)`: `num_layers` ops per image drawn with np.random.choice (uniform,
    with replacement), each applied with probability 0.5 at a Gaussian-jittered magnitude.  `plan(B, H, W)` returns, per layer, the
    list of per-image (op, iarg, farg, matrix) records; `__call__` runs them on a uint8 NCHW device batch."""

    def __init__(self, magnitude=9, magnitude_std=0.5, num_layers=2, prob=0.5, img_mean=IMAGENET_DEFAULT_MEAN, interpolation='bicubic',
                 translate_pct=0.45):
        self.magnitude, self.magnitude_std, self.num_layers, self.prob = magnitude, magnitude_std, num_layers, prob
        self.fill = tuple(min(255, round(255 * x)) for x in img_mean)
        self.interpolation = interpolation                    # 'random' -> random.choice((bilinear, bicubic)) per op, like timm
        self.translate_pct = translate_pct
        self.names = list(RAND_INCREASING_TRANSFORMS)

    def _resample(self):
        if self.interpolation == 'random':
            return random.choice((_PIL_RESAMPLE['bilinear'], _PIL_RESAMPLE['bicubic']))
        return _PIL_RESAMPLE[self.interpolation]

    def _one(self, name, H, W):
        """AugmentOp.__call__ for one image: (op code, iarg, farg, matrix | None); op 0 = not applied."""
        if self.prob < 1.0 and random.random() > self.prob:
            return (0, 0, 0.0, None)
        magnitude = self.magnitude
        if self.magnitude_std > 0:
            magnitude = random.gauss(magnitude, self.magnitude_std)
        magnitude = min(_MAX_LEVEL, max(0, magnitude))
        lv = magnitude / _MAX_LEVEL
        if name in ('AutoContrast', 'Equalize', 'Invert'):
            return (_OP_CODE[name], 0, 0.0, None)
        if name == 'PosterizeIncreasing':
            return (4, 4 - int(lv * 4), 0.0, None)
        if name == 'SolarizeIncreasing':
            return (5, 256 - int(lv * 256), 0.0, None)
        if name == 'SolarizeAdd':
            return (6, int(lv * 110), 0.0, None)
        if name in ('ColorIncreasing', 'ContrastIncreasing', 'BrightnessIncreasing', 'SharpnessIncreasing'):
            return (_OP_CODE[name], 0, 1.0 + _randomly_negate(lv * .9), None)
        if name == 'Rotate':
            deg = _randomly_negate(lv * 30.)
            return (11, self._resample(), 0.0, rotate_matrix(deg, W, H))
        if name in ('ShearX', 'ShearY'):
            f = _randomly_negate(lv * 0.3)
            m = [1, f, 0, 0, 1, 0] if name == 'ShearX' else [1, 0, 0, f, 1, 0]
            return (11, self._resample(), 0.0, [float(v) for v in m])
        if name in ('TranslateXRel', 'TranslateYRel'):
            pct = _randomly_negate(lv * self.translate_pct)
            m = [1, 0, pct * W, 0, 1, 0] if name == 'TranslateXRel' else [1, 0, 0, 0, 1, pct * H]
            return (11, self._resample(), 0.0, [float(v) for v in m])
        raise ValueError(name)

    def plan(self, B, H, W):
        per_image = []
        for _ in range(B):
            names = np.random.choice(self.names, self.num_layers, replace=True)
            per_image.append([self._one(str(n), H, W) for n in names])
        return [[per_image[b][l] for b in range(B)] for l in range(self.num_layers)]

    def __call__(self, x_u8, plan=None):
        if x_u8.dtype != torch.uint8 or not x_u8.is_contiguous() or x_u8.dim() != 4 or x_u8.shape[1] != 3:
            raise hip.OfbError('RandAugment expects a contiguous uint8 N x 3 x H x W device batch')
        B, _, H, W = x_u8.shape
        plan = self.plan(B, H, W) if plan is None else plan
        hist = torch.empty(B * 768, device=x_u8.device, dtype=torch.int32)
        lsum = torch.empty(B, device=x_u8.device, dtype=torch.int64)
        cur, other, keep = x_u8, torch.empty_like(x_u8), []
        for layer in plan:
            if not any(rec[0] for rec in layer):
                continue
            tab = (hip.AugOp * B)()
            for b, (op, iarg, farg, m) in enumerate(layer):
                t = tab[b]
                t.op, t.iarg, t.farg = op, iarg, farg
                t.fill[0], t.fill[1], t.fill[2] = self.fill
                for k in range(6):
                    t.m[k] = m[k] if m is not None else 0.0
            dev_tab, host = hip.upload_structs(tab, x_u8.device)
            hip.randaug_layer(cur, other, dev_tab, B, H, W, hist, lsum)
            keep.append((dev_tab, host))
            cur, other = other, cur
        self._keep = (keep, hist, lsum, other)
        return cur


class DeviceTransform:
    """uint8 HWC images -> normalized f32 NCHW batch on the device in two launches (resample rows, resample columns + flip +
    ToTensor + Normalize).  `images`: list of HxWx3 uint8 arrays / tensors (decoded elsewhere)."""

    def __init__(self, input_size=224, is_train=True, interpolation='bicubic', hflip=0.5, scale=(0.08, 1.0),
                 ratio=(3. / 4., 4. / 3.), mean=IMAGENET_DEFAULT_MEAN, std=IMAGENET_DEFAULT_STD, device='cuda', re_prob=0.0, seed=0,
                 auto_augment=None):
        self.S, self.is_train, self.cubic = input_size, is_train, int(interpolation == 'bicubic')
        self.hflip, self.scale, self.ratio, self.mean, self.std = hflip, scale, ratio, tuple(mean), tuple(std)
        self.device = torch.device(device)
        self.erase = RandomErasing(re_prob, seed=seed) if (is_train and re_prob > 0) else None
        # auto_augment='rand-m9-mstd0.5-inc1' (the reference default): RandAugment between the flip and ToTensor, as timm orders them
        self.randaug = None
        if is_train and auto_augment:
            cfg = {}
            for part in auto_augment.split('-')[1:]:                 # timm: key = leading letters, value = the number after them
                key = part.rstrip('0123456789.')
                cfg[key] = part[len(key):]
            if not auto_augment.startswith('rand') or cfg.get('inc', '0') in ('0', ''):
                raise NotImplementedError('only the increasing RandAugment family (rand-m*-mstd*-inc1) is built')
            self.randaug = RandAugment(magnitude=int(cfg.get('m', 10)), magnitude_std=float(cfg.get('mstd', 0)), img_mean=mean,
                                       interpolation=interpolation)

    def plan(self, sizes):
        out = []
        for (h, w) in sizes:
            if self.is_train:
                top, left, ch, cw = random_resized_crop_params(h, w, self.scale, self.ratio)
                flip = int(random.random() < self.hflip)
            else:
                top, left, ch, cw = center_crop_params(h, w, self.S)
                flip = 0
            out.append((top, left, ch, cw, flip))
        return out

    def __call__(self, images, plan=None, want_u8=False):
        """images: list of HxWx3 uint8 arrays / tensors, OR list of JPEG files as `bytes` (decoded on the way by JpegDecoder:
        host Huffman stage on worker threads, IDCT / upsampling / colour on the device - the pixels never visit the host)"""
        if len(images) and isinstance(images[0], (bytes, bytearray, memoryview)):
            if getattr(self, 'jpeg', None) is None:
                self.jpeg = JpegDecoder(self.device)
            src, offs, sizes = self.jpeg.decode(images)
            keep = (src,)
        else:
            src, offs, sizes, keep = self._pack_host(images)
        return self._run(src, offs, sizes, plan, want_u8, keep)

    def _pack_host(self, images):
        # pack the decoded images into ONE pinned staging buffer (two of them, used alternately, so the copy of the previous batch may
        # still be in flight) with plain numpy copies, then one asynchronous H2D transfer
        arrs = [im.numpy() if torch.is_tensor(im) else np.asarray(im) for im in images]
        sizes = [(int(a.shape[0]), int(a.shape[1])) for a in arrs]
        offs, total = [], 0
        for a in arrs:
            if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3:
                raise hip.OfbError('DeviceTransform expects HxWx3 uint8 images')
            offs.append(total)
            total += (a.size + 15) // 16 * 16
        if self.device.type != 'cuda':
            raise hip.OfbError('once-for-both_amd kernels need device tensors (no CPU fallback); DeviceTransform was built for ' + str(self.device))
        slot = getattr(self, '_slot', 0) ^ 1
        self._slot = slot
        pins = getattr(self, '_pins', None)
        if pins is None:
            pins = self._pins = [None, None]
        if pins[slot] is None or pins[slot][0].numel() < total:
            t = torch.empty(int(total * 1.25) + 4096, dtype=torch.uint8).pin_memory()
            pins[slot] = (t, t.numpy(), torch.cuda.Event())
        flat, flat_np, done = pins[slot]
        done.synchronize()                                   # the transfer that last used this staging buffer has finished
        for a, o in zip(arrs, offs):
            flat_np[o:o + a.size] = a.reshape(-1)
        src = flat[:total].to(self.device, non_blocking=True)
        done.record()
        return src, offs, sizes, (flat, src)

    def _run(self, src, offs, sizes, plan, want_u8, keep):
        if self.device.type != 'cuda':
            raise hip.OfbError('once-for-both_amd kernels need device tensors (no CPU fallback); DeviceTransform was built for ' + str(self.device))
        plan = self.plan(sizes) if plan is None else plan
        B = len(sizes)
        tab = (hip.CropParam * B)()
        for b, ((h, w), o, (top, left, ch, cw, flip)) in enumerate(zip(sizes, offs, plan)):
            t = tab[b]
            t.offset, t.src_h, t.src_w = o, h, w
            t.top, t.left, t.height, t.width, t.flip, t.cubic = top, left, ch, cw, flip, self.cubic
        dev_tab, host = hip.upload_structs(tab, self.device)
        max_h = max(h for h, _ in sizes)
        scratch = torch.empty(hip.crop_resize_scratch_bytes(B, self.S, max_h), device=self.device, dtype=torch.uint8)
        out = torch.empty(B, 3, self.S, self.S, device=self.device, dtype=torch.float32)
        out_u8 = torch.empty(B, 3, self.S, self.S, device=self.device, dtype=torch.uint8) if want_u8 else None
        if self.randaug is not None:
            # crop / resize / flip to uint8, RandAugment on the bytes, then ToTensor + Normalize
            out_u8 = torch.empty(B, 3, self.S, self.S, device=self.device, dtype=torch.uint8)
            hip.crop_resize_norm(src, dev_tab, B, self.S, max_h, self.mean, self.std, None, out_u8, scratch)
            out_u8 = self.randaug(out_u8)
            hip.normalize_u8(out_u8, out, B, self.S, self.S, self.mean, self.std)
        else:
            hip.crop_resize_norm(src, dev_tab, B, self.S, max_h, self.mean, self.std, out, out_u8, scratch)
        if self.erase is not None:
            self.erase(out)
        self._keep = (dev_tab, host, keep, scratch)
        return (out, out_u8) if want_u8 else out


def _cpu_share():
    """CPUs this process may really use: the cgroup quota (cpu.max) when there is one, else the affinity mask"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 8)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if quota != 'max':
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return n


class JpegDecoder:
    """JPEG files (bytes) -> decoded uint8 HWC RGB pixels in ONE device buffer: what PIL's default_loader hands the reference's
    transforms (datasets.py:90-125), produced without the pixels ever visiting the host.  Per batch: the frame headers are parsed,
    the Huffman stage of every file runs on a pool of host threads (csrc/jpeg.hip ofb_jpeg_decode_coefficients: plain C++, the
    ctypes call releases the GIL) straight into one pinned int16 staging buffer, ONE H2D copy moves the coefficients, two launches
    (IDCT per 8x8 block; chroma upsampling + YCbCr -> RGB per pixel) produce the pixels of the whole batch.  Arithmetic = libjpeg's
    default path (JDCT_ISLOW, fancy upsampling), bit-exact with Pillow on the committed fixtures.  Scope of the native stages:
    baseline / extended-sequential Huffman files with 1 or 3 YCbCr components.  The files of a batch that fall outside it
    (progressive, CMYK / YCCK, arithmetic-coded, 12-bit, RGB-stored Adobe files, PNGs with a .JPEG name - ImageNet-1k holds a few of
    each) do NOT fail the batch: each one goes through `fallback(bytes) -> HxWx3 uint8 array` - by default exactly the reference's
    loader, PIL `Image.open(...).convert('RGB')` (datasets.py:90-125 via torchvision's default_loader) - and its pixels are uploaded
    into the same output buffer.  `fallback=None` and no Pillow: hip.OfbError naming the indices of the rejected files."""

    def __init__(self, device='cuda', threads=None, fallback='pil'):
        self.device = torch.device(device)
        self.threads = int(threads) if threads else max(1, min(16, _cpu_share()))
        self._pins, self._slot = [None, None], 0
        self.fallback = fallback
        self.n_fallback = 0                                   # files decoded by the fallback so far (statistics)

    @staticmethod
    def _pil(blob):
        import io
        from PIL import Image
        with Image.open(io.BytesIO(blob)) as im:
            return np.asarray(im.convert('RGB'))

    def _split(self, blobs):
        """(indices the native stages accept, indices they reject) from the per-file header parse"""
        ok, bad = [], []
        for i, b in enumerate(blobs):
            try:
                hip.jpeg_parse(bytes(b))
                ok.append(i)
            except hip.OfbError:
                bad.append(i)
        return ok, bad

    def decode(self, blobs):
        """-> (flat uint8 device tensor, byte offset of each image's [H][W][3] pixels, [(H, W)])"""
        if self.device.type != 'cuda':
            raise hip.OfbError('once-for-both_amd kernels need device tensors (no CPU fallback); JpegDecoder was built for ' + str(self.device))
        try:
            return self._decode_native(blobs)
        except hip.OfbError:
            ok, bad = self._split(blobs)
            if not bad:
                raise                                          # every header parses: a corrupt entropy stream - not a scope question
        fb = self._pil if self.fallback == 'pil' else self.fallback
        if fb is None:
            raise hip.OfbError(f'JpegDecoder: files {bad} of the batch are outside the native decoder (progressive / CMYK / ...) and no '
                               'fallback decoder was given')
        try:
            extra = [np.ascontiguousarray(fb(bytes(blobs[i]))) for i in bad]
        except ImportError as e:
            raise hip.OfbError(f'JpegDecoder: files {bad} need the fallback decoder, and Pillow is not importable') from e
        for i, a in zip(bad, extra):
            if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3:
                raise hip.OfbError(f'JpegDecoder: the fallback decoder must return HxWx3 uint8 (file {i})')
        self.n_fallback += len(bad)
        tail = sum((a.size + 15) // 16 * 16 for a in extra)
        if ok:
            out, offs_ok, sizes_ok = self._decode_native([blobs[i] for i in ok], tail_bytes=tail)
            base = out.numel() - tail
        else:
            out, offs_ok, sizes_ok, base = torch.empty(tail, device=self.device, dtype=torch.uint8), [], [], 0
        host = torch.empty(tail, dtype=torch.uint8, pin_memory=True)
        hnp, offs_bad, o = host.numpy(), [], 0
        for a in extra:
            hnp[o:o + a.size] = a.reshape(-1)
            offs_bad.append(base + o)
            o += (a.size + 15) // 16 * 16
        out[base:base + tail].copy_(host, non_blocking=True)
        self._keep_fb = host
        offs, sizes = [0] * len(blobs), [None] * len(blobs)
        for i, oo, ss in zip(ok, offs_ok, sizes_ok):
            offs[i], sizes[i] = oo, ss
        for i, oo, a in zip(bad, offs_bad, extra):
            offs[i], sizes[i] = oo, (int(a.shape[0]), int(a.shape[1]))
        return out, offs, sizes

    def _decode_native(self, blobs, tail_bytes=0):
        pb = hip.jpeg_plan_batch(blobs)                       # headers of all files, batch layout, device job records (native)
        self._slot ^= 1
        pin = self._pins[self._slot]
        if pin is None or pin[0].numel() < pb.coef_total:
            t = torch.empty(int(pb.coef_total * 1.25) + 4096, dtype=torch.int16).pin_memory()
            pin = self._pins[self._slot] = (t, torch.cuda.Event())
        stage, done = pin
        done.synchronize()                                    # the copy that last read this staging buffer has finished
        hip.jpeg_decode_batch(pb, stage.data_ptr(), self.threads)
        coef = stage[:pb.coef_total].to(self.device, non_blocking=True)
        done.record()
        jobs_dev, host = hip.upload_structs(pb.jobs, self.device)
        planes = torch.empty(pb.plane_total, device=self.device, dtype=torch.uint8)
        out_total = (pb.out_total + 15) // 16 * 16
        out = torch.empty(out_total + tail_bytes, device=self.device, dtype=torch.uint8)       # + room for the fallback-decoded files
        hip.jpeg_decode_pixels(jobs_dev, pb.n, pb.max_blocks, pb.max_w, pb.max_h, coef, planes, out)
        self._keep = (jobs_dev, host, coef, planes)
        offs = [int(j.out_off) for j in pb.jobs]
        sizes = [(int(j.height), int(j.width)) for j in pb.jobs]
        return out, offs, sizes


class DeviceLoader:
    """Wraps an iterable of (list of uint8 images, int64 labels): the crop/resize of batch i+1 runs on a side stream while
    the step of batch i computes (the reference overlaps the same work in DataLoader worker processes)."""

    def __init__(self, batches, transform, mixup_fn=None):
        self.batches, self.transform, self.mixup_fn = batches, transform, mixup_fn
        self.stream = torch.cuda.Stream(device=transform.device)

    def _prepare(self, item):
        images, labels = item
        with torch.cuda.stream(self.stream):
            x = self.transform(images)
            y = torch.as_tensor(labels, dtype=torch.int64).to(self.transform.device, non_blocking=True)
            if self.mixup_fn is not None:
                x, y = self.mixup_fn(x, y)
        ev = torch.cuda.Event()
        ev.record(self.stream)
        return x, y, ev

    def __iter__(self):
        it = iter(self.batches)
        nxt = None
        for item in it:
            cur, nxt = nxt, self._prepare(item)
            if cur is not None:
                yield self._hand_over(cur)
        if nxt is not None:
            yield self._hand_over(nxt)

    @staticmethod
    def _hand_over(prep):
        x, y, ev = prep
        torch.cuda.current_stream().wait_event(ev)
        x.record_stream(torch.cuda.current_stream())
        y.record_stream(torch.cuda.current_stream())
        return x, y
