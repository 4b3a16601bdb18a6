"""JPEG fixtures for the decode stage: small files ENCODED by the Pillow installed in the build container and their
pixels as DECODED by that same Pillow (libjpeg-turbo, default settings: JDCT_ISLOW, fancy upsampling) - the decoder the reference
reads ImageNet with (datasets.py:90-125: ImageFolder -> PIL default_loader -> convert('RGB')).

Run:  python tests/golden/make_golden_jpeg.py        (writes tests/golden/jpeg_cases.npz: file bytes + expected RGB pixels)"""
import io
import os

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))


def picture(h, w, seed):
    """smooth gradients + edges + noise: exercises DC prediction, long zero runs, EOB / ZRL codes and saturated pixels"""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    img = np.stack([128 + 100 * np.sin(x / (3.0 + seed)) * np.cos(y / 5.0), 255.0 * x / max(w - 1, 1), 255.0 * ((x // 7 + y // 5) % 2)], -1)
    img += rng.normal(0, 12 + 4 * seed, img.shape)
    img[h // 3:h // 3 + 3, :, :] = 255
    img[:, w // 2:w // 2 + 2, :] = 0
    return np.clip(img, 0, 255).astype(np.uint8)


CASES = [  # name, (h, w), save kwargs, grayscale
    ('q75_420_53x37', (37, 53), dict(quality=75, subsampling=2), False),
    ('q90_422_64x48', (48, 64), dict(quality=90, subsampling=1), False),
    ('q95_444_31x17', (17, 31), dict(quality=95, subsampling=0), False),
    ('q50_420_opt_100x75', (75, 100), dict(quality=50, subsampling=2, optimize=True), False),
    ('q80_420_rst3_80x60', (60, 80), dict(quality=80, subsampling=2, restart_marker_blocks=3), False),
    ('q85_gray_45x33', (33, 45), dict(quality=85), True),
    ('q30_420_16x16', (16, 16), dict(quality=30, subsampling=2), False),
    ('q100_444_9x9', (9, 9), dict(quality=100, subsampling=0), False),
    ('q60_422_rst1_33x65', (65, 33), dict(quality=60, subsampling=1, restart_marker_blocks=1), False),
    ('q92_420_1x1', (1, 1), dict(quality=92, subsampling=2), False),
    ('q70_420_129x97', (97, 129), dict(quality=70, subsampling=2), False),
    ('q88_420_opt_rst_72x50', (50, 72), dict(quality=88, subsampling=2, optimize=True, restart_marker_rows=1), False),
]


def main():
    out = {}
    for i, (name, (h, w), kw, gray) in enumerate(CASES):
        img = picture(h, w, i)
        pil = Image.fromarray(img[:, :, 0] if gray else img)
        buf = io.BytesIO()
        pil.save(buf, 'JPEG', **kw)
        data = buf.getvalue()
        dec = np.asarray(Image.open(io.BytesIO(data)).convert('RGB'))
        out[f'{name}.jpg'] = np.frombuffer(data, np.uint8)
        out[f'{name}.rgb'] = dec
        print(f'{name}: {len(data)} bytes -> {dec.shape}')
    # a progressive file: must be REJECTED by the decoder (OFB_ELIMIT), not mis-decoded
    buf = io.BytesIO()
    Image.fromarray(picture(40, 40, 99)).save(buf, 'JPEG', quality=80, progressive=True)
    out['progressive.jpg'] = np.frombuffer(buf.getvalue(), np.uint8)
    # files OUTSIDE the native decoder's scope that ImageNet-1k really contains (round 4): the loader must route them to its fallback
    # (PIL, the reference's own loader) instead of failing the batch; 'oos_*.rgb' = Pillow's pixels
    oos = {}
    oos['oos_progressive'] = (picture(40, 40, 99), dict(quality=80, progressive=True), 'RGB')
    oos['oos_cmyk'] = (picture(35, 51, 7), dict(quality=85), 'CMYK')
    oos['oos_adobe_rgb'] = (picture(24, 40, 5), dict(quality=90, subsampling=0, keep_rgb=True), 'RGB')
    for name, (img, kw, mode) in oos.items():
        pil = Image.fromarray(img)
        if mode == 'CMYK':
            pil = pil.convert('CMYK')
        buf = io.BytesIO()
        pil.save(buf, 'JPEG', **kw)
        data = buf.getvalue()
        out[f'{name}.jpg'] = np.frombuffer(data, np.uint8)
        out[f'{name}.rgb'] = np.asarray(Image.open(io.BytesIO(data)).convert('RGB'))
        print(f'{name}: {len(data)} bytes -> {out[name + ".rgb"].shape}')
    path = os.path.join(HERE, 'jpeg_cases.npz')
    np.savez_compressed(path, **out)
    print(f'-> {path} {os.path.getsize(path) / 1024:.0f} KiB')


if __name__ == '__main__':
    main()
