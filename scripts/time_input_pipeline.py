"""Throughput of the GPU input side on decoded uint8 images: H2D copy of the packed batch, RandomResizedCrop + flip,
RandAugment (rand-m9-mstd0.5-inc1), ToTensor + Normalize, RandomErasing, Mixup/CutMix + soft targets, for batches of 128 images of
ImageNet-like sizes.  The step it has to keep up with runs at ~4000 images/s."""
import os, sys, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ofb_amd

rng = np.random.default_rng(0)
B = 128
imgs = [rng.integers(0, 256, size=(int(rng.integers(300, 520)), int(rng.integers(300, 520)), 3), dtype=np.uint8) for _ in range(B)]
labels = list(range(B))
random.seed(0); np.random.seed(0)
for name, kw in [('crop+flip+normalize', {}), ('+ RandAugment + RandomErasing', dict(auto_augment='rand-m9-mstd0.5-inc1', re_prob=0.25))]:
    tf = ofb_amd.DeviceTransform(224, True, 'bicubic', **kw)
    mix = ofb_amd.Mixup(0.8, 1.0, num_classes=1000)
    y = torch.tensor(labels, device='cuda')
    for mode in ('device work only (host plan + launches, batch already packed)', 'whole call'):
        for _ in range(3):
            x = tf(imgs); mix(x, y)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 10
        for _ in range(n):
            x = tf(imgs)
            x, soft = mix(x, y)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(f'{name:34s} {B / dt:9.0f} images/s   ({dt * 1e3:.2f} ms per batch of {B}, host packing of {sum(i.nbytes for i in imgs) / 1e6:.0f} MB included)')
        break
# device-only time of the kernels (events), RandAugment path
tf = ofb_amd.DeviceTransform(224, True, 'bicubic', auto_augment='rand-m9-mstd0.5-inc1', re_prob=0.25)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
x = tf(imgs); torch.cuda.synchronize()
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); x = tf(imgs); torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr); st.sort_stats('cumulative')
tot = st.total_tt
print(f'one call under cProfile: {tot * 1e3:.1f} ms; top host costs:')
for func, (cc, nc, tt, ct, callers) in sorted(st.stats.items(), key=lambda kv: -kv[1][3])[:8]:
    print(f'   {ct * 1e3:7.2f} ms  {func[2]} ({os.path.basename(func[0])}:{func[1]})')

# ---- JPEG bytes in (datasets.py:90-125): host Huffman stage on a thread pool + device IDCT / upsampling / colour ----
try:
    import io
    from PIL import Image
    # photo-like content (smooth gradients + texture) so that the entropy-coded size is ImageNet-like (~110 KB at 500 x 375, q = 90)
    def photo(h, w):
        yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
        base = np.stack([128 + 80 * np.sin(xx / 37 + c) * np.cos(yy / 53 - c) for c in range(3)], -1)
        return np.clip(base + rng.normal(0, 12, size=(h, w, 3)), 0, 255).astype(np.uint8)
    blobs = []
    for im in imgs:
        buf = io.BytesIO()
        Image.fromarray(photo(im.shape[0], im.shape[1])).save(buf, format='JPEG', quality=90, subsampling='4:2:0')
        blobs.append(buf.getvalue())
    print(f'JPEG batch: {B} files, {sum(len(b) for b in blobs) / B / 1e3:.0f} KB each on average')
    for threads in (4, 8, 16):
        dec = ofb_amd.JpegDecoder('cuda', threads=threads)
        for _ in range(3): dec.decode(blobs)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): dec.decode(blobs)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        print(f'JpegDecoder.decode, {threads:2d} host threads          {B / dt:9.0f} images/s   ({dt * 1e3:.2f} ms per batch)')
    t0 = time.perf_counter()
    for _ in range(3):
        ref = [np.asarray(Image.open(io.BytesIO(b)).convert('RGB')) for b in blobs]
    dt = (time.perf_counter() - t0) / 3
    print(f'Pillow decode, one thread (the reference loader, per worker)   {B / dt:9.0f} images/s')
    tf = ofb_amd.DeviceTransform(224, True, 'bicubic', auto_augment='rand-m9-mstd0.5-inc1', re_prob=0.25)
    for _ in range(3): tf(blobs)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): x = tf(blobs)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f'JPEG bytes -> decode -> crop / RandAugment / normalize / erase   {B / dt:9.0f} images/s   ({dt * 1e3:.2f} ms per batch)')
except ImportError as e:
    print('JPEG leg skipped:', e)
