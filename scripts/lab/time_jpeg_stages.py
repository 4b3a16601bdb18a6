"""Where a JpegDecoder.decode call spends its time (run on the GPU box)."""
import io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from PIL import Image
import ofb_amd
from ofb_amd import hip
rng = np.random.default_rng(0)
def photo(h, w):
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    base = np.stack([128 + 80 * np.sin(xx / 37 + c) * np.cos(yy / 53 - c) for c in range(3)], -1)
    return np.clip(base + rng.normal(0, 12, size=(h, w, 3)), 0, 255).astype(np.uint8)
blobs = []
for _ in range(128):
    buf = io.BytesIO()
    Image.fromarray(photo(int(rng.integers(300, 520)), int(rng.integers(300, 520)))).save(buf, format='JPEG', quality=90, subsampling='4:2:0')
    blobs.append(buf.getvalue())
pb = hip.jpeg_plan_batch(blobs)
print('coef MB', pb.coef_total * 2 / 1e6, 'planes MB', pb.plane_total / 1e6, 'pixels MB', pb.out_total / 1e6)
stage = torch.empty(pb.coef_total, dtype=torch.int16).pin_memory()
plain = np.empty(pb.coef_total, np.int16)
def t(f, n=5):
    f(); t0 = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t0) / n * 1e3
print('plan            %.2f ms' % t(lambda: hip.jpeg_plan_batch(blobs)))
for th in (1, 4, 8, 16):
    print('huffman %2d thr -> pinned  %.2f ms    -> pageable %.2f ms' % (th, t(lambda: hip.jpeg_decode_batch(pb, stage.data_ptr(), th)),
                                                                        t(lambda: hip.jpeg_decode_batch(pb, plain.ctypes.data, th))))
def h2d():
    stage.to('cuda', non_blocking=True); torch.cuda.synchronize()
print('H2D             %.2f ms' % t(h2d))
dec = ofb_amd.JpegDecoder('cuda')
def full():
    dec.decode(blobs); torch.cuda.synchronize()
print('decode() total  %.2f ms' % t(full))
print('cpu count', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)))
coef = stage.to('cuda')
jobs_dev, host = hip.upload_structs(pb.jobs, torch.device('cuda', 0))
planes = torch.empty(pb.plane_total, device='cuda', dtype=torch.uint8); out = torch.empty(pb.out_total, device='cuda', dtype=torch.uint8)
def px():
    hip.jpeg_decode_pixels(jobs_dev, pb.n, pb.max_blocks, pb.max_w, pb.max_h, coef, planes, out); torch.cuda.synchronize()
print('pixel kernels   %.2f ms' % t(px))
def up():
    hip.upload_structs(pb.jobs, torch.device('cuda', 0)); torch.cuda.synchronize()
print('upload jobs     %.2f ms' % t(up))
print('--- what makes the batch call slow inside decode()?')
def B():
    hip.jpeg_decode_batch(pb, stage.data_ptr(), 16); c = stage.to('cuda', non_blocking=True); torch.cuda.synchronize()
print('huffman + H2D + sync            %.2f ms' % t(B))
def C():
    p2 = hip.jpeg_plan_batch(blobs); hip.jpeg_decode_batch(p2, stage.data_ptr(), 16)
print('plan + huffman (fresh plan)     %.2f ms' % t(C))
st2 = torch.empty(int(pb.coef_total * 1.25) + 4096, dtype=torch.int16).pin_memory()
print('huffman -> bigger pinned buffer %.2f ms' % t(lambda: hip.jpeg_decode_batch(pb, st2.data_ptr(), 16)))
ev = torch.cuda.Event()
def D():
    ev.synchronize(); hip.jpeg_decode_batch(pb, st2.data_ptr(), 16); c = st2[:pb.coef_total].to('cuda', non_blocking=True); ev.record()
print('event sync + huffman + H2D      %.2f ms' % t(D))
import time as _t
t0 = _t.perf_counter(); hip.jpeg_decode_batch(pb, st2.data_ptr(), 16); print('single call after idle         %.2f ms' % ((_t.perf_counter() - t0) * 1e3))
_t.sleep(0.5)
t0 = _t.perf_counter(); hip.jpeg_decode_batch(pb, st2.data_ptr(), 16); print('single call after 0.5 s sleep   %.2f ms' % ((_t.perf_counter() - t0) * 1e3))
print('--- decode() line by line')
class T:
    def __init__(s): s.t = _t.perf_counter(); s.d = {}
    def lap(s, k): n = _t.perf_counter(); s.d[k] = s.d.get(k, 0) + (n - s.t) * 1e3; s.t = n
dec2 = ofb_amd.JpegDecoder('cuda')
for it in range(6):
    T1 = T()
    pbx = hip.jpeg_plan_batch(blobs); T1.lap('plan')
    dec2._slot ^= 1
    pin = dec2._pins[dec2._slot]
    if pin is None or pin[0].numel() < pbx.coef_total:
        tt = torch.empty(int(pbx.coef_total * 1.25) + 4096, dtype=torch.int16).pin_memory()
        pin = dec2._pins[dec2._slot] = (tt, torch.cuda.Event())
    stg, done = pin; T1.lap('pin')
    done.synchronize(); T1.lap('evsync')
    hip.jpeg_decode_batch(pbx, stg.data_ptr(), dec2.threads); T1.lap('huffman')
    cf = stg[:pbx.coef_total].to('cuda', non_blocking=True); done.record(); T1.lap('h2d issue')
    jd, hh = hip.upload_structs(pbx.jobs, torch.device('cuda', 0)); T1.lap('upload')
    pl = torch.empty(pbx.plane_total, device='cuda', dtype=torch.uint8); ou = torch.empty(pbx.out_total, device='cuda', dtype=torch.uint8); T1.lap('alloc')
    hip.jpeg_decode_pixels(jd, pbx.n, pbx.max_blocks, pbx.max_w, pbx.max_h, cf, pl, ou); T1.lap('launch')
    torch.cuda.synchronize(); T1.lap('sync')
    print(it, dec2.threads, {k: round(v, 2) for k, v in T1.d.items()})
