import io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from PIL import Image
from ofb_amd import hip
rng = np.random.default_rng(0)
blobs = []
for _ in range(128):
    buf = io.BytesIO()
    Image.fromarray(rng.integers(0, 256, size=(int(rng.integers(300, 520)), int(rng.integers(300, 520)), 3), dtype=np.uint8) // 8 + 100).save(buf, format='JPEG', quality=90)
    blobs.append(buf.getvalue())
pb = hip.jpeg_plan_batch(blobs)
use_gpu = os.environ.get('USE_GPU', '1') == '1'
host = torch.empty(pb.coef_total, dtype=torch.int16)
if use_gpu:
    host = host.pin_memory()
print('torch threads', torch.get_num_threads(), 'gpu', use_gpu)
for th in (16, 4, 1):
    d = []
    for i in range(40):
        t0 = time.perf_counter(); hip.jpeg_decode_batch(pb, host.data_ptr(), th); d.append((time.perf_counter() - t0) * 1e3)
    d = np.array(d)
    print(f'{th:2d} threads: median {np.median(d):.2f} ms  max {d.max():.2f}  >2x median: {(d > 2 * np.median(d)).sum()} of {len(d)}   ', np.round(np.sort(d)[-5:], 1))
