"""Sustained f32-MFMA rate of this device (diagnostic; run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ofb_amd import hip
out = torch.zeros(4, device='cuda')
for blocks_per_cu in (1, 2, 3):
    blocks, iters = 256 * blocks_per_cu, 20000
    hip.diag_mfma_peak(out, blocks, 1000); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); hip.diag_mfma_peak(out, blocks, iters); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    fl = blocks * 4 * iters * 4 * 4096.0
    print(f'{blocks_per_cu} blocks/CU: {ms:.2f} ms  {fl / ms / 1e9:.1f} TFLOP/s  (implied clock {fl / ms / 1e9 / 157.3 * 2.4:.2f} GHz)')
