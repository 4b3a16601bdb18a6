"""What a data-parallel exchange that HOLDS CUs does to the step (run on the GPU box): the DeiT-S bs-128 search step
timed while `n` stand-in workgroups sit on the chip on a third stream for the whole timed region (ofb_diag_cu_thief: they sleep, i.e.
they take a workgroup slot - and, with --lds, LDS - but almost no issue cycles; an RCCL ring kernel holds one workgroup per channel the
same way and also moves data).  Two questions: (1) how does the single-round "balanced + yield" GEMM schedule, which counts on two free
workgroup slots on every CU, degrade; (2) does planning the persistent grids for fewer CUs (OFB_TUNE_GEMM_CUS = CUs - n) win it back.

    python scripts/cu_thief.py [--steps 10]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--batch', type=int, default=128)
    args = ap.parse_args()
    import ctypes as C
    import ofb_amd
    from ofb_amd import engine, hip
    from ofb_amd.losses import OFBSearchLOSS, DistillationLoss, LabelSmoothingCrossEntropy
    dev = torch.device('cuda', 0)
    torch.cuda.set_stream(torch.cuda.Stream())
    torch.manual_seed(0)
    m = ofb_amd.create_model('deit_small_patch16_224_mim', method='search', num_classes=1000, drop_path_rate=0.1, patch_search=False,
                             mask_ratio=1.0).to(dev)
    m.correct_require_grad(0.5, 0.5, 0, 0.5)
    m.adjust_masking_ratio(0.0, 20, 100)
    m.train()
    crit = OFBSearchLOSS(DistillationLoss(LabelSmoothingCrossEntropy(0.1), None, 'none', 0.5, 1.0), dev, 0.5, 0.5, 0.0, 0.5, 5.0)
    opts = engine.build_optimizers(m, 2.5e-4 * args.batch / 256)
    g = torch.Generator(device=dev).manual_seed(1234)
    imgs = torch.randn(args.batch, 3, 224, 224, device=dev, generator=g)
    labels = torch.randint(0, 1000, (args.batch,), device=dev, generator=g)
    step = lambda: engine.search_step(m, crit, imgs, labels, 1.0, opts)
    for _ in range(8):
        step()
    torch.cuda.synchronize()
    thief_stream = torch.cuda.Stream()
    cus = torch.cuda.get_device_properties(dev).multi_processor_count

    def timed(n_thief, lds, plan_cus):
        hip.tune(hip.TUNE_GEMM_CUS, plan_cus)
        for _ in range(3):
            step()
        hip.join_side()
        torch.cuda.synchronize()
        if n_thief:
            usec = int((args.steps * 30 + 40) * 1000)                     # outlives the timed steps; leaves by itself
            rc = hip.lib().ofb_diag_cu_thief(C.c_int32(n_thief), C.c_int32(usec), C.c_int32(lds), C.c_void_p(thief_stream.cuda_stream))
            assert rc == 0, rc
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.steps):
            step()
        hip.join_side()
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / args.steps
        torch.cuda.synchronize()                                           # the thief has left
        hip.tune(hip.TUNE_GEMM_CUS, 0)
        return ms

    print(f'# DeiT-S bs {args.batch} search step, {cus} CUs, {args.steps} timed steps per row; thief = sleeping 256-thread workgroups on a third stream')
    print(f'{"thief workgroups":>18s} {"LDS each":>9s} {"GEMM plans for":>15s} {"ms/step":>9s}')
    base = None
    for rnd in range(2):                                                   # two passes over the table: drift of the box shows as a difference between them
        for n_thief, lds in ((0, 0), (16, 0), (32, 0), (16, 16384), (32, 16384), (64, 16384)):
            for plan in ([0] if n_thief == 0 else [0, cus - n_thief]):
                ms = timed(n_thief, lds, plan)
                base = ms if base is None else base
                print(f'{n_thief:18d} {lds:9d} {(plan or cus):15d} {ms:9.3f}   ({ms / base - 1:+.1%} vs the first row)')


if __name__ == '__main__':
    main()
