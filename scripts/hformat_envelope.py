"""Measured envelope of the H-format (csrc/hformat.h) on the DeiT-S bs-128 search step (run on the GPU box).

Every tensor that feeds a product is stored as two f16 planes of X 2^e with ONE exponent per tensor, chosen from an upper bound b of
max|X|: elements >= 2^-18 b keep 2^-23 relative accuracy, smaller ones an absolute 2^-39 b.  This script instruments ONE training step
(after --warm eager steps) and reports, per producer site, what the bound and the window cost on real data:
  looseness      b / measured max|X|                       (log2; 0 = exact)
  below window   share of the non-zero elements with |x| < 2^-18 b   (they carry fewer than 23 significant bits)
  zeros          share of exact zeros in the stored planes (true zeros + values flushed below 2^-39 b)
  worst row      min over the rows of log2(row max / b): a row far below the tensor's bound is the case a per-tensor exponent serves worst
usage: python scripts/hformat_envelope.py [--search-epoch 0|10] [--model deit_small] [--batch 128]  > profiles/r05_hformat_envelope_epochN.txt
"""
import argparse
import math
import os
import sys
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--search-epoch', type=float, default=0.0)
    ap.add_argument('--model', default='deit_small')
    ap.add_argument('--batch', type=int, default=128)
    ap.add_argument('--warm', type=int, default=3)
    ap.add_argument('--per-tensor', action='store_true', help='also list every tensor of the sites whose worst tensor has > 1 %% of its non-zeros below the window, in creation order')
    args = ap.parse_args()
    import ofb_amd
    from ofb_amd import engine, hip
    from ofb_amd.losses import OFBSearchLOSS, DistillationLoss, LabelSmoothingCrossEntropy
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    model = ofb_amd.create_model(f'{args.model}_patch16_224_mim', method='search', num_classes=1000, drop_path_rate=0.1, attn_search=True,
                                 mlp_search=True, embed_search=True, patch_search=False, mae=True, mask_ratio=1.0)
    model.correct_require_grad(0.5, 0.5, 0, 0.5)
    model.adjust_masking_ratio(args.search_epoch, 20, 100)
    for m in model.searchable_modules:
        m.update_w(args.search_epoch, 20)
    model.to(dev).train()
    opt_p, opt_a, opt_d = engine.build_optimizers(model, 2.5e-4 * args.batch / 256)
    crit = OFBSearchLOSS(DistillationLoss(LabelSmoothingCrossEntropy(0.1), None, 'none', 0.5, 1.0), dev, attn_w=0.5, mlp_w=0.5, patch_w=0.0,
                         embedding_w=0.5, flops_w=5.0)
    gen = torch.Generator(device=dev).manual_seed(1234)
    imgs = torch.randn(args.batch, 3, 224, 224, device=dev, generator=gen)
    labels = torch.randint(0, 1000, (args.batch,), device=dev, generator=gen)

    def step():
        return engine.search_step(model, crit, imgs, labels, 1.0, (opt_p, opt_a, opt_d))

    for _ in range(args.warm):
        step()
    hip.join_side()
    torch.cuda.synchronize()

    # ---- instrument: every HMat created during the step, tagged by the product-code frame that made it
    seen, order = {}, []
    real_init = hip.HMat.__init__
    pkg = os.path.join('once-for-both_amd', '')

    def tag_of():
        frames = [f for f in traceback.extract_stack()[:-2] if pkg in f.filename]
        names = [f'{os.path.basename(f.filename)[:-3]}.{f.name}' for f in frames if f.name not in ('__init__', 'for_rows_written_by_kernel', '_pm', '_P')]
        keep = [n for n in names if not n.startswith('hip.')] or names
        return ' < '.join(reversed(keep[-2:]))

    def init(self, R, C_, device, buf=None):
        real_init(self, R, C_, device, buf)
        key = self.buf.data_ptr()
        if key not in seen:
            seen[key] = (self, tag_of())
            order.append(key)

    hip.HMat.__init__ = init
    try:
        step()
    finally:
        hip.HMat.__init__ = real_init
    hip.join_side()
    torch.cuda.synchronize()

    rows_out = {}
    for key in order:
        pm, tag = seen[key]
        e, b, _, _ = pm.header()
        if not (b > 0) or not math.isfinite(b):
            continue
        x = pm.to_f32()
        ax = x.abs()
        mx = float(ax.max())
        if mx == 0.0:
            continue
        nz = ax > 0
        n_nz = int(nz.sum())
        below = int((nz & (ax < b * 2.0 ** -18)).sum())
        rowmax = ax.max(1).values
        rowmax = rowmax[rowmax > 0]
        rec = dict(loose=math.log2(b / mx), below=below / max(n_nz, 1), zeros=1.0 - n_nz / x.numel(),
                   worst_row=float(torch.log2(rowmax.min() / b)), shape=(pm.R, pm.C))
        rows_out.setdefault((tag, (pm.R, pm.C)), []).append(rec)
        del x, ax, nz

    print(f'# H-format envelope of ONE {args.model} bs-{args.batch} search step at epoch {args.search_epoch:g} (w_p {model.searchable_modules[0].w_p:.3f}, keep ratio '
          f'{model.patch_ratio_list[0]:.3f}); {sum(len(v) for v in rows_out.values())} plane tensors; scripts/hformat_envelope.py')
    print('# looseness = log2(bound / measured max); below window = share of non-zero elements under 2^-18 bound; worst row = min log2(row max / bound)')
    print(f'{"producer site (callee < caller)":62s} {"shape":>14s} {"n":>3s}  {"looseness min / med / max":>26s}  {"below window max":>16s}  {"zeros max":>10s}  {"worst row":>9s}')
    worst_loose, worst_below = 0.0, 0.0
    for (tag, shape), recs in sorted(rows_out.items(), key=lambda kv: -max(r['loose'] for r in kv[1])):
        lo = sorted(r['loose'] for r in recs)
        med = lo[len(lo) // 2]
        bl, zr, wr = max(r['below'] for r in recs), max(r['zeros'] for r in recs), min(r['worst_row'] for r in recs)
        worst_loose, worst_below = max(worst_loose, lo[-1]), max(worst_below, bl)
        print(f'{tag[:62]:62s} {str(shape):>14s} {len(recs):3d}  {lo[0]:8.2f} / {med:6.2f} / {lo[-1]:6.2f}  {bl:16.2e}  {zr:10.2e}  {wr:9.1f}')
    print(f'# worst looseness 2^{worst_loose:.2f}; largest share below the window {worst_below:.2e}')
    if args.per_tensor:
        print('# per tensor, in creation order (forward: block 0 first; backward: block 11 first), for the sites above 1 %:')
        for (tag, shape), recs in rows_out.items():
            if max(r['below'] for r in recs) > 0.01:
                print(f'{tag[:62]} {shape}: below window ' + ' '.join(f'{r["below"]:.3f}' for r in recs))
                print(f'{"":10s} zeros ' + ' '.join(f'{r["zeros"]:.3f}' for r in recs) + '   worst row ' + ' '.join(f'{r["worst_row"]:.0f}' for r in recs))


if __name__ == '__main__':
    main()
