"""The fp64 oracle at batch sizes it cannot hold at once: one search micro-step (engine.py:131-144 + losses.py:80-106) evaluated in
sample chunks.  Exact, not an approximation: the batch enters the loss only through (i) the label-smoothing CE, a mean of per-sample
terms, (ii) the PMIM term, a ratio whose numerator is a sum of per-sample terms and whose denominator (the number of masked pixels)
does not depend on the parameters, (iii) the detached mixing factor base / decoder_loss - two batch-global scalars, obtained in a
first pass without a graph.  Every per-sample operation of the model (patch masking from that sample's noise, DropPath from that
sample's uniform, attention over that sample's tokens) is chunk-local.  `tests/test_oracle_golden.py` checks the chunked step against
the one-shot oracle."""
import torch

from oracle import ofb_oracle as O


def _chunk_terms(cfg, p, st, imgs, labels, noise, dp_u):
    out = O.search_forward(cfg, p, st, imgs, noise, dp_u, training=True)
    b = imgs.shape[0]
    ce_sum = O.label_smoothing_ce(out['logits'], labels) * b
    mask = out['mask']
    if mask is None:
        return out, ce_sum, None, 0.0
    gh, Pz = cfg.img_size // cfg.patch_size, cfg.patch_size
    Mpix = mask.view(b, gh, gh).repeat_interleave(Pz, 1).repeat_interleave(Pz, 2).unsqueeze(1)
    num = ((out['targets'] - out['x_rec']).abs() * Mpix).sum()
    return out, ce_sum, num, float(Mpix.sum())


def chunked_oracle_step(cfg, p, st, imgs, labels, patch_noise, droppath_u, chunk, target_flops=1.0, w=(0.5, 0.5, 0.0, 0.5, 5.0)):
    """Accumulates d(loss_total)/d(p[k]) into p[k].grad; returns dict(logits, base, arch, decoder_loss, loss_total) (detached)."""
    B = imgs.shape[0]
    cuts = [(lo, min(lo + chunk, B)) for lo in range(0, B, chunk)]
    sl = lambda lo, hi: (imgs[lo:hi], labels[lo:hi], patch_noise[lo:hi], None if droppath_u is None else droppath_u[:, lo:hi])
    ce_tot, num_tot, msum, logits = 0.0, 0.0, 0.0, []
    with torch.no_grad():
        for lo, hi in cuts:
            out, ce_sum, num, ms = _chunk_terms(cfg, p, st, *sl(lo, hi))
            ce_tot, msum = ce_tot + float(ce_sum), msum + ms
            num_tot += 0.0 if num is None else float(num)
            logits.append(out['logits'])
    base = ce_tot / B
    has_dec = msum > 0
    dec = num_tot / (msum + 1e-5) / cfg.in_chans if has_dec else 0.0
    c = base / dec if has_dec else 0.0                                       # stopgrad(base / decoder_loss), engine.py:140-143
    for lo, hi in cuts:
        out, ce_sum, num, _ = _chunk_terms(cfg, p, st, *sl(lo, hi))
        part = ce_sum / B
        if has_dec:
            part = part + c * num / (msum + 1e-5) / cfg.in_chans
        part.backward()
    gates = O.gates_for(cfg, p, st)
    l_attn, l_mlp, l_patch, l_emb = O.sparsity_losses(cfg, p, st, gates)
    tot_f, sea_f = O.flops_G(cfg, gates, st, p)
    arch = w[0] * l_attn + w[1] * l_mlp + w[2] * l_patch + w[3] * l_emb + w[4] * ((sea_f - target_flops) / tot_f) ** 2
    arch.backward()
    total = base + float(arch.detach()) + (c * dec if has_dec else 0.0)
    f64 = lambda v: torch.tensor(v, dtype=torch.float64)
    return dict(logits=torch.cat(logits), base=f64(base), arch=arch.detach(), decoder_loss=f64(dec), loss_total=f64(total))
