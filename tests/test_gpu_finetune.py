"""Finetune path: plain VisionTransformer with heterogeneous pruned per-block dims, forward + backward
against the fp64 oracle (reference vision_transformer.py:332-358, Block :144-160, Attention layers.py:382-394)."""
import pytest
import torch
import torch.nn as nn

from oracle import fill
from oracle import ofb_oracle as O

pytestmark = pytest.mark.gpu


def _pruned_model(D, heads, dhs, hids, ncls):
    """what finetune.intersect (finetune.py:182-249) leaves behind: per-block num_heads / qkv width / hidden width."""
    import ofb_amd
    depth = len(heads)
    m = ofb_amd.VisionTransformer(embed_dim=D, depth=depth, num_heads=6, num_classes=ncls, drop_path_rate=0.0)
    for i, blk in enumerate(m.blocks):
        hd = heads[i] * dhs[i]
        blk.attn.qkv = nn.Linear(D, 3 * hd)
        blk.attn.proj = nn.Linear(hd, D)
        blk.attn.num_heads = heads[i]
        blk.attn.scale = 64 ** -0.5            # never re-derived after pruning (SURVEY D-2)
        blk.mlp.fc1 = nn.Linear(D, hids[i])
        blk.mlp.fc2 = nn.Linear(hids[i], D)
    return m


@pytest.mark.parametrize('D,heads,dhs,hids', [(288, [4, 6, 2], [40, 64, 16], [768, 1152, 384]),
                                               (192, [3, 2], [64, 24], [768, 192])])
def test_pruned_vit_forward_backward(D, heads, dhs, hids):
    B, ncls = 3, 10
    m = _pruned_model(D, heads, dhs, hids, ncls)
    sd = {k: torch.from_numpy(fill.param_value(k, tuple(v.shape))) for k, v in m.state_dict().items()}
    m.load_state_dict(sd)
    m.cuda().train(False)                                   # finetune.py:445: eval-mode semantics with --finetune
    imgs = torch.from_numpy(fill.images(B))
    labels = torch.from_numpy(fill.labels(B, ncls))
    p = {k: v.double().requires_grad_(True) for k, v in sd.items()}
    ref = O.vit_forward(p, imgs.double(), len(heads), heads, 64 ** -0.5)
    loss_ref = O.label_smoothing_ce(ref, labels)
    loss_ref.backward()
    from ofb_amd.losses import DistillationLoss, LabelSmoothingCrossEntropy
    crit = DistillationLoss(LabelSmoothingCrossEntropy(0.1), None, 'none', 0.5, 1.0)
    out = m(imgs.cuda())
    loss = crit(imgs, out, labels.cuda())
    loss.backward()
    torch.cuda.synchronize()
    e = float((out.detach().cpu().double() - ref.detach()).norm() / ref.detach().norm())
    print(f'logits rel err {e:.2e}, loss {float(loss.detach()):.6f} vs {float(loss_ref.detach()):.6f}')
    assert e < 1e-3 and abs(float(loss.detach()) - float(loss_ref.detach())) < 1e-4
    worst = 0.0
    for k, prm in m.named_parameters():
        g_ref = p[k].grad
        if float(g_ref.norm()) < 1e-10:
            continue
        err = float((prm.grad.detach().cpu().double() - g_ref).norm() / g_ref.norm())
        worst = max(worst, err)
        assert err < 1e-3, (k, err)
    print(f'worst grad rel err {worst:.2e}')


def test_pruned_vit_bench_config4_shapes():
    """the shape set `bench.py --mode finetune` reports (configs[4]: FT_EMBED = 264, blocks (heads, head dim, hidden) of
    bench.FT_BLOCKS) against the fp64 oracle, forward + backward: N = 264 is 1.375 GEMM column tiles, head dims 48 / 40 / 32 / 56."""
    import bench
    blocks = bench.FT_BLOCKS[:4]
    test_pruned_vit_forward_backward(bench.FT_EMBED, [h for h, _, _ in blocks], [dh for _, dh, _ in blocks], [hid for _, _, hid in blocks])
