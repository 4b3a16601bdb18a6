"""Guards around the fast path (round 3): stale P-format planes after `.data` / load_state_dict edits, whole-object checkpoints
(torch.save(model), search.py:671-740) staying plane-free, side-stream ordering of the gate gradients of the restricted attention
spaces and of bias-free layers under gradient accumulation, and the device-side non-finite gate (engine.py:146-150)."""
import io

import pytest
import torch

from oracle import ofb_oracle as O
from tests.golden_util import load_case, rel_err
from tests.test_gpu_model import build_product, run_step

pytestmark = pytest.mark.gpu


def _oracle_logits(cfg, st, inputs, sd):
    p = {k: v.detach().cpu().double() for k, v in sd.items()}
    out = O.search_forward(cfg, p, st, inputs['imgs'].double(), inputs['patch_noise'].double(), inputs['droppath_u'].double(), training=True)
    return out['logits']


def test_planes_follow_data_and_load_state_dict_edits():
    """a weight written through `.data` (the reference's compress / resume idiom: moves neither Tensor._version nor any hook) or
    through load_state_dict between two forwards must reach the next forward: every model forward re-makes the planes"""
    z, cfg, st, inputs, lr = load_case('micro_b')
    m = build_product(cfg, st, inputs)
    imgs = inputs['imgs'].cuda()
    with torch.no_grad():
        l0 = m(imgs)[0]
        assert rel_err(l0.cpu(), _oracle_logits(cfg, st, inputs, m.state_dict())) < 1e-4
        v0 = m.blocks[0].mlp.fc1.weight._version
        m.blocks[0].mlp.fc1.weight.data.mul_(1.5)                              # invisible to _version
        m.blocks[1].attn.qkv.weight.data.copy_(m.blocks[1].attn.qkv.weight.data * 0.5)
        m.head.weight.data.add_(0.01)
        assert m.blocks[0].mlp.fc1.weight._version == v0
        l1 = m(imgs)[0]
        assert rel_err(l1.cpu(), _oracle_logits(cfg, st, inputs, m.state_dict())) < 1e-4
        assert rel_err(l1.cpu(), l0.cpu()) > 1e-3                              # the edit does change the function
        sd = {k: (v * 0.9 if k.endswith('proj.weight') else v) for k, v in m.state_dict().items()}
        m.load_state_dict(sd)
        l2 = m(imgs)[0]
        assert rel_err(l2.cpu(), _oracle_logits(cfg, st, inputs, sd)) < 1e-4
        assert rel_err(l2.cpu(), l1.cpu()) > 1e-4


def test_whole_object_checkpoint_has_no_planes_and_reloads_on_the_fast_path():
    """torch.save(model) after a forward + backward: no P-format planes ride along (they live in a registry, not on the
    Parameters), and the reloaded model registers its weights again (ONE multi-tensor refresh per forward)"""
    from ofb_amd import hip
    z, cfg, st, inputs, lr = load_case('micro_a')
    m = build_product(cfg, st, inputs)
    run_step(m, inputs)
    assert hip.weight_registry_size() >= 2 + 4 * cfg.depth
    for p in m.parameters():
        assert not [k for k in vars(p) if k.startswith('_ofb')], 'plane cache attribute on a Parameter'
    buf = io.BytesIO()
    torch.save(m, buf)
    n_param_bytes = sum(v.numel() * v.element_size() for v in m.state_dict().values())
    n_grad_bytes = sum(p.grad.numel() * 4 for p in m.parameters() if p.grad is not None)
    assert buf.tell() < 1.15 * (n_param_bytes + n_grad_bytes) + (1 << 20), (buf.tell(), n_param_bytes, n_grad_bytes)
    buf.seek(0)
    m2 = torch.load(buf, weights_only=False)
    m2._forced = m._forced
    before = hip.weight_registry_size()
    singles = []
    real_into = hip.to_hformat
    hip.to_hformat = lambda *a, **k: (singles.append(1) if k.get('into') is not None else None, real_into(*a, **k))[1]
    try:
        with torch.no_grad():
            l2 = m2(inputs['imgs'].cuda())[0]
            n_first = len(singles)
            l2 = m2(inputs['imgs'].cuda())[0]                                  # second forward: every weight is registered by now
            l1 = m(inputs['imgs'].cuda())[0]
    finally:
        hip.to_hformat = real_into
    assert torch.equal(l1, l2)
    assert hip.weight_registry_size() > before                                 # the reloaded weights hold their own planes
    assert len(singles) == n_first, 'registered weights fell back to per-tensor conversions instead of the multi-tensor refresh'


@pytest.mark.parametrize('tag', ['micro_h', 'micro_c'])
def test_restricted_space_gate_gradients_do_not_race_the_side_stream(tag):
    """head-only / channel-only attention gates are broadcast INSIDE the branch op and their gradient is reduced on the stream
    that produced it: gradients with the side stream on equal those with it off, at a token count above the switch-on point"""
    from ofb_amd import hip, ops
    z, cfg, st, inputs, lr = load_case(tag)
    B = 64                                                                     # 64 x 197 = 12608 tokens >= 12000
    torch.manual_seed(0)
    big = dict(imgs=torch.randn(B, 3, 224, 224), labels=torch.randint(0, cfg.num_classes, (B,)),
               patch_noise=torch.rand(B, cfg.num_patches), droppath_u=torch.rand(2 * cfg.depth, B))
    assert B * (cfg.num_patches + 1) >= ops._SIDE_MIN_TOKENS
    grads = {}
    old = hip.SIDE_STREAM
    try:
        for side in (False, True, True, True):
            hip.SIDE_STREAM = side
            m = build_product(cfg, st, big)
            run_step(m, big)
            g = {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}
            if side in grads:
                assert all(torch.equal(g[k], grads[side][k]) for k in g), 'side-stream run is not reproducible'
            grads[side] = g
    finally:
        hip.SIDE_STREAM = old
    bad = [k for k in grads[False] if not torch.equal(grads[False][k], grads[True][k])]
    assert not bad, bad


def test_bias_free_layer_under_accumulation_waits_for_the_side_stream():
    """a bias-free Linear whose weight gradient is being accumulated (second micro-step): the guard looks at the Parameter's
    .grad (not at the reshaping view's, which is always None) and keeps the product off the side stream"""
    from ofb_amd import hip, ops
    torch.manual_seed(0)
    M, K, N = 12608, 192, 384
    x = torch.randn(M, K, device='cuda')
    W = torch.nn.Parameter(torch.randn(N, K, device='cuda') * 0.05)
    dy = torch.randn(M, N, device='cuda')
    old = hip.SIDE_STREAM
    try:
        res = {}
        for side in (False, True):
            hip.SIDE_STREAM = side
            W.grad = None
            for _ in range(3):                                                 # three micro-steps accumulate into W.grad
                y = ops.Linear.apply(x, W, None)
                y.backward(dy)
            hip.join_side()
            torch.cuda.synchronize()
            res[side] = W.grad.detach().clone()
        assert torch.equal(res[False], res[True])
        ref = 3.0 * (dy.double().t() @ x.double())
        assert rel_err(res[True].cpu(), ref.cpu()) < 1e-5
    finally:
        hip.SIDE_STREAM = old


def test_nonfinite_loss_freezes_optimizer_and_ema():
    """reference engine.py:146-150 stops BEFORE backward when the loss is not finite.  Here the loss is never read per step: a
    device counter trips instead and every later AdamW / EMA launch leaves the tensors untouched"""
    from ofb_amd import engine, hip
    from ofb_amd.utils import ModelEma
    from tests.test_gpu_dp import _crit
    z, cfg, st, inputs, lr = load_case('micro_a')
    m = build_product(cfg, st, inputs)
    opts = engine.build_optimizers(m, 1e-3)
    ema = ModelEma(m, decay=0.9)
    crit = _crit()
    imgs, labels = inputs['imgs'].cuda(), inputs['labels'].cuda()
    hip.reset_nonfinite()
    try:
        engine.search_step(m, crit, imgs, labels, 1.0, opts)
        ema.update(m)
        torch.cuda.synchronize()
        assert int(hip.nonfinite_flag('cuda')) == 0
        good = {k: v.detach().clone() for k, v in m.state_dict().items()}
        good_ema = {k: v.detach().clone() for k, v in ema.ema.state_dict().items()}
        bad_imgs = imgs.clone()
        bad_imgs[0, 0, 0, 0] = float('nan')
        out = engine.search_step(m, crit, bad_imgs, labels, 1.0, opts)
        ema.update(m)
        engine.search_step(m, crit, imgs, labels, 1.0, opts)                   # a later finite step stays frozen too (sticky)
        ema.update(m)
        torch.cuda.synchronize()
        assert not bool(torch.isfinite(out[3]))
        assert int(hip.nonfinite_flag('cuda')) >= 1
        assert all(torch.equal(v, good[k]) for k, v in m.state_dict().items())
        assert all(torch.equal(v, good_ema[k]) for k, v in ema.ema.state_dict().items())
    finally:
        hip.reset_nonfinite()


def test_layernorm_module_used_twice_and_hooked_parameters_get_whole_gradients():
    """LayerNorm's dgamma / dbeta (and the upstream branch's bias gradient) are normally handed to autograd UNREDUCED
    and filled by one multi-job launch at the end of backward.  That is only legal while nobody reads them earlier: a module used
    twice in one graph (autograd adds the two buffers at once), a parameter with a tensor hook, a parameter that already holds a
    gradient - all must take the immediate reduction."""
    from ofb_amd import hip, layers
    torch.manual_seed(0)
    D, rows = 96, 400
    ln = layers.LayerNorm(D, eps=1e-6).cuda()
    with torch.no_grad():
        ln.weight.normal_(1.0, 0.2)
        ln.bias.normal_(0.0, 0.1)
    x1, x2 = torch.randn(rows, D, device='cuda'), torch.randn(rows, D, device='cuda') * 2 + 1
    w1, w2 = torch.randn(rows, D, device='cuda'), torch.randn(rows, D, device='cuda')

    def ref():
        g, b = ln.weight.detach().double().requires_grad_(True), ln.bias.detach().double().requires_grad_(True)
        y = (torch.nn.functional.layer_norm(x1.double(), (D,), g, b, 1e-6) * w1.double()).sum() + \
            (torch.nn.functional.layer_norm(x2.double(), (D,), g, b, 1e-6) * w2.double()).sum()
        y.backward()
        return g.grad, b.grad

    gw, gb = ref()
    seen = []
    for hooked in (False, True):
        ln.weight.grad = ln.bias.grad = None
        h = ln.weight.register_hook(lambda g_: seen.append(g_.clone())) if hooked else None
        ((ln(x1) * w1).sum() + (ln(x2) * w2).sum()).backward()
        torch.cuda.synchronize()
        if h is not None:
            h.remove()
        assert (ln.weight.grad.double() - gw.cuda()).abs().max().item() < 1e-3 * gw.abs().max().item()
        assert (ln.bias.grad.double() - gb.cuda()).abs().max().item() < 1e-3 * gb.abs().max().item()
    # the hook saw complete gradients (their sum is the total), never an unfilled buffer
    assert seen and (sum(seen).double() - gw.cuda()).abs().max().item() < 1e-3 * gw.abs().max().item()
    # a backward that raises must not poison the next step: the queued jobs and the once-per-pass flag are dropped at the next forward
    from ofb_amd import ops
    ops._cb_queued[0] = True
    hip._deferred.append((x1, 1, 1, 1, x1))
    hip.begin_forward()
    assert not ops._cb_queued[0] and not hip._deferred


def test_gradient_accumulation_equals_the_sum_of_the_micro_steps():
    """two micro-steps accumulated into .grad (engine.py:152-169 with accum_iter 2) against the same two backward passes run
    separately and added: every leaf, bit for bit where the addition order is the same.  Regression (round 5): the cls_token
    gradient was returned as a VIEW of the pos_embed gradient; autograd adopted both as the leaves' .grad, which then shared
    storage, and from the second micro-step on each of them accumulated into the other (found by the reference-pinned epoch test,
    tests/test_gpu_engine.py::test_search_one_epoch_matches_reference_run)."""
    z, cfg, st, inputs, lr = load_case('micro_a')
    m = build_product(cfg, st, inputs)
    singles = []
    for rep in range(2):
        for p in m.parameters():
            p.grad = None
        run_step(m, inputs)
        singles.append({k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None})
    for p in m.parameters():
        p.grad = None
    run_step(m, inputs)
    run_step(m, inputs)                                                           # accumulates on top of the first pass
    for k, p in m.named_parameters():
        if p.grad is None:
            continue
        exp = singles[0][k] + singles[1][k]
        err = float((p.grad - exp).abs().max()) / (float(exp.abs().max()) + 1e-30)
        assert err < 1e-5, (k, err)
    spans = sorted((p.grad.data_ptr(), p.grad.data_ptr() + p.grad.numel() * 4, k) for k, p in m.named_parameters() if p.grad is not None)
    for (a0, a1, ka), (b0, b1, kb) in zip(spans, spans[1:]):                      # no two leaves' gradients overlap in memory
        assert a1 <= b0, (ka, kb)
