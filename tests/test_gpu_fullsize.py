"""BASELINE configs[1]-[4] at FULL size (DeiT-S bs 128; the same under a one-rank RCCL exchange with the default 25-MB buckets;
DeiT-B bs 64; the pruned finetune subnet at bs 256).  configs[1] - the size the metric is quoted on -, configs[3] and configs[4] are
compared with the fp64 oracle ELEMENT by element (the oracle walks the batch in chunks: tests/fullsize_util.py, about a minute of host
time each, once); all four are also checked through size-independent properties - run-to-run bit-identity (every reduction has a fixed order), batch-chunk consistency of the
forward, and central finite differences of the full loss with respect to parameters."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(seed=0, drop_path=0.1, arch='deit_small', B=128):
    import ofb_amd
    from ofb_amd.losses import OFBSearchLOSS, DistillationLoss, LabelSmoothingCrossEntropy
    torch.manual_seed(seed)
    dev = torch.device('cuda')
    m = ofb_amd.create_model(f'{arch}_patch16_224_mim', method='search', num_classes=1000, drop_path_rate=drop_path,
                             patch_search=False, mask_ratio=1.0).to(dev)
    m.correct_require_grad(0.5, 0.5, 0, 0.5)
    m.adjust_masking_ratio(0.0, 20, 100)
    crit = OFBSearchLOSS(DistillationLoss(LabelSmoothingCrossEntropy(0.1), None, 'none', 0.5, 1.0), dev, 0.5, 0.5, 0.0, 0.5, 5.0)
    g = torch.Generator(device=dev).manual_seed(1234)
    imgs = torch.randn(B, 3, 224, 224, device=dev, generator=g)
    labels = torch.randint(0, 1000, (B,), device=dev, generator=g)
    L, depth = 196, 12
    m._forced = dict(patch_noise=torch.rand(B, L, device=dev, generator=g), droppath_u=torch.rand(2 * depth, B, device=dev, generator=g))
    return m, crit, imgs, labels


def _loss(m, crit, imgs, labels):
    logits, (dec, _) = m(imgs)
    base, arch = crit(imgs, logits, labels, m, 'arch', 1.0, False)
    return base + arch + (base / dec).detach() * dec, logits


def elementwise_report(got, ref, rel=1e-3, floor_bits=20):
    """the two per-tensor measures of VERDICT r5 #2: norm-wise ||a - b|| / ||b||, and the worst element against the bound
    |a - b| <= rel * |b| + 2^-floor_bits * max|b| (returned as a fraction of the bound: <= 1 passes)."""
    a, b = got.double().reshape(-1), ref.double().reshape(-1)
    diff = (a - b).abs()
    bound = rel * b.abs() + 2.0 ** -floor_bits * float(b.abs().max())
    return float(diff.norm() / (b.norm() + 1e-300)), float((diff / bound).max())


def test_deit_small_bs128_matches_the_fp64_oracle_element_by_element():
    """configs[1] at the size `bench.py` times: one search micro-step of DeiT-S at batch 128 (epoch-0 state: w_p 0.99, keep ratio 0.95,
    DropPath 0.1, random-normal images, the constructor's random initialisation with a non-zero classifier so that both losses send
    gradients into the trunk) against the fp64 oracle on the same parameters and inputs.  Every gradient tensor must meet (a) the
    norm-wise 1e-3 of north_star AND (b) the element-wise bound |a - b| <= 1e-3 |b| + 2^-20 max|b|; the weight gradients fed by the
    two plane tensors that DESIGN section 3 lists as mostly below the H-format window (`dH` -> fc1.weight / fc1.bias, the LayerNorm-backward
    planes -> proj / fc2 / qkv weights) are also checked per group of 128 rows."""
    import time
    from oracle import ofb_oracle as O
    from ofb_amd.layers import trunc_normal_
    from tests.fullsize_util import chunked_oracle_step
    m, crit, imgs, labels = _setup()
    with torch.no_grad():
        trunc_normal_(m.head.weight, std=.02)                # (the constructor zeroes it: a pretrained classifier is not zero)
    m.train()
    total, logits = _loss(m, crit, imgs, labels)
    total.backward()
    torch.cuda.synchronize()
    cfg = O.Config(**O.DEIT_SMALL, num_classes=1000, drop_path_rate=0.1)
    st = O.SearchState(w_p=0.99, keep_ratio=0.95)
    assert abs(m.patch_ratio_list[0] - 0.95) < 1e-12 and all(abs(x.w_p - 0.99) < 1e-12 for x in m.searchable_modules)
    p = {k: v.detach().cpu().double().requires_grad_(k != 'alpha_patch') for k, v in m.state_dict().items()}
    threads = torch.get_num_threads()
    torch.set_num_threads(min(16, __import__('os').cpu_count() or 16))
    t0 = time.time()
    try:
        ref = chunked_oracle_step(cfg, p, st, imgs.cpu().double(), labels.cpu(), m._forced['patch_noise'].cpu().double(),
                                  m._forced['droppath_u'].cpu().double(), chunk=16)
    finally:
        torch.set_num_threads(threads)
    print(f'fp64 oracle, DeiT-S bs 128 in chunks of 16: {time.time() - t0:.1f} s on the host')
    lg_n, lg_e = elementwise_report(logits.detach().cpu(), ref['logits'])
    print(f'logits: norm-wise {lg_n:.2e}, worst element {lg_e:.3f} of its bound')
    assert lg_n < 1e-4 and lg_e <= 1.0
    got_total = float(total.detach())
    assert abs(got_total - float(ref['loss_total'])) <= 1e-4 * abs(float(ref['loss_total'])), (got_total, float(ref['loss_total']))
    _check_gradients(m, p, 180)


def _check_gradients(m, p, min_checked, groups=('fc1.weight', 'fc2.weight', 'proj.weight', 'qkv.weight'), floor_bits=20):
    """every gradient tensor of `m` against the oracle's p[k].grad: norm-wise 1e-3, element-wise |a - b| <= 1e-3 |b| + 2^-20 max|b|,
    and (2-D weights with a multiple of 128 rows) every group of 128 rows norm-wise"""
    worst_n, worst_e, worst_g = (0.0, ''), (0.0, ''), (0.0, '')
    checked = 0
    for k, prm in m.named_parameters():
        rg = p[k].grad
        if rg is None or float(rg.norm()) < 1e-12 or k.endswith('qkv.bias'):       # (k third of a qkv bias gradient: a mathematical zero)
            continue
        g = prm.grad.detach().cpu()
        n_err, e_err = elementwise_report(g, rg, floor_bits=floor_bits)
        checked += 1
        worst_n = max(worst_n, (n_err, k))
        worst_e = max(worst_e, (e_err, k))
        assert n_err < 1e-3, (k, n_err)
        assert e_err <= 1.0, (k, e_err)
        if g.dim() == 2 and g.shape[0] % 128 == 0 and any(t in k for t in groups):
            a, b = g.double().view(-1, 128, g.shape[1]), rg.view(-1, 128, g.shape[1])
            grp = (a - b).flatten(1).norm(dim=1) / b.flatten(1).norm(dim=1).clamp_min(1e-300)
            worst_g = max(worst_g, (float(grp.max()), k))
            assert float(grp.max()) < 1e-3, (k, float(grp.max()))
    assert checked > min_checked, checked
    print(f'{checked} gradient tensors; worst norm-wise {worst_n[0]:.2e} ({worst_n[1]}); worst element {worst_e[0]:.3f} of its bound '
          f'(floor 2^-{floor_bits} max|b|; {worst_e[1]}); worst 128-row group of a weight gradient {worst_g[0]:.2e} ({worst_g[1]})')


def _host_threads():
    import os
    return min(16, os.cpu_count() or 16)


def test_deit_base_bs64_matches_the_fp64_oracle_element_by_element():
    """configs[3] at its full size (DeiT-B, 64 images per GPU, epoch-0 state): the same comparison as the DeiT-S one above - logits,
    total loss and every gradient tensor element by element against the fp64 oracle walking the batch in chunks of 8."""
    import time
    from oracle import ofb_oracle as O
    from ofb_amd.layers import trunc_normal_
    from tests.fullsize_util import chunked_oracle_step
    m, crit, imgs, labels = _setup(arch='deit_base', B=64)
    with torch.no_grad():
        trunc_normal_(m.head.weight, std=.02)
    m.train()
    total, logits = _loss(m, crit, imgs, labels)
    total.backward()
    torch.cuda.synchronize()
    cfg = O.Config(**O.DEIT_BASE, num_classes=1000, drop_path_rate=0.1)
    st = O.SearchState(w_p=0.99, keep_ratio=0.95)
    p = {k: v.detach().cpu().double().requires_grad_(k != 'alpha_patch') for k, v in m.state_dict().items()}
    threads = torch.get_num_threads()
    torch.set_num_threads(_host_threads())
    t0 = time.time()
    try:
        ref = chunked_oracle_step(cfg, p, st, imgs.cpu().double(), labels.cpu(), m._forced['patch_noise'].cpu().double(),
                                  m._forced['droppath_u'].cpu().double(), chunk=8)
    finally:
        torch.set_num_threads(threads)
    print(f'fp64 oracle, DeiT-B bs 64 in chunks of 8: {time.time() - t0:.1f} s on the host')
    lg_n, lg_e = elementwise_report(logits.detach().cpu(), ref['logits'])
    print(f'logits: norm-wise {lg_n:.2e}, worst element {lg_e:.3f} of its bound')
    assert lg_n < 1e-4 and lg_e <= 1.0
    got_total = float(total.detach())
    assert abs(got_total - float(ref['loss_total'])) <= 1e-4 * abs(float(ref['loss_total'])), (got_total, float(ref['loss_total']))
    _check_gradients(m, p, 180)


def test_finetune_subnet_bs256_matches_the_fp64_oracle_element_by_element():
    """configs[4] at its full size: the pruned subnet of `bench.py --mode finetune` (embed 264, ragged heads / hidden widths, eval-mode
    semantics as finetune.py:445 leaves them) at 256 images with a soft-target cross entropy (what Mixup feeds engine.train_one_epoch,
    engine.py:42-44): logits, loss and every gradient tensor element by element against the oracle's plain-ViT forward in fp64 (chunks
    of 32 samples: the loss is a mean of per-sample terms).  The weight gradients here contract over 50 432 token rows, twice configs[1]'s:
    the absolute floor of the element-wise bound is taken as 2^-19 max|b| (measured: 0.996 of the 2^-20 bound on one element of
    blocks.7.mlp.fc1.weight, every tensor norm-wise <= 1.3e-6 - too close to the edge for a gate, so the floor is one bit wider here)."""
    import sys, os, time
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    import ofb_amd
    from oracle import ofb_oracle as O
    from ofb_amd.layers import trunc_normal_
    dev = torch.device('cuda')
    torch.manual_seed(0)
    model, _, _ = bench.build_finetune_subnet(ofb_amd, dev, 1000)
    with torch.no_grad():
        trunc_normal_(model.head.weight, std=.02)            # (zero at construction: no gradient would reach the trunk)
    ofb_amd.hip.bump_weight_epoch()
    model.train(False)
    B = 256
    g = torch.Generator(device=dev).manual_seed(4321)
    imgs = torch.randn(B, 3, 224, 224, device=dev, generator=g)
    target = torch.softmax(torch.randn(B, 1000, device=dev, generator=g) * 3, -1)
    logits = model(imgs)
    loss = ofb_amd.data.SoftTargetCrossEntropy()(logits, target)
    loss.backward()
    torch.cuda.synchronize()
    heads = [int(blk.attn.num_heads) for blk in model.blocks]
    assert heads == [h for h, _, _ in bench.FT_BLOCKS]
    scale = float(model.blocks[0].attn.scale)
    assert all(float(blk.attn.scale) == scale for blk in model.blocks)
    p = {k: v.detach().cpu().double().requires_grad_(True) for k, v in model.state_dict().items()}
    im64, tg64 = imgs.cpu().double(), target.cpu().double()
    threads = torch.get_num_threads()
    torch.set_num_threads(_host_threads())
    t0 = time.time()
    ref_logits, ref_loss = [], 0.0
    try:
        for lo in range(0, B, 32):
            out = O.vit_forward(p, im64[lo:lo + 32], len(heads), heads, scale)
            part = -(tg64[lo:lo + 32] * torch.log_softmax(out, -1)).sum() / B
            part.backward()
            ref_logits.append(out.detach())
            ref_loss += float(part.detach())
    finally:
        torch.set_num_threads(threads)
    print(f'fp64 oracle, finetune subnet bs 256 in chunks of 32: {time.time() - t0:.1f} s on the host')
    lg_n, lg_e = elementwise_report(logits.detach().cpu(), torch.cat(ref_logits))
    print(f'logits: norm-wise {lg_n:.2e}, worst element {lg_e:.3f} of its bound; loss {float(loss.detach()):.6f} vs {ref_loss:.6f}')
    assert lg_n < 1e-4 and lg_e <= 1.0
    assert abs(float(loss.detach()) - ref_loss) <= 1e-5 * abs(ref_loss)
    _check_gradients(model, p, 100, floor_bits=19)


def test_full_size_step_is_bit_reproducible():
    m, crit, imgs, labels = _setup()
    m.train()
    grads = []
    for _ in range(2):
        for p in m.parameters():
            p.grad = None
        total, logits = _loss(m, crit, imgs, labels)
        total.backward()
        torch.cuda.synchronize()
        grads.append({k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None})
    assert len(grads[0]) > 190
    for k in grads[0]:
        assert torch.equal(grads[0][k], grads[1][k]), k               # stream-K partial sums, dQ partials, column sums: fixed order
    assert all(bool(torch.isfinite(g).all()) for g in grads[0].values())


def test_full_size_step_on_the_cus_a_data_parallel_exchange_leaves():
    """configs[2] on one GPU: with more than one rank `dp.GradAllReducer` plans every GEMM for 256 - 32 = 224 CUs (OFB_TUNE_GEMM_CUS; RCCL's
    kernels hold the rest during backward).  Other rounds, tails and K cuts, the same arithmetic: loss and every gradient tensor of the
    bs-128 step agree with the 256-CU run to summation-order level, and the step stays bit-reproducible on the smaller grid."""
    from ofb_amd import hip
    m, crit, imgs, labels = _setup()
    m.train()
    runs = []
    try:
        for cus in (0, 224, 224):
            hip.tune(hip.TUNE_GEMM_CUS, cus)
            for p in m.parameters():
                p.grad = None
            total, logits = _loss(m, crit, imgs, labels)
            total.backward()
            torch.cuda.synchronize()
            runs.append((float(total.detach()), logits.detach().clone(), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}))
    finally:
        hip.tune(hip.TUNE_GEMM_CUS, 0)
    (l0, lg0, g0), (l1, lg1, g1), (l2, lg2, g2) = runs
    assert abs(l0 - l1) <= 1e-6 * abs(l0), (l0, l1)
    assert float((lg0 - lg1).norm() / lg0.norm()) < 1e-6
    worst = (0.0, '')
    for k in g0:
        nrm = float(g0[k].double().norm())
        if nrm < 1e-12:
            continue
        err = float((g0[k].double() - g1[k].double()).norm()) / nrm
        worst = max(worst, (err, k))
        assert err < 2e-5, (k, err)
        assert torch.equal(g1[k], g2[k]), k
    print(f'224-CU plans against 256-CU plans: worst gradient tensor {worst[0]:.2e} ({worst[1]})')


def test_full_size_forward_is_batch_separable():
    """every sample's logits depend on that sample alone: the 128-image forward equals four 32-image forwards (different GEMM
    tilings / stream-K cuts, so equal to rounding, not bitwise)."""
    m, crit, imgs, labels = _setup()
    m.eval()
    with torch.no_grad():
        full = m(imgs)[0]
        parts = torch.cat([m(imgs[i:i + 32])[0] for i in range(0, 128, 32)])
    err = float((full - parts).norm() / full.norm())
    print(f'batch-chunk rel err {err:.2e}')
    assert err < 2e-6


def test_full_size_gradients_match_finite_differences():
    m, crit, imgs, labels = _setup(drop_path=0.1)
    m.train()
    total, _ = _loss(m, crit, imgs, labels)
    total.backward()
    probes = [('blocks.5.attn.alpha', (1, 3)), ('blocks.7.mlp.alpha', (0, 4)), ('patch_embed.alpha', (0, 9)), ('blocks.2.mlp.score', (0, 77)),
              ('blocks.9.attn.score', (3, 20))]
    params = dict(m.named_parameters())
    for name, idx in probes:
        p = params[name]
        g = float(p.grad[idx])
        eps = 2e-2 if 'alpha' in name else 1e-1
        vals = []
        for sgn in (1.0, -1.0):
            with torch.no_grad():
                p[idx] += sgn * eps
            with torch.no_grad():
                vals.append(float(_loss(m, crit, imgs, labels)[0]))
            with torch.no_grad():
                p[idx] -= sgn * eps
        fd = (vals[0] - vals[1]) / (2 * eps)
        print(f'{name}{idx}: grad {g:.5e}  finite difference {fd:.5e}')
        assert abs(fd - g) <= 2e-2 * max(abs(g), abs(fd)) + 2e-4, name


def test_deit_base_bs64_step_properties():
    """configs[3] (reference search.py:617-620 with --model deit_base..., bs 64 per GPU): M = 12608 token rows, D = 768 - other
    stream-K plans and tile counts than DeiT-S.  Bit-reproducible gradients, batch-separable forward, finite differences."""
    m, crit, imgs, labels = _setup(arch='deit_base', B=64)
    m.train()
    grads = []
    for _ in range(2):
        for p in m.parameters():
            p.grad = None
        total, _ = _loss(m, crit, imgs, labels)
        total.backward()
        torch.cuda.synchronize()
        grads.append({k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None})
    assert len(grads[0]) > 190
    for k in grads[0]:
        assert torch.equal(grads[0][k], grads[1][k]), k
        assert bool(torch.isfinite(grads[0][k]).all()), k
    params = dict(m.named_parameters())
    for name, idx in [('blocks.4.attn.alpha', (1, 2)), ('blocks.8.mlp.alpha', (0, 3)), ('patch_embed.alpha', (0, 7)), ('blocks.1.mlp.score', (0, 100))]:
        p = params[name]
        g = float(grads[0][name][idx])
        eps = 2e-2 if 'alpha' in name else 1e-1
        vals = []
        for sgn in (1.0, -1.0):
            with torch.no_grad():
                p[idx] += sgn * eps
                vals.append(float(_loss(m, crit, imgs, labels)[0]))
                p[idx] -= sgn * eps
        fd = (vals[0] - vals[1]) / (2 * eps)
        print(f'deit-b {name}{idx}: grad {g:.5e}  finite difference {fd:.5e}')
        assert abs(fd - g) <= 2e-2 * max(abs(g), abs(fd)) + 2e-4, name
    m.eval()
    with torch.no_grad():
        full = m(imgs)[0]
        parts = torch.cat([m(imgs[i:i + 16])[0] for i in range(0, 64, 16)])
    err = float((full - parts).norm() / full.norm())
    print(f'deit-b batch-chunk rel err {err:.2e}')
    assert err < 2e-6


def test_finetune_subnet_bs256_step_properties():
    """configs[4] (reference finetune.py:421-424, engine.py:18-72): the pruned OFB-DeiT-C-like subnet of bench.py --mode finetune
    (embed 264, ragged heads / hidden widths) at bs 256: M = 50432 token rows on 264-wide tiles (the 256 x 96 tile and its tail
    policy).  Bit-reproducible gradients, batch-separable forward, finite differences on weights."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    import ofb_amd
    dev = torch.device('cuda')
    torch.manual_seed(0)
    model, macs, nparams = bench.build_finetune_subnet(ofb_amd, dev, 1000)
    model.train(False)                                       # finetune.py:445: eval-mode semantics during finetune
    g = torch.Generator(device=dev).manual_seed(4321)
    imgs = torch.randn(256, 3, 224, 224, device=dev, generator=g)
    target = torch.softmax(torch.randn(256, 1000, device=dev, generator=g) * 3, -1)
    crit = ofb_amd.data.SoftTargetCrossEntropy()

    def loss_of():
        return crit(model(imgs), target)

    grads = []
    for _ in range(2):
        for p in model.parameters():
            p.grad = None
        loss_of().backward()
        torch.cuda.synchronize()
        grads.append({k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None})
    assert len(grads[0]) > 100
    for k in grads[0]:
        assert torch.equal(grads[0][k], grads[1][k]), k
        assert bool(torch.isfinite(grads[0][k]).all()), k
    params = dict(model.named_parameters())
    from ofb_amd import hip
    for name in ('blocks.3.mlp.fc1.weight', 'blocks.7.attn.qkv.weight', 'head.weight', 'blocks.0.norm1.weight'):
        p = params[name]
        idx = tuple(min(3, s - 1) for s in p.shape)
        gnum = float(grads[0][name][idx])
        eps = 0.05
        vals = []
        for sgn in (1.0, -1.0):
            with torch.no_grad():
                p[idx] += sgn * eps
                hip.bump_weight_epoch()
                vals.append(float(loss_of()))
                p[idx] -= sgn * eps
        hip.bump_weight_epoch()
        fd = (vals[0] - vals[1]) / (2 * eps)
        print(f'finetune {name}{idx}: grad {gnum:.5e}  finite difference {fd:.5e}')
        assert abs(fd - gnum) <= 3e-2 * max(abs(gnum), abs(fd)) + 2e-5, name
    with torch.no_grad():
        full = model(imgs)
        parts = torch.cat([model(imgs[i:i + 64]) for i in range(0, 256, 64)])
    err = float((full - parts).norm() / full.norm())
    print(f'finetune batch-chunk rel err {err:.2e}')
    assert err < 2e-6


def test_deit_small_bs128_one_rank_rccl_exchange_is_transparent():
    """configs[2] (reference search.py:617-620: DistributedDataParallel around the model): DeiT-S bs 128 under a ONE-rank RCCL group with
    the reducer's DEFAULT 25-MB buckets, side stream on, two optimizer steps - the parameters must be BIT-identical to the run
    without a reducer (world size 1: the average IS the gradient)."""
    import os
    import socket
    import torch.distributed as dist
    import ofb_amd
    from ofb_amd import engine, hip
    dev = torch.device('cuda', 0)
    created = False
    if not dist.is_initialized():
        s = socket.socket()
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
        s.close()
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        hip.ensure_side_stream(dev)
        dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{port}', world_size=1, rank=0, device_id=dev)
        created = True
    try:
        finals = []
        for use_reducer in (False, True):
            m, crit, imgs, labels = _setup(seed=3)
            m.train()
            opts = engine.build_optimizers(m, 2.5e-4 * 128 / 256)
            red = ofb_amd.dp.GradAllReducer(list(m.parameters()), force_collective=True) if use_reducer else None
            if red is not None:
                assert len(red.buckets) >= 3 and red.bucket_bytes == 25 * 1024 * 1024
            for _ in range(2):
                engine.search_step(m, crit, imgs, labels, 1.0, opts, reducer=red)
            torch.cuda.synchronize()
            if red is not None:
                assert red.collectives >= 2 * len(red.buckets)
                red.close()
            finals.append({k: v.detach().clone() for k, v in m.state_dict().items()})
        for k in finals[0]:
            assert torch.equal(finals[0][k], finals[1][k]), k
    finally:
        if created:
            torch.cuda.synchronize()
            dist.destroy_process_group()
