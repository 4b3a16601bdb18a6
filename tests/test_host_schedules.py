"""Host-side scalar bookkeeping of the drivers (CPU): the per-iteration cosine schedule with warm-up prefix and `lr_scale` groups
(reference lr_sched.py:14-122 on timm's CosineLRScheduler, restated: SURVEY 8c), the layer-wise decay groups of finetune.py:378-383
(lr_decay.py:15-76) and the loss-scaler stand-in the drivers checkpoint."""
import math
import types

import pytest
import torch


def _opt(groups):
    ps = [torch.nn.Parameter(torch.zeros(2)) for _ in groups]
    return torch.optim.SGD([dict(params=[p], lr=lr, **extra) for p, (lr, extra) in zip(ps, groups)], lr=0.1)


def test_cosine_schedule_with_warmup_prefix_and_lr_scale():
    from ofb_amd.lr_sched import create_scheduler
    opt = _opt([(1e-3, {}), (1e-3, {'lr_scale': 0.25})])
    args = types.SimpleNamespace(sched='cosine', cooldown_epochs=0, seed=0)
    n_iter, epochs, warm = 10, 5, 1
    sched, n_ep = create_scheduler(epochs, warm, 1e-6, 1e-5, args, opt, n_iter)
    assert n_ep == (epochs - warm) * n_iter                                  # get_cycle_length() + cooldown (in updates: t_in_epochs=False)
    assert opt.param_groups[0]['lr'] == 1e-6 and opt.param_groups[1]['lr'] == 0.25e-6     # starts at the warm-up rate
    sched.step(3)                                                            # per-EPOCH calls do nothing (search.py never relies on them)
    assert opt.param_groups[0]['lr'] == 1e-6
    T, W = (epochs - warm) * n_iter, warm * n_iter
    for k in (0, 3, 9, 10, 11, 25, 49, 50, 70):
        sched.step_update(k)
        if k < W:
            exp = 1e-6 + k * (1e-3 - 1e-6) / W
        elif k - W < T:                                                      # one cycle (lr_cycle_limit 1), then lr_min
            exp = 1e-5 + 0.5 * (1e-3 - 1e-5) * (1 + math.cos(math.pi * (k - W) / T))
        else:
            exp = 1e-5
        assert opt.param_groups[0]['lr'] == pytest.approx(exp, rel=1e-12), k
        assert opt.param_groups[1]['lr'] == pytest.approx(0.25 * exp, rel=1e-12), k
    with pytest.raises(NotImplementedError):
        create_scheduler(epochs, warm, 1e-6, 1e-5, types.SimpleNamespace(sched='step', cooldown_epochs=0), opt, n_iter)


def test_layer_decay_groups():
    import ofb_amd
    from ofb_amd import lr_decay as lrd
    m = ofb_amd.VisionTransformer(embed_dim=64, depth=3, num_heads=2, num_classes=5)
    groups = lrd.param_groups_lrd(m, 0.05, no_weight_decay_list=m.no_weight_decay(), layer_decay=0.5)
    L = 4                                                                    # depth + 1
    name_of = {id(p): n for n, p in m.named_parameters()}
    seen = 0
    for g in groups:
        names = [name_of[id(p)] for p in g['params']]
        seen += len(names)
        layers = {lrd.get_layer_id_for_vit(n, L) for n in names}
        assert len(layers) == 1
        layer = layers.pop()
        assert g['lr_scale'] == 0.5 ** (L - layer)
        for n in names:
            p = dict(m.named_parameters())[n]
            assert (g['weight_decay'] == 0.0) == (p.ndim == 1 or n in m.no_weight_decay()), n
    assert seen == len(list(m.parameters()))
    assert lrd.get_layer_id_for_vit('cls_token', L) == 0 and lrd.get_layer_id_for_vit('patch_embed.proj.weight', L) == 0
    assert lrd.get_layer_id_for_vit('blocks.2.mlp.fc1.weight', L) == 3 and lrd.get_layer_id_for_vit('head.bias', L) == L


def test_loss_scaler_stand_in_and_rank_helpers():
    from ofb_amd import utils, dp
    sc = utils.NativeScalerWithGradNormCount()
    st = sc.state_dict()
    assert {'scale', 'growth_factor', 'backoff_factor', 'growth_interval', '_growth_tracker'} <= set(st)      # GradScaler's keys
    sc.load_state_dict(dict(st, scale=128.0))
    assert sc.state_dict()['scale'] == 128.0
    lin = torch.nn.Linear(3, 1)
    opt = torch.optim.SGD(lin.parameters(), lr=0.1)
    w0 = lin.weight.detach().clone()
    norm = sc(lin(torch.ones(2, 3)).sum(), opt, clip_grad=None, parameters=lin.parameters())
    assert float(norm) > 0 and not torch.equal(w0, lin.weight)
    assert utils.get_rank() == 0 and utils.get_world_size() == 1 and utils.is_main_process() and not utils.is_dist_avail_and_initialized()
    assert dp.common_length(7) == 7                                          # no process group: the loader's own length
    a = types.SimpleNamespace(gpu=None, dist_url='env://')
    import os
    saved = {k: os.environ.pop(k) for k in ('RANK', 'WORLD_SIZE', 'SLURM_PROCID') if k in os.environ}
    try:
        utils.init_distributed_mode(a)
    finally:
        os.environ.update(saved)
    assert a.distributed is False
