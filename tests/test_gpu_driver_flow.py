"""The reference's two DRIVERS, end to end, on this package - written against the reference's own call sequence and nothing else:
search.py main() (:374-792: create_model -> correct_require_grad -> the three AdamW instances from its parameter grouping ->
create_scheduler x 3 -> DistributedDataParallel -> DistillationLoss / OFBSearchLOSS -> epochs of search_one_epoch with the periodic
compress() inside -> whole-object checkpoints -> evaluate -> get_flops / give_alphas logging -> the finish_search switch-over (reset
mask ratio, freeze decoder, Mixup + SoftTargetCrossEntropy) -> best.pth -> fuse() -> evaluate -> model_fused.pth) and finetune.py main()
(:251-490: create_model -> intersect(model, searched model) -> ModelEma -> layer-decay groups -> torch.optim.AdamW -> create_scheduler
-> DistributedDataParallel -> get_flops -> train_one_epoch -> evaluate_finetune) at micro scale (DeiT-T, batch 2, a few iterations).
Every name the drivers import is taken from `ofb_amd` as INTEGRATION.md's import swap says; the statements are the drivers'."""
import json
import os
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _args(**kw):
    a = dict(model='deit_tiny_patch16_224_mim', nb_classes=10, drop=0.0, drop_path=0.1, batch_size=2, accum_iter=1, epochs=4, warmup_epochs=1,
             warmup_lr=1e-6, min_lr=1e-5, sched='cosine', cooldown_epochs=0, seed=0, lr=None, blr=2.5e-4, lr_arch=None, blr_arch=2.5e-4,
             lr_decoder=None, blr_decoder=2.5e-4, weight_decay=1e-3, weight_decay_decoder=1e-3, w_head=0.5, w_mlp=0.5, w_patch=0.0, w_embedding=0.5,
             w_flops=5.0, target_flops=0.5, smoothing=0.1, distillation_type='none', distillation_alpha=0.5, distillation_tau=1.0, use_amp=False,
             clip_grad=None, no_entropy=False, no_var=False, no_norm=False, no_progressive=False, max_ratio=0.95, min_ratio=0.75, mae=True,
             fuse_point=50, model_ema=False, model_ema_decay=0.99996, model_ema_force_cpu=False, cutmix_minmax=None, mixup_prob=1.0,
             mixup_switch_prob=0.5, mixup_mode='batch', layer_decay=0.75, finetune='searched', distributed=True, gpu=0)
    a.update(kw)
    return types.SimpleNamespace(**a)


def _loader(n, bs, ncls, seed):
    g = torch.Generator().manual_seed(seed)
    return [(torch.randn(bs, 3, 224, 224, generator=g), torch.randint(0, ncls, (bs,), generator=g)) for _ in range(n)]


def test_search_py_and_finetune_py_flows(tmp_path):
    # ---- the import swap of INTEGRATION.md 1 --------------------------------------------------------------------------------
    from ofb_amd import create_model, Mixup, SoftTargetCrossEntropy
    from ofb_amd.losses import DistillationLoss, OFBSearchLOSS, LabelSmoothingCrossEntropy
    from ofb_amd.optim import AdamW
    from ofb_amd.engine import search_one_epoch, evaluate, train_one_epoch, evaluate_finetune
    from ofb_amd.lr_sched import create_scheduler
    from ofb_amd.utils import ModelEma, NativeScalerWithGradNormCount as NativeScaler, intersect, install_reference_aliases
    from ofb_amd.dp import DistributedDataParallel
    from ofb_amd import utils, lr_decay as lrd

    args = _args()
    output_dir = tmp_path
    device = torch.device('cuda')
    torch.manual_seed(args.seed + utils.get_rank())                                      # search.py:381
    np.random.seed(args.seed)
    data_loader_train, data_loader_val = _loader(6, args.batch_size, args.nb_classes, 1), _loader(2, 3, args.nb_classes, 2)

    # ---- search.py:393-419 ---------------------------------------------------------------------------------------------------
    model = create_model(args.model, pretrained=False, num_classes=args.nb_classes, drop_rate=args.drop, drop_path_rate=args.drop_path,
                         drop_block_rate=None, mae=args.mae, pretrained_strict=False, head_search=False, channel_search=False, method='search',
                         norm_pix_loss=False, attn_search=True, mlp_search=True, embed_search=True, patch_search=False, mask_ratio=1.0)
    with torch.no_grad():                                  # (stands in for the pretrained DeiT the driver loads: a non-zero classifier)
        model.head.weight.normal_(std=.02)
    model.to(device)
    model.correct_require_grad(args.w_head, args.w_mlp, args.w_patch, args.w_embedding)
    model_ema = None
    finish_search = model.finish_search
    mixup_fn = None

    # ---- search.py:486-559: parameter grouping and the three optimizers ---------------------------------------------------------
    groups = {k: ([], []) for k in ('no_decay', 'decay', 'no_decay_decoder', 'decay_decoder', 'archs')}
    skip = model.no_weight_decay() if hasattr(model, 'no_weight_decay') else {}
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        if len(p.shape) == 1 or name.endswith('.bias') or any(ele in name for ele in skip):
            key = 'no_decay' if 'decoder' not in name else 'no_decay_decoder'
        elif 'alpha' in name:
            key = 'archs'
        else:
            key = 'decay' if 'decoder' not in name else 'decay_decoder'
        groups[key][0].append(p)
        groups[key][1].append(name)
    eff_batch_size = args.batch_size * args.accum_iter * utils.get_world_size()
    args.lr, args.lr_arch, args.lr_decoder = (b * eff_batch_size / 256 for b in (args.blr, args.blr_arch, args.blr_decoder))
    optimizer_param = AdamW([{'params': groups['no_decay'][0], 'weight_decay': 0.}, {'params': groups['decay'][0], 'weight_decay': args.weight_decay}],
                            {0: groups['no_decay'][1], 1: groups['decay'][1]}, lr=args.lr)
    assert len(groups['decay_decoder'][0]) and len(groups['archs'][0])
    optimizer_decoder = AdamW([{'params': groups['no_decay_decoder'][0], 'weight_decay': 0.},
                               {'params': groups['decay_decoder'][0], 'weight_decay': args.weight_decay_decoder}],
                              {0: groups['no_decay_decoder'][1], 1: groups['decay_decoder'][1]}, lr=args.lr_decoder)
    optimizer_arch = AdamW(groups['archs'][0], {0: groups['archs'][1]}, lr=args.lr_arch, weight_decay=1e-3)
    loss_scaler = NativeScaler()

    # ---- search.py:572-631 -------------------------------------------------------------------------------------------------------
    lr_scheduler_params, _ = create_scheduler(args.epochs, args.warmup_epochs, args.warmup_lr, args.min_lr, args, optimizer_param, len(data_loader_train))
    lr_scheduler_arch, _ = create_scheduler(args.epochs, args.warmup_epochs, args.warmup_lr, args.min_lr, args, optimizer_arch, len(data_loader_train))
    lr_scheduler_decoder, _ = create_scheduler(args.epochs, args.warmup_epochs, args.warmup_lr, args.min_lr, args, optimizer_decoder, len(data_loader_train))
    assert optimizer_param.param_groups[0]['lr'] == args.warmup_lr               # the schedule starts at the warm-up rate
    criterion = LabelSmoothingCrossEntropy(smoothing=args.smoothing)
    model_without_ddp = model
    if args.distributed:
        model = DistributedDataParallel(model, device_ids=[args.gpu], find_unused_parameters=True)
        model_without_ddp = model.module
    n_parameters = sum(p.numel() for p in model.parameters() if p.requires_grad)
    assert n_parameters > 5e6
    criterion = DistillationLoss(criterion, None, args.distillation_type, args.distillation_alpha, args.distillation_tau)
    criterion = OFBSearchLOSS(criterion, device, attn_w=args.w_head, mlp_w=args.w_mlp, patch_w=args.w_patch, embedding_w=args.w_embedding,
                              flops_w=args.w_flops, entropy=not args.no_entropy, var=not args.no_var, norm=not args.no_norm)

    # (test input: every module's alpha leans on ONE cell, so that the first compress() of epoch 0 finishes the search - the event
    #  the driver's later branches wait for; a real run gets there after tens of epochs)
    pe = model_without_ddp.patch_embed
    for mod, (i, j) in [(pe, (0, 10))] + [(m, (0, 4)) for b in model_without_ddp.blocks for m in (b.attn, b.mlp)]:
        a = torch.full_like(mod.alpha.data, -8.0)
        a[i, j] = 0.0
        mod.alpha.data.copy_(a)

    # ---- search.py:633-773: the epoch loop --------------------------------------------------------------------------------------------
    target_flops, max_soft_accuracy, flag, execute_prune, log = args.target_flops, 0.0, True, False, []
    for epoch in range(0, 2):
        if finish_search and flag:
            flag = False
            if hasattr(model, 'module'):
                model.module.reset_mask_ratio(1.0)
                model.module.freeze_decoder()
            else:
                model.reset_mask_ratio(1.0)
                model.freeze_decoder()
            optimizer_decoder = None
            lr_scheduler_decoder = None
            mixup_fn = Mixup(mixup_alpha=0.8, cutmix_alpha=1.0, cutmix_minmax=args.cutmix_minmax, prob=args.mixup_prob,
                             switch_prob=args.mixup_switch_prob, mode=args.mixup_mode, label_smoothing=args.smoothing, num_classes=args.nb_classes)
            criterion.base_criterion.base_criterion = SoftTargetCrossEntropy()
            max_soft_accuracy = 0.0
        torch.cuda.synchronize()
        train_stats, finish_search, execute_prune, optimizer_param, optimizer_decoder, optimizer_arch = search_one_epoch(
            model, criterion, target_flops, data_loader_train, optimizer_param, optimizer_decoder, optimizer_arch, lr_scheduler_params,
            lr_scheduler_arch, lr_scheduler_decoder, device, epoch, args.clip_grad, model_ema, mixup_fn, use_amp=args.use_amp,
            finish_search=finish_search, args=args, progressive=not args.no_progressive, max_ratio=args.max_ratio, min_ratio=args.min_ratio)
        torch.cuda.synchronize()
        ckpt = {'model': model_without_ddp, 'optimizer_param': optimizer_param.state_dict(),
                'optimizer_arch': optimizer_arch.state_dict() if optimizer_arch is not None else None,
                'optimizer_decoder': optimizer_decoder.state_dict() if optimizer_decoder is not None else None, 'epoch': epoch,
                'model_ema': model_ema.ema.state_dict() if args.model_ema else model_ema, 'scaler': loss_scaler.state_dict(), 'args': args}
        if finish_search and execute_prune:
            utils.save_on_master(ckpt, output_dir / 'model_pruned.pth')
        utils.save_on_master(ckpt, output_dir / 'running_ckpt.pth')
        test_stats = evaluate(data_loader_val, model, device, use_amp=False)
        max_soft_accuracy = max(max_soft_accuracy, test_stats['acc1'])
        if test_stats['acc1'] >= max_soft_accuracy:
            utils.save_on_master({'model': model_without_ddp, 'epoch': epoch, 'model_ema': model_ema, 'scaler': loss_scaler.state_dict(), 'args': args},
                                 output_dir / 'best.pth')
        n_parameters_updated = sum(p.numel() for name, p in model.named_parameters()
                                   if p.requires_grad and 'decoder' not in name and 'alpha' not in name and 'score' not in name)
        flops = model.module.get_flops()[1].item() if hasattr(model, 'module') else model.get_flops()[1].item()
        log_stats = {**{f'train_{k}': v for k, v in train_stats.items()}, **{f'soft_test_{k}': v for k, v in test_stats.items()}, 'epoch': epoch,
                     'n_parameters': n_parameters_updated, 'n_gflops': flops}
        log.append(json.dumps(log_stats))
        if not finish_search:
            alphas_attn, alphas_mlp, alphas_patch, alphas_embed = model.module.give_alphas()
            json.dumps({'epoch': epoch, 'attn': alphas_attn, 'mlp': alphas_mlp, 'patch': alphas_patch, 'embed': alphas_embed})
        if epoch == args.fuse_point and hasattr(model_without_ddp, 'fused') and not model_without_ddp.fused:
            break
    # what the flow must have gone through
    s0, s1 = json.loads(log[0]), json.loads(log[1])
    assert finish_search and not execute_prune                        # epoch 0 cut and finished; epoch 1 ran the finished model
    assert (output_dir / 'model_pruned.pth').exists() and optimizer_arch is None and optimizer_decoder is None
    assert {'train_loss_total', 'train_loss_param', 'train_lr_param', 'train_loss_arch', 'train_loss_decoder', 'soft_test_acc1', 'n_gflops'} <= set(s0)
    assert 'train_loss_arch' not in s1 and 'train_loss_decoder' not in s1
    assert all(np.isfinite(v) for v in s0.values()) and all(np.isfinite(v) for v in s1.values())
    assert s1['n_gflops'] < 0.7 and s1['n_parameters'] < s0['n_parameters'] * 1.0 + 1       # (DeiT-T is 1.25 GMACs; the forced cells cut it)
    assert optimizer_param.param_groups[0]['lr'] > args.warmup_lr                        # the per-iteration schedule moved the rate

    # ---- search.py:775-787: fuse the best model, evaluate, model_fused.pth ------------------------------------------------------------
    assert utils.is_main_process() and finish_search and not execute_prune and not model_without_ddp.fused
    best_state = torch.load(output_dir / 'best.pth', map_location='cpu', weights_only=False)
    best_model = best_state['model']
    best_model = best_model.cuda()
    with torch.no_grad():
        before = evaluate(data_loader_val, best_model, device, use_amp=False)
    best_model.fuse()
    test_stats = evaluate(data_loader_val, best_model, device, use_amp=False)
    assert abs(test_stats['loss'] - before['loss']) < 1e-4 * max(1.0, abs(before['loss'])) and test_stats['acc1'] == before['acc1']
    utils.save_on_master({'model': best_model, 'epoch': best_state['epoch']}, output_dir / 'model_fused.pth')
    model.reducer.close()

    # ---- finetune.py:251-490 ----------------------------------------------------------------------------------------------------------
    fargs = _args(model='deit_tiny_patch16_224_finetune', model_ema=True, lr=5e-4, weight_decay=0.05, warmup_epochs=0, epochs=2)
    model = create_model(fargs.model, num_classes=fargs.nb_classes, drop_rate=fargs.drop, drop_path_rate=fargs.drop_path, drop_block_rate=None)
    state_dict = torch.load(output_dir / 'model_pruned.pth', map_location='cpu', weights_only=False)['model']
    model = intersect(model, state_dict)                                                  # finetune.py:322-323 (--pretrained-path)
    model.to(device)
    model_ema = ModelEma(model, decay=fargs.model_ema_decay, device='cpu' if fargs.model_ema_force_cpu else '', resume='')
    param_groups = lrd.param_groups_lrd(model, fargs.weight_decay, no_weight_decay_list=model.no_weight_decay(), layer_decay=fargs.layer_decay)
    optimizer_param = torch.optim.AdamW(param_groups, lr=fargs.lr)                        # finetune.py:383: torch's own AdamW
    loss_scaler = NativeScaler()
    lr_scheduler, _ = create_scheduler(fargs.epochs, fargs.warmup_epochs, fargs.warmup_lr, fargs.min_lr, fargs, optimizer_param, len(data_loader_train))
    mixup_fn = Mixup(mixup_alpha=0.8, cutmix_alpha=1.0, cutmix_minmax=None, prob=1.0, switch_prob=0.5, mode='batch', label_smoothing=0.1,
                     num_classes=fargs.nb_classes)
    criterion = DistillationLoss(SoftTargetCrossEntropy(), None, 'none', 0.5, 1.0)
    model_without_ddp = model
    if fargs.distributed:
        model = DistributedDataParallel(model, device_ids=[fargs.gpu], find_unused_parameters=True)
        model_without_ddp = model.module
    n_flops = model_without_ddp.get_flops()
    assert 0 < n_flops < 1.25e9                                                           # the searched shapes, not DeiT-T's
    w0 = model_without_ddp.blocks[0].mlp.fc1.weight.detach().clone()
    e0 = model_ema.ema.blocks[0].mlp.fc1.weight.detach().clone()
    train_stats = train_one_epoch(model, criterion, data_loader_train, optimizer_param, lr_scheduler, device, 0, loss_scaler, fargs.clip_grad,
                                  model_ema, mixup_fn, use_amp=fargs.use_amp, args=fargs, set_training_mode=fargs.finetune == '')
    test_stats = evaluate_finetune(data_loader_val, model, device, use_amp=False)
    torch.cuda.synchronize()
    assert set(train_stats) == {'loss', 'lr'} and np.isfinite(train_stats['loss']) and np.isfinite(test_stats['loss'])
    assert not torch.equal(w0, model_without_ddp.blocks[0].mlp.fc1.weight) and not torch.equal(e0, model_ema.ema.blocks[0].mlp.fc1.weight)
    assert optimizer_param.param_groups[0]['lr'] != optimizer_param.param_groups[-1]['lr']      # layer-wise decay reached the groups
    assert not model_without_ddp.training                                                 # finetune.py:445: eval-mode semantics with --finetune
    torch.save({'model': model_without_ddp.state_dict(), 'model_ema': model_ema.ema.state_dict(), 'scaler': loss_scaler.state_dict()},
               output_dir / 'checkpoint.pth')
    model.reducer.close()
