"""Epoch engines with the reference's protocol (reference engine.py:18-219) on the fused HIP path.

`search_step` is one micro-step without any host synchronisation; `search_one_epoch` wraps it in the
reference's per-iteration schedule (progressive masking ratio, w_p warm-up, 3 optimizers, periodic compress).
"""
import math
import os
import sys
import time

import torch

# backward() on the CALLING thread: the autograd engine otherwise hands the graph to its per-device worker thread, and every one of the
# ~100 Python backward functions of a step then takes the GIL from a thread without a lasting interpreter state - measured 9.5 vs
# 7.9 ms of host time per search step (scripts/lab/host_profile.py), which is what decides the step time of the post-compress shapes
# and of small batches.  OFB_BACKWARD_THREAD=engine restores torch's default.
_BACKWARD_ON_CALLER = os.environ.get('OFB_BACKWARD_THREAD', 'caller') != 'engine'


def run_backward(loss):
    """loss.backward() of the epoch engines (and of bench.py's restatement of them)"""
    if _BACKWARD_ON_CALLER:
        with torch.autograd.set_multithreading_enabled(False):
            loss.backward()
    else:
        loss.backward()


def mix_losses(loss, decoder_loss):
    """engine.py:134-144: base + arch + stopgrad(base / decoder_loss) * decoder_loss."""
    if isinstance(loss, tuple):
        base, arch = loss
    else:
        base, arch = loss, None
    dec = None if isinstance(decoder_loss, float) else decoder_loss
    if (arch is not None or dec is not None) and isinstance(base, torch.Tensor) and base.is_cuda:
        from . import ops
        if ops._scalar_ok(base, arch, dec):
            return base, arch, ops.TotalLoss.apply(base, arch, dec)       # one launch instead of four one-element ATen kernels
    total = base if arch is None else base + arch
    if dec is not None:
        total = total + (base / dec).detach() * dec
    return base, arch, total


def search_step(model, criterion, samples, targets, target_flops, optimizers, finish_search=False, accum_iter=1,
                do_step=True, reducer=None):
    """forward + OFBSearchLOSS + backward (+ DP exchange) + optimizer steps.  Returns device scalars."""
    outputs, (decoder_loss, _) = model(samples)
    loss = criterion(samples, outputs, targets, model, 'arch', target_flops, finish_search)
    base, arch, total = mix_losses(loss, decoder_loss)
    if total.is_cuda:
        # engine.py:146-150 without the per-step host sync: a non-finite loss bumps a device counter that makes every later AdamW /
        # EMA launch a no-op; the epoch loop reads it at its print points and stops
        from . import hip
        hip.watch_nonfinite(total)
    scale = 1.0 / accum_iter
    if reducer is not None:
        reducer.prescaled = True
        reducer.sync = do_step                           # exchange once per accumulation window, on its closing micro-step
        scale *= reducer.grad_scale                      # SUM all-reduce of (loss / world) gradients == average
    run_backward(total * scale if scale != 1.0 else total)
    if reducer is not None:
        reducer.finalize()
    if do_step:
        for opt in optimizers:
            if opt is not None:
                opt.step()
        for opt in optimizers:
            if opt is not None:
                opt.zero_grad(set_to_none=True)
    return base, arch, decoder_loss, total


def _agreed_length(data_loader, device, reducer):
    """iterations of this epoch: len(data_loader), or - with a data-parallel reducer - the smallest length among the ranks (dp.common_length:
    every rank runs the same number of steps, so nobody waits in a collective for a rank whose loader came up short; the schedules
    t = epoch + it / n_iter of engine.py:102 then use that common length on every rank)"""
    n = len(data_loader)
    if reducer is None:
        return n
    from .dp import common_length
    return common_length(n, device, reducer.group)


class GraphedStep:
    """One whole training step (forward, losses, backward, optimizer steps) captured ONCE into a hipGraph and replayed: ~650
    kernel launches per step leave the host as a single hipGraphLaunch (the launch-bound sizes - small batches, pruned models - are
    host bound otherwise).  `fn()` must be the step as it runs eagerly, on static input tensors, after a few eager executions
    (allocator / workspaces warm); it may use the side stream (it is forked from and joined into the capturing stream).  The AdamW
    instances passed in read their learning rate and bias corrections from device memory inside the graph (optim.AdamW.begin_capture)
    and are refreshed before every replay; random draws (DropPath, patch masking) come from torch's graph-safe generator.
    With a data-parallel reducer (pass it as `reducer`): the bucketed RCCL all-reduces of the step are captured with it (RCCL
    collectives are stream-ordered kernels; every rank captures and replays the same sequence) - the launch-bound sizes keep the
    graph's gain when a second rank exists.  Not across compress() (shapes and buckets change: capture again)."""

    def __init__(self, fn, optimizers, reducer=None, side_stream=False):
        self.fn, self.opts = fn, [o for o in optimizers if o is not None]
        self.reducer, self.side_stream = reducer, side_stream
        self.graph, self.out, self.k, self._keep = None, None, 0, None

    def capture(self, warm_steps=2):
        from . import hip
        hip.join_side()
        torch.cuda.synchronize()
        # Autograd runs every parameter's AccumulateGrad on the stream the parameter was set up on and syncs it with the stream of
        # backward.  For a model built on the legacy default stream that pulls the (uncapturable) null stream into the capture and
        # hipStreamEndCapture faults.  So the whole job must live on ONE non-default stream: create it before the model is built
        # (`torch.cuda.set_stream(torch.cuda.Stream())`, as bench.py does) and the capture uses that same stream.
        cur = torch.cuda.current_stream()
        if cur == torch.cuda.default_stream():
            raise RuntimeError('GraphedStep: build the model and run the step on a non-default stream '
                               '(torch.cuda.set_stream(torch.cuda.Stream()) before creating it)')
        self.stream = cur
        # The captured step is a SINGLE-stream chain unless side_stream=True was asked for: how the runtime places a graph's parallel
        # branches on hardware queues depends on which streams exist in the process (queues x priorities x RCCL: scripts/lab/
        # queue_matrix.sh) - the same bs-128 capture replays in 25.9 or 37.3 ms - and the sizes graphs are for (launch-bound: few
        # tokens) do not use the side stream anyway (ops._side_ok).  One stream: 25.7 ms at bs 128, always.
        # (A process-wide switch on purpose: ops._side_ok / dp._launch run inside autograd's per-device worker thread, which does not
        # inherit Python thread-locals of the thread that calls backward().  Capture is a set-up phase: do not run another model's
        # backward in a second thread while it is in progress.)
        side_was = hip.SIDE_STREAM
        hip.SIDE_STREAM = side_was and self.side_stream
        try:
            self._capture(warm_steps)
        finally:
            hip.SIDE_STREAM = side_was

    def _capture(self, warm_steps):
        from . import hip
        for _ in range(warm_steps):
            self.out = self.fn()
        hip.join_side()
        self.stream.synchronize()
        self.out = None
        keep = hip.begin_capture_arena()
        for o in self.opts:
            o.begin_capture()
        graph = torch.cuda.CUDAGraph()
        # With a process group in the process its watchdog thread polls the events of earlier collectives while this thread
        # captures: under the default 'global' capture mode a runtime call from ANOTHER thread can invalidate the capture (or abort
        # the process from that thread); 'thread_local' restricts the checks to the capturing thread.
        import torch.distributed as dist
        mode = 'thread_local' if (dist.is_available() and dist.is_initialized()) else 'global'
        try:
            with torch.cuda.graph(graph, stream=self.stream, capture_error_mode=mode):
                self.out = self.fn()
        finally:
            hip.end_capture_arena()
            for o in self.opts:
                o.end_capture()
        self.graph, self._keep, self.k = graph, keep, 0

    def __call__(self):
        if self.graph is None:
            self.capture()
        for o in self.opts:
            o.refresh_hyper(self.k)
        self.graph.replay()
        self.k += 1
        return self.out


def search_one_epoch(model, criterion, target_flops, data_loader, optimizer_param, optimizer_decoder, optimizer_arch,
                     lr_scheduler_param, lr_scheduler_arch, lr_scheduler_decoder, device, epoch, max_norm=0, model_ema=None,
                     mixup_fn=None, set_training_mode=True, use_amp=False, finish_search=False, args=None, progressive=True,
                     max_ratio=0.95, min_ratio=0.75, reducer=None, print_freq=10):
    """reference engine.py:75-219 (fp32 path; --use-amp is not supported)."""
    if use_amp:
        raise NotImplementedError('apex AMP path is off in the reference workflow')
    model.train(set_training_mode)
    net = model.module if hasattr(model, 'module') else model
    if reducer is None:
        reducer = getattr(model, 'reducer', None)        # dp.DistributedDataParallel carries its own
    accum_iter = args.accum_iter
    n_iter = _agreed_length(data_loader, device, reducer)
    for opt in (optimizer_param, optimizer_decoder, optimizer_arch):
        if opt is not None:
            opt.zero_grad(set_to_none=True)
    execute_pruned = False
    stats, t0 = {}, time.time()
    if device is not None and torch.device(device).type == 'cuda':
        from . import hip
        hip.reset_nonfinite(device)                      # the device-side NaN gate is scoped to this epoch loop, not to the process
    # epoch statistics stay on the device (MetricLogger of the reference, engine.py:87-90,186-199: one meter per key, global_avg = sum
    # of the values a meter received / how many it received): running sums of the four losses and of the three learning rates, a
    # count of non-finite totals and the number of updates of the three meter families; the host reads them at print points only
    sums, stats, hsum = None, {}, [0.0] * 6
    for it, (samples, targets) in zip(range(n_iter), data_loader):
        samples = samples.to(device, non_blocking=True)
        targets = targets.to(device, non_blocking=True)
        if mixup_fn is not None:
            samples, targets = mixup_fn(samples, targets)
        if it % accum_iter == 0:
            t = it / n_iter + epoch
            if progressive:
                net.adjust_masking_ratio(t, args.warmup_epochs, args.epochs, max_ratio=max_ratio, min_ratio=min_ratio)
            for m in net.searchable_modules:
                if not m.finish_search:
                    m.update_w(t, args.warmup_epochs)
        boundary = (it + 1) % accum_iter == 0
        has_arch = optimizer_arch is not None and not finish_search
        opts = (optimizer_param, optimizer_arch if has_arch else None, optimizer_decoder)
        base, arch, dec, total = search_step(net, criterion, samples, targets, target_flops, opts, finish_search, accum_iter,
                                             do_step=boundary, reducer=reducer)
        if boundary:
            gstep = epoch * n_iter + it
            lr_scheduler_param.step_update(gstep)
            if has_arch:
                lr_scheduler_arch.step_update(gstep)
            if optimizer_decoder is not None:
                lr_scheduler_decoder.step_update(gstep)
        if model_ema is not None:
            model_ema.update(model)
        has_dec = optimizer_decoder is not None and not isinstance(dec, float)
        with torch.no_grad():
            tf = total.detach().float()
            zero = torch.zeros_like(tf)
            vals = torch.stack([tf, base.detach().float(), arch.detach().float() if (arch is not None and has_arch) else zero,
                                dec.detach().float() if has_dec else zero])
            # device part [loss_total, loss_param, loss_arch, loss_decoder, non-finite]; host part [lr_param, lr_arch, lr_decoder, n, n_arch,
            # n_decoder] (known without a sync: joined to the device part at the print points only)
            row = torch.cat([torch.nan_to_num(vals, nan=0.0, posinf=0.0, neginf=0.0).double(), (~torch.isfinite(tf)).double().reshape(1)])
            sums = row if sums is None else sums + row
        for i, v in enumerate((optimizer_param.param_groups[0]['lr'], optimizer_arch.param_groups[0]['lr'] if has_arch else 0.0,
                               optimizer_decoder.param_groups[0]['lr'] if has_dec else 0.0, 1.0, float(has_arch), float(has_dec))):
            hsum[i] += v
        if it % print_freq == 0 or it == n_iter - 1:          # the only host syncs of the loop
            last = it == n_iter - 1
            # every rank must take the same decision (a rank that left alone would leave the others in the next collective): the
            # sums - the non-finite count among them - are reduced over the ranks at every look; at the epoch's end the reduced
            # vector is the epoch's statistics over ALL ranks (engine.py:216, utils.py:41-52), the iteration counts included
            from .dp import sum_across_ranks
            tot, _ = sum_across_ranks(torch.cat([sums, torch.tensor(hsum, dtype=torch.float64, device=sums.device)]),
                                      reducer.group if reducer is not None else None)
            host = tot.tolist()
            lv = float(total.detach())
            # no parameter, moment or EMA update was applied after the first non-finite loss: the device-side watch (search_step)
            # froze the AdamW / EMA kernels (reference: a host check every micro-step, engine.py:146-148)
            if host[4] > 0:
                print('Loss is {}, stopping training'.format(lv if not math.isfinite(lv) else 'non-finite on a rank / in an earlier micro-step'))
                sys.exit(1)
            n, n_arch, n_dec = host[8], host[9], host[10]
            stats = dict(loss_param=host[1] / n, loss_total=host[0] / n, lr_param=host[5] / n)
            if n_arch > 0:
                stats.update(loss_arch=host[2] / n_arch, lr_arch=host[6] / n_arch)
            if n_dec > 0:
                stats.update(loss_decoder=host[3] / n_dec, lr_decoder=host[7] / n_dec)
            print(f'Epoch: [{epoch}] [{it}/{n_iter}] loss_total: {lv:.5f} ' + ' '.join(f'{k}(avg): {v:.6f}' for k, v in stats.items())
                  + f' time: {(time.time() - t0) / (it + 1):.4f}' + (' (all ranks)' if last else ''))
        every = max(1, n_iter // 3 // accum_iter)
        if not finish_search and boundary and ((it + 1) // accum_iter) % every == 0 and hasattr(net, 'compress'):
            finish_search, execute_prune, optimizer_param, optimizer_decoder, optimizer_arch = net.compress(
                0.2, optimizer_param, optimizer_decoder, optimizer_arch)
            execute_pruned |= execute_prune
            if reducer is not None and execute_prune:            # compress() replaced Parameters: re-bucket the exchange
                reducer.rebuild([p for p in net.parameters()])
            if model_ema is not None:                            # engine.py:212-213: the EMA adopts the cut shapes right away
                model_ema.update(model)
            if finish_search:
                optimizer_arch, lr_scheduler_arch = None, None
    return stats, finish_search, execute_pruned, optimizer_param, optimizer_decoder, optimizer_arch


def train_one_epoch(model, criterion, data_loader, optimizer, lr_schedule, device, epoch, loss_scaler=None, max_norm=0,
                    model_ema=None, mixup_fn=None, set_training_mode=True, use_amp=False, args=None, reducer=None, print_freq=10):
    """reference engine.py:18-72 (finetune of the pruned subnet)."""
    if use_amp:
        raise NotImplementedError('apex AMP path is off in the reference workflow')
    model.train(set_training_mode)
    if reducer is None:
        reducer = getattr(model, 'reducer', None)        # dp.DistributedDataParallel carries its own
    accum_iter = args.accum_iter
    optimizer.zero_grad(set_to_none=True)
    n_iter, stats = _agreed_length(data_loader, device, reducer), {}
    if device is not None and torch.device(device).type == 'cuda':
        from . import hip
        hip.reset_nonfinite(device)
    sums, hsum = None, [0.0, 0.0]                        # device: [sum of losses, non-finite losses]; host: [sum of lr, iterations]
    for it, (samples, targets) in zip(range(n_iter), data_loader):
        samples, targets = samples.to(device, non_blocking=True), targets.to(device, non_blocking=True)
        if mixup_fn is not None:
            samples, targets = mixup_fn(samples, targets)
        loss = criterion(samples, model(samples), targets)
        if loss.is_cuda:
            from . import hip
            hip.watch_nonfinite(loss)                    # engine.py:48-50 without a host sync: freezes AdamW / EMA once non-finite
        if reducer is not None:
            reducer.sync = (it + 1) % accum_iter == 0    # one exchange per accumulation window
            reducer.prescaled = False                    # plain SUM, divided by the world size in finalize()
        run_backward(loss / accum_iter if accum_iter != 1 else loss)
        if reducer is not None:
            reducer.finalize()
        if (it + 1) % accum_iter == 0:
            optimizer.step()
            optimizer.zero_grad(set_to_none=True)
            lr_schedule.step_update(epoch * n_iter + it)
        if model_ema is not None:
            model_ema.update(model)
        with torch.no_grad():
            lf = loss.detach().double()
            row = torch.stack([torch.nan_to_num(lf, nan=0.0, posinf=0.0, neginf=0.0), (~torch.isfinite(lf)).double()])
            sums = row if sums is None else sums + row
        hsum[0] += optimizer.param_groups[0]['lr']       # engine.py:66 (after the window's scheduler update)
        hsum[1] += 1.0
        if it % print_freq == 0 or it == n_iter - 1:
            # the ranks decide together (see search_one_epoch); at the epoch's end the reduced sums are the statistics of ALL ranks
            from .dp import sum_across_ranks
            tot, _ = sum_across_ranks(torch.cat([sums, torch.tensor(hsum, dtype=torch.float64, device=sums.device)]),
                                      reducer.group if reducer is not None else None)
            host = tot.tolist()
            lv = float(loss.detach())
            if host[1] > 0:
                print('Loss is {}, stopping training'.format(lv if not math.isfinite(lv) else 'non-finite on a rank / in an earlier micro-step'))
                sys.exit(1)
            stats = dict(loss=host[0] / host[3], lr=host[2] / host[3])                     # MetricLogger.global_avg (engine.py:72)
    return stats


def _evaluate(data_loader, model, device, use_amp, search_model, across_ranks):
    """shared body of evaluate / evaluate_finetune.  The meters of the reference's MetricLogger (engine.py:246-249, :283-286) are
    kept as ONE device vector - [sum of per-batch mean losses, batches, top-1 hits, top-5 hits, samples] - that the host reads once,
    after the last batch (the reference reads three scalars per batch)."""
    if use_amp:
        raise NotImplementedError('apex / autocast evaluation is off in the reference workflow (search.py:726, finetune.py:461 pass use_amp=False)')
    model.eval()
    acc = None
    for images, target in data_loader:
        images, target = images.to(device, non_blocking=True), target.to(device, non_blocking=True)
        output = model(images)
        if search_model and isinstance(output, tuple):          # engine.py:241 `output, _ = model(images)`
            output = output[0]
        output = output.float()
        if output.is_cuda:
            from . import ops
            loss = ops.LabelSmoothingCE.apply(output, target, 0.0)            # torch.nn.CrossEntropyLoss(): mean NLL of the batch
        else:
            loss = torch.nn.functional.cross_entropy(output, target)
        hit = output.topk(min(5, output.shape[1]), 1).indices == target.view(-1, 1)      # timm.utils.accuracy(topk=(1, 5))
        one = torch.ones((), dtype=torch.float64, device=output.device)
        row = torch.stack([loss.double(), one, hit[:, 0].sum().double(), hit.sum().double(), one * target.numel()])
        acc = row if acc is None else acc + row
    if acc is None:
        acc = torch.zeros(5, dtype=torch.float64, device=device)
    if across_ranks:                                            # engine.py:288 metric_logger.synchronize_between_processes() (C4)
        from .dp import sum_across_ranks
        acc, _ = sum_across_ranks(acc, getattr(getattr(model, 'reducer', None), 'group', None))
    loss_sum, batches, c1, c5, n = acc.tolist()
    # MetricLogger.global_avg: the loss meter is updated with n = 1 per batch (mean of the batch means), the accuracy meters with
    # n = batch size (sample-weighted)
    stats = {'loss': loss_sum / max(batches, 1.0), 'acc1': 100.0 * c1 / max(n, 1.0), 'acc5': 100.0 * c5 / max(n, 1.0)}
    print('* Acc@1 {acc1:.3f} Acc@5 {acc5:.3f} loss {loss:.3f}'.format(**stats))
    return stats


@torch.no_grad()
def evaluate(data_loader, model, device, use_amp=False):
    """reference engine.py:222-257: top-1 / top-5 accuracy and CE loss of the SEARCH model (it returns (logits, aux)) in eval mode,
    on this rank's loader only (the reference's rank reduction is commented out there, :253; search.py:725-726 calls it on rank 0)."""
    return _evaluate(data_loader, model, device, use_amp, True, False)


@torch.no_grad()
def evaluate_finetune(data_loader, model, device, use_amp=False):
    """reference engine.py:260-290: the same for the plain (pruned) ViT of finetune.py, whose forward returns the logits; the meters
    are summed over the ranks (:288)."""
    return _evaluate(data_loader, model, device, use_amp, False, True)


def param_groups(model, weight_decay=1e-3):
    """search.py:486-508 grouping -> dict of lists (params / decoder / archs optimizers)."""
    skip = model.no_weight_decay() if hasattr(model, 'no_weight_decay') else []
    g = dict(nodecay=[], decay=[], decoder_nodecay=[], decoder_decay=[], arch=[])
    names = {k: [] for k in g}
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        if len(p.shape) == 1 or name.endswith('.bias') or any(s in name for s in skip):
            key = 'decoder_nodecay' if 'decoder' in name else 'nodecay'
        elif 'alpha' in name:
            key = 'arch'
        else:
            key = 'decoder_decay' if 'decoder' in name else 'decay'
        g[key].append(p)
        names[key].append(name)
    return g, names


def build_optimizers(model, lr, lr_arch=None, lr_decoder=None, weight_decay=1e-3):
    """the three AdamW instances of search.py:549-559."""
    from .optim import AdamW
    g, names = param_groups(model)
    opt_p = AdamW([{'params': g['nodecay'], 'weight_decay': 0.}, {'params': g['decay'], 'weight_decay': weight_decay}],
                  {0: names['nodecay'], 1: names['decay']}, lr=lr)
    opt_d = None
    if g['decoder_decay']:
        opt_d = AdamW([{'params': g['decoder_nodecay'], 'weight_decay': 0.}, {'params': g['decoder_decay'], 'weight_decay': weight_decay}],
                      {0: names['decoder_nodecay'], 1: names['decoder_decay']}, lr=lr_decoder or lr)
    opt_a = AdamW(g['arch'], {0: names['arch']}, lr=lr_arch or lr, betas=(0.5, 0.999), weight_decay=1e-3) if g['arch'] else None
    return opt_p, opt_a, opt_d
