"""Data-parallel gradient exchange for the search step: bucketed all-reduce (RCCL over xGMI on MI355X;
backend 'nccl' IS RCCL on ROCm) launched from post-accumulate-grad hooks so it overlaps with the rest of
backward.  Replaces DistributedDataParallel(find_unused_parameters=True) of reference search.py:617-620
(collective C1 of SURVEY.md 2.2).  One process per GPU; no model or sequence sharding exists on this path.

* Construction (and `rebuild()` after `compress()` replaced Parameters) broadcasts rank 0's copy of EVERY tensor
  handed in - trainable or frozen, plus `extra_tensors` such as a finished module's plain-tensor score - exactly as
  DDP's constructor does (search.py:619), after checking that all ranks hold the same shapes.  Ranks that were seeded
  per rank (search.py:381) or resumed from different files therefore start from identical replicas.
* Buckets are persistent flat fp32 buffers filled in reverse parameter order (gradients become ready
  last-block-first).  The weight-gradient GEMMs write straight into their slice of the bucket (`grad_slot`, used by
  ops.py) so the big tensors are never copied; small gradients (biases, LayerNorm, alpha / score) are copied in when
  the bucket is launched.  After the reduce each parameter's .grad IS its slice of the reduced bucket.
* Parameters whose gradient never arrives in a window (frozen alphas, a finished decoder) are skipped: their slice
  travels as zeros and is not installed.
"""
import contextlib
import os

import torch
import torch.distributed as dist

_active = None        # the reducer whose buckets ops.py may write weight gradients into (one model per process)


def grad_slot(param):
    """Slice of the active reducer's flat bucket where `param`'s gradient should be written, or None (no reducer,
    unknown tensor, a gradient is already being accumulated in this window, or the slot was already handed out in this
    backward pass: a weight used twice in one graph - tied weights, a module called twice - gets ONE direct write, the
    second product goes to a fresh buffer and autograd adds the two)."""
    r = _active
    if r is None or not r.sync:
        return None
    key = param.data_ptr()
    ent = r._slot_of.get(key)
    if ent is None:
        return None
    p, view = ent
    if p.grad is not None or p.numel() != param.numel() or key in r._handed:
        return None
    r._handed.add(key)
    return view.view(param.shape)


def _flat_chunks(tensors, limit=64 * 1024 * 1024):
    """groups of tensors of at most `limit` bytes (one collective each)"""
    cur, size = [], 0
    for t in tensors:
        n = t.numel() * t.element_size()
        if cur and size + n > limit:
            yield cur
            cur, size = [], 0
        cur.append(t)
        size += n
    if cur:
        yield cur


def broadcast_tensors(tensors, src=0, process_group=None):
    """rank `src`'s values into every rank's tensors, a few coalesced collectives (DDP's constructor broadcast)."""
    if not dist.is_initialized() or dist.get_world_size(process_group) == 1:
        return
    ts = [t for t in tensors if t is not None and t.numel()]
    # replicas must agree on structure before any payload collective (a mismatch would hang or corrupt silently)
    sig = torch.tensor([len(ts), sum(t.numel() for t in ts), sum((i + 1) * t.numel() for i, t in enumerate(ts)) % (2 ** 31)],
                       dtype=torch.int64, device=ts[0].device if ts else 'cpu')
    lo, hi = sig.clone(), sig.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=process_group)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=process_group)
    if not torch.equal(lo, hi):
        raise RuntimeError('data-parallel replicas disagree on parameter count / shapes: '
                           f'min {lo.tolist()} max {hi.tolist()} (did compress() cut different cells per rank?)')
    by_type = {}
    for t in ts:
        by_type.setdefault((t.dtype, t.device), []).append(t)
    for group in by_type.values():
        for chunk in _flat_chunks(group):
            flat = torch.cat([t.detach().reshape(-1) for t in chunk])
            dist.broadcast(flat, src=src, group=process_group)
            off = 0
            with torch.no_grad():
                for t in chunk:
                    n = t.numel()
                    t.copy_(flat[off:off + n].view_as(t))
                    off += n
    from . import hip
    hip.bump_weight_epoch()                              # `.data` was overwritten: P-format copies of the weights are stale


class GradAllReducer:
    def __init__(self, params, bucket_bytes=25 * 1024 * 1024, process_group=None, force_collective=False, broadcast=True,
                 extra_tensors=(), first_bucket_bytes=4 * 1024 * 1024, reserve_cus=None):
        global _active
        self.group = process_group
        self.reserve_cus = reserve_cus
        # force_collective: issue the all-reduce even in a one-rank group (rehearses the RCCL path on a single GPU)
        self.force_collective = bool(force_collective) and dist.is_initialized()
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.bucket_bytes = bucket_bytes
        # the FIRST bucket to fill (the last layers' gradients: backward produces them first) is small, so that the exchange starts
        # early in backward instead of after 25 MB of gradients exist (DeiT-B: 349 MB in 14 buckets + this one); the others keep
        # the size that amortises a ring all-reduce over xGMI
        self.first_bucket_bytes = min(first_bucket_bytes, bucket_bytes)
        all_params = list(params)
        if broadcast:
            broadcast_tensors([p.data for p in all_params] + list(extra_tensors), 0, process_group)
        self.params = [p for p in all_params if p.requires_grad]
        self.buckets, cur, size = [], [], 0
        for p in reversed(self.params):
            cur.append(p)
            size += p.numel() * 4
            if size >= (self.first_bucket_bytes if not self.buckets else bucket_bytes):
                self.buckets.append(cur)
                cur, size = [], 0
        if cur:
            self.buckets.append(cur)
        self._where, self._slot_of, self._flat, self._views = {}, {}, [], []
        for bi, b in enumerate(self.buckets):
            flat = torch.zeros(sum(p.numel() for p in b), device=b[0].device, dtype=torch.float32)
            views, off = [], 0
            for p in b:
                n = p.numel()
                v = flat[off:off + n].view_as(p)
                views.append(v)
                self._where[p] = bi
                self._slot_of[p.data_ptr()] = (p, v)
                off += n
            self._flat.append(flat)
            self._views.append(views)
        self._ready = [0] * len(self.buckets)
        self._launched = [False] * len(self.buckets)
        self._works = []
        self._handed = set()                             # data pointers whose bucket slot grad_slot() gave out in this window
        self._handles = [p.register_post_accumulate_grad_hook(self._hook) for p in self.params]
        # averaging: callers scale the loss by `grad_scale` (= 1/world, exact for power-of-two worlds) before backward and
        # the exchange is a plain SUM, so no extra pass over the gradients is needed; prescaled=False divides afterwards.
        self.grad_scale = 1.0 / self.world
        self.prescaled = False
        self.wait_seconds, self.collectives, self.finalized = 0.0, 0, 0
        # gradient accumulation: with sync = False a backward only accumulates into the local .grad (DDP's no_sync()); the
        # micro-step that closes the accumulation window exchanges the accumulated gradients once.  Exchanging every
        # micro-step (as the reference does, SURVEY D-7) would SUM the already-reduced part again, because .grad is re-pointed
        # at the reduced bucket and autograd keeps accumulating into it.
        self.sync = True
        _active = self
        self._reserve()

    def _reserve(self, on=True):
        """With more than one rank the exchange's kernels sit on some CUs for the length of backward.  The GEMM's persistent grids count on
        two workgroup slots per CU and use all of a CU's LDS: one resident foreign workgroup that holds ANY LDS keeps a GEMM workgroup
        off its CU, and a single-round launch then ends a whole tile late - measured +5 % per step for 16, 32 or 64 such workgroups
        alike, and back to +0.5...1 % when the GEMM plans for that many CUs fewer (profiles/r06_cu_thief_step_vs_resident_workgroups.txt;
        planning for 64 fewer against 64 residents: +11 %, the cure worse than the disease).  Default: 32 CUs left to the exchange when
        world > 1 (`reserve_cus=` / OFB_DP_RESERVE_CUS: another count, 0 = none); nothing is reserved in a one-rank group."""
        if not torch.cuda.is_available():
            return
        n = self.reserve_cus
        if n is None:
            n = int(os.environ.get('OFB_DP_RESERVE_CUS', '32')) if self.world > 1 else 0
        from . import hip
        if not on or n <= 0:
            if getattr(self, '_reserved', 0):
                hip.tune(hip.TUNE_GEMM_CUS, 0)
            self._reserved = 0
            return
        cus = torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count
        hip.tune(hip.TUNE_GEMM_CUS, max(cus - n, cus // 2))
        self._reserved = n

    def close(self):
        global _active
        for h in self._handles:
            h.remove()
        self._handles = []
        self._reserve(False)
        if _active is self:
            _active = None

    def rebuild(self, params, extra_tensors=()):
        """call after compress() replaced Parameters (fixes the reference's silent de-sync, SURVEY D-6).  Re-broadcasts rank
        0's replica (after a shape check) and keeps the `sync` / `prescaled` settings of the running loop."""
        sync, prescaled = self.sync, self.prescaled
        self.close()
        self.__init__(params, bucket_bytes=self.bucket_bytes, process_group=self.group, force_collective=self.force_collective,
                      extra_tensors=extra_tensors, first_bucket_bytes=self.first_bucket_bytes, reserve_cus=self.reserve_cus)
        self.sync, self.prescaled = sync, prescaled

    def bucket_sizes_mb(self):
        """bucket sizes in launch order (MB)"""
        return [round(f.numel() * 4 / 2 ** 20, 2) for f in self._flat]

    def measure_collectives(self, reps=3):
        """SERIAL cost of one step's exchange: every bucket's all-reduce run alone, back to back, waited for (ms per step; the best of
        `reps`).  Next to the exposed wait of the timed steps it tells how much of the exchange backward hid.  Collective: every rank
        calls it at the same point.  The bucket contents are garbage afterwards (call it after the timed region)."""
        import time
        if not dist.is_initialized() or (self.world == 1 and not self.force_collective):
            return 0.0
        best = None
        for _ in range(reps):
            if self._flat and self._flat[0].is_cuda:
                torch.cuda.synchronize()
            t0 = time.perf_counter()
            for flat in self._flat:
                dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            if self._flat and self._flat[0].is_cuda:
                torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) * 1e3
            best = dt if best is None else min(best, dt)
        return best

    def _launch(self, bi):
        self._launched[bi] = True
        from . import hip
        flat = self._flat[bi]
        if flat.is_cuda and hip.SIDE_STREAM:
            # The bucket is finished ON THE SIDE STREAM, after the weight-gradient GEMMs that write its slices (ops.py): the side
            # stream first waits for the main stream's position (the small gradients and the queued column sums are complete there),
            # then runs the column sums, the one copy launch and hands the bucket to RCCL, whose stream orders itself behind the
            # stream the call is made on.  The main stream never waits in the middle of backward (joining it here, four buckets per
            # step, serialised the side-stream GEMMs with the critical path: 27.2 vs 25.8 ms at one rank); it meets the side stream
            # at the end of backward and the exchange in finalize().
            ctx = hip.side_work(flat.device)
        else:
            hip.join_side()
            ctx = contextlib.nullcontext()
        on_side = not isinstance(ctx, contextlib.nullcontext)
        with ctx:
            if flat.is_cuda:
                hip.flush_deferred()                     # LayerNorm / bias gradients whose reduction was queued (ops._ln_colsum)
                if on_side:                              # its inputs were allocated on the main stream: referenced until the join
                    hip._side_keep.append(hip._deferred_keep[0])
            live, jobs = [], []
            for p, v in zip(self.buckets[bi], self._views[bi]):
                if p.grad is None:
                    jobs.append((None, v))               # travels as zeros, is not installed afterwards
                    continue
                if p.grad.data_ptr() != v.data_ptr():    # small gradients / accumulated windows: copied into the bucket
                    g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                    jobs.append((g, v))
                    p.grad = v
                live.append(p)
            if jobs:
                if flat.is_cuda:                         # ONE launch for all of them (was ~150 copy_ / zero_ launches per step)
                    self._copy_keep = (hip.multi_copy(jobs, flat.device), [g for g, _ in jobs])
                    if on_side:
                        hip._side_keep.append(self._copy_keep)
                else:                                    # CPU tensors exist only in the gloo rehearsal of the launcher / protocol
                    for g, v in jobs:
                        v.zero_() if g is None else v.copy_(g)
            if not live:
                return
            if self.world > 1 or self.force_collective:
                work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            else:
                work = None
            self._works.append((bi, work))

    def _hook(self, p):
        if not self.sync:
            return
        bi = self._where[p]
        self._ready[bi] += 1
        if self._ready[bi] == len(self.buckets[bi]):
            self._launch(bi)

    def finalize(self):
        """wait for the exchanges of this backward and install the averaged gradients."""
        if not self.sync:
            return
        # buckets with frozen / unused members never fill up; under gradient accumulation a bucket may hold gradients from
        # earlier micro-steps only.  Every rank sees the same set (it follows from requires_grad and the graph), so launch
        # every bucket that holds any gradient.
        for bi in range(len(self.buckets)):
            if not self._launched[bi] and any(p.grad is not None for p in self.buckets[bi]):
                self._launch(bi)
        import time
        t0 = time.perf_counter()
        for bi, work in self._works:
            if work is not None:
                work.wait()
                if not self.prescaled:
                    self._flat[bi].div_(self.world)
        # host-side time spent waiting for the exchanges of this step (the EXPOSED part: what backward did not hide) and the number of
        # collectives, for bench.py's per-rank report
        self.wait_seconds += time.perf_counter() - t0
        self.collectives += sum(1 for _, w in self._works if w is not None)
        self.finalized += 1
        if self._flat and self._flat[0].is_cuda:
            # late bucket launches (above) and one-rank / no-collective runs finish their buckets on the side stream and nothing else
            # joins it before the optimizer: whoever reads .grad after finalize() (clipping, a norm log, a test) must see them complete
            from . import hip
            hip.join_side()
        self._works = []
        self._ready = [0] * len(self.buckets)
        self._launched = [False] * len(self.buckets)
        self._handed.clear()


def sum_across_ranks(vec, process_group=None):
    """Collective C4 of SURVEY 2.2 (reference engine.py:216 / :70 `metric_logger.synchronize_between_processes()`, utils.py:41-52): the
    epoch's running sums (losses, counts) of every rank in ONE fused all-reduce; returns (summed vector, world size).  A no-op
    without a process group."""
    if not dist.is_initialized() or dist.get_world_size(process_group) == 1:
        return vec, 1
    out = vec.detach().clone()
    dist.all_reduce(out, op=dist.ReduceOp.SUM, group=process_group)
    return out, dist.get_world_size(process_group)


def common_length(n, device=None, process_group=None):
    """the number of iterations EVERY rank can run: min over the ranks of `n` (one tiny all-reduce per epoch).  The reference relies on
    its samplers handing every rank the same count (RASampler / DistributedSampler truncate or pad, samplers.py:35-53); a rank whose
    loader came up one batch short would otherwise leave the others waiting in the gradient exchange of a step it never runs (C1) or
    in the statistics reduction at its own early epoch end (C4).  A no-op without a process group."""
    if not dist.is_initialized() or dist.get_world_size(process_group) == 1:
        return n
    dev = device if (device is not None and dist.get_backend(process_group) == 'nccl') else 'cpu'
    t = torch.tensor([int(n)], dtype=torch.int64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=process_group)
    return int(t.item())


def average_scalars(tensors, process_group=None):
    """one fused all-reduce for a list of tiny tensors (alphas at compress time: collective C3)."""
    if not dist.is_initialized() or dist.get_world_size(process_group) == 1:
        return tensors
    flat = torch.cat([t.reshape(-1) for t in tensors])
    dist.all_reduce(flat, group=process_group)
    flat /= dist.get_world_size(process_group)
    out, off = [], 0
    for t in tensors:
        out.append(flat[off:off + t.numel()].view_as(t))
        off += t.numel()
    return out


class DistributedDataParallel(torch.nn.Module):
    """The `.module`-exposing wrapper the reference's drivers swap in (`search.py:617-620`, `finetune.py:421-424`:
    `model = torch.nn.parallel.DistributedDataParallel(model, device_ids=[args.gpu], find_unused_parameters=True)`,
    `model_without_ddp = model.module`, then `model.module.reset_mask_ratio(...)`, `model.module.get_flops()`, ... `:644-645,743,754`).

    Same constructor surface for the arguments those call sites pass; the exchange itself is a `GradAllReducer` (persistent flat
    buckets, asynchronous RCCL all-reduce from post-accumulate hooks) that the epoch engines pick up from `model.reducer` -
    `engine.search_one_epoch(model, ...)` / `train_one_epoch(model, ...)` then finalize the exchange after every backward and
    rebuild the buckets after `compress()` (which the reference's DDP silently does not, SURVEY D-6).  A hand-written loop calls
    `model.reducer.finalize()` after `backward()`.  `state_dict()` keys carry the `module.` prefix, as torch's wrapper's do."""

    def __init__(self, module, device_ids=None, output_device=None, dim=0, broadcast_buffers=True, process_group=None,
                 bucket_cap_mb=25, find_unused_parameters=False, check_reduction=False, gradient_as_bucket_view=False,
                 static_graph=False, first_bucket_bytes=4 * 1024 * 1024, force_collective=False):
        super().__init__()
        self.module = module
        self.device_ids, self.output_device, self.find_unused_parameters = device_ids, output_device, find_unused_parameters
        self.reducer = GradAllReducer(module.parameters(), bucket_bytes=int(bucket_cap_mb * 1024 * 1024), process_group=process_group,
                                      force_collective=force_collective, first_bucket_bytes=first_bucket_bytes)

    def forward(self, *inputs, **kwargs):
        return self.module(*inputs, **kwargs)

    def __getstate__(self):
        d = self.__dict__.copy()
        d['reducer'] = None                              # hooks and buckets are per-process state, not checkpoint content
        return d

    def no_sync(self):
        """torch DDP's context manager: backward passes inside it only accumulate locally."""
        import contextlib

        @contextlib.contextmanager
        def ctx():
            was = self.reducer.sync
            self.reducer.sync = False
            try:
                yield
            finally:
                self.reducer.sync = was
        return ctx()
