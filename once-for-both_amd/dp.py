"""Data-parallel gradient exchange for the search step: bucketed all-reduce (RCCL over xGMI on MI355X;
backend 'nccl' IS RCCL on ROCm) launched from post-accumulate-grad hooks so it overlaps with the rest of
backward.  Replaces DistributedDataParallel(find_unused_parameters=True) of reference search.py:617-620
(collective C1 of SURVEY.md 2.2).  One process per GPU; no model or sequence sharding exists on this path.

Buckets are filled in reverse parameter order (gradients become ready last-block-first).  After the
reduce, each parameter's .grad is re-pointed at its slice of the reduced bucket (no copy back).
Parameters whose gradient never arrives in a step (frozen alphas, a finished decoder) are skipped.
"""
import torch
import torch.distributed as dist


class GradAllReducer:
    def __init__(self, params, bucket_bytes=25 * 1024 * 1024, process_group=None, force_collective=False):
        self.group = process_group
        # force_collective: issue the all-reduce even in a one-rank group (rehearses the RCCL path on a single GPU)
        self.force_collective = bool(force_collective) and dist.is_initialized()
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.params = [p for p in params if p.requires_grad]
        self.buckets, cur, size = [], [], 0
        for p in reversed(self.params):
            cur.append(p)
            size += p.numel() * 4
            if size >= bucket_bytes:
                self.buckets.append(cur)
                cur, size = [], 0
        if cur:
            self.buckets.append(cur)
        self._where = {}
        for bi, b in enumerate(self.buckets):
            for p in b:
                self._where[p] = bi
        self._ready = [0] * len(self.buckets)
        self._works = []
        self._flat = [None] * len(self.buckets)
        self._handles = [p.register_post_accumulate_grad_hook(self._hook) for p in self.params]
        # averaging: callers scale the loss by `grad_scale` (= 1/world, exact for power-of-two worlds) before backward and
        # the exchange is a plain SUM, so no extra pass over the gradients is needed; prescaled=False divides afterwards.
        self.grad_scale = 1.0 / self.world
        self.prescaled = False
        # gradient accumulation: with sync = False a backward only accumulates into the local .grad (DDP's no_sync()); the
        # micro-step that closes the accumulation window exchanges the accumulated gradients once.  Exchanging every
        # micro-step (as the reference does, SURVEY D-7) would SUM the already-reduced part again, because .grad is re-pointed
        # at the reduced bucket and autograd keeps accumulating into it.
        self.sync = True

    def rebuild(self, params):
        """call after compress() replaced Parameters (fixes the reference's silent de-sync, SURVEY D-6)."""
        for h in self._handles:
            h.remove()
        self.__init__(params, process_group=self.group, force_collective=self.force_collective)

    def _launch(self, bi):
        ps = [p for p in self.buckets[bi] if p.grad is not None]
        if not ps:
            return
        flat = torch.cat([p.grad.reshape(-1) for p in ps])
        if self.world > 1 or self.force_collective:
            work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            work = None
        self._works.append((bi, ps, flat, work))

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
        for bi in range(len(self.buckets)):            # buckets with frozen / unused members never filled up
            if 0 < self._ready[bi] < len(self.buckets[bi]):
                self._launch(bi)
        for bi, ps, flat, work in self._works:
            if work is not None:
                work.wait()
                if not self.prescaled:
                    flat.div_(self.world)
            off = 0
            for p in ps:
                n = p.numel()
                p.grad = flat[off:off + n].view_as(p)
                off += n
        self._works = []
        self._ready = [0] * len(self.buckets)


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
