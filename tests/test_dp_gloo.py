"""world_size-2 gloo test of the data-parallel gradient exchange (runs on CPU; on MI355X the same code uses RCCL)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import ofb_amd
    from ofb_amd.dp import GradAllReducer, average_scalars
    torch.manual_seed(0)
    params = [torch.nn.Parameter(torch.randn(n)) for n in (5, 300, 7, 1000, 64)]
    frozen = torch.nn.Parameter(torch.randn(3), requires_grad=False)
    unused = torch.nn.Parameter(torch.randn(11))
    red = GradAllReducer(params + [frozen, unused], bucket_bytes=1024)          # several small buckets
    assert len(red.buckets) >= 2
    for step in range(2):
        for p in params:
            p.grad = None
        loss = sum((p * (rank + 1 + i + step)).sum() for i, p in enumerate(params))
        loss.backward()
        red.finalize()
        for i, p in enumerate(params):
            exp = sum(r + 1 + i + step for r in range(world)) / world
            assert torch.allclose(p.grad, torch.full_like(p, exp)), (rank, i, step)
        assert unused.grad is None
    # gradient accumulation over 3 micro-steps (engine.search_step: prescaled SUM; engine.train_one_epoch: SUM then / world):
    # only the closing micro-step exchanges, and the result is the rank-average of the accumulated gradients
    for prescaled in (True, False):
        red.prescaled = prescaled
        for p in params:
            p.grad = None
        for micro in range(3):
            red.sync = micro == 2
            loss = sum((p * (rank + 1 + i + 10 * micro)).sum() for i, p in enumerate(params))
            (loss * red.grad_scale if prescaled else loss).backward()
            red.finalize()
        for i, p in enumerate(params):
            exp = sum(sum(r + 1 + i + 10 * micro for micro in range(3)) for r in range(world)) / world
            assert torch.allclose(p.grad, torch.full_like(p, float(exp))), (rank, i, prescaled, p.grad[0].item(), exp)
    red.sync, red.prescaled = True, False
    avg = average_scalars([torch.full((2, 3), float(rank)), torch.full((4,), 10.0 * rank)])
    assert torch.allclose(avg[0], torch.full((2, 3), 0.5)) and torch.allclose(avg[1], torch.full((4,), 5.0))
    out[rank] = 1
    dist.destroy_process_group()


def test_bucketed_allreduce_world2():
    mp.set_start_method('spawn', force=True)
    with mp.Manager() as mgr:
        out = mgr.dict()
        port = _free_port()
        ctx = mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
        assert dict(out) == {0: 1, 1: 1}
