"""world_size-2 gloo test of the data-parallel gradient exchange (runs on CPU; on MI355X the same code uses RCCL)."""
import os

import pytest
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
    # a bucket whose members received gradients only in EARLIER micro-steps of the window must still be exchanged
    red.prescaled = False
    for p in params:
        p.grad = None
    red.sync = False
    sum((p * (rank + 1)).sum() for p in params).backward()
    red.finalize()
    red.sync = True
    (params[0] * (rank + 1)).sum().backward()                  # closing micro-step touches one parameter only
    red.finalize()
    for i, p in enumerate(params):
        exp = sum((r + 1) * (2 if i == 0 else 1) for r in range(world)) / world
        assert torch.allclose(p.grad, torch.full_like(p, float(exp))), (rank, i, p.grad[0].item(), exp)
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


def _worker_broadcast_rebuild(rank, world, port, out):
    """divergent per-rank initialisation (search.py:381 seeds with seed + rank) -> construction broadcasts rank 0's replica,
    frozen tensors and extra plain tensors included; rebuild() after Parameters were REPLACED (compress()) re-buckets the new
    tensors, keeps the loop's flags and broadcasts again; replicas that disagree on shapes raise instead of hanging."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import ofb_amd
    from ofb_amd import dp
    from ofb_amd.dp import GradAllReducer
    torch.manual_seed(1000 + rank)
    params = [torch.nn.Parameter(torch.randn(n)) for n in (40, 3000, 17)]
    frozen = torch.nn.Parameter(torch.randn(9), requires_grad=False)
    extra = torch.randn(5)
    red = GradAllReducer(params + [frozen], bucket_bytes=2048, extra_tensors=[extra])
    everything = torch.cat([p.detach().reshape(-1) for p in params + [frozen]] + [extra])
    ref = everything.clone()
    dist.broadcast(ref, 0)
    assert torch.equal(everything, ref), 'construction must leave every rank with rank 0 values'
    # gradients land in the persistent flat buckets: after finalize() every .grad is a view of its bucket (no torch.cat copy)
    sum((p * (rank + 1)).sum() for p in params).backward()
    red.finalize()
    for p in params:
        bi = red._where[p]
        lo, hi = red._flat[bi].data_ptr(), red._flat[bi].data_ptr() + 4 * red._flat[bi].numel()
        assert lo <= p.grad.data_ptr() < hi
        assert torch.allclose(p.grad, torch.full_like(p, 1.5))
    # direct-write slots (ops.py writes weight gradients straight into the bucket): only while no gradient is pending
    assert dp.grad_slot(params[1]) is None
    for p in params:
        p.grad = None
    slot = dp.grad_slot(params[1])
    assert slot is not None and slot.shape == params[1].shape and slot.data_ptr() == red._slot_of[params[1].data_ptr()][1].data_ptr()
    # ---- compress(): Parameters are replaced by differently-shaped ones, rank-locally perturbed
    torch.manual_seed(2000 + rank)
    new = [torch.nn.Parameter(torch.randn(n)) for n in (24, 1000, 17, 8)]
    red.sync, red.prescaled = True, True
    red.rebuild(new + [frozen])
    assert red.sync is True and red.prescaled is True, 'rebuild must keep the loop flags'
    cat = torch.cat([p.detach() for p in new])
    ref = cat.clone()
    dist.broadcast(ref, 0)
    assert torch.equal(cat, ref)
    for p in params:                                            # hooks of the old tensors are gone
        p.grad = None
    (sum((p * (rank + 1)).sum() for p in new) * red.grad_scale).backward()
    red.finalize()
    for p in new:
        assert torch.allclose(p.grad, torch.full_like(p, 1.5)), (rank, p.grad[:3])
    assert dp.grad_slot(params[1]) is None                      # stale tensors are no longer known
    # ---- shape disagreement is an error, not a hang
    bad = [torch.nn.Parameter(torch.randn(10 + rank))]
    try:
        GradAllReducer(bad)
        raised = False
    except RuntimeError as e:
        raised = 'disagree' in str(e)
    assert raised
    out[rank] = 1
    dist.destroy_process_group()


def test_broadcast_and_rebuild_world2():
    mp.set_start_method('spawn', force=True)
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker_broadcast_rebuild, args=(2, _free_port(), out), nprocs=2, join=True)
        assert dict(out) == {0: 1, 1: 1}


def _worker_epoch_stats(rank, world, port, out):
    """collective C4 (reference engine.py:216 / :70, utils.py:41-52): the epoch's loss averages are over ALL ranks.  Drives
    engine.train_one_epoch on a CPU stand-in model over gloo: every rank sees different losses, all ranks must report the same
    global average at the end of the epoch."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import types
    import ofb_amd
    from ofb_amd import engine
    from ofb_amd.dp import GradAllReducer, sum_across_ranks
    torch.manual_seed(0)
    lin = torch.nn.Linear(4, 1)
    red = GradAllReducer(list(lin.parameters()), bucket_bytes=1024)

    class Opt:
        param_groups = [{'lr': 0.1}]
        def step(self): pass
        def zero_grad(self, set_to_none=True):
            for p in lin.parameters():
                p.grad = None

    class Sched:
        def step_update(self, k): pass

    data = [(torch.full((2, 4), float(rank + 1 + i)), torch.zeros(2)) for i in range(3)]
    crit = lambda x, y, t: y.mean() * 0 + x.mean()             # the loss IS the batch mean: rank r, iteration i -> r + 1 + i
    stats = engine.train_one_epoch(lin, crit, data, Opt(), Sched(), torch.device('cpu'), 0, args=types.SimpleNamespace(accum_iter=1), reducer=red)
    exp = sum(r + 1 + i for r in range(world) for i in range(3)) / (3 * world)
    assert abs(stats['loss'] - exp) < 1e-6, (rank, stats, exp)
    tot, w = sum_across_ranks(torch.tensor([1.0, float(rank)]))
    assert w == world and tot.tolist() == [float(world), float(sum(range(world)))]
    out[rank] = 1
    dist.destroy_process_group()


def test_epoch_statistics_are_averaged_over_ranks_world2():
    mp.set_start_method('spawn', force=True)
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker_epoch_stats, args=(2, _free_port(), out), nprocs=2, join=True)
        assert dict(out) == {0: 1, 1: 1}


def _worker_uneven_loaders(rank, world, port, out):
    """one rank's loader is ONE iteration short (3 vs 4 batches): without an agreement on the epoch's length rank 1 would wait forever in
    the gradient exchange of step 3 (C1) and rank 0 in the statistics reduction at its epoch end (C4).  Both engines: every rank runs
    the 3 common iterations, the statistics are those of 2 x 3 steps, nobody hangs (the spawn below has a timeout)."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import types
    import ofb_amd
    from ofb_amd import engine
    from ofb_amd.dp import GradAllReducer
    torch.manual_seed(0)
    lin = torch.nn.Linear(4, 1)
    red = GradAllReducer(list(lin.parameters()), bucket_bytes=1024)
    steps = []

    class Opt:
        param_groups = [{'lr': 0.1}]
        def step(self): steps.append(1)
        def zero_grad(self, set_to_none=True):
            for p in lin.parameters():
                p.grad = None

    class Sched:
        def step_update(self, k): pass

    n_mine = 4 - rank                                           # rank 0: 4 batches, rank 1: 3
    data = [(torch.full((2, 4), float(rank + 1 + i)), torch.zeros(2)) for i in range(n_mine)]
    crit = lambda x, y, t: y.mean() * 0 + x.mean()
    stats = engine.train_one_epoch(lin, crit, data, Opt(), Sched(), torch.device('cpu'), 0, args=types.SimpleNamespace(accum_iter=1), reducer=red)
    assert len(steps) == 3, (rank, len(steps))
    exp = sum(r + 1 + i for r in range(world) for i in range(3)) / (3 * world)
    assert abs(stats['loss'] - exp) < 1e-6, (rank, stats, exp)
    # the search engine takes the same decision (its loop is driven without a model here: the agreed length is what matters)
    assert engine._agreed_length(data, torch.device('cpu'), red) == 3
    out[rank] = 1
    dist.destroy_process_group()


def test_uneven_loaders_do_not_hang_world2():
    mp.set_start_method('spawn', force=True)
    with mp.Manager() as mgr:
        out = mgr.dict()
        ctx = mp.spawn(_worker_uneven_loaders, args=(2, _free_port(), out), nprocs=2, join=False)
        import time
        t0 = time.time()
        while not ctx.join(timeout=5):
            assert time.time() - t0 < 120, 'a rank is waiting in a collective'
        assert dict(out) == {0: 1, 1: 1}


@pytest.mark.parametrize('n', [4, 8])
def test_bench_self_launch_four_and_eight_ranks(n):
    """the self-launch path with FOUR and with EIGHT ranks (gloo rehearsal of the driver's N = 4 / 8 launches): rendezvous, broadcast,
    bucketed exchange, the fused statistics collective and the per-rank exchange report (bucket sizes, exposed wait, serial cost)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env['OFB_BENCH_REHEARSAL'] = 'gloo'
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', str(n), '--steps', '2', '--warmup', '1'], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    res = json.loads(lines[0])
    assert res['n_gpus'] == n and res['config']['collective']['ranks'] == n and res['config']['exchange_ok'] is True
    rep = [l for l in r.stderr.splitlines() if l.startswith('[exchange] rank')]
    assert len(rep) == n and all('exposed wait' in l and 'serial' in l for l in rep), r.stderr[-1500:]
    assert any(l.startswith('[exchange] buckets in launch order') for l in r.stderr.splitlines())


def test_bench_self_launch_two_ranks():
    """`python bench.py --gpus 2` from a bare shell (no torchrun, WORLD_SIZE unset) must start two ranks that rendezvous, run the
    reducer and print ONE JSON line with n_gpus = 2; rehearsed on CPU / gloo (OFB_BENCH_REHEARSAL), on MI355X the same launcher
    starts the RCCL ranks."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env['OFB_BENCH_REHEARSAL'] = 'gloo'
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1'], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    res = json.loads(lines[0])
    assert res['n_gpus'] == 2 and res['config']['parallelism'] == 'dp2' and res['config']['collective']['ranks'] == 2
    assert res['config']['exchange_ok'] is True
    # without the rehearsal switch the ranks must fail loudly on a GPU-less host (no CPU fallback), and the launcher hands the
    # failure back as a non-zero exit code
    if torch.cuda.device_count() == 0:
        env.pop('OFB_BENCH_REHEARSAL')
        r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'], env=env,
                           capture_output=True, text=True, timeout=300)
        assert r.returncode != 0 and 'MI355X' in r.stderr
