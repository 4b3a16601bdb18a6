"""The data-parallel exchange on the device (configs[2]; reference search.py:617-620 wraps the model in DistributedDataParallel):
ONE process, one MI355X, a one-rank RCCL group.  `GradAllReducer(force_collective=True)` then really all-reduces its flat buckets
through RCCL, the weight-gradient GEMMs write straight into their bucket slices from the side stream, small gradients are gathered
by one multi-tensor launch, `.grad` is re-pointed at the bucket - and the parameters must come out BIT-IDENTICAL to a run without
any reducer (world size 1: the average is the gradient itself)."""
import os
import socket

import pytest
import torch

from oracle import ofb_oracle as O
from tests.golden_util import load_case
from tests.test_gpu_model import build_product

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def rccl_world_of_one():
    import torch.distributed as dist
    if dist.is_initialized():
        yield
        return
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    from ofb_amd import hip
    hip.ensure_side_stream(torch.device('cuda', 0))          # before RCCL's streams exist (hardware-queue mapping)
    dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{port}', world_size=1, rank=0, device_id=torch.device('cuda', 0))
    yield
    torch.cuda.synchronize()
    dist.destroy_process_group()


def _crit():
    from ofb_amd.losses import OFBSearchLOSS, DistillationLoss, LabelSmoothingCrossEntropy
    return OFBSearchLOSS(DistillationLoss(LabelSmoothingCrossEntropy(0.1), None, 'none', 0.5, 1.0), torch.device('cuda'),
                         attn_w=0.5, mlp_w=0.5, patch_w=0.0, embedding_w=0.5, flops_w=5.0)


def _run(tag, use_reducer, accum_iter, side_stream, compress_after=None, steps=3):
    """`steps` optimizer steps (each `accum_iter` micro-steps) of engine.search_step on the golden case's model; optionally a
    forced compress() + reducer.rebuild() after optimizer step `compress_after`.  Returns the final state_dict."""
    import ofb_amd
    from ofb_amd import engine, hip, ops
    z, cfg, st, inputs, lr = load_case(tag)
    m = build_product(cfg, st, inputs)
    opts = engine.build_optimizers(m, 1e-3)
    red = ofb_amd.dp.GradAllReducer(list(m.parameters()), bucket_bytes=256 * 1024, force_collective=True) if use_reducer else None
    crit = _crit()
    imgs, labels = inputs['imgs'].cuda(), inputs['labels'].cuda()
    old_min, old_side = ops._SIDE_MIN_TOKENS, hip.SIDE_STREAM
    ops._SIDE_MIN_TOKENS, hip.SIDE_STREAM = 0, side_stream          # micro models are far below the 12k-token switch-on point
    try:
        if red is not None:
            assert red.force_collective and len(red.buckets) >= 2
        opts = list(opts)
        for k in range(steps):
            for a in range(accum_iter):
                engine.search_step(m, crit, imgs, labels, 1.0, opts, accum_iter=accum_iter, do_step=(a == accum_iter - 1), reducer=red)
            if compress_after is not None and k == compress_after:
                mods = dict(zip(O.module_names(cfg), m.searchable_modules))
                a7 = torch.full((1, 7), -6.0)
                a7[0, 2] = 0.0
                mods['blocks.0.mlp'].alpha.data.copy_(a7)          # one cell survives: fc1 / fc2 are cut, alpha leaves the optimizer
                fin, ex, opts[0], opts[2], opts[1] = m.compress(0.2, opts[0], opts[2], opts[1])
                assert ex
                if red is not None:
                    red.rebuild(list(m.parameters()))
        torch.cuda.synchronize()
        return {k: v.detach().clone() for k, v in m.state_dict().items()}
    finally:
        ops._SIDE_MIN_TOKENS, hip.SIDE_STREAM = old_min, old_side
        if red is not None:
            red.close()


@pytest.mark.parametrize('accum_iter,side', [(1, True), (2, True), (1, False)])
def test_rccl_bucket_exchange_is_bit_identical_to_no_reducer(rccl_world_of_one, accum_iter, side):
    ref = _run('micro_b', False, accum_iter, side)
    got = _run('micro_b', True, accum_iter, side)
    bad = [k for k in ref if not torch.equal(ref[k], got[k])]
    assert not bad, bad


def test_rccl_exchange_survives_compress_and_rebuild(rccl_world_of_one):
    ref = _run('micro_a', False, 1, True, compress_after=0)
    got = _run('micro_a', True, 1, True, compress_after=0)
    assert tuple(got['blocks.0.mlp.fc1.weight'].shape) == (128, 64)
    bad = [k for k in ref if not torch.equal(ref[k], got[k])]
    assert not bad, bad


def test_rccl_all_reduce_moves_bytes(rccl_world_of_one):
    """the forced collective really runs on RCCL: the returned work handles are real and a bucket all-reduced in a group of one
    comes back unchanged"""
    import torch.distributed as dist
    import ofb_amd
    ps = [torch.nn.Parameter(torch.randn(n, device='cuda')) for n in (70000, 33, 4096)]
    red = ofb_amd.dp.GradAllReducer(ps, bucket_bytes=64 * 1024, force_collective=True)
    try:
        (sum((p * (i + 1)).sum() for i, p in enumerate(ps))).backward()
        assert red._works and all(w is not None for _, w in red._works)
        red.prescaled = True
        red.finalize()
        for i, p in enumerate(ps):
            assert torch.equal(p.grad, torch.full_like(p, float(i + 1)))
        assert dist.get_backend() == 'nccl' and dist.get_world_size() == 1
    finally:
        red.close()


def test_graphed_step_with_rccl_exchange():
    """hipGraph capture of the whole step INCLUDING the bucketed RCCL exchange (engine.GraphedStep with a reducer): the replays must
    walk the parameters like eager steps with the same reducer.  One rank: the collective is issued for real (force_collective).

    Runs in a process of its own (round 6): late in a LONG-lived test process - a process group whose watchdog thread has been polling
    the events of hundreds of earlier collectives, dozens of streams and graph pools behind it - the first replay of a graph that
    contains RCCL work aborted the process without a message in 2 of 4 whole-suite runs on fresh boxes (never under a debugger, never
    with the files up to this one alone: profiles/r06_graph_rccl_abort_notes.txt); a training job captures its step in a fresh process
    right after start-up, which is what the child process reproduces."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    out = subprocess.run([sys.executable, '-c', 'import tests.test_gpu_dp as t; t.graphed_step_with_rccl_exchange_in_this_process()'],
                         cwd=root, env=env, capture_output=True, text=True, timeout=600)
    print(out.stdout[-2000:])
    assert out.returncode == 0, out.stderr[-4000:]
    assert 'graphed + RCCL vs eager + RCCL' in out.stdout


def graphed_step_with_rccl_exchange_in_this_process():
    import torch.distributed as dist
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    from ofb_amd import hip
    hip.ensure_side_stream(torch.device('cuda', 0))          # before RCCL's streams exist (hardware-queue mapping)
    dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{port}', world_size=1, rank=0, device_id=torch.device('cuda', 0))
    try:
        _graphed_step_with_rccl_exchange_body()
    finally:
        torch.cuda.synchronize()
        dist.destroy_process_group()


def _graphed_step_with_rccl_exchange_body():
    import ofb_amd
    from ofb_amd import engine
    z, cfg, st, inputs, lr = load_case('micro_a')
    prev = torch.cuda.current_stream()
    torch.cuda.set_stream(torch.cuda.Stream())
    try:
        res = []
        for graphed in (False, True):
            m = build_product(cfg, st, inputs)
            opts = engine.build_optimizers(m, 1e-3)
            red = ofb_amd.dp.GradAllReducer(list(m.parameters()), bucket_bytes=256 * 1024, force_collective=True)
            crit = _crit()
            imgs, labels = inputs['imgs'].cuda(), inputs['labels'].cuda()
            step = lambda: engine.search_step(m, crit, imgs, labels, 1.0, opts, reducer=red)
            try:
                if graphed:
                    gs = engine.GraphedStep(step, opts, reducer=red)
                    gs.capture(warm_steps=2)
                    for _ in range(3):
                        gs()
                else:
                    for _ in range(5):
                        step()
                torch.cuda.synchronize()
                res.append({k: v.detach().clone() for k, v in m.state_dict().items()})
            finally:
                red.close()
        worst = max(float((res[0][k].double() - res[1][k].double()).abs().max() / (res[0][k].double().abs().max() + 1e-12)) for k in res[0])
        print('graphed + RCCL vs eager + RCCL: worst relative parameter difference', worst)
        assert worst <= 1e-5
    finally:
        torch.cuda.set_stream(prev)
