"""bench.py — OFB search-step throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one search micro-step (engine.py:131-184 of the reference): forward of the bi-mask gated
DeiT-S + PMIM branch, OFBSearchLOSS, backward, gradient all-reduce (N > 1), 3x AdamW — on one synthetic
batch of 128 images per GPU that is resident in HBM before the timed region.  Prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BASELINE_GFLOP_PER_IMG = {'deit_small': 27.83, 'deit_tiny': 7.64, 'deit_base': 105.85}      # BASELINE.md section 2 (fwd+bwd)
# what the step really computes: the decoder GEMM (2 * 196 * 768 * D flops per image and pass, x3 for fwd + both gradients) runs on
# the masked patches only - 5 % of the rows at keep ratio 0.95 (exact: unmasked patches contribute 0 to the PMIM loss)
GFLOP_PER_IMG = {'deit_small': 27.49, 'deit_tiny': 7.48, 'deit_base': 105.19}
PEAK_F32_MFMA_TFLOPS = 157.3                                                       # MI355X_MICROARCH.md, f32-input MFMA
PEAK_F16_MFMA_TFLOPS = 2500.0                                                      # MI355X_MICROARCH.md, dense bf16 / f16 MFMA
# The GEMM computes every f32 product as three f16 MFMA terms (two-plane operand split of a power-of-two scaled copy, f32
# accumulate, csrc/hformat.h): the matrix pipe executes 3 hardware flops per algorithmic flop, so the ceiling for ALGORITHMIC f32
# flops is the f16 peak / 3.  (Rounds 1-3: six bf16 terms, ceiling 416.7.)
GEMM_MFMA_TERMS = 3
INIT_STEPS = 6
PEAK_GEMM_TFLOPS = PEAK_F16_MFMA_TFLOPS / GEMM_MFMA_TERMS
PEAK_HBM_TBS, HBM_COPY_TBS = 8.0, 6.29                                              # MI355X_MICROARCH.md: HBM3E spec / measured copy rate
# algorithmic bytes of all ofb_gemm_h calls of one step (4-B H-format operands + outputs + epilogue side inputs, each
# moved once), GB
GEMM_ALGORITHMIC_GB = {('deit_small', 128): 28.5}
ARITHMETIC = ('f32 inputs / outputs / accumulation / elementwise; every f32 PRODUCT of the GEMMs and of attention is three f16 MFMA terms of operands '
              'split into two f16 planes of a power-of-two scaled copy - ONE exponent per tensor: 23 significant bits for elements within 2^18 of '
              "the tensor's bound, an absolute 2^-39 of the bound below (measured weakest row of the step: ~17 bits; DESIGN 3)")


def csrc_hash():
    """sha256 over the kernel sources the library is built from: a profile summary under profiles/ names the build it was taken on"""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, 'once-for-both_amd', 'csrc', '*.hip')) + glob.glob(os.path.join(ROOT, 'once-for-both_amd', 'csrc', '*.h'))
                    + glob.glob(os.path.join(ROOT, 'include', '*.h'))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, 'rb').read())
    return h.hexdigest()[:16]
PROF_TAGS = ['gemm_f32', 'attention_fwd', 'layernorm_fwd', 'layernorm_bwd', 'attention_bwd', 'norm_targets', 'adamw']


def gemm_traffic_per_launch(launches_per_step):
    """HBM-side bytes per ofb_gemm_h call from the committed PMC summary (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate
    passes over this same bench command, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950; scripts/round_profiles.sh
    -> scripts/profile_summaries.py).  Hardware counters cannot be read from inside the process, so the figure is the newest
    profiled one for configs[1] - and ONLY if that summary was taken on THIS build (its CSRC_SHA line = csrc_hash()): a summary of
    other kernel sources yields traffic: null."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*pmc_traffic_summary*.txt')),
                   key=lambda f: [int(x) for x in re.findall(r'\d+', os.path.basename(f))])
    here = csrc_hash()
    for f in reversed(files):
        txt = open(f).read()
        m, h = re.search(r'GEMM_BYTES_PER_STEP (\d+)', txt), re.search(r'CSRC_SHA (\w+)', txt)
        if m and h and h.group(1) == here:
            return int(m.group(1)) / launches_per_step, os.path.relpath(f, ROOT)
    return None, None


def _cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or 'unknown'


def cpu_baseline(log):
    """The CPU oracle (a parity-pinned port of the reference's engine.py search step) timed on this host's cores on a bounded
    sample of the same workload, as SURVEY.md 8(d) prescribes: FULL steps (forward + OFBSearchLOSS + backward + the three
    AdamW instances), fp32, 3 warm-up + 5 timed, at configs[0] (DeiT-T, 2 classes, bs 8) and at a reduced-batch DeiT-S point
    (bs 32).  `value` is the DeiT-S figure (the architecture of the metric); both cases are listed."""
    from oracle import ofb_oracle as O
    # 16 threads = this job's CPU share on a one-GPU box; with all 128 hardware threads the small batches spend their time in
    # thread hand-offs (1.4-3.5 images/s, varying run to run) instead of arithmetic
    torch.set_num_threads(min(16, os.cpu_count() or 16))
    warm, timed = 3, 5

    def run_case(arch, ncls, bs):
        torch.manual_seed(0)
        cfg = O.Config(**arch, num_classes=ncls, drop_path_rate=0.1)
        p = {k: v.requires_grad_(k != 'alpha_patch') for k, v in O.formula_params(cfg, torch.float32).items()}
        opt = O.OptimState(p, frozen=('alpha_patch',))
        st = O.SearchState(w_p=0.99, keep_ratio=0.95)
        g = torch.Generator().manual_seed(1234)
        imgs = torch.randn(bs, 3, 224, 224, generator=g)
        labels = torch.randint(0, ncls, (bs,), generator=g)
        lr = 2.5e-4 * bs / 256

        def step():
            for v in p.values():
                v.grad = None
            out = O.search_step_loss(cfg, p, st, imgs, labels, torch.rand(bs, 196, generator=g), torch.rand(2 * cfg.depth, bs, generator=g))
            out['loss_total'].backward()
            with torch.no_grad():
                grads = {k: v.grad for k, v in p.items()}
                q = {k: v.detach() for k, v in p.items()}
                opt.step(q, grads, lr)                                   # the three reference AdamW instances (optim.py:56-120)
            for k in p:
                p[k] = q[k].requires_grad_(k != 'alpha_patch')

        for _ in range(warm):
            step()
        t0 = time.time()
        for _ in range(timed):
            step()
        dt = (time.time() - t0) / timed
        log(f'cpu_baseline: {arch["embed_dim"]}-wide, bs {bs}: {dt:.3f} s/step on {torch.get_num_threads()} threads')
        return dict(s_per_step=round(dt, 4), images_per_s=round(bs / dt, 3), batch=bs)

    tiny = run_case(O.DEIT_TINY, 2, 8)
    small = run_case(O.DEIT_SMALL, 1000, 32)
    # how the port compares with the reference's own Python on one CPU (build container, 8 threads, same inputs:
    # profiles/r01_reference_cpu_timing.json): the port is ~2x FASTER than the reference it restates (it fuses the gate / loss
    # micro-ops), so the reference itself would sit at about value x reference_ratio on this host
    ratio = None
    try:
        cases = json.load(open(os.path.join(ROOT, 'profiles', 'r01_reference_cpu_timing.json')))['cases']
        ds = [c for c in cases if 'DeiT-S' in c['case']][0]
        ratio = round(ds['reference_images_per_s'] / ds['oracle_images_per_s'], 3)
    except (OSError, KeyError, IndexError, ValueError):
        pass
    return dict(value=small['images_per_s'], unit='images/s', cores=torch.get_num_threads(), kind='port', cpu_model=_cpu_model(),
                reference_ratio=ratio,
                reference_ratio_note='reference engine.py search step / this port, DeiT-S bs 8, 8 threads of the build container '
                                     '(profiles/r01_reference_cpu_timing.json): the port is the FASTER of the two',
                sample=f'full OFB search steps (fwd + loss + bwd + 3x AdamW) of the oracle, fp32, {warm} warm-up + {timed} timed: '
                       f'DeiT-S bs 32 (value) and configs[0] DeiT-T 2-class bs 8',
                cases={'deit_small_bs32': small, 'configs[0]_deit_tiny_2cls_bs8': tiny})


# configs[4]: the released OFB-DeiT-C shapes are not available, so a subnet at the same budget (~1.7 GMAC, ~8 M
# parameters) is synthesised by forcing every module's alpha onto one cell and letting compress() cut the search model.
FT_EMBED = 264
FT_BLOCKS = [(4, 48, 576), (4, 40, 768), (6, 32, 768), (4, 56, 960)] * 3          # (heads, head dim, mlp hidden) per block


def build_finetune_subnet(ofb_amd, dev, ncls):
    import torch
    from ofb_amd import utils
    search = ofb_amd.create_model('deit_small_patch16_224_mim', method='search', num_classes=ncls, drop_path_rate=0.1, attn_search=True,
                                  mlp_search=True, embed_search=True, patch_search=False, mae=True, mask_ratio=1.0)
    search.correct_require_grad(0.5, 0.5, 0, 0.5)
    search.to(dev)

    def force(mod, i, j):
        a = torch.full_like(mod.alpha.data, -8.0)
        a[i, j] = 0.0
        mod.alpha.data.copy_(a)

    pe = search.patch_embed
    force(pe, 0, pe._chan_thr().index(FT_EMBED))
    for blk, (h, dh, hid) in zip(search.blocks, FT_BLOCKS):
        force(blk.attn, list(blk.attn.head_num_list).index(h), blk.attn._chan_thr().index(dh))
        force(blk.mlp, 0, blk.mlp._chan_thr().index(hid))
    finish, *_ = search.compress(1.0)
    assert finish, 'forced alphas must finish the search in one compress()'
    model = ofb_amd.create_model('deit_small_patch16_224_finetune', num_classes=ncls, drop_path_rate=0.1).to(dev)
    utils.intersect(model, search)
    n, d = 197, FT_EMBED
    macs = 196 * 768 * d + d * ncls
    for h, dh, hid in FT_BLOCKS:
        hd = h * dh
        macs += n * d * 3 * hd + 2 * n * n * hd + n * hd * d + 2 * n * d * hid
    params = sum(p.numel() for p in model.parameters())
    return model, macs, params


def build_pruned_search(ofb_amd, dev, ncls):
    """The SEARCH model as it looks for almost the whole search (reference engine.py:201-205 calls compress() three times per epoch
    from epoch 0 on): the largest options of every module are dead and cut away, two cells per module are still competing.  The
    alphas are forced (the shapes of the configs[4] subnet: embed 264, ragged heads / hidden widths), ONE compress() cuts weights,
    gates and optimizer state; the step that is timed afterwards is the full search step (bi-mask gates, PMIM branch, arch loss)."""
    import torch
    search = ofb_amd.create_model('deit_small_patch16_224_mim', method='search', num_classes=ncls, drop_path_rate=0.1, attn_search=True,
                                  mlp_search=True, embed_search=True, patch_search=False, mae=True, mask_ratio=1.0)
    search.correct_require_grad(0.5, 0.5, 0, 0.5)
    search.to(dev)

    def force(mod, cells):
        a = torch.full_like(mod.alpha.data, -12.0)
        for (i, j) in cells:
            a[i, j] = 0.0
        mod.alpha.data.copy_(a)

    pe = search.patch_embed
    j = pe._chan_thr().index(FT_EMBED)
    force(pe, [(0, j), (0, j - 1)])
    for blk, (h, dh, hid) in zip(search.blocks, FT_BLOCKS):
        i, jj = list(blk.attn.head_num_list).index(h), blk.attn._chan_thr().index(dh)
        force(blk.attn, [(i, jj), (i, jj - 1)])
        jm = blk.mlp._chan_thr().index(hid)
        force(blk.mlp, [(0, jm), (0, jm - 1)])
    finish, pruned, *_ = search.compress(0.2)
    assert pruned and not finish, 'forced alphas must cut the weights and leave the search running'
    shapes = dict(embed_dim=int(search.pos_embed.shape[-1]),
                  blocks_heads_headdim_hidden=[(int(b.attn.num_heads), int(b.attn.qkv.weight.shape[0] // 3 // b.attn.num_heads), int(b.mlp.fc1.weight.shape[0]))
                                               for b in search.blocks][:4], pattern_repeats=3)
    return search, shapes


def self_launch(n):
    """`python bench.py --gpus N` from a bare shell: start N ranks (one per GPU) under torch.distributed.run and hand their exit
    code back.  This parent never touches the GPU (no HIP call, not even torch.cuda.is_available()): a process that has
    initialised the GPU must not be replaced or forked on this pool, so the ranks are CHILD processes started first."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')        # dmabuf IPC: RCCL needs it on this driver
    env.setdefault('OMP_NUM_THREADS', '4')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print('bench.py: launching', ' '.join(cmd), file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)                     # the ranks inherit stdout: rank 0 prints the one JSON line


def report_exchange(reducer, rank, world, dev, seconds, steps):
    """every rank's view of the gradient exchange on stderr (rank order): buckets, collectives per step and the EXPOSED exchange
    time per step - the host-side wait in finalize() that backward did not hide - next to its step time: the first real multi-GPU
    run is diagnosable from the tail of its log"""
    import torch.distributed as dist
    if reducer is None or not dist.is_initialized():
        return
    n = max(reducer.finalized, 1)
    serial = reducer.measure_collectives()                   # (after the timed region: every bucket's all-reduce alone, back to back)
    mine = torch.tensor([rank, len(reducer.buckets), reducer.collectives / n, reducer.wait_seconds / n * 1e3, seconds / max(steps, 1) * 1e3,
                         serial], dtype=torch.float64, device=dev if dev is not None else 'cpu')
    rows = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(rows, mine)
    if rank == 0:
        print(f'[exchange] buckets in launch order (MB): {reducer.bucket_sizes_mb()}; CUs the GEMM plans leave to the exchange kernels: '
              f'{getattr(reducer, "_reserved", 0)} (OFB_DP_RESERVE_CUS)', file=sys.stderr, flush=True)
        for r in rows:
            r = r.tolist()
            hid = f', {100.0 * (1.0 - r[3] / r[5]):.0f} % of it hidden under backward' if r[5] > 0 else ''
            print(f'[exchange] rank {int(r[0])}: {int(r[1])} buckets, {r[2]:.1f} collectives/step, exposed wait {r[3]:.3f} ms/step of '
                  f'{r[4]:.3f} ms/step; the collectives alone, serial: {r[5]:.3f} ms/step{hid}', file=sys.stderr, flush=True)


def rehearsal(args, rank, world, real_stdout):
    """OFB_BENCH_REHEARSAL=gloo: the launcher, rendezvous, broadcast, bucketed exchange and JSON plumbing of the N-rank run on
    CPU tensors over gloo (no GPU in the build container; tests/test_dp_gloo.py drives it).  Not a benchmark."""
    import torch.distributed as dist
    dist.init_process_group('gloo', init_method='env://')
    import ofb_amd
    torch.manual_seed(100 + rank)                            # deliberately different replicas: the reducer must broadcast rank 0
    params = [torch.nn.Parameter(torch.randn(n)) for n in (4096, 33, 70000, 512)]
    red = ofb_amd.dp.GradAllReducer(params, bucket_bytes=64 * 1024)
    ref = [p.detach().clone() for p in params]
    dist.broadcast(ref[0], 0)
    same = all(bool(torch.equal(a, b)) for a, b in zip(params[:1], ref[:1]))
    t0 = time.perf_counter()
    for k in range(args.steps):
        for p in params:
            p.grad = None
        (sum((p * (rank + 1)).sum() for p in params) * red.grad_scale).backward()
        red.prescaled = True
        red.finalize()
    dt = time.perf_counter() - t0
    ok = same and all(torch.allclose(p.grad, torch.full_like(p, sum(range(1, world + 1)) / world)) for p in params)
    # collective C4 (epoch statistics): the fused sum of every rank's running sums
    tot, w = ofb_amd.dp.sum_across_ranks(torch.tensor([float(rank + 1), 1.0]))
    ok = ok and w == world and tot.tolist() == [float(sum(range(1, world + 1))), float(world)]
    report_exchange(red, rank, world, None, dt, args.steps)
    flag = torch.tensor([1.0 if ok else 0.0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0:
        res = dict(metric='REHEARSAL of the N-rank launch on CPU/gloo (not a benchmark)', value=0.0, unit='images/s', n_gpus=world,
                   steps=args.steps, warmup=args.warmup, ms_per_step=round(dt / max(args.steps, 1) * 1e3, 3), higher_is_better=True,
                   scaling='weak', vs_baseline=None, dtype='f32', data='synthetic',
                   config=dict(workload='gradient exchange of 4 CPU tensors', parallelism=f'dp{world}',
                               collective=dict(backend='gloo', ranks=dist.get_world_size()), exchange_ok=bool(flag.item())))
        os.write(real_stdout, (json.dumps(res) + '\n').encode())
    dist.destroy_process_group()
    return 0 if flag.item() else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=128, help='images per GPU (BASELINE config: 128)')
    ap.add_argument('--model', default='deit_small', choices=list(GFLOP_PER_IMG))
    ap.add_argument('--mode', default='search', choices=['search', 'finetune'],
                    help="search: the BASELINE metric (default); finetune: configs[4], a pruned OFB-DeiT-C-like subnet (not a bench line)")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-prof', action='store_true', help='skip per-kernel HIP-event timing')
    ap.add_argument('--prof-all', action='store_true', help='bracket every tagged kernel family in the sampled steps (default: the GEMM only)')
    ap.add_argument('--graph', action='store_true', help='search mode: capture the whole step (with N > 1: incl. the RCCL exchange) into a hipGraph and replay it '
                    '(engine.GraphedStep); for launch-bound sizes (small batches); the sampled profile steps stay eager')
    ap.add_argument('--force-dp', action='store_true', help='exercise the DP bucket path even with one rank (debug)')
    ap.add_argument('--pruned', action='store_true', help='search mode: time the search step of a model that compress() has already cut '
                    '(ragged shapes, what most of a real search runs on; reference engine.py:201-205) - NOT the BASELINE line')
    ap.add_argument('--search-epoch', type=float, default=0.0, help='search mode: the epoch whose warm-up state (w_p, patch keep ratio) is timed; '
                    '0 = the state the BASELINE metric is quoted on, 10 = SURVEY 8(d) second point (w_p 0.545, keep ratio 0.85)')
    args = ap.parse_args()

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:     # bare `python bench.py --gpus N`: become the launcher
        raise SystemExit(self_launch(args.gpus))
    # stdout carries exactly ONE line (the JSON): library banners written to fd 1 (RCCL prints its version block there when
    # the communicator comes up) are sent to stderr instead, and the JSON goes to the saved descriptor at the end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}')
    if os.environ.get('OFB_BENCH_REHEARSAL') == 'gloo':
        raise SystemExit(rehearsal(args, rank, world, real_stdout))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the once-for-both_amd hot path has no CPU fallback')
    # the step's host side is one Python thread; ATen's intra-op pool defaults to every core of the node (x N ranks), and a parallel
    # CPU op above its grain size wakes all of them to spin (on a CPU-quota cgroup that throttled the whole process: DESIGN section 0,
    # JPEG row).  The CPU baseline sets its own thread count.
    torch.set_num_threads(max(1, min(torch.get_num_threads(), 16)))
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    import ofb_amd
    ofb_amd.hip.ensure_side_stream(dev)                      # before RCCL creates its streams (hardware-queue mapping)
    import torch.distributed as dist
    if world > 1 or (args.force_dp and 'MASTER_ADDR' in os.environ):      # --force-dp under torchrun: a one-rank RCCL group
        dist.init_process_group('nccl', init_method='env://', device_id=dev)

    def log(*a):
        if rank == 0:
            print(*a, file=sys.stderr, flush=True)

    import ofb_amd
    from ofb_amd import engine, hip
    from ofb_amd.losses import OFBSearchLOSS, DistillationLoss, LabelSmoothingCrossEntropy

    use_graph = args.graph and args.mode == 'search'
    if use_graph:
        torch.cuda.set_stream(torch.cuda.Stream(device=dev))     # the whole job on ONE non-default stream: capturable (GraphedStep)
    elif os.environ.get('OFB_MAIN_STREAM_PRIORITY'):             # lab: the critical path on a high-priority stream, dW on the normal side stream
        torch.cuda.set_stream(torch.cuda.Stream(device=dev, priority=int(os.environ['OFB_MAIN_STREAM_PRIORITY'])))
    torch.manual_seed(0 + rank)                              # per-rank init streams (search.py:381); the reducer broadcasts rank 0's replica
    ncls = 1000
    eff_bs = args.batch * world
    gflop_img = GFLOP_PER_IMG[args.model]
    ft_info = None
    if args.mode == 'search':
        if args.pruned:
            model, ft_info = build_pruned_search(ofb_amd, dev, ncls)
        else:
            model = ofb_amd.create_model(f'{args.model}_patch16_224_mim', method='search', num_classes=ncls, drop_path_rate=0.1,
                                         attn_search=True, mlp_search=True, embed_search=True, patch_search=False, mae=True,
                                         mask_ratio=1.0)
            model.correct_require_grad(0.5, 0.5, 0, 0.5)
        # epoch-0 state by default: keep ratio 0.95, w_p 0.99; --search-epoch 10 = SURVEY 8(d)'s second point (0.85 / 0.545): more
        # masked patches for the decoder / PMIM branch, gates further from the sigmoid scores
        model.adjust_masking_ratio(args.search_epoch, 20, 100)
        for m in model.searchable_modules:
            m.update_w(args.search_epoch, 20)
        model.to(dev).train()
        lr = 2.5e-4 * eff_bs / 256
        opt_p, opt_a, opt_d = engine.build_optimizers(model, lr)
        crit = OFBSearchLOSS(DistillationLoss(LabelSmoothingCrossEntropy(0.1), None, 'none', 0.5, 1.0), dev, attn_w=0.5, mlp_w=0.5,
                             patch_w=0.0, embedding_w=0.5, flops_w=5.0)
    else:
        # finetune.py's recipe on the pruned subnet: eval-mode semantics (finetune.py:445), Mixup(0.8, 1.0) + SoftTargetCrossEntropy
        # (:310, :390-393), AdamW (lr 2.5e-4 * eff_bs / 512, wd 0.05), ModelEma every micro-step (:52, engine.py:62-63)
        import numpy as np
        from ofb_amd.optim import AdamW
        from ofb_amd.utils import ModelEma
        model, macs, nparams = build_finetune_subnet(ofb_amd, dev, ncls)
        model.train(False)
        gflop_img = 3 * 2 * macs / 1e9                       # forward + 2x backward, 2 flops per MAC
        ft_info = dict(gmacs_per_img=round(macs / 1e9, 3), params_m=round(nparams / 1e6, 2), embed_dim=FT_EMBED,
                       blocks_heads_headdim_hidden=FT_BLOCKS[:4], pattern_repeats=3)
        opt_ft = AdamW(model.parameters(), None, lr=2.5e-4 * eff_bs / 512, weight_decay=0.05)
        crit_ft = DistillationLoss(ofb_amd.SoftTargetCrossEntropy(), None, 'none', 0.5, 1.0)
        mixup_fn = ofb_amd.Mixup(mixup_alpha=0.8, cutmix_alpha=1.0, label_smoothing=0.1, num_classes=ncls)
        np.random.seed(1234 + rank)
    # replaces DDP's wrap (search.py:617-620): broadcasts rank 0's replica, then exchanges gradients in persistent flat buckets
    reducer = ofb_amd.dp.GradAllReducer(list(model.parameters()), force_collective=args.force_dp) if (world > 1 or args.force_dp) else None
    if args.mode == 'finetune':
        ema = ModelEma(model, decay=0.99996)                 # after the broadcast: the EMA copies rank 0's weights on every rank

    torch.manual_seed(1234 + rank)                           # per-rank data / mask / DropPath streams (search.py:381)
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    imgs = torch.randn(args.batch, 3, 224, 224, device=dev, generator=gen)
    labels = torch.randint(0, ncls, (args.batch,), device=dev, generator=gen)

    if args.mode == 'search':
        def step():
            return engine.search_step(model, crit, imgs, labels, 1.0, (opt_p, opt_a, opt_d), reducer=reducer)
    else:
        def step():                                          # one micro-step of engine.train_one_epoch (engine.py:30-63)
            x, soft = mixup_fn(imgs, labels)                 # in place on the resident batch (it stays a valid mixed batch)
            loss = crit_ft(x, model(x), soft)
            if reducer is not None:
                reducer.prescaled = True
                loss = loss * reducer.grad_scale
            engine.run_backward(loss)
            if reducer is not None:
                reducer.finalize()
            opt_ft.step()
            opt_ft.zero_grad(set_to_none=True)
            ema.update(model)
            return None, None, None, loss

    # one-time initialisation, not steady state: the first step loads the code objects and grows the caching allocator to its
    # final 8 GiB, and the 5th step of a fresh process pays a single ~100 ms host-side stall (scripts/step_times.py shows it
    # once in 60 steps).  These INIT_STEPS run before the W warm-up steps so that the timed region starts from a settled process.
    for _ in range(INIT_STEPS):
        step()
    eager_step = step
    if use_graph:
        gstep = engine.GraphedStep(eager_step, (opt_p, opt_a, opt_d), reducer=reducer)   # with a reducer the RCCL exchange is captured too
        gstep.capture()                                      # two more eager steps on the capture stream, then the capture
        step = gstep
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    # per-launch HIP events (on the launch stream) bracket the kernels of the LAST `prof_steps` timed steps only: each
    # bracket costs a few microseconds of inter-kernel bubble, so sampling keeps `value` honest while still measuring
    # inside the timed region
    prof_steps = 0 if args.no_prof else min(2, args.steps)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        if prof_steps and i == args.steps - prof_steps:
            # the sampled steps run with the side stream off: per-launch durations of kernels that overlap on two streams are not
            # attributable to one kernel (each stretches while the other shares the chip); `value` includes these slower steps
            hip.prof_enable(True if args.prof_all else 1)     # bit 0: the GEMM (the roofline kernel); --prof-all: every tagged kernel
            side_was, hip.SIDE_STREAM = hip.SIDE_STREAM, False
            step = eager_step                                 # events cannot be recorded inside a replayed graph
        out = step()
    if prof_steps:
        hip.SIDE_STREAM = side_was
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof = None
    if not args.no_prof:
        prof = hip.prof_collect(len(PROF_TAGS))
        hip.prof_enable(False)
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax)
    loss_val = float(out[3].detach())
    report_exchange(reducer, rank, world, dev, dt, args.steps)
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    ms_step = dt / args.steps * 1e3
    value = eff_bs * args.steps / dt
    roof = None
    if prof is not None:
        for name, (n, ms, work) in zip(PROF_TAGS, prof):
            if n:
                log(f'  {name:14s} launches/step {n / prof_steps:7.1f}  ms/step {ms / prof_steps:8.3f}  '
                    f'{(work / (ms * 1e-3) / 1e12) if ms and name != "adamw" else 0:7.2f} T(FLOP|B)/s')
        n, ms, work = prof[0]
        if n:
            ach = work / (ms * 1e-3) / 1e12
            roof = dict(bound='mfma', kernel='gemm_h_kernel (f32 products as three v_mfma_f32_16x16x32_f16 terms of operands pre-split into two f16 planes '
                                             'of a power-of-two scaled copy, LDS-DMA staged; bound pre-kernel + main + stream-K tail + fix-up launches of all '
                                             'ofb_gemm_h calls)',
                        achieved=round(ach, 2), peak=round(PEAK_GEMM_TFLOPS, 1), unit='TFLOP/s', frac=round(ach / PEAK_GEMM_TFLOPS, 4),
                        traffic=None, peak_basis='2500 TFLOP/s dense f16 MFMA / 3 MFMA terms per f32 product (achieved = algorithmic f32 flops; '
                                                 'rounds 1-3 issued six bf16 terms: ceiling 416.7, against which this kernel reads '
                                                 f'{ach / 416.7:.3f}; the f32-input MFMA peak would be 157.3)',
                        mfma_issued_tflops=round(ach * GEMM_MFMA_TERMS, 1),
                        launches_per_step=round(n / prof_steps, 1), avg_launch_us=round(ms / n * 1e3, 2),
                        share_of_step=round(ms / prof_steps / ms_step, 3), sampled_steps=prof_steps)
            if args.mode == 'search' and not args.pruned and (args.model, args.batch) in GEMM_ALGORITHMIC_GB:
                # the step's GEMMs sit ON the ridge with 4-byte operands (arithmetic intensity 95-151 flop/B against 104 = 833 TFLOP/s /
                # 8 TB/s): their HBM floor is not below their matrix-pipe floor, so both fractions are quoted
                gb = GEMM_ALGORITHMIC_GB[(args.model, args.batch)]
                gemm_s = ms / prof_steps * 1e-3
                roof['hbm_frac'] = round(gb * 1e9 / gemm_s / (PEAK_HBM_TBS * 1e12), 4)
                roof['hbm_achieved_GBps'] = round(gb / gemm_s, 1)
                roof['bound_detail'] = (f'ridge: matrix-pipe floor {work / prof_steps / (PEAK_GEMM_TFLOPS * 1e12) * 1e3:.2f} ms, HBM floor '
                                        f'{gb / (PEAK_HBM_TBS * 1e3) * 1e3:.2f} ms at {PEAK_HBM_TBS:.0f} TB/s ({gb / HBM_COPY_TBS:.2f} ms at the '
                                        f'{HBM_COPY_TBS} TB/s a copy kernel reaches), measured {gemm_s * 1e3:.2f} ms per step')
            if args.mode == 'search' and not args.pruned and (args.model, args.batch) == ('deit_small', 128):
                tr, src = gemm_traffic_per_launch(n / prof_steps)
                if tr:
                    roof['traffic'] = round(tr)
                    roof['traffic_unit'] = 'bytes per GEMM call (FETCH_SIZE x2 + WRITE_SIZE of its kernels, PMC)'
                    roof['traffic_source'] = src
                    roof['algorithmic_bytes_note'] = ('H-format operands (4 B / element) + outputs + epilogue side inputs of the 152 calls, each moved '
                                                      'once: 28.5 GB per step = 187 MB per call')
    step_tflops = value * gflop_img / 1e3 / world
    log(f'loss_total {loss_val:.4f}; step {ms_step:.2f} ms; whole-step {step_tflops:.1f} TFLOP/s/GPU '
        f'({step_tflops / PEAK_GEMM_TFLOPS:.1%} of the three-term f16 ceiling {PEAK_GEMM_TFLOPS:.1f})')
    cfg_tag = 'configs[1]' if (args.model, args.batch) == ('deit_small', 128) else ('configs[3]' if args.model == 'deit_base' else 'off-config size')
    if args.mode == 'search' and args.pruned:
        metric = 'images/sec OFB-search step of a compress()-ed model (ragged shapes; NOT the BASELINE metric)'
        workload = (f'deit_small OFB search step + PMIM branch AFTER one compress() (embed 264, ragged heads / hidden widths, two live cells per '
                    f'module): bs {args.batch}/GPU, fwd + OFBSearchLOSS + bwd + 3x AdamW')
        step_tflops = 0.0                                        # no FLOP model for the cut shapes: only the GEMM's own work counter is quoted
    elif args.mode == 'search':
        metric = ('images/sec OFB-search step, DeiT-S bs=128/GPU @1/2/4/8 MI355X' if (args.model, args.batch, args.search_epoch) == ('deit_small', 128, 0.0) else
                  f'images/sec OFB-search step, {args.model} bs={args.batch}/GPU ({cfg_tag}; NOT the BASELINE metric)')
        workload = (f'{args.model} OFB search step + PMIM branch ({cfg_tag}): bs {args.batch}/GPU, 224x224 synthetic images, fwd + '
                    f'OFBSearchLOSS + bwd + 3x AdamW, drop_path 0.1, w_p {0.99 - 0.89 * min(args.search_epoch, 20) / 20:.3f}, '
                    f'keep ratio {0.95 - 0.2 * min(args.search_epoch, 20) / 20:.3f}')
    else:
        metric = 'images/sec finetune step, pruned OFB-DeiT-C-like subnet (configs[4]; NOT the BASELINE metric)'
        workload = (f'configs[4]: finetune micro-step of a synthesised ~1.7 GMAC subnet (search model cut by compress()), bs {args.batch}/GPU, '
                    'Mixup/CutMix in place + SoftTargetCrossEntropy + bwd + AdamW + ModelEma, eval-mode semantics')
    res = dict(metric=metric, value=round(value, 2), unit='images/s', n_gpus=world,
               steps=args.steps, warmup=args.warmup, ms_per_step=round(ms_step, 3), higher_is_better=True, scaling='weak',
               vs_baseline=None, dtype='f32', data='synthetic',
               config=dict(workload=workload, global_batch=eff_bs, parallelism=f'dp{world}',
                           collective=dict(backend='nccl (RCCL over xGMI)' if dist.is_initialized() else 'none',
                                           ranks=dist.get_world_size() if dist.is_initialized() else 1,
                                           buckets=len(reducer.buckets) if reducer is not None else 0), init_steps=INIT_STEPS,
                           hip_graph=bool(use_graph),
                           step_tflops_per_gpu=round(step_tflops, 2), step_frac_of_3term_f16_ceiling=round(step_tflops / PEAK_GEMM_TFLOPS, 4),
                           arithmetic=ARITHMETIC,
                           gflop_per_image=gflop_img, baseline_gflop_per_image=BASELINE_GFLOP_PER_IMG[args.model]),
               roofline=roof)
    if ft_info:
        res['config']['subnet'] = ft_info
    if world == 1 and not args.no_cpu_baseline and args.mode == 'search' and not args.pruned:
        res['cpu_baseline'] = cpu_baseline(log)
    sys.stdout.flush()
    os.write(real_stdout, (json.dumps(res) + '\n').encode())
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
