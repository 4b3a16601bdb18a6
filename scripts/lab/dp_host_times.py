"""Host-side cost of the DP path at one rank (run on the GPU box under MASTER_ADDR/RANK/WORLD_SIZE env): wall time spent inside
GradAllReducer._launch / finalize / dist.all_reduce / work.wait per step, and the step time with and without the reducer."""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.distributed as dist
import ofb_amd
from ofb_amd import engine, dp
from ofb_amd.losses import OFBSearchLOSS, DistillationLoss, LabelSmoothingCrossEntropy
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29544')
os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
if os.environ.get('PRE_SIDE') == '1':                       # lab: the side stream exists before RCCL creates its own streams
    from ofb_amd import hip as _hip
    _hip._side_streams[torch.device('cuda', 0)] = torch.cuda.Stream(device=dev)
    torch.zeros(1, device=dev)
if os.environ.get('NO_RCCL') != '1':
    dist.init_process_group('nccl', init_method='env://', device_id=dev)
torch.manual_seed(0)
model = ofb_amd.create_model('deit_small_patch16_224_mim', method='search', num_classes=1000, drop_path_rate=0.1, attn_search=True,
                             mlp_search=True, embed_search=True, patch_search=False, mae=True, mask_ratio=1.0)
model.correct_require_grad(0.5, 0.5, 0, 0.5); model.adjust_masking_ratio(0.0, 20, 100); model.to(dev).train()
opts = engine.build_optimizers(model, 2.5e-4 * 128 / 256)
crit = OFBSearchLOSS(DistillationLoss(LabelSmoothingCrossEntropy(0.1), None, 'none', 0.5, 1.0), dev, attn_w=0.5, mlp_w=0.5, patch_w=0.0,
                     embedding_w=0.5, flops_w=5.0)
imgs = torch.randn(128, 3, 224, 224, device=dev); labels = torch.randint(0, 1000, (128,), device=dev)
T = collections.defaultdict(float)
def timed(name, f):
    def g(*a, **k):
        t = time.perf_counter(); r = f(*a, **k); T[name] += time.perf_counter() - t; return r
    return g
def run(reducer, n=20):
    for _ in range(6): engine.search_step(model, crit, imgs, labels, 1.0, opts, reducer=reducer)
    torch.cuda.synchronize(); T.clear(); t = time.perf_counter()
    for _ in range(n): engine.search_step(model, crit, imgs, labels, 1.0, opts, reducer=reducer)
    host = time.perf_counter() - t
    torch.cuda.synchronize(); tot = time.perf_counter() - t
    return host / n * 1e3, tot / n * 1e3
print('plain     host-enqueue %.2f ms  step %.2f ms' % run(None))
if os.environ.get('NO_RCCL') == '1':
    raise SystemExit(0)
red = dp.GradAllReducer(list(model.parameters()), force_collective=True)
red._launch = timed('_launch', red._launch); red.finalize = timed('finalize', red.finalize)
_ar = dist.all_reduce; dist.all_reduce = timed('all_reduce', _ar); dp.dist.all_reduce = dist.all_reduce
h, s = run(red)
print('force-dp  host-enqueue %.2f ms  step %.2f ms' % (h, s), {k: round(v / 20 * 1e3, 3) for k, v in T.items()}, 'ms per step')
