"""lab: where does ofb_amd.engine.search_one_epoch leave the oracle's epoch (tests/epoch_util inputs)?  Snapshots of every parameter
when the loader hands out batch i, on both sides."""
import sys, os, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import ofb_oracle as O
from tests import epoch_util as E
from tests.test_gpu_model import build_product
from ofb_amd import engine
from ofb_amd.losses import OFBSearchLOSS, DistillationLoss, LabelSmoothingCrossEntropy
z = E.load()
cfg = O.Config(**E.MINI, drop_path_rate=0.0)
dev = torch.device('cuda')
# ---- oracle
st = O.SearchState(); st.frozen.add('alpha_patch')
p = O.formula_params(cfg, torch.float64)
opt = O.OptimState(p, frozen=st.frozen)
snap_o = {}
def hook_o(i):
    snap_o[i] = {k: v.detach().clone() for k, v in p.items()}
    stage = E.CRAFT_AT.get(i)
    if stage:
        for name, a in E.crafted(z, stage).items():
            p[name + '.alpha'] = a.double()
O.search_epoch(cfg, p, st, opt, E.N_ITER, lambda i: (lambda b: (b[0].double(), b[1]))(E.batch_of(i, cfg.num_classes)),
               lambda i: E.noise_of(i, cfg.num_patches).double(), epoch=0, accum_iter=E.ACCUM, warmup_epochs=E.WARMUP_EPOCHS, lr=E.LR0,
               lr_sched=E.lr_at, hook=hook_o)
# ---- product
inputs = dict(patch_noise=E.noise_of(0, cfg.num_patches), droppath_u=torch.zeros(2 * cfg.depth, E.BATCH))
m = build_product(cfg, O.SearchState(), inputs)
by_name = dict(zip(O.module_names(cfg), m.searchable_modules))
opt_p, opt_a, opt_d = engine.build_optimizers(m, lr=E.LR0['p'], lr_arch=E.LR0['a'], lr_decoder=E.LR0['d'], weight_decay=1e-3)
class Sched:
    def __init__(self, opt, which): self.opt, self.which = opt, which
    def step_update(self, g):
        for grp in self.opt.param_groups: grp['lr'] = E.lr_at(self.which, g)
snap_h = {}
class Loader:
    def __len__(self): return E.N_ITER
    def __iter__(self):
        for i in range(E.N_ITER):
            torch.cuda.synchronize()
            snap_h[i] = {k: v.detach().cpu().double().clone() for k, v in m.named_parameters()}
            stage = E.CRAFT_AT.get(i)
            if stage:
                for name, a in E.crafted(z, stage).items():
                    by_name[name].alpha.data.copy_(a)
            m._forced = dict(patch_noise=E.noise_of(i, cfg.num_patches).to(dev), droppath_u=torch.zeros(2 * cfg.depth, E.BATCH, device=dev))
            yield E.batch_of(i, cfg.num_classes)
crit = OFBSearchLOSS(DistillationLoss(LabelSmoothingCrossEntropy(0.1), None, 'none', 0.5, 1.0), dev, attn_w=0.5, mlp_w=0.5, patch_w=0.0, embedding_w=0.5, flops_w=5.0)
args = types.SimpleNamespace(accum_iter=E.ACCUM, warmup_epochs=E.WARMUP_EPOCHS, epochs=E.EPOCHS)
engine.search_one_epoch(m, crit, 1.0, Loader(), opt_p, opt_d, opt_a, Sched(opt_p, 'p'), Sched(opt_a, 'a'), Sched(opt_d, 'd'), dev, epoch=0, args=args, print_freq=100)
for i in sorted(snap_h):
    worst = []
    for k, v in snap_h[i].items():
        o = snap_o[i][k]
        if tuple(o.shape) != tuple(v.shape):
            worst.append((9e9, k, tuple(o.shape), tuple(v.shape))); continue
        d = float((v - o).abs().max()); s = float(o.abs().max()) + 1e-30
        worst.append((d, k, d / s))
    worst.sort(reverse=True, key=lambda t: t[0])
    print('before batch', i, 'largest differences:', [(k, f'{d:.2e}') for d, k, *_ in worst[:6]])
