"""AdamW with the reference's interface (reference optim.py:7-182) on one fused multi-tensor kernel per group."""
import math
import struct

import torch
from torch.optim.optimizer import Optimizer

from . import hip


class AdamW(Optimizer):
    def __init__(self, params, param_names=None, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, amsgrad=False):
        if amsgrad:
            raise NotImplementedError('amsgrad is never enabled on the OFB path')
        if lr < 0 or eps < 0 or weight_decay < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1:
            raise ValueError('invalid AdamW hyper-parameter')
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad))
        self.param_names = param_names
        self._tables = {}         # (group, tensors) -> (addresses, device table, pinned copy, largest tensor): step()
        self.table_hits = 0       # launches that reused their table (tests / diagnostics)
        self._cap = None          # while / after a hipGraph capture of step(): device-resident {lr, bc1, 1/sqrt(bc2)} per launch

    # ---- hipGraph support (engine.GraphedStep) ---------------------------------------------------------------------------------
    # A captured step() must not bake the learning rate or the step count into its launch arguments: while capturing, every launch
    # reads them from a slot of a small device tensor; refresh_hyper(k) recomputes the slots on the host (same double-precision
    # formulas as the eager path) for the k-th replay and copies them over, outside the graph.
    def begin_capture(self, max_launches=16):
        dev = next(p for g in self.param_groups for p in g['params']).device
        self._cap = dict(active=True, slots=[], dev=torch.zeros(4 * max_launches, device=dev), max=max_launches)

    def end_capture(self):
        if self._cap is not None:
            self._cap['active'] = False

    def refresh_hyper(self, k):
        cap = self._cap
        vals = [0.0] * (4 * cap['max'])
        for i, (group, step0, plist) in enumerate(cap['slots']):
            t = step0 + k
            # the eager entry point receives the betas as C floats and forms the corrections in double: same here, bit for bit
            b1, b2 = (struct.unpack('f', struct.pack('f', b))[0] for b in group['betas'])
            vals[4 * i], vals[4 * i + 1], vals[4 * i + 2] = group['lr'], 1.0 - math.pow(b1, t), 1.0 / math.sqrt(1.0 - math.pow(b2, t))
            for p in plist:
                self.state[p]['step'] = t
        # a fresh pinned staging tensor per refresh: the host may run several replays ahead of the GPU, and torch's pinned
        # allocator does not hand a block out again before the copy that reads it has completed
        cap['dev'].copy_(torch.tensor(vals, dtype=torch.float32).pin_memory(), non_blocking=True)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        hip.join_side()                          # weight gradients are produced on the side stream (ops.py)
        hip.flush_deferred()                     # queued LayerNorm / bias gradient reductions (normally done at the end of backward)
        for group in self.param_groups:
            by_step = {}
            state = self.state
            for p in group['params']:
                g = p.grad
                if g is None:
                    continue
                st = state[p]
                if len(st) == 0:
                    st['step'] = 0
                    st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                t = st['step'] = st['step'] + 1
                lst = by_step.get(t)
                if lst is None:
                    lst = by_step[t] = []
                lst.append((p, g, st))
            b1, b2 = group['betas']
            for step, triples in by_step.items():
                # the device-side tensor table of a launch is kept while every address in it stays what it was: no table build, no upload.
                # That needs gradients at FIXED addresses (the data-parallel buckets' slots, gradients written in place) - plain
                # autograd gradients get a different permutation of the caching allocator's blocks every step and rebuild the table
                # (scripts/lab/table_reuse_count.py: 0 reuses in 20 steps).  While a capture is recorded every launch gets its own.
                keep, maxn, addr, plist = [], 0, [], []
                for p, g, st in triples:
                    if not g.is_contiguous():
                        g = g.contiguous()
                    if not p.is_contiguous():
                        raise hip.OfbError('AdamW needs contiguous parameters')
                    addr += (p.data_ptr(), g.data_ptr(), st['exp_avg'].data_ptr(), st['exp_avg_sq'].data_ptr(), p.numel())
                    keep.append(g)
                    plist.append(p)
                addr = tuple(addr)
                slot = (id(group), len(plist))
                capturing = self._cap is not None and self._cap['active']
                cached = None if capturing else self._tables.get(slot)
                if cached is not None and cached[0] == addr:
                    dev_tab, host, maxn = cached[1], cached[2], cached[3]
                    self.table_hits += 1
                else:
                    tab = (hip.AdamwTensor * len(plist))()
                    for i, p in enumerate(plist):
                        t = tab[i]
                        t.p, t.g, t.m, t.v, t.n = addr[5 * i:5 * i + 5]
                        maxn = max(maxn, addr[5 * i + 4])
                    dev_tab, host = hip.upload_structs(tab, plist[0].device)
                    if not capturing:
                        self._tables[slot] = (addr, dev_tab, host, maxn)
                if self._cap is not None and self._cap['active']:
                    cap = self._cap
                    i = len(cap['slots'])
                    if i >= cap['max']:
                        raise hip.OfbError('AdamW capture: more launches than hyper-parameter slots')
                    cap['slots'].append((group, step, list(plist)))
                    hip.adamw_step_dev(dev_tab, len(plist), maxn, cap['dev'][4 * i:4 * i + 4], b1, b2, group['eps'], group['weight_decay'])
                else:
                    hip.adamw_step(dev_tab, len(plist), maxn, group['lr'], b1, b2, group['eps'], group['weight_decay'], step)
                self._keep = (dev_tab, host, keep)
        hip.bump_weight_epoch()                  # the H-format copies of the weights (hip.weight_h) are stale now
        return loss

    def update(self, ori_w, cur_w, w_name, group_idx, keep_idx, dim, initialize=False):
        """compress() hook (reference optim.py:122-182): parameter `w_name` of group `group_idx` was replaced by `cur_w`.

        * cur_w frozen (requires_grad False): the slot and its state leave the optimizer.
        * initialize=True: the moments restart from zero at step 0 (re-seeded alpha / finished score).
        * otherwise the moments are cut exactly like the weight was: `keep_idx` is a 1-D index (index_select along `dim`)
          or an index tensor of the state's rank (gather along `dim`); lists of both apply the cuts in sequence.
        A parameter that never received a step has no moments to carry; its slot is simply re-pointed."""
        names = self.param_names[group_idx]
        slot = names.index(w_name)
        plist = self.param_groups[group_idx]['params']
        old = self.state.pop(ori_w, None)
        if not cur_w.requires_grad:
            del names[slot]
            del plist[slot]
            return
        plist[slot] = cur_w
        if old is None or len(old) == 0:
            return
        if initialize:
            self.state[cur_w] = {'step': 0, 'exp_avg': torch.zeros_like(cur_w, memory_format=torch.preserve_format),
                                 'exp_avg_sq': torch.zeros_like(cur_w, memory_format=torch.preserve_format)}
            return
        cuts = list(zip(keep_idx, dim)) if isinstance(keep_idx, (list, tuple)) else [(keep_idx, dim)]
        m, v = old['exp_avg'], old['exp_avg_sq']
        for idx, d in cuts:
            m, v = take(m, idx, d), take(v, idx, d)
        if tuple(m.shape) != tuple(cur_w.shape):
            raise hip.OfbError(f'AdamW.update({w_name}): cut moments {tuple(m.shape)} do not match the parameter {tuple(cur_w.shape)}')
        self.state[cur_w] = {'step': old['step'], 'exp_avg': m, 'exp_avg_sq': v}


def take(t, index, dim):
    """the slicing vocabulary of compress(): 1-D `index` -> t.index_select(dim, index); same-rank `index` (2-D) ->
    torch.gather(t, dim, index).  Runs on the device through ofb_index_select."""
    index = torch.as_tensor(index)
    if index.dim() == 1:
        return hip.index_select(t, index, dim)
    if index.dim() != 2 or t.dim() != 2:
        raise hip.OfbError('gather-style cuts are defined for 2-D state only')
    rows, cols = t.shape
    index = index.to(torch.int64)
    if dim % 2 == 1:                                   # out[r][c] = t[r][index[r][c]]
        flat = torch.arange(index.shape[0], device=index.device).view(-1, 1) * cols + index
    else:                                              # out[r][c] = t[index[r][c]][c]
        flat = index * cols + torch.arange(index.shape[1], device=index.device).view(1, -1)
    return hip.index_select(t.reshape(-1), flat.reshape(-1), 0).view(index.shape)
