"""AdamW with the reference's interface (reference optim.py:7-182) on one fused multi-tensor kernel per group."""
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

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            by_step = {}
            for p in group['params']:
                if p.grad is None:
                    continue
                st = self.state[p]
                if len(st) == 0:
                    st['step'] = 0
                    st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st['step'] += 1
                by_step.setdefault(st['step'], []).append(p)
            b1, b2 = group['betas']
            for step, plist in by_step.items():
                tab = (hip.AdamwTensor * len(plist))()
                keep, maxn = [], 0
                for i, p in enumerate(plist):
                    g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                    if not p.is_contiguous():
                        raise hip.OfbError('AdamW needs contiguous parameters')
                    st = self.state[p]
                    tab[i].p, tab[i].g = p.data_ptr(), g.data_ptr()
                    tab[i].m, tab[i].v, tab[i].n = st['exp_avg'].data_ptr(), st['exp_avg_sq'].data_ptr(), p.numel()
                    maxn = max(maxn, p.numel())
                    keep.append(g)
                dev_tab, host = hip.upload_structs(tab, plist[0].device)
                hip.adamw_step(dev_tab, len(plist), maxn, group['lr'], b1, b2, group['eps'], group['weight_decay'], step)
                self._keep = (dev_tab, host, keep)
        return loss

    def update(self, ori_w, cur_w, w_name, group_idx, keep_idx, dim, initialize=False):
        raise NotImplementedError('optimizer-state surgery belongs to compress() (SURVEY 8f-1)')
