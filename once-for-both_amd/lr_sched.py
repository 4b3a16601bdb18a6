"""Per-iteration learning-rate schedules of the drivers (reference lr_sched.py:14-122 on timm's CosineLRScheduler; search.py:572-579,
finetune.py:387).  Host-side scalar bookkeeping: one float per parameter group per optimizer step, honouring the `lr_scale` a group
may carry (layer-wise decay, lr_decay.py).

timm (the unpinned fork of requirements.txt:4) is absent from /root/reference: this restates the published CosineLRScheduler rule the
reference configures - linear warm-up from `warmup_lr_init` over `warmup_t` updates, then (`warmup_prefix=True`) a half cosine over
`t_initial` updates from each group's initial lr down to `lr_min`, `lr_min` after `cycle_limit` cycles; `t_in_epochs=False`, so
`step_update(global_step)` moves the rate and `step(epoch)` does nothing.  Parity unpinned for this third-party piece.
"""
import math

import torch


class CosineLRSchedulerwithLayerDecay:
    """reference lr_sched.py:14-41 (`update_groups` multiplies by a group's `lr_scale`) over timm's cosine schedule."""

    def __init__(self, optimizer, t_initial, lr_min=0., warmup_t=0, warmup_lr_init=0, warmup_prefix=False, cycle_limit=0,
                 t_in_epochs=True, noise_range_t=None, noise_pct=0.67, noise_std=1.0, noise_seed=42, initialize=True):
        assert t_initial > 0 and lr_min >= 0
        self.optimizer, self.param_group_field = optimizer, 'lr'
        self.t_initial, self.lr_min, self.warmup_t, self.warmup_lr_init = t_initial, lr_min, warmup_t, warmup_lr_init
        self.warmup_prefix, self.cycle_limit, self.t_in_epochs = warmup_prefix, cycle_limit, t_in_epochs
        self.noise_range_t, self.noise_pct, self.noise_std, self.noise_seed = noise_range_t, noise_pct, noise_std, noise_seed
        for i, g in enumerate(optimizer.param_groups):
            if initialize:
                g.setdefault('initial_lr', g['lr'])
            elif 'initial_lr' not in g:
                raise KeyError(f'param_groups[{i}] has no initial_lr')
        self.base_values = [g['initial_lr'] for g in optimizer.param_groups]
        if warmup_t:
            self.warmup_steps = [(v - warmup_lr_init) / warmup_t for v in self.base_values]
            self.update_groups(warmup_lr_init)
        else:
            self.warmup_steps = [1 for _ in self.base_values]
            self.update_groups(self.base_values)

    def state_dict(self):
        return {k: v for k, v in self.__dict__.items() if k != 'optimizer'}

    def load_state_dict(self, state):
        self.__dict__.update(state)

    def _get_lr(self, t):
        if t < self.warmup_t:
            return [self.warmup_lr_init + t * s for s in self.warmup_steps]
        if self.warmup_prefix:
            t = t - self.warmup_t
        cycle, t_cur = t // self.t_initial, t - self.t_initial * (t // self.t_initial)
        if self.cycle_limit == 0 or cycle < self.cycle_limit:
            return [self.lr_min + 0.5 * (v - self.lr_min) * (1 + math.cos(math.pi * t_cur / self.t_initial)) for v in self.base_values]
        return [self.lr_min for _ in self.base_values]

    def get_cycle_length(self, cycles=0):
        return self.t_initial * max(1, cycles or self.cycle_limit)

    def _noisy(self, lrs, t):
        r = self.noise_range_t
        if r is None or not ((r[0] <= t < r[1]) if isinstance(r, (list, tuple)) else t >= r):
            return lrs
        g = torch.Generator()
        g.manual_seed(self.noise_seed + t)
        while True:                                      # one truncated-normal factor for all groups
            noise = torch.randn(1, generator=g).item()
            if abs(noise) < self.noise_pct:
                break
        return [v + v * noise for v in lrs]

    def update_groups(self, values):
        if not isinstance(values, (list, tuple)):
            values = [values] * len(self.optimizer.param_groups)
        for g, v in zip(self.optimizer.param_groups, values):
            g[self.param_group_field] = v * g['lr_scale'] if 'lr_scale' in g else v

    def step(self, epoch, metric=None):
        if self.t_in_epochs:
            self.update_groups(self._noisy(self._get_lr(epoch), epoch))

    def step_update(self, num_updates, metric=None):
        if not self.t_in_epochs:
            self.update_groups(self._noisy(self._get_lr(num_updates), num_updates))


def create_scheduler(num_epochs, warmup_epochs, warmup_lr, min_lr, args, optimizer, n_iter_per_epoch):
    """reference lr_sched.py:44-122 for `--sched cosine` (the only schedule the OFB workflow uses, exp_sh/run_exp.sh); returns
    (scheduler, epochs incl. cool-down)."""
    if args.sched != 'cosine':
        raise NotImplementedError(f'--sched {args.sched}: the OFB workflow runs the cosine schedule only')
    num_steps, warmup_steps = int(num_epochs * n_iter_per_epoch), int(warmup_epochs * n_iter_per_epoch)
    noise = getattr(args, 'lr_noise', None)
    if noise is not None:
        if isinstance(noise, (list, tuple)):
            noise = [n * num_epochs for n in noise]
            noise = noise[0] if len(noise) == 1 else noise
        else:
            noise = noise * num_epochs
    sched = CosineLRSchedulerwithLayerDecay(
        optimizer, t_initial=num_steps - warmup_steps, lr_min=min_lr, warmup_lr_init=warmup_lr, warmup_t=warmup_steps,
        cycle_limit=getattr(args, 'lr_cycle_limit', 1), t_in_epochs=False, warmup_prefix=True, noise_range_t=noise,
        noise_pct=getattr(args, 'lr_noise_pct', 0.67), noise_std=getattr(args, 'lr_noise_std', 1.), noise_seed=getattr(args, 'seed', 42))
    return sched, sched.get_cycle_length() + args.cooldown_epochs


def adjust_learning_rate(warmup_epochs, lr, min_lr, optimizer, epoch, total_epochs, args):
    """reference lr_sched.py:124-137: linear warm-up then half cosine, per call, honouring `lr_scale`."""
    if epoch < args.warmup_epochs:
        lr = lr * epoch / warmup_epochs
    else:
        lr = min_lr + (lr - min_lr) * 0.5 * (1. + math.cos(math.pi * (epoch - warmup_epochs) / (total_epochs - warmup_epochs)))
    for g in optimizer.param_groups:
        g['lr'] = lr * g['lr_scale'] if 'lr_scale' in g else lr
    return lr
