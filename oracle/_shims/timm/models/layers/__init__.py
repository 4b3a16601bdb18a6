import math
import torch
import torch.nn as nn
from .helpers import to_2tuple


def trunc_normal_(tensor, mean=0., std=1., a=-2., b=2.):
    # standard truncated normal init (same call as torch.nn.init.trunc_normal_)
    return torch.nn.init.trunc_normal_(tensor, mean=mean, std=std, a=a, b=b)


def lecun_normal_(tensor):
    fan_in = tensor.shape[1] if tensor.dim() > 1 else tensor.shape[0]
    return trunc_normal_(tensor, std=math.sqrt(1.0 / fan_in) / .87962566103423978)


# Stochastic depth: per-sample Bernoulli(keep)/keep scaling of the residual branch.
# The keep-mask draw goes through `DropPath.rand` so the golden generator can capture it.
class DropPath(nn.Module):
    rand = staticmethod(torch.rand)

    def __init__(self, drop_prob=None):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        if self.drop_prob == 0. or not self.training:
            return x
        keep_prob = 1 - self.drop_prob
        shape = (x.shape[0],) + (1,) * (x.ndim - 1)
        random_tensor = keep_prob + DropPath.rand(shape, dtype=x.dtype, device=x.device)
        random_tensor.floor_()
        return x.div(keep_prob) * random_tensor
