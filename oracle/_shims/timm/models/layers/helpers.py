import collections.abc
from itertools import repeat


def to_2tuple(x):
    if isinstance(x, collections.abc.Iterable):
        return tuple(x)
    return tuple(repeat(x, 2))
