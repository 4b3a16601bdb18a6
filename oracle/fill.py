"""Deterministic closed-form tensors shared by the golden generator, the oracle and the tests.

TEST INFRASTRUCTURE ONLY (see oracle/README.md).  Every parity case fills parameters and
inputs with these formulas on BOTH sides (reference / oracle / HIP path), so fixtures only
need to hold expected outputs.  numpy float64 -> float32, no RNG, identical on every host.
"""
import zlib

import numpy as np


def _phase(name: str) -> float:
    return (zlib.crc32(name.encode()) % 100003) * 0.6180339887


def wave(name: str, shape, scale: float = 1.0, offset: float = 0.0) -> np.ndarray:
    """Quasi-random, all-distinct values in offset + scale*[-1, 1]."""
    n = int(np.prod(shape)) if len(shape) else 1
    i = np.arange(n, dtype=np.float64)
    v = np.sin(i * 12.9898 + _phase(name)) * 43758.5453
    v = (v - np.floor(v)) * 2.0 - 1.0          # classic shader hash -> U(-1,1)
    return (offset + scale * v).reshape(shape).astype(np.float32)


def uniform01(name: str, shape) -> np.ndarray:
    return ((wave(name, shape).astype(np.float64) + 1.0) * 0.5).astype(np.float32)


def param_value(name: str, shape) -> np.ndarray:
    """Per-parameter fill rule keyed on the reference's state_dict names."""
    leaf = name.split('.')[-1]
    if leaf == 'alpha' or name == 'alpha_patch':
        return uniform01(name, shape)                       # alpha ~ U[0,1)  (layers.py:455)
    if leaf == 'score':
        return wave(name, shape, 0.4)                       # trunc-normal(.2) range (layers.py:467)
    if 'norm' in name and leaf == 'weight':
        return wave(name, shape, 0.2, 1.0)
    if leaf == 'bias':
        return wave(name, shape, 0.05)
    if name in ('cls_token', 'mask_token', 'pos_embed'):
        return wave(name, shape, 0.05)
    fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else int(shape[0])
    return wave(name, shape, 1.7 / np.sqrt(fan_in))         # ~xavier-sized weights


def images(batch: int, chans: int = 3, size: int = 224, tag: str = 'imgs') -> np.ndarray:
    """Smooth structure + hash noise, roughly unit variance (stand-in for normalised ImageNet)."""
    y, x = np.meshgrid(np.arange(size, dtype=np.float64), np.arange(size, dtype=np.float64), indexing='ij')
    out = np.empty((batch, chans, size, size), dtype=np.float32)
    for b in range(batch):
        for c in range(chans):
            smooth = np.sin(x * (0.031 + 0.007 * c) + b) * np.cos(y * (0.023 + 0.005 * b) + c)
            out[b, c] = (0.8 * smooth + 0.9 * wave(f'{tag}.{b}.{c}', (size, size)).astype(np.float64)).astype(np.float32)
    return out


def labels(batch: int, ncls: int) -> np.ndarray:
    return ((np.arange(batch, dtype=np.int64) * 7 + 3) % ncls).astype(np.int64)


def patch_noise(batch: int, length: int = 196, tag: str = 'patch_noise') -> np.ndarray:
    return uniform01(tag, (batch, length))


def droppath_noise(n_calls: int, batch: int, tag: str = 'droppath') -> np.ndarray:
    return uniform01(tag, (n_calls, batch))
