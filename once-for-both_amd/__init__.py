"""once-for-both_amd — MI355X-native implementation of the Once-for-Both (OFB) search-training hot path.

Host side: Python modules that mirror the reference's module API.
Device side: hand-written HIP kernels for gfx950 behind the C ABI in include/ofb_hip.h
(csrc/ -> csrc/libofb_hip.so, loaded with ctypes by `hip.py`).  There is no CPU fallback:
every compute entry point raises if the HIP library or a GPU is missing.
"""
from . import hip, ops, layers, vision_transformer, model, losses, optim, dp, engine, data, utils, lr_sched, lr_decay  # noqa: F401
from .layers import (MAEPatchEmbed, MAESparseAttention, MAESparseMlp, SearchableAttention, SearchableMlp,  # noqa: F401
                     ModuleInjection, LayerNorm, PatchEmbed, Attention, Mlp)
from .vision_transformer import (MIMVisionTransformer, VisionTransformerSearched, VisionTransformer, MAEBlock, Block,  # noqa: F401
                                 norm_targets)
from .model import create_model  # noqa: F401

from .data import (Mixup, SoftTargetCrossEntropy, RASampler, RandomErasing, RandAugment, DeviceTransform, DeviceLoader,  # noqa: F401
                   JpegDecoder)
