"""once-for-both_amd — MI355X-native implementation of the Once-for-Both (OFB) search-training hot path.

Host side: Python modules that mirror the reference's module API.
Device side: hand-written HIP kernels for gfx950 behind the C ABI in include/ofb_hip.h
(csrc/ -> csrc/libofb_hip.so, loaded with ctypes by `hip.py`).  There is no CPU fallback:
every compute entry point raises if the HIP library or a GPU is missing.
"""
import os as _os
# HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) in creation order.  With RCCL's streams in the process the
# weight-gradient side stream can land on the main stream's queue and lose its overlap (hip.ensure_side_stream); eight queues keep
# them apart.  Effective only if the HIP runtime is not initialised yet (import this package before the first GPU call).
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
from . import hip, ops, layers, vision_transformer, model, losses, optim, dp, engine, data  # noqa: F401
from .layers import (MAEPatchEmbed, MAESparseAttention, MAESparseMlp, SearchableAttention, SearchableMlp,  # noqa: F401
                     ModuleInjection, LayerNorm, PatchEmbed, Attention, Mlp)
from .vision_transformer import (MIMVisionTransformer, VisionTransformerSearched, VisionTransformer, MAEBlock, Block,  # noqa: F401
                                 norm_targets)
from .model import create_model  # noqa: F401

from .data import (Mixup, SoftTargetCrossEntropy, RASampler, RandomErasing, RandAugment, DeviceTransform, DeviceLoader,  # noqa: F401
                   JpegDecoder)
