"""Model factories with the reference's names and keyword surface (reference models/model.py:88-173).

`create_model(name, **kwargs)` mirrors timm's registry call used by search.py:393-411 (None-valued kwargs dropped).
Pretrained DeiT downloads are out of scope (no network): `pretrained=True` raises.
"""
from functools import partial

from .layers import LayerNorm, ModuleInjection, PatchEmbed
from .vision_transformer import MIMVisionTransformer, VisionTransformer

_REGISTRY = {}


def register_model(fn):
    _REGISTRY[fn.__name__] = fn
    return fn


def create_model(model_name, pretrained=False, **kwargs):
    kwargs = {k: v for k, v in kwargs.items() if v is not None}
    return _REGISTRY[model_name](pretrained=pretrained, **kwargs)


_DIMS = {'tiny': dict(embed_dim=192, num_heads=3), 'small': dict(embed_dim=384, num_heads=6), 'base': dict(embed_dim=768, num_heads=12)}


def _mim(size, pretrained, mae, head_search, channel_search, kwargs):
    if pretrained:
        raise NotImplementedError('pretrained DeiT weights need network access; load a state_dict explicitly')
    kwargs.pop('pretrained_strict', None)
    ModuleInjection.method = kwargs.pop('method', 'full')
    ModuleInjection.searchable_modules = []
    model = MIMVisionTransformer(patch_size=16, depth=12, mlp_ratio=4, qkv_bias=True, norm_layer=partial(LayerNorm, eps=1e-6),
                                 embed_layer=PatchEmbed, mae=mae, head_search=head_search, channel_search=channel_search,
                                 **_DIMS[size], **kwargs)
    model.searchable_modules = [m for m in model.modules() if hasattr(m, 'alpha')]
    return model


def _finetune(size, pretrained, kwargs):
    if pretrained:
        raise NotImplementedError('pretrained weights need network access')
    return VisionTransformer(patch_size=16, depth=12, mlp_ratio=4, qkv_bias=True, norm_layer=partial(LayerNorm, eps=1e-6),
                             embed_layer=PatchEmbed, **_DIMS[size], **kwargs)


@register_model
def deit_tiny_patch16_224_mim(pretrained=False, mae=True, pretrained_strict=False, head_search=False, channel_search=False, **kwargs):
    """not registered by the reference; provided for BASELINE config 1."""
    return _mim('tiny', pretrained, mae, head_search, channel_search, kwargs)


@register_model
def deit_small_patch16_224_mim(pretrained=False, mae=True, pretrained_strict=False, head_search=False, channel_search=False, **kwargs):
    return _mim('small', pretrained, mae, head_search, channel_search, kwargs)


@register_model
def deit_base_patch16_224_mim(pretrained=False, mae=True, pretrained_strict=False, head_search=False, channel_search=False, **kwargs):
    return _mim('base', pretrained, mae, head_search, channel_search, kwargs)


@register_model
def deit_tiny_patch16_224_finetune(pretrained=False, **kwargs):
    """not registered by the reference; the plain counterpart of deit_tiny_patch16_224_mim (BASELINE config 1 plumbing)."""
    return _finetune('tiny', pretrained, kwargs)


@register_model
def deit_small_patch16_224_finetune(pretrained=False, **kwargs):
    return _finetune('small', pretrained, kwargs)


@register_model
def deit_base_patch16_224_finetune(pretrained=False, **kwargs):
    return _finetune('base', pretrained, kwargs)
