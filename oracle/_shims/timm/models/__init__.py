from .registry import register_model, _REGISTRY


def create_model(name, pretrained=False, **kwargs):
    # timm drops None-valued kwargs before calling the factory (search.py:393-411)
    kwargs = {k: v for k, v in kwargs.items() if v is not None}
    return _REGISTRY[name](pretrained=pretrained, **kwargs)
