"""timm-style model registry (the reference's plugin boundary: `@register_model` factories looked up by
`create_model(name, pretrained, num_classes, drop_rate, drop_path_rate, ...)`, GA/train.py:407-420)."""
import sys

_entrypoints = {}


def register_model(fn):
    _entrypoints[fn.__name__] = fn
    mod = sys.modules[fn.__module__]
    if hasattr(mod, '__all__') and fn.__name__ not in mod.__all__:
        mod.__all__.append(fn.__name__)
    return fn


def is_model(name):
    return name in _entrypoints


def list_models(filter=''):
    return sorted(n for n in _entrypoints if filter in n)


def model_entrypoint(name):
    return _entrypoints[name]


def create_model(model_name, pretrained=False, checkpoint_path='', scriptable=None, **kwargs):
    """timm.create_model semantics: kwargs whose value is None are dropped before reaching the factory."""
    if not is_model(model_name):
        raise RuntimeError(f'Unknown model ({model_name}); known: {list_models()}')
    kwargs = {k: v for k, v in kwargs.items() if v is not None}
    model = _entrypoints[model_name](pretrained=pretrained, **kwargs)
    if checkpoint_path:
        from .checkpoint import load_checkpoint
        load_checkpoint(model, checkpoint_path)
    return model
