"""timm-style model registry (the reference's plugin boundary: `@register_model` factories looked up by
`create_model(name, pretrained, num_classes, drop_rate, drop_path_rate, ...)`, GA/train.py:407-420)."""
import sys

import warnings

_entrypoints = {}
_unsupported = {}   # name -> reason: registered for state_dict / checkpoint compatibility, refused by the HIP engine


def register_model(fn, name=None):
    name = name or fn.__name__
    _entrypoints[name] = fn
    mod = sys.modules.get(fn.__module__)
    if name == fn.__name__ and mod is not None and hasattr(mod, '__all__') and name not in mod.__all__:
        mod.__all__.append(name)
    return fn


def _unregister(name):
    _entrypoints.pop(name, None)
    _unsupported.pop(name, None)


def is_model(name):
    return name in _entrypoints


def mark_unsupported(name, reason):
    _unsupported[name] = reason


def is_supported(name):
    return name in _entrypoints and name not in _unsupported


def list_models(filter='', include_unsupported=False):
    """names the HIP engine can run; include_unsupported adds the ones that only construct (parameter layout, checkpoints)"""
    return sorted(n for n in _entrypoints if filter in n and (include_unsupported or n not in _unsupported))


def model_entrypoint(name):
    return _entrypoints[name]


def create_model(model_name, pretrained=False, checkpoint_path='', scriptable=None, **kwargs):
    """timm.create_model semantics: kwargs whose value is None are dropped before reaching the factory."""
    if not is_model(model_name):
        raise RuntimeError(f'Unknown model ({model_name}); known: {list_models()}')
    kwargs = {k: v for k, v in kwargs.items() if v is not None}
    if model_name in _unsupported:   # say so at creation time, not at the first training step
        warnings.warn(f'{model_name}: {_unsupported[model_name]}', stacklevel=2)
    model = _entrypoints[model_name](pretrained=pretrained, **kwargs)
    if checkpoint_path:
        from .checkpoint import load_checkpoint
        load_checkpoint(model, checkpoint_path)
    return model
