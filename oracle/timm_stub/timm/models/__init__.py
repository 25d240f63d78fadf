"""Stub of timm.models: registry + builder helpers (test infrastructure only)."""
import sys

_model_entrypoints = {}


def register_model(fn):
    _model_entrypoints[fn.__name__] = fn
    mod = sys.modules[fn.__module__]
    if hasattr(mod, '__all__') and fn.__name__ not in mod.__all__:
        mod.__all__.append(fn.__name__)
    return fn


def register_notrace_module(module):
    return module


def named_apply(fn, module, name='', depth_first=True, include_root=False):
    # timm semantics: depth-first over named_children, children before parent
    if not depth_first and include_root:
        fn(module=module, name=name)
    for child_name, child_module in module.named_children():
        child_name = '.'.join((name, child_name)) if name else child_name
        named_apply(fn=fn, module=child_module, name=child_name, depth_first=depth_first, include_root=True)
    if depth_first and include_root:
        fn(module=module, name=name)
    return module


def build_model_with_cfg(model_cls, variant, pretrained, **kwargs):
    for k in ('pretrained_cfg', 'pretrained_cfg_overlay', 'features_only', 'default_cfg'):
        kwargs.pop(k, None)
    if pretrained:
        raise RuntimeError('no pretrained weights are obtainable offline')
    return model_cls(**kwargs)


def create_model(model_name, pretrained=False, checkpoint_path='', scriptable=None, **kwargs):
    kwargs = {k: v for k, v in kwargs.items() if v is not None}
    return _model_entrypoints[model_name](pretrained=pretrained, **kwargs)


from . import layers  # noqa: E402,F401
from . import registry  # noqa: E402,F401
