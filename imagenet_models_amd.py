"""Import shim: the package directory is named `imagenet-models_amd/` (not a valid Python identifier);
`import imagenet_models_amd` loads it from there."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'imagenet-models_amd')
_spec = importlib.util.spec_from_file_location('imagenet_models_amd', os.path.join(_dir, '__init__.py'),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules['imagenet_models_amd'] = _mod
_spec.loader.exec_module(_mod)
