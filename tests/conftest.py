import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture
def knobs():
    """set library tuning knobs for one test (`knobs(NT_DMA=2)`; include/gaext.h ga_set_knob) and restore the defaults after it:
    the library reads the GAEXT_* environment once only, so tests switch kernel forms through the C ABI"""
    from imagenet_models_amd import _lib
    touched = []

    def set_(**kv):
        lib = _lib.load()
        for k, v in kv.items():
            _lib.check(lib.ga_set_knob(k.encode(), int(v)), f'ga_set_knob({k})')
            touched.append(k)
    yield set_
    for k in touched:
        _lib.load().ga_unset_knob(k.encode())


def pytest_collection_modifyitems(config, items):
    # GPU tests are skipped automatically when no device is visible (keeps `pytest tests/` usable on CPU).
    try:
        import torch
        have = torch.cuda.is_available()
    except Exception:
        have = False
    if have:
        return
    skip = pytest.mark.skip(reason='no GPU visible')
    for it in items:
        if 'gpu' in it.keywords:
            it.add_marker(skip)
