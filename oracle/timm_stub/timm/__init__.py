"""TEST INFRASTRUCTURE ONLY -- minimal stand-in for the handful of `timm` symbols the
reference model files import (timm itself is not installed and cannot be fetched).

Used only by oracle/gen_golden.py in the build container to import
/root/reference/GA/ga_convnext.py and friends and dump golden vectors.  This is the
build's own code restating timm's *published* semantics (timm 0.9.x); it is "parity
unpinned" with respect to timm itself.  Never imported by the
product package.
"""
from .models import create_model  # noqa: F401
