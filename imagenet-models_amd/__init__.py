"""MI355X-native engine for the GA-ConvNeXt / GA / MAP ImageNet training hot path.

Python host code (timm-style registry, model containers, train/validate glue) over hand-written HIP kernels for
gfx950 reached through the C ABI in include/gaext.h.  There is no CPU or eager-PyTorch fallback: compute entry
points raise if csrc/libgaext.so is missing.
"""
from . import _lib  # noqa: F401
from . import ops  # noqa: F401

__version__ = '0.1.0'

from .registry import create_model, is_model, list_models, register_model  # noqa: E402,F401
from . import ga_convnext  # noqa: E402,F401  (registers the ga_convnext_* entry points)
from .ga_convnext import GA_ConvNeXt  # noqa: E402,F401
from . import ga_cswin  # noqa: E402,F401  (registers the ga_CSWin_* entry points)
from .ga_cswin import GA_CSWinTransformer  # noqa: E402,F401
from . import map_convnext  # noqa: E402,F401  (registers the map_convnext_* entry points)
from .map_convnext import MAP_ConvNeXt  # noqa: E402,F401
from . import map_vit  # noqa: E402,F401  (registers the map_vit_* entry points)
from .map_vit import MAP_ViT  # noqa: E402,F401
from . import map_pit  # noqa: E402,F401  (registers map_pit_s)
from . import convnext  # noqa: E402,F401  (registers the plain convnext_tiny / convnext_small of map_convnext.py)
from .convnext import ConvNeXt  # noqa: E402,F401
from .map_pit import MAP_PiT  # noqa: E402,F401
from .loss import ga_loss, heads_topk, accuracy_from_topk, map_loss, heads_mean_topk  # noqa: E402,F401
from .optim import create_optimizer_v2, FusedSGD, FusedAdamW, FusedLamb, CosineLRScheduler  # noqa: E402,F401
from .mixup import Mixup  # noqa: E402,F401
from .trainer import TrainStep, distribute_bn, make_buckets  # noqa: E402,F401
from .comm import NativeComm  # noqa: E402,F401
from .checkpoint import save_checkpoint, load_checkpoint, ModelEma  # noqa: E402,F401
