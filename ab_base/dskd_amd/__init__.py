"""dskd_amd -- MI355X-native implementation of the DSKD distillation hot path.

Importing the package registers every module under the reference's registry names
(``DeformableDETR_il``, ``GFLDeformableDETRHead_il``, ``GFLHungarianAssigner`` ...), so the
reference's config files resolve to these classes."""
from . import (backbones, bbox, deformable_detr_il, gfl_deformable_detr_head_il, gfl_head, losses, necks, swin,  # noqa: F401
               transformer)
from .builder import build_detector  # noqa: F401
from .config import Config  # noqa: F401

__version__ = "0.1.0"
