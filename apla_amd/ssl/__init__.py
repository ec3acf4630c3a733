"""Pieces of the DINOv2-APLA self-supervised step that exist so far: the self-distillation losses, the
projection head, KoLeo, the EMA teacher update and the multi-crop (packed) backbone forward.  The masking collate, the
step glue (dinov2/models.py:207-441) and the SSL trainer are not built yet."""
from .backbone import DinoVisionTransformer  # noqa: F401
from .heads import DINOHead, KoLeoLoss, update_teacher  # noqa: F401
from .losses import DINOLoss, iBOTPatchLoss  # noqa: F401
