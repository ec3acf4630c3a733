"""The DINOv2-APLA self-supervised step (reference self_supervised/dinov2/): losses, projection head, KoLeo,
the multi-crop (packed) backbone, the masking collate and schedules, the student/teacher meta-architecture and the
training iteration."""
from .backbone import DinoVisionTransformer, MemEffAttention  # noqa: F401
from .collate import CosineScheduler, MaskingGenerator, build_schedulers, collate_data_and_cast  # noqa: F401
from .heads import DINOHead, KoLeoLoss, update_teacher  # noqa: F401
from .losses import DINOLoss, iBOTPatchLoss  # noqa: F401
from .models import DINOv2, build_model  # noqa: F401
from .trainer import Dinov2Trainer  # noqa: F401
