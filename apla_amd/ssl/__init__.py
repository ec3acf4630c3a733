"""Pieces of the DINOv2-APLA self-supervised step that exist so far: the self-distillation losses."""
from .losses import DINOLoss, iBOTPatchLoss  # noqa: F401
