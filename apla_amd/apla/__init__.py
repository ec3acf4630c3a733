"""Same import surface as the reference's ``apla`` package."""
from .apla_vit import build_apla, indices_from_trainable, replace_attn_with_apla  # noqa: F401
from .appla_attn import APLA_Attention  # noqa: F401
from .appla_attn_mem_eff import APLA_MemEffAttention  # noqa: F401
