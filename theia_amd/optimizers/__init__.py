from .fused_adamw import FusedAdamW  # noqa: F401
from .utils import param_groups_weight_decay  # noqa: F401
