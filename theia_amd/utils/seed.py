"""seed_everything (reference utils/seed.py:14-48): python / numpy / torch seeding."""
import os
import random
from typing import Any, Optional

import numpy as np
import torch


def seed_everything(seed: Optional[Any] = None, workers: bool = False) -> int:
    try:
        seed = int(os.environ.get("PL_GLOBAL_SEED", 0)) if seed is None else int(seed)
    except ValueError:
        seed = 0
    if not (0 <= seed <= np.iinfo(np.uint32).max):
        seed = 0
    os.environ["PL_GLOBAL_SEED"] = str(seed)
    os.environ["PYTHON_SEED"] = str(seed)
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    os.environ["PL_SEED_WORKERS"] = f"{int(workers)}"
    return seed
