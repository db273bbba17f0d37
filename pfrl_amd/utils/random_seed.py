"""``set_random_seed`` (reference pfrl/utils/random_seed.py:7-22): seeds the
three host RNG streams that define "identical seeds" (SURVEY.md 7.3)."""
import random

import numpy as np
import torch


def set_random_seed(seed):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
