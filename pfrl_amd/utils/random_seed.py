"""``set_random_seed`` (reference pfrl/utils/random_seed.py:7-22): seeds the
three host RNG streams that define "identical seeds" (SURVEY.md 7.3), and every
device generator."""
import random

import numpy as np
import torch

_HOST_SEEDERS = (random.seed, np.random.seed, torch.manual_seed)


def set_random_seed(seed):
    for seeder in _HOST_SEEDERS:
        seeder(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
