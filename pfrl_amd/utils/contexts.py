"""``evaluating(net)`` context (reference pfrl/utils/contexts.py:4-13)."""
from contextlib import contextmanager


@contextmanager
def evaluating(net):
    """Temporarily switch a module to evaluation mode."""
    was_training = net.training
    try:
        net.eval()
        yield net
    finally:
        if was_training:
            net.train()
