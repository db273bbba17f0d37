"""``evaluating(module)``: run a block with the module in eval mode and put it
back into its previous mode afterwards (reference pfrl/utils/contexts.py)."""


class evaluating(object):
    def __init__(self, module):
        self.module = module
        self._restore_train = False

    def __enter__(self):
        self._restore_train = self.module.training
        self.module.eval()
        return self.module

    def __exit__(self, exc_type, exc, tb):
        if self._restore_train:
            self.module.train()
        return False
