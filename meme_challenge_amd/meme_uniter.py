"""MemeUniter: UNITER encoder -> pooler -> Linear(hidden, n_classes)
(mirror of model/meme_uniter.py:6-21)."""
from torch import nn

from .model import UniterModel, HipLinear, ensure_store, pool_head


class MemeUniter(nn.Module):

    def __init__(self, uniter_model: UniterModel, hidden_size: int, n_classes: int):
        super().__init__()
        self.uniter_model = uniter_model
        self.n_classes = n_classes
        self.linear = HipLinear(hidden_size, n_classes)

    def param_store(self):
        """Flat parameter/gradient storage shared by encoder and head."""
        st = ensure_store(self)
        self.uniter_model._ensure_handle()
        return st

    def forward(self, **kwargs):
        ensure_store(self)
        out = self.uniter_model(**kwargs)
        # pooler -> linear (model/meme_uniter.py:19-21), one launch each way
        return pool_head(out, self.uniter_model.pooler, self.linear)
