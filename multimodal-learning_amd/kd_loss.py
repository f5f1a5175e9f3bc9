"""Drop-in for the reference's MICCAI-2022/KD_loss.py."""
import torch.nn as nn

from . import ops


class DistillKL(nn.Module):
    """Distilling the Knowledge in a Neural Network (KD_loss.py:7-17).  `batch_norm_size` lets a
    data-parallel caller divide by the GLOBAL batch (DataParallel semantics); default = local batch."""

    def __init__(self, T):
        super().__init__()
        self.T = T
        self.batch_norm_size = None

    def forward(self, y_s, y_t):
        bn = self.batch_norm_size or y_s.shape[0]
        return ops.KLFn.apply(y_s, y_t.detach(), float(self.T), float(bn))
