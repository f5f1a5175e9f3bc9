"""Drop-in for the reference's MICCAI-2022/CL_utils/CRD_criterion.py (stage-1 mean-teacher trainer, `--CRD_distill 1`,
train_test_MT.py:74-76,157-165): the vanilla CRD memory bank (one exact positive + K sampled negatives) with a
two-layer projection head.  Same module / buffer names: CRDLoss(opt){.embed_s, .embed_t, .contrast{params[5],
memory_v1, memory_v2}}, forward(f_s, f_t, idx, contrast_idx) -> tensor of shape [1].  The bank arithmetic is the one of
CRD_criterion_v3 (MIA-2022) with unit sample weights: same fused kernels."""
import torch
import torch.nn as nn

from .. import ops
from .CRD_loss import Normalize
from .CRD_criterion_v3 import ContrastMemory, ContrastLoss   # noqa: F401  (:8-81, :190-217 are identical up to the weights)
from .memory_new import _CRDCoreFn, draw_uniform_indices


class Embed(nn.Module):
    """CRD_criterion.py:219-234: Linear -> ReLU -> Linear, then L2 normalisation (state_dict keys linear.0.*, linear.2.*)."""

    def __init__(self, dim_in=1024, dim_out=128):
        super().__init__()
        self.linear = nn.Sequential(nn.Linear(dim_in, dim_out), nn.ReLU(), nn.Linear(dim_out, dim_out))
        self.l2norm = Normalize(2)

    def forward(self, x):
        x = x.view(x.shape[0], -1)
        h = ops.LinearActFn.apply(x, self.linear[0].weight, self.linear[0].bias, ops.ACT_RELU)
        y = ops.LinearFn.apply(h, self.linear[2].weight, self.linear[2].bias)
        return self.l2norm(y)


class CRDLoss(nn.Module):
    """CRD_criterion.py:143-189 (reads opt.n_data)."""

    def __init__(self, opt):
        super().__init__()
        self.embed_s = Embed(opt.s_dim, opt.feat_dim)
        self.embed_t = Embed(opt.t_dim, opt.feat_dim)
        self.contrast = ContrastMemory(opt.feat_dim, opt.n_data, opt.nce_k, opt.nce_t, opt.nce_m)
        self.criterion_t = ContrastLoss(opt.n_data)
        self.criterion_s = ContrastLoss(opt.n_data)

    def forward(self, f_s, f_t, idx, contrast_idx=None):
        if contrast_idx is None:      # CRD_criterion.py:37-39: K + 1 rows per sample from the AliasMethod table, column 0 := idx
            contrast_idx = draw_uniform_indices(self.contrast, idx, self.contrast.K + 1)
        if contrast_idx.shape[1] != self.contrast.K + 1:
            raise RuntimeError("contrast_idx must be [B, nce_k + 1] (CRD_criterion.py:42 views it so)")
        f_s = self.embed_s(f_s)
        f_t = self.embed_t(f_t)
        sample_loss = _CRDCoreFn.apply(f_s, f_t, self.contrast, idx, contrast_idx, None, True)   # [B], already / bsz
        return sample_loss.sum(0, keepdim=True)                                                  # s_loss + t_loss, [1]
