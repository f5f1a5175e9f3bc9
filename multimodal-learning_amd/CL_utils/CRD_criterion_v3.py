"""Drop-in for the reference's "MIA 2022/CL_utils/CRD_criterion_v3.py" (SURVEY row a17): the vanilla CRD
memory bank (one exact positive + K negatives, no discrepancy selection) with a per-sample-weighted NCE loss.
Same module / buffer names: CRDLoss(.embed_s, .embed_t, .contrast{params[5], memory_v1, memory_v2}),
forward(sample_weights, f_s, f_t, idx, contrast_idx) -> tensor of shape [1]."""
import math

import torch
import torch.nn as nn

from .CRD_loss import Embed, Normalize   # noqa: F401  (same classes as CRD_criterion_v3.py:227-250)
from .memory_new import _CRDCoreFn, draw_uniform_indices

eps = 1e-7


class ContrastMemory(nn.Module):
    """CRD_criterion_v3.py:8-81.  params = [K, T, Z_v1, Z_v2, momentum]; scores use the pre-update bank, Z is set
    from the first batch, the rows mem[y] are momentum-updated afterwards - all inside the fused CRD kernels
    (ph_crd_score / ph_crd_select with P = P2 = 1 and selection off / ph_crd_loss_grad / ph_crd_update)."""

    def __init__(self, inputSize, outputSize, K, T=0.07, momentum=0.5):
        super().__init__()
        self.nLem = outputSize
        self.K = K
        self.P, self.P2, self.K2, self.T = 1, 1, K, T
        self.select_neg_pairs = "False"
        self.register_buffer("params", torch.tensor([K, T, -1, -1, momentum], dtype=torch.float32))
        stdv = 1.0 / math.sqrt(inputSize / 3)
        self.register_buffer("memory_v1", torch.rand(outputSize, inputSize).mul_(2 * stdv).add_(-stdv))
        self.register_buffer("memory_v2", torch.rand(outputSize, inputSize).mul_(2 * stdv).add_(-stdv))
        self._z_set = False
        self.sync = None
        self.batch_norm_size = None
        self.verbose = True
        self.last = None

    def _load_from_state_dict(self, *a, **k):
        super()._load_from_state_dict(*a, **k)
        self._z_set = bool((self.params[2:4] > 0).all().item())


class CRDLoss(nn.Module):
    """CRD_criterion_v3.py:150-189."""

    def __init__(self, opt, n_data):
        super().__init__()
        self.embed_s = Embed(opt.s_dim, opt.feat_dim)
        self.embed_t = Embed(opt.t_dim, opt.feat_dim)
        self.contrast = ContrastMemory(opt.feat_dim, n_data, opt.nce_k, opt.nce_t, opt.nce_m)
        self.criterion_t = ContrastLoss(n_data)
        self.criterion_s = ContrastLoss(n_data)

    def forward(self, sample_weights, f_s, f_t, idx, contrast_idx=None):
        if contrast_idx is None:      # CRD_criterion_v3.py:37-39: K + 1 rows per sample from the AliasMethod table, column 0 := idx
            contrast_idx = draw_uniform_indices(self.contrast, idx, self.contrast.K + 1)
        if contrast_idx.shape[1] != self.contrast.K + 1:
            raise RuntimeError("contrast_idx must be [B, nce_k + 1] (CRD_criterion_v3.py:42 views it so)")
        f_s = self.embed_s(f_s)
        f_t = self.embed_t(f_t)
        sample_loss = _CRDCoreFn.apply(f_s, f_t, self.contrast, idx, contrast_idx, None, True)   # [B], already /bsz
        if torch.is_tensor(sample_weights):
            sample_weights = sample_weights.to(sample_loss.device).reshape(-1)
        # s_loss + t_loss of :186-188; the reference's result has shape [1] (sum over dim 0 of a [B,1] tensor)
        return (sample_weights * sample_loss).sum(0, keepdim=True)


class ContrastLoss(nn.Module):
    """Kept for API compatibility; its arithmetic (:200-224) is fused into ph_crd_loss_grad."""

    def __init__(self, n_data):
        super().__init__()
        self.n_data = n_data

    def forward(self, sample_weights, x):
        raise NotImplementedError("ContrastLoss is fused into the CRD loss kernel; call CRDLoss.forward")
