"""Drop-in for the reference's CL_utils/CRD_loss.py: CRDLoss (:127-175), ContrastLoss_v2 (:212-252),
Embed (:256-267), Normalize (:270-279)."""
import torch
import torch.nn as nn

from .. import ops
from .memory_new import ContrastMemory_v3

eps = 1e-7


class CRDLoss(nn.Module):
    """CRD loss with DC-Distill pair selection.  forward(epoch, f_s, f_t, idx, contrast_idx) -> 0-d loss (the [B] per-sample
    losses with opt.sample_KD == "True"), differentiable w.r.t. f_s and both Embed layers; mutates the banks and Z exactly
    like the reference."""

    def __init__(self, opt, n_data):
        super().__init__()
        self.P = opt.nce_p
        self.P2 = opt.nce_p2
        self.embed_s = Embed(opt.s_dim, opt.feat_dim)
        self.embed_t = Embed(opt.t_dim, opt.feat_dim)
        self.contrast = ContrastMemory_v3(opt.feat_dim, n_data, opt.nce_p, opt.nce_k, opt.nce_t, opt.nce_m,
                                          opt.select_pos_pairs, opt.nce_p2, opt.select_neg_pairs, opt.nce_k2)
        # --sample_KD True (options.py:69): both criteria return the [B] per-sample losses of ContrastLoss_v2's second branch
        # (:246-250) and forward() their sum s_loss + t_loss, shape [B] - the same fused kernel, rows left un-normalised
        self.sample_KD = getattr(opt, "sample_KD", "False")
        self.contrast.sample_KD = self.sample_KD == "True"
        self.criterion_t = ContrastLoss_v2(n_data, sample_KD=opt.sample_KD)
        self.criterion_s = ContrastLoss_v2(n_data, sample_KD=opt.sample_KD)
        self.select_pos_mode = opt.select_pos_mode

    def forward(self, epoch, f_s, f_t, idx, contrast_idx=None, ranks=None):
        # (contrast_idx=None: ContrastMemory_v3 draws the rows itself, memory_new.py:265-267)
        if self.sample_KD not in ("False", "True"):
            raise UnboundLocalError("local variable 'loss' referenced before assignment")   # what :252 raises there
        f_s = self.embed_s(f_s)
        f_t = self.embed_t(f_t)
        return self.contrast.loss(epoch, f_s, f_t, idx, contrast_idx, self.select_pos_mode, ranks)


class _ContrastLossV2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, P, n_data, per_sample):
        from .._lib import lib, check, ptr, stream
        B, S = x.shape[0], x.shape[1]
        xf = ops._f32(x).reshape(B, S).contiguous()
        rows = torch.empty(B, device=x.device, dtype=torch.float32)
        dx = torch.empty(B, S, device=x.device, dtype=torch.float32)
        check(lib().ph_contrast_loss_v2(ptr(xf), ptr(rows), ptr(dx), B, S, int(P), float(n_data), stream()),
              "ph_contrast_loss_v2")
        ctx.save_for_backward(dx)
        ctx.shape, ctx.per_sample = tuple(x.shape), per_sample
        if per_sample:
            return rows
        out = torch.empty((), device=x.device, dtype=torch.float32)
        check(lib().ph_sum(ptr(rows), ptr(out), B, 1.0 / B, stream()), "ph_sum")
        return out

    @staticmethod
    def backward(ctx, g):
        (dx,) = ctx.saved_tensors
        B = dx.shape[0]
        g = g.reshape(-1, 1) if ctx.per_sample else g / B
        return (dx * g).reshape(ctx.shape), None, None, None


class ContrastLoss_v2(nn.Module):
    """CRD_loss.py:212-252.  Inside CRDLoss its arithmetic is fused into ph_crd_loss_grad; called directly it evaluates
    the same formulas on x [B, P+N, 1] (positives first): a 0-d loss for sample_KD == "False" (:240-244), the per-sample
    losses [B] for sample_KD == "True" (:246-250)."""

    def __init__(self, n_data, sample_KD):
        super().__init__()
        self.n_data = n_data
        self.sample_KD = sample_KD

    def forward(self, x, P):
        if self.sample_KD not in ("False", "True"):
            raise UnboundLocalError("local variable 'loss' referenced before assignment")   # what :252 raises there
        return _ContrastLossV2Fn.apply(x, P, self.n_data, self.sample_KD == "True")


class Embed(nn.Module):
    """Linear + L2 normalisation (CRD_loss.py:256-267)."""

    def __init__(self, dim_in=1024, dim_out=128):
        super().__init__()
        self.linear = nn.Linear(dim_in, dim_out)
        self.l2norm = Normalize(2)

    def forward(self, x):
        x = x.view(x.shape[0], -1)
        x = ops.LinearFn.apply(x, self.linear.weight, self.linear.bias)
        return self.l2norm(x)


class Normalize(nn.Module):
    def __init__(self, power=2):
        super().__init__()
        if power != 2:
            raise NotImplementedError("only the L2 norm is used (CRD_loss.py:261)")
        self.power = power

    def forward(self, x):
        return ops.L2NormFn.apply(x)
