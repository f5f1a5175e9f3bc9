"""Drop-in for the reference's MICCAI-2022/CL_utils/orthogonal_loss.py (stage-1 trainer, `--orth_loss True`,
train_test_MT.py:79,216-218): mean squared cross-correlation of the L2-scaled features of two modalities."""
import torch
import torch.nn as nn

from .. import ops
from .._lib import lib, check, ptr, stream


class _OrthFn(torch.autograd.Function):
    """loss = mean((A^T B)^2), A = x1 / (||x1|| + 1e-6), B = x2 / (||x2|| + 1e-6) with the norms detached
    (orthogonal_loss.py:24-30): row scaling kernels + fp32 GEMMs + a squared-sum reduction."""

    @staticmethod
    def forward(ctx, x1, x2, sync=None):
        x1, x2 = ops._f32(x1), ops._f32(x2)
        B, D1 = x1.shape
        D2 = x2.shape[1]
        a, b = torch.empty_like(x1), torch.empty_like(x2)
        i1 = torch.empty(B, device=x1.device, dtype=torch.float32); i2 = torch.empty_like(i1)
        check(lib().ph_row_invnorm_scale(ptr(x1), ptr(a), ptr(i1), B, D1, 1e-6, stream()), "ph_row_invnorm_scale")
        check(lib().ph_row_invnorm_scale(ptr(x2), ptr(b), ptr(i2), B, D2, 1e-6, stream()), "ph_row_invnorm_scale")
        m = torch.empty(D1, D2, device=x1.device, dtype=torch.float32)
        ops.sgemm(a, b, None, m, D1, D2, B, 1, D1, D2, 1)                  # M = A^T B
        if sync is not None:
            sync.all_reduce_sum(m)            # data parallel: M sums over the global batch; the row gradients stay local
        zero = torch.zeros_like(m)
        out = torch.empty(1, device=x1.device, dtype=torch.float32)
        check(lib().ph_sqdiff_sum(ptr(m), ptr(zero), ptr(out), m.numel(), 1.0 / m.numel(), stream()), "ph_sqdiff_sum")
        ctx.save_for_backward(a, b, i1, i2, m)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        a, b, i1, i2, m = ctx.saved_tensors
        B, D1 = a.shape
        D2 = b.shape[1]
        # dL/dM = 2 g M / (D1 D2)
        gm = torch.empty_like(m)
        zero = torch.zeros_like(m)
        check(lib().ph_scaled_diff(ptr(m), ptr(zero), ptr(ops._f32(g).reshape(1)), 2.0 / m.numel(), ptr(gm), m.numel(),
                                   stream()), "ph_scaled_diff")
        d1 = d2 = None
        if ctx.needs_input_grad[0]:
            da = torch.empty_like(a)
            ops.sgemm(b, gm, None, da, B, D1, D2, D2, 1, 1, D2)            # dA = B dM^T
            d1 = torch.empty_like(a)
            check(lib().ph_row_scale(ptr(da), ptr(i1), ptr(d1), B, D1, stream()), "ph_row_scale")
        if ctx.needs_input_grad[1]:
            db = torch.empty_like(b)
            ops.sgemm(a, gm, None, db, B, D2, D1, D1, 1, D2, 1)            # dB = A dM
            d2 = torch.empty_like(b)
            check(lib().ph_row_scale(ptr(db), ptr(i2), ptr(d2), B, D2, stream()), "ph_row_scale")
        return d1, d2, None


class OrthLoss(nn.Module):
    """orthogonal_loss.py:11-32."""

    sync = None   # a ReplicaSync under data parallelism

    def forward(self, input1, input2):
        bsz = input1.size(0)
        return _OrthFn.apply(input1.reshape(bsz, -1), input2.reshape(bsz, -1), self.sync)
