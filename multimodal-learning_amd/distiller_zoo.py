"""Feature-distillation baselines of the reference's distiller zoo that the trainers can select with `--distill`
(part): `Similarity` ("MIA 2022/distiller_zoo/SP.py": similarity-preserving KD; also
`opt.distill == "sp"` in the MIA-2023 stage-2 trainer, train_test_path_multi_distill.py:369-370) and `feats_KL`
("MIA 2022/distiller_zoo/feats_KL.py": KL divergence between softmaxed feature vectors).  Both are compositions of
kernels of the hot path (fp32 GEMM, row L2 normalisation, squared-difference reduction, the KL kernels); the teacher
side is treated as a constant (the trainers pass it detached).  `RKDLoss` ("MIA 2022/distiller_zoo/RKD.py") and `PKT`
("MIA 2022/distiller_zoo/PKT.py"), the `--distill rkd|pkt` choices of train_test_path_multi_distill_v2.py:339-342, are
closed-form loss + gradient kernels of their own (csrc/zoo.hip).  The other zoo members are not built."""
import torch
import torch.nn as nn

from . import ops
from ._lib import lib, check, ptr, stream
from .tsvd import _SqDiffFn


class Similarity(nn.Module):
    """SP.py:9-30: G = rownormalize(f f^T) for student and teacher, loss = ||G_t - G_s||_F^2 / B^2 (shape [1])."""

    def forward(self, f_s, f_t):
        bsz = f_s.shape[0]
        f_s = ops._f32(f_s).reshape(bsz, -1)
        f_t = ops._f32(f_t.detach()).reshape(bsz, -1)
        g_s = ops.L2NormFn.apply(ops.LinearFn.apply(f_s, f_s, None))
        with torch.no_grad():
            g_t = ops.L2NormFn.apply(ops.LinearFn.apply(f_t, f_t, None))
        return _SqDiffFn.apply(g_s, g_t, 1.0 / (bsz * bsz)).reshape(1)


class feats_KL(nn.Module):
    """feats_KL.py:13-20: sum KL(softmax(f_t) || softmax(f_s)) / B."""

    def forward(self, f_s, f_t):
        return ops.KLFn.apply(f_s, f_t.detach(), 1.0, float(f_s.shape[0]))


class _LossGradFn(torch.autograd.Function):
    """loss(f_s; f_t) whose kernel returns the gradient with respect to f_s together with the value."""

    @staticmethod
    def forward(ctx, f_s, f_t, kind, w_d, w_a):
        f_s, f_t = ops._f32(f_s), ops._f32(f_t.detach())
        B, D = f_s.shape
        loss = torch.empty(1, device=f_s.device, dtype=torch.float32)
        dx = torch.empty_like(f_s)
        if kind == "pkt":
            ws = torch.empty(lib().ph_pkt_workspace_bytes(B, D), device=f_s.device, dtype=torch.uint8)
            check(lib().ph_pkt_loss_grad(ptr(f_s), ptr(f_t), ptr(loss), ptr(dx), B, D, ptr(ws), stream()), "ph_pkt_loss_grad")
        else:
            ws = torch.empty(lib().ph_rkd_workspace_bytes(B, D), device=f_s.device, dtype=torch.uint8)
            check(lib().ph_rkd_loss_grad(ptr(f_s), ptr(f_t), ptr(loss), ptr(dx), B, D, float(w_d), float(w_a), ptr(ws),
                                         stream()), "ph_rkd_loss_grad")
        ctx.save_for_backward(dx)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        dx, = ctx.saved_tensors
        return dx * g, None, None, None, None


class PKT(nn.Module):
    """PKT.py:7-46: KL between the row-normalised cosine-similarity distributions of teacher and student (0-d)."""

    def forward(self, f_s, f_t):
        return _LossGradFn.apply(f_s.reshape(f_s.shape[0], -1), f_t.reshape(f_t.shape[0], -1), "pkt", 0.0, 0.0)


class RKDLoss(nn.Module):
    """RKD.py:8-45: w_d * distance-wise + w_a * angle-wise relational losses (0-d).  B <= 128."""

    def __init__(self, w_d=25, w_a=50):
        super().__init__()
        self.w_d, self.w_a = w_d, w_a

    def forward(self, f_s, f_t):
        return _LossGradFn.apply(f_s.reshape(f_s.shape[0], -1), f_t.reshape(f_t.shape[0], -1), "rkd", self.w_d, self.w_a)
