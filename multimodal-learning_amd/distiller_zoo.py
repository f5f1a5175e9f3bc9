"""Feature-distillation baselines of the reference's distiller zoo that the trainers can select with `--distill`
(part): `Similarity` ("MIA 2022/distiller_zoo/SP.py": similarity-preserving KD; also
`opt.distill == "sp"` in the MIA-2023 stage-2 trainer, train_test_path_multi_distill.py:369-370) and `feats_KL`
("MIA 2022/distiller_zoo/feats_KL.py": KL divergence between softmaxed feature vectors).  Both are compositions of
kernels of the hot path (fp32 GEMM, row L2 normalisation, squared-difference reduction, the KL kernels); the teacher
side is treated as a constant (the trainers pass it detached).  RKD / PKT and the other zoo members are not built."""
import torch
import torch.nn as nn

from . import ops
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
