"""Drop-in for the reference's MICCAI-2022/fusion.py BilinearFusion (fusion.py:6-63): same constructor,
parameter names and forward(vec1, vec2) -> [B, mmhid]; the arithmetic runs through the C-ABI dense
kernels.  The Kronecker product o1 (x) o2 is formed by ``ph_outer`` and contracted by a split-K SGEMM.

The frozen stage-2 teacher (train_test_path_multi_distill.py:170-173) runs the fused forward-only calls; when a gradient
is required (stage-1 teacher training, row f-1) the same arithmetic runs taped through ops.*Fn.
"""
import torch
import torch.nn as nn

from . import ops
from .utils import init_max_weights


class BilinearFusion(nn.Module):
    def __init__(self, skip=1, use_bilinear=1, gate1=1, gate2=1, dim1=32, dim2=32, scale_dim1=1, scale_dim2=1,
                 mmhid=64, dropout_rate=0.25):
        super().__init__()
        self.skip, self.use_bilinear, self.gate1, self.gate2 = skip, use_bilinear, gate1, gate2
        self.relu = nn.ReLU(inplace=False)
        dim1_og, dim2_og, dim1, dim2 = dim1, dim2, dim1 // scale_dim1, dim2 // scale_dim2
        skip_dim = dim1 + dim2 + 2 if skip else 0
        self.linear_h1 = nn.Sequential(nn.Linear(dim1_og, dim1), nn.ReLU())
        self.linear_z1 = nn.Bilinear(dim1_og, dim2_og, dim1) if use_bilinear else nn.Sequential(nn.Linear(dim1_og + dim2_og, dim1))
        self.linear_o1 = nn.Sequential(nn.Linear(dim1, dim1), nn.ReLU(), nn.Dropout(p=dropout_rate))
        self.linear_h2 = nn.Sequential(nn.Linear(dim2_og, dim2), nn.ReLU())
        self.linear_z2 = nn.Bilinear(dim1_og, dim2_og, dim2) if use_bilinear else nn.Sequential(nn.Linear(dim1_og + dim2_og, dim2))
        self.linear_o2 = nn.Sequential(nn.Linear(dim2, dim2), nn.ReLU(), nn.Dropout(p=dropout_rate))
        self.post_fusion_dropout = nn.Dropout(p=dropout_rate)
        self.encoder1 = nn.Sequential(nn.Linear((dim1 + 1) * (dim2 + 1), mmhid), nn.BatchNorm1d(mmhid), nn.ReLU(),
                                      nn.Dropout(p=dropout_rate))
        self.encoder2 = nn.Sequential(nn.Linear(mmhid + skip_dim, mmhid), nn.BatchNorm1d(mmhid), nn.ReLU(),
                                      nn.Dropout(p=dropout_rate))
        self.dropout_rate = dropout_rate
        self._rng_offset = 0
        self.rng_seed = 0x5EED
        self.register_buffer("rng_step", torch.zeros(1, dtype=torch.int64), persistent=False)
        init_max_weights(self)
        if not (use_bilinear and gate1 and gate2) or skip:
            raise NotImplementedError("hot path uses use_bilinear=1, gates on, skip=0 (options.py:143-146)")

    def _drop(self, x):
        if self.training and self.dropout_rate > 0:
            ops.dropout_(x, self.dropout_rate, self.rng_seed, self._rng_offset, self.rng_step, alpha=False)
            self._rng_offset += x.numel()
        return x

    def _bn_relu(self, x, bn):
        if not self.training:
            if torch.is_grad_enabled() and x.requires_grad:
                return ops.BN1dEvalFn.apply(x, bn, True)      # a gradient flows through the eval-mode net to its inputs
            return ops.bn1d_eval(x, bn, relu=True)
        return ops.BN1dFn.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked,
                                True, True)

    def forward(self, vec1, vec2):
        if torch.is_grad_enabled() and (vec1.requires_grad or vec2.requires_grad or
                                        any(p.requires_grad for p in self.parameters())):
            return self._forward_autograd(vec1, vec2)     # stage-1 teacher training (row f-1)
        self._rng_offset = 0     # per-call-site offsets are static; the device step counter makes steps differ
        v1 = ops.eltwise(vec1, None, ops.EW_RELU)                                   # fusion.py:38-39
        v2 = ops.eltwise(vec2, None, ops.EW_RELU)
        Bn, D1 = v1.shape
        D2 = v2.shape[1]
        v12 = ops.outer(v1, v2, 0)                                                  # nn.Bilinear operand
        h1 = ops.linear_fwd(v1, self.linear_h1[0].weight, self.linear_h1[0].bias, ops.ACT_RELU)
        z1 = ops.linear_fwd(v12, self.linear_z1.weight.view(-1, D1 * D2), self.linear_z1.bias)
        o1 = self._drop(ops.linear_fwd(ops.eltwise(z1, h1, ops.EW_GATE), self.linear_o1[0].weight,
                                       self.linear_o1[0].bias, ops.ACT_RELU))
        h2 = ops.linear_fwd(v2, self.linear_h2[0].weight, self.linear_h2[0].bias, ops.ACT_RELU)
        z2 = ops.linear_fwd(v12, self.linear_z2.weight.view(-1, D1 * D2), self.linear_z2.bias)
        o2 = self._drop(ops.linear_fwd(ops.eltwise(z2, h2, ops.EW_GATE), self.linear_o2[0].weight,
                                       self.linear_o2[0].bias, ops.ACT_RELU))
        o12 = self._drop(ops.outer(o1, o2, 1))                                      # fusion.py:56-59
        out = ops.linear_fwd(o12, self.encoder1[0].weight, self.encoder1[0].bias)
        out = self._drop(self._bn_relu(out, self.encoder1[1]))
        out = ops.linear_fwd(out, self.encoder2[0].weight, self.encoder2[0].bias)
        out = self._drop(self._bn_relu(out, self.encoder2[1]))
        if self.training and self.dropout_rate > 0:
            ops.counter_inc(self.rng_step)
        return out

    # ---- taped form of the same arithmetic (same dropout call sites and offsets, so both paths draw the same masks)
    def _drop_ag(self, x):
        if self.training and self.dropout_rate > 0:
            x = ops.DropoutFn.apply(x, self.dropout_rate, self.rng_seed, self._rng_offset, self.rng_step, False)
            self._rng_offset += x.numel()
        return x

    def _forward_autograd(self, vec1, vec2):
        self._rng_offset = 0
        v1 = ops.ReluFn.apply(ops._f32(vec1))
        v2 = ops.ReluFn.apply(ops._f32(vec2))
        D1, D2 = v1.shape[1], v2.shape[1]
        v12 = ops.OuterFn.apply(v1, v2, 0)
        h1 = ops.LinearActFn.apply(v1, self.linear_h1[0].weight, self.linear_h1[0].bias, ops.ACT_RELU)
        z1 = ops.LinearFn.apply(v12, self.linear_z1.weight.view(-1, D1 * D2), self.linear_z1.bias)
        o1 = self._drop_ag(ops.LinearActFn.apply(ops.GateFn.apply(z1, h1), self.linear_o1[0].weight,
                                                 self.linear_o1[0].bias, ops.ACT_RELU))
        h2 = ops.LinearActFn.apply(v2, self.linear_h2[0].weight, self.linear_h2[0].bias, ops.ACT_RELU)
        z2 = ops.LinearFn.apply(v12, self.linear_z2.weight.view(-1, D1 * D2), self.linear_z2.bias)
        o2 = self._drop_ag(ops.LinearActFn.apply(ops.GateFn.apply(z2, h2), self.linear_o2[0].weight,
                                                 self.linear_o2[0].bias, ops.ACT_RELU))
        o12 = self._drop_ag(ops.OuterFn.apply(o1, o2, 1))
        out = ops.LinearFn.apply(o12, self.encoder1[0].weight, self.encoder1[0].bias)
        out = self._drop_ag(self._bn_relu(out, self.encoder1[1]))
        out = ops.LinearFn.apply(out, self.encoder2[0].weight, self.encoder2[0].bias)
        out = self._drop_ag(self._bn_relu(out, self.encoder2[1]))
        if self.training and self.dropout_rate > 0:
            ops.counter_inc(self.rng_step)
        return out
