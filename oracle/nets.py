"""Functional CPU restatement of the networks on the hot path (TEST INFRASTRUCTURE).

All functions take a reference-layout ``state_dict`` (OIHW fp32 weights, reference key names) and
run in train mode (batch-statistic BatchNorm) exactly like the reference's hot loop does for the
student, the EMA net **and** the frozen teacher (train_test_path_multi_distill.py:231-232).

``Rounding`` emulates the product's *perf mode* operand rounding so that the bf16 HIP path can be
compared like-for-like (SURVEY.md section 7 "Hard parts"): conv operands and stored activations
are rounded to bf16, accumulation / BN statistics / heads stay fp32.
"""
from contextlib import contextmanager

import torch
import torch.nn.functional as F


class Rounding:
    """mode 'fp32' = the reference's arithmetic; 'bf16' = perf-mode emulation.
    `training` False = module.eval(): BatchNorm normalises with the running statistics (reference test())."""
    mode = "fp32"
    training = True

    @classmethod
    def q(cls, x):
        if cls.mode == "bf16":
            return x.to(torch.bfloat16).to(torch.float32)
        return x

    @classmethod
    @contextmanager
    def use(cls, mode):
        old = cls.mode
        cls.mode = mode
        try:
            yield
        finally:
            cls.mode = old


class _QuantSTE(torch.autograd.Function):
    """bf16 rounding with a straight-through gradient (activation *storage* rounding)."""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return g


def _store(x):
    return _QuantSTE.apply(x) if Rounding.mode == "bf16" else x


def _bn_train(x, sd, name, update_running=True, eps=1e-5, momentum=0.1):
    """nn.BatchNorm{1,2}d in training mode (resnets.py:148, fusion.py:29-31) = F.batch_norm with
    training=True (the op the reference's modules call; its CPU backward accumulates in double).

    In bf16 emulation the normalised tensor is the bf16-stored copy of the conv output (the HIP
    conv epilogue takes the statistics from the unrounded accumulators; the difference is the mean
    of ~1e4+ rounding errors and is far below the comparison tolerance).
    """
    xs = _store(x)
    if not Rounding.training:
        return F.batch_norm(xs, sd[name + ".running_mean"], sd[name + ".running_var"], sd[name + ".weight"],
                            sd[name + ".bias"], False, momentum, eps)
    if update_running:
        rm, rv = sd[name + ".running_mean"], sd[name + ".running_var"]
        with torch.no_grad():
            sd[name + ".num_batches_tracked"] += 1
    else:
        rm = rv = None
    return F.batch_norm(xs, rm, rv, sd[name + ".weight"], sd[name + ".bias"], True, momentum, eps)


def _conv(x, w, stride, pad):
    return F.conv2d(Rounding.q(x), Rounding.q(w), None, stride, pad)


def _basic_block(x, sd, p, stride, has_ds, upd):
    """BasicBlock.forward (resnets.py:58-74)."""
    out = _conv(x, sd[p + ".conv1.weight"], stride, 1)
    out = _store(F.relu(_bn_train(out, sd, p + ".bn1", upd)))
    out = _conv(out, sd[p + ".conv2.weight"], 1, 1)
    out = _bn_train(out, sd, p + ".bn2", upd)
    if has_ds:
        idn = _conv(x, sd[p + ".downsample.0.weight"], stride, 0)
        idn = _bn_train(idn, sd, p + ".downsample.1", upd)
    else:
        idn = x
    return _store(F.relu(out + idn))


def resnet_trunk(x, sd, prefix="", update_running=True, return_all=False):
    """ResNet._forward_impl trunk (resnets.py:217-236): returns pooled (f3[B,256], f4[B,512])."""
    p = prefix
    inter = {}
    x = _store(x)
    y = _conv(x, sd[p + "conv1.weight"], 2, 3)
    y = _store(F.relu(_bn_train(y, sd, p + "bn1", update_running)))
    y = F.max_pool2d(y, 3, 2, 1)
    inter["pool"] = y
    inpl = 64
    f3 = None
    for li, planes in enumerate([64, 128, 256, 512], start=1):
        for bi in range(2):
            stride = 2 if (li > 1 and bi == 0) else 1
            has_ds = stride != 1 or inpl != planes
            y = _basic_block(y, sd, f"{p}layer{li}.{bi}", stride, has_ds, update_running)
            inpl = planes
        inter[f"layer{li}"] = y
        if li == 3:
            f3 = y.mean(dim=(2, 3))
    f4 = y.mean(dim=(2, 3))
    if return_all:
        return f3, f4, inter
    return f3, f4


def resnet_head(f4, sd, prefix="", update_running=True):
    """fc_new1 / fc_new2 / LogSoftmax (resnets.py:165-169,239-253; networks_new.py:139-140)."""
    p = prefix
    h = F.linear(f4, sd[p + "fc_new1.0.weight"], sd[p + "fc_new1.0.bias"])
    feat = F.relu(_bn_train(h, sd, p + "fc_new1.1", update_running))
    hazard = F.linear(feat, sd[p + "fc_new2.weight"], sd[p + "fc_new2.bias"])
    pred = F.log_softmax(hazard, dim=1)
    return feat, hazard, pred


def resnet_forward(x_path, sd, prefix="", update_running=True):
    """ResNet.forward (resnets.py:267-272) -> (feat_f3, features, hazard, pred, None)."""
    f3, f4 = resnet_trunk(x_path, sd, prefix, update_running)
    feat, hazard, pred = resnet_head(f4, sd, prefix, update_running)
    return f3, feat, hazard, pred, None


def _alpha_dropout(x, p, gen):
    """nn.AlphaDropout in training mode (networks_new.py:193); identity for p == 0."""
    if p <= 0:
        return x
    alpha_p = -1.7580993408473766
    a = ((1 - p) * (1 + p * alpha_p ** 2)) ** -0.5
    b = -a * alpha_p * p
    keep = (torch.rand(x.shape, generator=gen) >= p).to(x.dtype)
    return a * (x * keep + alpha_p * (1 - keep)) + b


def _dropout(x, p, gen):
    if p <= 0:
        return x
    keep = (torch.rand(x.shape, generator=gen) >= p).to(x.dtype)
    return x * keep / (1 - p)


def maxnet_forward(x_omic, sd, prefix="", dropout_rate=0.0, gen=None):
    """MaxNet.forward (networks_new.py:223-251): 4x(Linear-ELU-AlphaDropout), ReLU, Linear, LSM."""
    p = prefix
    h = x_omic
    for i in range(4):
        h = F.elu(F.linear(h, sd[f"{p}encoder.{i}.0.weight"], sd[f"{p}encoder.{i}.0.bias"]))
        h = _alpha_dropout(h, dropout_rate, gen)
    feat = F.relu(h)
    out = F.linear(feat, sd[p + "classifier.0.weight"], sd[p + "classifier.0.bias"])
    pred = F.log_softmax(out, dim=1)
    return feat, out, pred, None


def bilinear_fusion_forward(vec1, vec2, sd, prefix="fusion.", dropout_rate=0.0, gen=None,
                            update_running=True):
    """BilinearFusion.forward (fusion.py:36-63), skip=0, use_bilinear=1, both gates on."""
    p = prefix
    v1 = F.relu(vec1)
    v2 = F.relu(vec2)
    h1 = F.relu(F.linear(v1, sd[p + "linear_h1.0.weight"], sd[p + "linear_h1.0.bias"]))
    z1 = F.bilinear(v1, v2, sd[p + "linear_z1.weight"], sd[p + "linear_z1.bias"])
    o1 = _dropout(F.relu(F.linear(torch.sigmoid(z1) * h1, sd[p + "linear_o1.0.weight"],
                                  sd[p + "linear_o1.0.bias"])), dropout_rate, gen)
    h2 = F.relu(F.linear(v2, sd[p + "linear_h2.0.weight"], sd[p + "linear_h2.0.bias"]))
    z2 = F.bilinear(v1, v2, sd[p + "linear_z2.weight"], sd[p + "linear_z2.bias"])
    o2 = _dropout(F.relu(F.linear(torch.sigmoid(z2) * h2, sd[p + "linear_o2.0.weight"],
                                  sd[p + "linear_o2.0.bias"])), dropout_rate, gen)
    one = torch.ones(o1.shape[0], 1, dtype=o1.dtype)
    o1 = torch.cat((o1, one), 1)
    o2 = torch.cat((o2, one), 1)
    o12 = torch.bmm(o1.unsqueeze(2), o2.unsqueeze(1)).flatten(start_dim=1)
    out = _dropout(o12, dropout_rate, gen)
    out = F.linear(out, sd[p + "encoder1.0.weight"], sd[p + "encoder1.0.bias"])
    out = _dropout(F.relu(_bn_train(out, sd, p + "encoder1.1", update_running)), dropout_rate, gen)
    out = F.linear(out, sd[p + "encoder2.0.weight"], sd[p + "encoder2.0.bias"])
    out = _dropout(F.relu(_bn_train(out, sd, p + "encoder2.1", update_running)), dropout_rate, gen)
    return out


def pathomic_forward(x_path, x_omic, sd, dropout_rate=0.0, gen=None, update_running=True, cut_fuse_grad=True):
    """PathomicNet.forward (networks_new.py:294-353) -> the reference's 11-tuple.  cut_fuse_grad (:302-306): the fusion
    branch sees detached unimodal features (the shipped stage-2 flag); False lets the fused loss train both encoders."""
    f3, path_vec, h_path, pred_path, _ = resnet_forward(x_path, sd, "path_net.", update_running)
    omic_vec, h_omic, pred_omic, _ = maxnet_forward(x_omic, sd, "omic_net.", dropout_rate, gen)
    feats = bilinear_fusion_forward(path_vec.detach() if cut_fuse_grad else path_vec,
                                    omic_vec.detach() if cut_fuse_grad else omic_vec, sd, "fusion.",
                                    dropout_rate, gen, update_running)
    hazard = F.linear(feats, sd["classifier.0.weight"], sd["classifier.0.bias"])
    pred = F.log_softmax(hazard, dim=1)
    return (feats, path_vec, omic_vec, f3, [h_path, h_omic, hazard], pred, pred_path, pred_omic,
            None, None, None)
