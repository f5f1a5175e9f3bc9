"""Deterministic weight recipe shared by the golden generator, the oracle and the tests.

TEST INFRASTRUCTURE.  ResNet-18 has 11.2 M parameters (45 MB) - too large for a fixture - so
fixtures store a *recipe*: every tensor of a state_dict is drawn from a CPU generator seeded by
(seed, crc32(key)).  The key sets / shapes below are the reference's own state_dict layout
(probed from /root/reference/MICCAI-2022/{resnets,networks_new,fusion}.py
and CL_utils/CRD_loss.py:256-267).
"""
import math
import zlib
from collections import OrderedDict

import torch


def _resnet_shapes(prefix="", path_dim=128, label_dim=3):
    s = OrderedDict()
    p = prefix
    s[p + "output_range"] = (1,)
    s[p + "output_shift"] = (1,)

    def bn(name, c):
        s[name + ".weight"] = (c,)
        s[name + ".bias"] = (c,)
        s[name + ".running_mean"] = (c,)
        s[name + ".running_var"] = (c,)
        s[name + ".num_batches_tracked"] = ()

    s[p + "conv1.weight"] = (64, 3, 7, 7)
    bn(p + "bn1", 64)
    inpl = 64
    for li, planes in enumerate([64, 128, 256, 512], start=1):
        for bi in range(2):
            stride = 2 if (li > 1 and bi == 0) else 1
            b = f"{p}layer{li}.{bi}"
            s[b + ".conv1.weight"] = (planes, inpl, 3, 3)
            bn(b + ".bn1", planes)
            s[b + ".conv2.weight"] = (planes, planes, 3, 3)
            bn(b + ".bn2", planes)
            if stride != 1 or inpl != planes:
                s[b + ".downsample.0.weight"] = (planes, inpl, 1, 1)
                bn(b + ".downsample.1", planes)
            inpl = planes
    s[p + "fc_new1.0.weight"] = (path_dim, 512)
    s[p + "fc_new1.0.bias"] = (path_dim,)
    bn(p + "fc_new1.1", path_dim)
    s[p + "fc_new2.weight"] = (label_dim, path_dim)
    s[p + "fc_new2.bias"] = (label_dim,)
    return s


def student_shapes(path_dim=128, label_dim=3):
    """state_dict layout of the student / EMA ResNet (reference resnets.py:126-174)."""
    return _resnet_shapes("", path_dim, label_dim)


def teacher_shapes(input_size_omic=320, path_dim=128, omic_dim=128, mmhid=128, label_dim=3):
    """state_dict layout of PathomicNet (reference networks_new.py:267-292, fusion.py:7-34)."""
    s = OrderedDict()
    s["output_range"] = (1,)
    s["output_shift"] = (1,)
    s.update(_resnet_shapes("path_net.", path_dim, label_dim))
    s["omic_net.output_range"] = (1,)
    s["omic_net.output_shift"] = (1,)
    hidden = [64, 48, 32, omic_dim]
    d = input_size_omic
    for i, h in enumerate(hidden):
        s[f"omic_net.encoder.{i}.0.weight"] = (h, d)
        s[f"omic_net.encoder.{i}.0.bias"] = (h,)
        d = h
    s["omic_net.classifier.0.weight"] = (label_dim, omic_dim)
    s["omic_net.classifier.0.bias"] = (label_dim,)
    d1, d2 = path_dim, omic_dim
    for k in ("1", "2"):
        dk = d1 if k == "1" else d2
        s[f"fusion.linear_h{k}.0.weight"] = (dk, dk)
        s[f"fusion.linear_h{k}.0.bias"] = (dk,)
        s[f"fusion.linear_z{k}.weight"] = (dk, d1, d2)
        s[f"fusion.linear_z{k}.bias"] = (dk,)
        s[f"fusion.linear_o{k}.0.weight"] = (dk, dk)
        s[f"fusion.linear_o{k}.0.bias"] = (dk,)
    s["fusion.encoder1.0.weight"] = (mmhid, (d1 + 1) * (d2 + 1))
    s["fusion.encoder1.0.bias"] = (mmhid,)
    for nm, c in (("fusion.encoder1.1", mmhid),):
        s[nm + ".weight"] = (c,); s[nm + ".bias"] = (c,)
        s[nm + ".running_mean"] = (c,); s[nm + ".running_var"] = (c,)
        s[nm + ".num_batches_tracked"] = ()
    s["fusion.encoder2.0.weight"] = (mmhid, mmhid)   # skip=0 (options.py:143)
    s["fusion.encoder2.0.bias"] = (mmhid,)
    for nm, c in (("fusion.encoder2.1", mmhid),):
        s[nm + ".weight"] = (c,); s[nm + ".bias"] = (c,)
        s[nm + ".running_mean"] = (c,); s[nm + ".running_var"] = (c,)
        s[nm + ".num_batches_tracked"] = ()
    s["classifier.0.weight"] = (label_dim, mmhid)
    s["classifier.0.bias"] = (label_dim,)
    return s


def embed_shapes(dim_in=128, dim_out=128):
    """Embed (reference CL_utils/CRD_loss.py:256-267)."""
    return OrderedDict([("linear.weight", (dim_out, dim_in)), ("linear.bias", (dim_out,))])


def _gen(seed, key):
    g = torch.Generator(device="cpu")
    g.manual_seed((int(seed) * 1000003 + zlib.crc32(key.encode())) % (2 ** 31 - 1))
    return g


def make_state_dict(shapes, seed):
    """Draw every tensor of `shapes` deterministically.

    Scales are chosen to look like a *trained-ish* network rather than a fresh init so the golden
    vectors exercise every term: conv/linear ~ N(0, 1/fan_in)*sqrt(2), BN weight ~ U(0.5,1.5),
    BN bias ~ N(0,0.1), running_mean ~ N(0,0.1), running_var ~ U(0.5,1.5), biases ~ N(0,0.05).
    """
    sd = OrderedDict()
    for key, shape in shapes.items():
        g = _gen(seed, key)
        if key.endswith("num_batches_tracked"):
            t = torch.zeros((), dtype=torch.long)
        elif key.endswith("output_range"):
            t = torch.tensor([6.0])
        elif key.endswith("output_shift"):
            t = torch.tensor([-3.0])
        elif key.endswith("running_mean"):
            t = torch.randn(shape, generator=g) * 0.1
        elif key.endswith("running_var"):
            t = torch.rand(shape, generator=g) + 0.5
        elif len(shape) == 1 and key.endswith(".weight"):      # BN weight
            t = torch.rand(shape, generator=g) + 0.5
        elif key.endswith(".bias"):
            is_bn = (key[:-5] + ".running_mean") in shapes
            t = torch.randn(shape, generator=g) * (0.1 if is_bn else 0.05)
        else:                                                  # conv / linear / bilinear weight
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            t = torch.randn(shape, generator=g) * math.sqrt(2.0 / fan_in)
        sd[key] = t
    return sd


def adam_moments(named_shapes, scales, seed):
    """Deterministic, PLAUSIBLE Adam moments for a "mid-training" optimiser state (tests/golden/make_golden_midstate.py):
    exp_avg ~ 0.3 c N(0, 1) and exp_avg_sq = c^2 U(0.5, 1.5) per tensor, c = that tensor's gradient scale (stored in the
    fixture).  With sqrt(v) ~ c the update lr * m_hat / (sqrt(v_hat) + eps) is a SMOOTH function of the fresh gradient -
    unlike the first steps from zero moments, where it is sign(g) and amplifies rounding into 1e-2 logit differences."""
    out = OrderedDict()
    for name, shape in named_shapes:
        c = float(scales[name])
        m = torch.randn(shape, generator=_gen(seed, "m:" + name)) * (0.3 * c)
        v = (torch.rand(shape, generator=_gen(seed, "v:" + name)) + 0.5) * (c * c)
        out[name] = (m, v)
    return out
