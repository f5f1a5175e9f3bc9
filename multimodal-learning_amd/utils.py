"""The three helpers of the reference's MICCAI-2022/utils.py that are on the hot path
(init_max_weights :239-244, init_net :247-270, count_parameters :542)."""
import math

import torch
import torch.nn as nn


def init_max_weights(module):
    """utils.py:239-244 - exact-type check: nn.Bilinear is NOT re-initialised (SURVEY trap 10)."""
    for m in module.modules():
        if type(m) == nn.Linear:
            stdv = 1.0 / math.sqrt(m.weight.size(1))
            m.weight.data.normal_(0, stdv)
            m.bias.data.zero_()


def init_net(net, init_type="normal", init_gain=0.02, gpu_ids=[]):
    """utils.py:247-270.  The reference wraps in nn.DataParallel when gpu_ids is non-empty; the MI355X
    design is one process per GPU (RCCL data parallelism lives in multimodal_learning_amd.dist), so the net
    is only moved to the device.  Callers that do `.module` get the net itself through the alias below."""
    if len(gpu_ids) > 0:
        if not torch.cuda.is_available():
            raise RuntimeError("gpu_ids given but no GPU is visible")
        net.to(torch.device("cuda", torch.cuda.current_device()))
    if init_type not in ("max", "none"):
        raise NotImplementedError("init_type '%s': the shipped commands use 'max' (options.py:149)" % init_type)
    if not hasattr(net, "module"):
        net.__dict__["module"] = net   # reference call sites unwrap DataParallel with `.module` (:163,:393)
    return net


def count_parameters(model):
    return sum(p.numel() for p in model.parameters() if p.requires_grad)
