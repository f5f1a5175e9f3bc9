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


# ---- regularisation (utils.py:57-198).  Same attribute walks as the reference, including what they do NOT find: the
# helpers unwrap DataParallel with `model.module` and probe sub-modules with the `__hasattr__` method that only
# PathomicNet defines (networks_new.py:356-369), so on a network without it (the ResNet student) `path` / `mm` / `omic`
# raise AttributeError exactly as the reference does.
def _l1(param_lists):
    from . import ops
    tensors = [W for ps in param_lists for W in ps]
    return ops.l1_norm_sum(tensors)


def regularize_weights(model, reg_type=None):
    """utils.py:60-68 - every parameter of the model (output_range / output_shift included, as there)."""
    return _l1([model.parameters()])


def regularize_path_weights(model, reg_type=None):
    """utils.py:71-86 - `classifier` then `linear` of the unwrapped model; a model without `.linear` (PathomicNet)
    raises AttributeError in the reference and here."""
    return _l1([model.module.classifier.parameters(), model.module.linear.parameters()])


_MM_PARTS = ("omic_net", "linear_h_path", "linear_h_omic", "linear_h_grph", "linear_z_path", "linear_z_omic", "linear_z_grph",
             "linear_o_path", "linear_o_omic", "linear_o_grph", "encoder1", "encoder2", "classifier")


def regularize_MM_weights(model, reg_type=None):
    """utils.py:88-183 - the listed direct sub-modules of the unwrapped model that exist (PathomicNet: omic_net and
    classifier; the gating / encoder layers live inside `fusion` and are not direct children)."""
    m = model.module
    return _l1([getattr(m, name).parameters() for name in _MM_PARTS if m.__hasattr__(name)])


def regularize_MM_omic(model, reg_type=None):
    """utils.py:186-198 - the genomic SNN's parameters (default --reg_type of the stage-1 command)."""
    m = model.module
    return _l1([m.omic_net.parameters()] if m.__hasattr__("omic_net") else [])


def count_parameters(model):
    return sum(p.numel() for p in model.parameters() if p.requires_grad)


class _CoxFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, theta, survtime, censor):
        from . import ops
        from ._lib import lib, check, ptr, stream
        theta = ops._f32(theta).reshape(-1).contiguous()
        t = ops._f32(survtime.to(theta.device)).reshape(-1).contiguous()
        c = ops._f32(censor.to(theta.device)).reshape(-1).contiguous()
        loss = torch.empty(1, device=theta.device, dtype=torch.float32)
        d = torch.empty_like(theta)
        check(lib().ph_cox_loss_grad(ptr(theta), ptr(t), ptr(c), ptr(loss), ptr(d), theta.shape[0], stream()), "ph_cox_loss_grad")
        ctx.save_for_backward(d)
        ctx.shape = None
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        d, = ctx.saved_tensors
        return d * g, None, None


def CoxLoss(survtime, censor, hazard_pred, device=None):
    """utils.py:361-376 (same argument order): the Cox partial-likelihood loss of the survival task, one kernel instead of
    a B x B host loop.  The survival TRAINERS are out of scope; this is the loss function alone."""
    g = _CoxFn.apply(hazard_pred.reshape(-1), survtime, censor)
    return g
