"""Superpixel attention masks of the MIA-2023 stage-1 trainer, reference
"MIA 2023/stage1_multi_modal_teacher/train_test_MT_SP_Masking.py":42-102 (`superpixel_attention_mask`).

Built here: everything after the input gradients exist (:77-98) - the per-superpixel aggregation of the image gradient
(the reference moves a one-hot [B, N, H*W] tensor to the HOST and runs the bmm there), the top-`Path_K` superpixel mask
and the top-`Omic_K` omic mask - as two kernels (csrc/superpixel.hip).  NOT built: the gradients themselves (:45-75
need an eval-mode backward down to the image, i.e. a stem dgrad kernel, which the distillation hot path never needs)."""
import torch

from . import ops
from ._lib import lib, check, ptr, stream, require_cuda


def superpixel_topk_mask(x_path_grad, sp_mask, path_k, num_superpixels=None, return_mean=False):
    """x_path_grad [B, C, H, W] float, sp_mask [B, H, W] integer labels -> [B, H, W] float mask of the `path_k`
    superpixels with the largest mean gradient (:77-93).  `num_superpixels` (labels are < it) avoids the device->host
    read of sp_mask.max() that F.one_hot does in the reference."""
    g = ops._f32(require_cuda(x_path_grad, "x_path_grad").detach()).contiguous()
    sp = sp_mask.to(g.device).long().contiguous()
    B, C, H, W = g.shape
    if tuple(sp.shape) != (B, H, W):
        raise RuntimeError("sp_mask must be [B, H, W]")
    N = int(num_superpixels) if num_superpixels is not None else int(sp.max().item()) + 1
    mask = torch.empty(B, H, W, device=g.device, dtype=torch.float32)
    mean = torch.empty(B, N, device=g.device, dtype=torch.float32) if return_mean else None
    check(lib().ph_superpixel_mask(ptr(g), ptr(sp), ptr(mask), ptr(mean), B, C, H, W, N, int(path_k), stream()),
          "ph_superpixel_mask")
    return (mask, mean) if return_mean else mask


def omic_topk_mask(x_omic_grad, omic_k):
    """x_omic_grad [B, D] -> float mask of the entries >= the omic_k-th largest of their row (:96)."""
    g = ops._f32(require_cuda(x_omic_grad, "x_omic_grad").detach()).contiguous()
    B, D = g.shape
    mask = torch.empty_like(g)
    check(lib().ph_topk_threshold_mask(ptr(g), ptr(mask), B, D, int(omic_k), stream()), "ph_topk_threshold_mask")
    return mask


def apply_mask(x, mask):
    """x * (1 - mask) with the mask broadcast over the channel axis (:201-202): x [B, C, H, W] with mask [B, H, W], or
    x [B, D] with mask [B, D]."""
    x = ops._f32(require_cuda(x, "x")).contiguous()
    mask = ops._f32(mask.to(x.device)).contiguous()
    B = x.shape[0]
    C = x.shape[1] if x.dim() == 4 else 1
    P = mask[0].numel()
    out = torch.empty_like(x)
    check(lib().ph_apply_mask(ptr(x), ptr(mask), ptr(out), B, C, P, stream()), "ph_apply_mask")
    return out


def masks_from_input_gradients(opt, x_path_grad, x_omic_grad, sp_mask, num_superpixels=None):
    """The tail of superpixel_attention_mask (:77-101): returns (x_path_super_mask, x_omic_super_mask), both float."""
    return (superpixel_topk_mask(x_path_grad, sp_mask, opt.Path_K, num_superpixels), omic_topk_mask(x_omic_grad, opt.Omic_K))


def superpixel_attention_mask(opt, optimizer, model, x_path, x_grph, x_omic, sp_mask, grade, device, num_superpixels=None):
    """The reference's function, same signature (:42-102): switch `model` to eval mode, take the gradient of the fused
    branch's NLL with respect to the image and the omic vector (eval-mode backward of the trunk down to the image:
    `ph_resnet_backward_input`), build the two masks, switch back to train mode.  Unlike the reference this does not
    leave the eval-mode cost's parameter gradients in `.grad` (the trainer zeroes them before its own backward)."""
    dev = torch.device(device)
    x_path_adv = ops._f32(x_path.detach().to(dev)).clone().requires_grad_(True)            # :46-49
    x_omic_adv = ops._f32(x_omic.detach().to(dev)).clone().requires_grad_(True)
    grade = grade.to(dev)
    model.eval()                                                                           # :62
    try:
        out = model(x_path=x_path_adv, x_grph=x_grph, x_omic=x_omic_adv)                  # :63
        pred = out[5]
        if not pred.requires_grad:
            raise RuntimeError("the fused prediction does not depend on the inputs (cut_fuse_grad?): the reference's "
                               "gradient hooks would never fire either")
        cost = ops.NLLFn.apply(pred, grade, float(pred.shape[0]))                          # :66
        x_path_grad, x_omic_grad = torch.autograd.grad(cost, [x_path_adv, x_omic_adv])     # :67-75
        masks = masks_from_input_gradients(opt, x_path_grad, x_omic_grad, sp_mask, num_superpixels)
    finally:
        model.train()                                                                      # :99
    return masks
