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


def masks_from_input_gradients(opt, x_path_grad, x_omic_grad, sp_mask, num_superpixels=None):
    """The tail of superpixel_attention_mask (:77-101): returns (x_path_super_mask, x_omic_super_mask), both float."""
    return (superpixel_topk_mask(x_path_grad, sp_mask, opt.Path_K, num_superpixels), omic_topk_mask(x_omic_grad, opt.Omic_K))
