"""ctypes binding of libpathomic_hip.so (include/pathomic_hip.h).

The product path has NO fallback: if the shared library is missing or a call fails, a RuntimeError is
raised (never a silent eager-PyTorch path).
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# PH_LIB_VARIANT=<tag>: load libpathomic_hip<tag>.so instead (same-box A/B of two builds, profiles/scripts/ab_lib.sh)
LIB_PATH = os.path.join(_HERE, "libpathomic_hip%s.so" % os.environ.get("PH_LIB_VARIANT", ""))
CSRC = os.path.join(_HERE, "csrc")

vp, i32, i64, f32, f64, sz, u64, lng = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_double, C.c_size_t, C.c_uint64, C.c_long

# name -> (restype, argtypes); mirrors include/pathomic_hip.h one to one
SIGNATURES = {
    "ph_abi_version": (i32, []),
    "ph_resnet_plan_create": (vp, [i32, i32, i32, i32]),
    "ph_resnet_plan_destroy": (None, [vp]),
    "ph_resnet_workspace_bytes": (sz, [vp]),
    "ph_resnet_packed_bytes": (sz, [vp]),
    "ph_resnet_num_units": (i32, [vp]),
    "ph_resnet_unit_shape": (i32, [vp, i32, vp]),
    "ph_resnet_pack_weights": (i32, [vp, vp, vp, vp]),
    "ph_resnet_plan_set_backward_prec": (i32, [vp, i32]),
    "ph_resnet_plan_set_backward_overlap": (i32, [vp, i32]),
    "ph_pack_input": (i32, [vp, vp, i32, i32, i32, i32, vp]),
    "ph_resnet_forward": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, vp]),
    "ph_resnet_backward": (i32, [vp, vp, vp, vp, vp, vp, vp, vp]),
    "ph_bn1d_eval_bwd": (i32, [vp, vp, vp, vp, vp, i32, i32, f32, i32, vp]),
    "ph_resnet_backward_input": (i32, [vp, vp, vp, vp, vp, vp, vp, vp]),
    "ph_stem_dgrad": (i32, [vp, vp, vp, i32, i32, i32, i32, vp]),
    "ph_resnet_backward_part": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, vp]),
    "ph_resnet_backward_debug": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, vp]),
    "ph_resnet_tensor_info": (i32, [vp, i32, i32, vp, vp]),
    "ph_sgemm": (i32, [vp, vp, vp, vp, i32, i32, i32, lng, lng, lng, lng, lng, i32, i32, vp]),
    "ph_sgemm_splitk": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, lng, lng, lng, lng, lng, i32, vp]),
    "ph_bn1d_fwd": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, f32, f32, i32, vp]),
    "ph_bn1d_eval": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, f32, i32, vp]),
    "ph_bn1d_bwd": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp]),
    "ph_log_softmax": (i32, [vp, vp, i32, i32, vp]),
    "ph_log_softmax_bwd": (i32, [vp, vp, vp, i32, i32, vp]),
    "ph_nll_fwd": (i32, [vp, vp, vp, i32, i32, f32, vp]),
    "ph_nll_bwd": (i32, [vp, vp, vp, i32, i32, f32, vp]),
    "ph_kl_fwd": (i32, [vp, vp, vp, i32, i32, f32, f32, vp]),
    "ph_kl_bwd": (i32, [vp, vp, vp, vp, i32, i32, f32, f32, vp]),
    "ph_l2norm_fwd": (i32, [vp, vp, vp, i32, i32, vp]),
    "ph_l2norm_bwd": (i32, [vp, vp, vp, vp, i32, i32, vp]),
    "ph_eltwise": (i32, [vp, vp, vp, sz, i32, vp]),
    "ph_outer": (i32, [vp, vp, vp, i32, i32, i32, i32, vp]),
    "ph_dropout": (i32, [vp, sz, f32, u64, u64, i32, vp]),
    "ph_dropout_dev": (i32, [vp, sz, f32, u64, u64, vp, i32, vp]),
    "ph_counter_inc": (i32, [vp, vp]),
    "ph_sum": (i32, [vp, vp, i32, f32, vp]),
    "ph_dropout_bwd_dev": (i32, [vp, sz, f32, u64, u64, vp, i32, vp]),
    "ph_dropout_dev_to": (i32, [vp, vp, sz, f32, u64, u64, vp, i32, vp]),
    "ph_dropout_bwd_dev_to": (i32, [vp, vp, sz, f32, u64, u64, vp, i32, vp]),
    "ph_gate_bwd": (i32, [vp, vp, vp, vp, vp, sz, vp]),
    "ph_outer_bwd": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "ph_contrast_sampler": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, u64, vp, vp, vp]),
    "ph_row_invnorm_scale": (i32, [vp, vp, vp, i32, i32, f32, vp]),
    "ph_row_scale": (i32, [vp, vp, vp, i32, i32, vp]),
    "ph_sqdiff_sum": (i32, [vp, vp, vp, sz, f32, vp]),
    "ph_scaled_diff": (i32, [vp, vp, vp, f32, vp, sz, vp]),
    "ph_cox_loss_grad": (i32, [vp, vp, vp, vp, vp, i32, vp]),
    "ph_pkt_workspace_bytes": (sz, [i32, i32]),
    "ph_pkt_loss_grad": (i32, [vp, vp, vp, vp, i32, i32, vp, vp]),
    "ph_rkd_workspace_bytes": (sz, [i32, i32]),
    "ph_rkd_loss_grad": (i32, [vp, vp, vp, vp, i32, i32, f32, f32, vp, vp]),
    "ph_superpixel_mask": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "ph_shuffle_indices": (i32, [vp, i32, i32, u64, vp, vp]),
    "ph_alias_uniform_draw": (i32, [vp, vp, i32, i32, i32, u64, vp, vp]),
    "ph_augment_params": (i32, [vp, i32, u64, vp, i32, i32, i32, f32, f32, f32, f32, vp]),
    "ph_augment_apply": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "ph_apply_mask": (i32, [vp, vp, vp, i32, i32, sz, vp]),
    "ph_topk_threshold_mask": (i32, [vp, vp, i32, i32, i32, vp]),
    "ph_maxnorm_mix": (i32, [vp, vp, vp, sz, f32, f32, vp]),
    "ph_tsvd_workspace_bytes": (sz, [i32, i32]),
    "ph_tsvd_update_aux": (i32, [vp, vp, vp, i32, i32, f32, vp, vp]),
    "ph_tsvd_update_aux_dev": (i32, [vp, vp, vp, i32, i32, vp, vp, vp]),
    "ph_crd_score": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, vp]),
    "ph_crd_select": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    "ph_crd_bank_topk_workspace_bytes": (sz, [i32, i32]),
    "ph_crd_bank_topk": (i32, [vp, vp, vp, vp, i32, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp]),
    "ph_kl_rows_fwd": (i32, [vp, vp, vp, i32, i32, f32, vp]),
    "ph_kl_rows_bwd": (i32, [vp, vp, vp, vp, i32, i32, f32, vp]),
    "ph_conf_discrepancy": (i32, [vp, vp, vp, vp, i32, i32, f32, vp]),
    "ph_gk_rows": (i32, [vp, i32, i32, i32, i32, f32, vp, vp]),
    "ph_crd_zsum": (i32, [vp, vp, vp, i32, vp]),
    "ph_crd_setz": (i32, [vp, vp, f32, f32, vp]),
    "ph_crd_loss_grad_workspace_bytes": (sz, [i32]),
    "ph_crd_loss_grad": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, f32, f32, vp, vp]),
    "ph_crd_neg_hist": (i32, [vp, lng, i32, i32, i32, i32, vp, vp]),
    "ph_crd_scan_neg_workspace_bytes": (sz, [i32, i32]),
    "ph_crd_scan_neg": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, i32, vp]),
    "ph_crd_loss_grad_pos": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, f32, vp]),
    "ph_crd_update": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, vp]),
    "ph_crd_outputs": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "ph_crd_outputs_bwd": (i32, [vp, vp, vp, vp, vp, vp, f32, vp, vp, i32, i32, i32, vp]),
    "ph_contrast_loss_v2": (i32, [vp, vp, vp, i32, i32, i32, f32, vp]),
    "ph_crd_class_centers_workspace_bytes": (sz, [i32, i32]),
    "ph_crd_class_centers": (i32, [vp, vp, vp, i32, i32, i32, i32, vp, vp]),
    "ph_gram": (i32, [vp, vp, i32, i32, vp]),
    "ph_gk_scale": (i32, [vp, vp, i32, i32, f32, vp, vp, vp]),
    "ph_gk_scale_momentum": (i32, [vp, i32, i32, f32, f32, vp, vp, vp]),
    "ph_logit_losses": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, i32, f32, f32, vp]),
    "ph_gk_finish": (i32, [vp, vp, vp, vp, vp, f32, vp, vp, vp, vp, vp, vp]),
    "ph_gk_finish_momentum": (i32, [vp, vp, f32, f32, vp, f32, f32, i32, f32, f32, vp, vp, vp, vp, vp, vp, vp]),
    "ph_adam_ema_step": (i32, [vp, vp, vp, vp, vp, sz, f64, f64, f64, f64, f64, i32, f64, vp]),
    "ph_adam_ema_step_dev": (i32, [vp, vp, vp, vp, vp, sz, f64, f64, f64, f64, vp, vp]),
    "ph_adagrad_ema_step_dev": (i32, [vp, vp, vp, vp, sz, f64, f64, vp, vp]),
    "ph_ema_update": (i32, [vp, vp, sz, f32, vp]),
    "ph_l1_sum": (i32, [vp, sz, vp, vp, i32, vp]),
    "ph_l1_sign_axpy": (i32, [vp, vp, sz, vp, f32, vp]),
    "ph_prof_enable": (i32, [i32]),
    "ph_prof_reset": (i32, []),
    "ph_prof_summary": (i32, [vp, i32]),
    "ph_prof_summary4": (i32, [vp, i32]),
    "ph_prof_stamp": (i32, [vp, vp]),
    "ph_conv2d_workspace_bytes": (sz, [i32, i32, i32, i32, i32, i32, i32, i32]),
    "ph_hp_pack": (i32, [vp, vp, sz, f32, vp]),
    "ph_hp_unpack": (i32, [vp, vp, sz, vp]),
    "ph_conv2d_fwd": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp]),
    "ph_conv2d_dgrad": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp]),
    "ph_conv2d_dgrad_res": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp]),
    "ph_conv2d_fwd_fused_in": (i32, [vp] * 7 + [i32] * 5 + [vp, vp]),
    "ph_conv2d_dgrad_bnstat": (i32, [vp] * 13 + [i32] * 5 + [vp, vp]),
    "ph_conv2d_wgrad": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp]),
}

_lib = None


def build(force=False):
    """Compile every HIP source for gfx950 into libpathomic_hip.so (hipcc cross-compiles without a GPU)."""
    if force:
        subprocess.run(["make", "-C", CSRC, "clean"], check=True, capture_output=True)
    r = subprocess.run(["make", "-C", CSRC, "-j8"], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("building libpathomic_hip.so failed:\n" + r.stdout[-4000:] + r.stderr[-4000:])
    return LIB_PATH


def lib():
    """The loaded library; raises RuntimeError (loudly) when it is absent - there is no CPU fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: the HIP extension is required (no CPU/eager fallback). "
                "Build it with `python -c 'import __graft_entry__ as g; g.build()'` or `make -C "
                f"{CSRC}`.")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)       # AttributeError if the .so lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"libpathomic_hip: {what} failed with code {rc}")


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else t.data_ptr()


def stream():
    import torch
    return torch.cuda.current_stream().cuda_stream


def require_cuda(t, name="tensor"):
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on the GPU: the HIP path has no CPU fallback")
    return t
