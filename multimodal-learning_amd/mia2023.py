"""MIA-2023 stage-2 additions to the hot path, reference
"MIA 2023/stage2_unimodal_student/{KD_loss.py, train_test_path_multi_distill.py}"."""
import torch
import torch.nn as nn

from . import ops
from ._lib import lib, check, ptr, stream


class _KLRowsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y_s, y_t, T):
        y_s, y_t = ops._f32(y_s), ops._f32(y_t)
        rows = torch.empty(y_s.shape[0], device=y_s.device, dtype=torch.float32)
        check(lib().ph_kl_rows_fwd(ptr(y_s), ptr(y_t), ptr(rows), y_s.shape[0], y_s.shape[1], T, stream()), "ph_kl_rows_fwd")
        ctx.save_for_backward(y_s, y_t)
        ctx.T = T
        return rows

    @staticmethod
    def backward(ctx, g):
        y_s, y_t = ctx.saved_tensors
        g = ops._f32(g)
        d = torch.empty_like(y_s)
        check(lib().ph_kl_rows_bwd(ptr(g), ptr(y_s), ptr(y_t), ptr(d), y_s.shape[0], y_s.shape[1], ctx.T, stream()),
              "ph_kl_rows_bwd")
        return d, None, None


class DistillKL(nn.Module):
    """MIA-2023 KD_loss.py:8-20: forward(y_s, y_t) -> (loss, sample_loss[B])."""

    def __init__(self, T):
        super().__init__()
        self.T = T
        self.batch_norm_size = None    # global batch under data parallelism

    def forward(self, y_s, y_t):
        sample_loss = _KLRowsFn.apply(y_s, y_t.detach(), float(self.T))
        return sample_loss.sum() / float(self.batch_norm_size or y_s.shape[0]), sample_loss


def assign_sample_weights(pred_s, pred_t, gt, discrep_scale, max_discrep, from_logits=False):
    """train_test_path_multi_distill.py:131-158.  The reference passes softmax probabilities; the kernel works on
    logits (log-probability margins are softmax-invariant up to the same normaliser), so probabilities are
    converted with log() - pass from_logits=True to skip that."""
    ls = pred_s if from_logits else torch.log(pred_s)
    lt = pred_t if from_logits else torch.log(pred_t)
    ls, lt = ops._f32(ls.detach()), ops._f32(lt.detach())
    out = torch.empty(ls.shape[0], device=ls.device, dtype=torch.float32)
    check(lib().ph_conf_discrepancy(ptr(ls), ptr(lt), ptr(gt.contiguous()), ptr(out), ls.shape[0], ls.shape[1],
                                    float(max_discrep), stream()), "ph_conf_discrepancy")
    return out


def GK_refine_thresh(opt, optimizer, main_loss, feat_s, loss_t_list, batch_norm_size=None, sync=None):
    """train_test_path_multi_distill.py:81-128: per-sample gradient-agreement weights.  The reference loops over
    the batch on the host with sklearn; here one wave per sample (ph_gk_rows).  Returns (scale[n], total_KD_loss).
    The weights are per sample (a cosine between rows of this replica's gradients: nothing to exchange under data
    parallelism); `batch_norm_size` is the global batch the total is averaged over, `sync` averages the reported
    mean weights over the replicas."""
    losses = [l.sum() for l in loss_t_list] + ([main_loss] if opt.CE_grads else [])
    grads = [torch.autograd.grad(l, feat_s, retain_graph=True)[0] for l in losses]
    ng = len(grads)
    G = torch.stack(grads).contiguous()                       # [ng, B, D]
    B, D = G.shape[1], G.shape[2]
    all_scale = torch.empty(B, ng, device=G.device, dtype=torch.float32)
    check(lib().ph_gk_rows(ptr(G), ng, B, D, 1 if opt.use_grads_thresh == "True" else 0, float(opt.grads_thresh),
                           ptr(all_scale), stream()), "ph_gk_rows")
    total = torch.sum(all_scale[:, :-1].transpose(0, 1) * torch.stack(list(loss_t_list))) / float(batch_norm_size or B)
    mean_scale = all_scale.mean(0)
    if sync is not None:
        mean_scale = sync.all_reduce_sum(mean_scale) / sync.world_size
    return mean_scale, total
