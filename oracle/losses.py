"""CPU restatement of the distillation losses on the hot path (TEST INFRASTRUCTURE).

Follows /root/reference/MICCAI-2022: KD_loss.py:7-17, CL_utils/CRD_loss.py:127-279,
CL_utils/memory_new.py:225-397, train_test_path_multi_distill.py:41-70.  The "semantic traps" of
SURVEY.md section 8-a are mirrored, not fixed.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

EPS = 1e-7   # CRD_loss.py:5


def nll_loss(pred, grade):
    """F.nll_loss(pred_path, grade), mean reduction (train_test_path_multi_distill.py:262)."""
    return F.nll_loss(pred, grade)


def distill_kl(y_s, y_t, T=1.0):
    """DistillKL.forward (KD_loss.py:13-17)."""
    p_s = F.log_softmax(y_s / T, dim=1)
    p_t = F.softmax(y_t / T, dim=1)
    return F.kl_div(p_s, p_t, reduction="sum") * (T ** 2) / y_s.shape[0]


def embed_forward(x, w, b):
    """Embed.forward + Normalize (CRD_loss.py:263-267,276-279): Linear then x/||x||_2 (no eps)."""
    x = F.linear(x.view(x.shape[0], -1), w, b)
    return x / x.pow(2).sum(1, keepdim=True).pow(0.5)


class CRDState:
    """Buffers of ContrastMemory_v3 (memory_new.py:244-247) + the two Embed layers."""

    def __init__(self, n_data, feat_dim=128, P=300, K=700, T=0.07, momentum=0.5, seed=0,
                 embed_s=None, embed_t=None):
        g = torch.Generator(device="cpu")
        g.manual_seed(seed)
        stdv = 1.0 / math.sqrt(feat_dim / 3)
        self.params = torch.tensor([K, T, -1, -1, momentum, P], dtype=torch.float32)
        self.memory_v1 = torch.rand(n_data, feat_dim, generator=g).mul_(2 * stdv).add_(-stdv)
        self.memory_v2 = torch.rand(n_data, feat_dim, generator=g).mul_(2 * stdv).add_(-stdv)
        self.embed_s = embed_s   # dict linear.weight / linear.bias
        self.embed_t = embed_t
        self.n_data = n_data


def contrast_memory_v3(st, v1, v2, y, idx, P2, K2, select_pos_mode="mid", mid_ranks=None,
                       select_neg_pairs="True"):
    """ContrastMemory_v3.forward (memory_new.py:249-397).

    v1 = student embedding, v2 = teacher embedding (CRD_loss.py:167).  `mid_ranks` is the host-RNG
    draw of memory_new.py:311 passed in explicitly.  Returns (out_v1, out_v2) of shape
    [B, P2+K2, 1]; mutates st.params[2:4] on the first call and both banks.
    """
    K = int(st.params[0].item()); T = st.params[1].item()
    Z_v1 = st.params[2].item(); Z_v2 = st.params[3].item()
    momentum = st.params[4].item(); P = int(st.params[5].item())
    B = v1.size(0); D = st.memory_v1.size(1); n_out = st.memory_v1.size(0)

    weight_v1 = torch.index_select(st.memory_v1, 0, idx.view(-1)).detach().view(B, K + P, D)
    out_v2 = torch.exp(torch.bmm(weight_v1, v2.view(B, D, 1)) / T)              # :270-273
    weight_v2 = torch.index_select(st.memory_v2, 0, idx.view(-1)).detach().view(B, K + P, D)
    out_v1 = torch.exp(torch.bmm(weight_v2, v1.view(B, D, 1)) / T)              # :275-278

    # names are swapped in the reference (trap 1): t_relation is the STUDENT side.  :288-292
    t_rel = torch.bmm(weight_v1 / weight_v1.norm(dim=2, keepdim=True),
                      (v1 / v1.norm(dim=1, keepdim=True)).view(B, D, 1))
    s_rel = torch.bmm(weight_v2 / weight_v2.norm(dim=2, keepdim=True),
                      (v2 / v2.norm(dim=1, keepdim=True)).view(B, D, 1))

    diff_pos = (t_rel.narrow(1, 0, P) - s_rel.narrow(1, 0, P))
    indices = torch.sort(diff_pos, dim=1, descending=True)[1]                    # :303
    if select_pos_mode == "hard":
        sel = indices[:, :P2, :].squeeze(-1).clone()                             # :308
    elif select_pos_mode in ("mid", "random", "curriculum"):
        index = torch.as_tensor(np.asarray(mid_ranks), dtype=torch.long)
        sel = indices.index_select(1, index).squeeze(-1).clone()                 # :315
    else:
        raise NotImplementedError(select_pos_mode)
    sel[:, 0] = 0                                                                # :325
    flat = (torch.arange(B).view(-1, 1) * (K + P) + sel).view(-1)
    out_v2_pos = out_v2.view(-1, 1).index_select(0, flat).view(-1, P2, 1)
    out_v1_pos = out_v1.view(-1, 1).index_select(0, flat).view(-1, P2, 1)

    if select_neg_pairs == "True":
        diff_neg = (t_rel.narrow(1, P, K) - s_rel.narrow(1, P, K))
        indn = torch.sort(diff_neg, dim=1, descending=False)[1]                  # :342
        seln = P + indn[:, :K2, :].squeeze(-1)                                   # :345
        flatn = (torch.arange(B).view(-1, 1) * (K + P) + seln).view(-1)
        out_v2_neg = out_v2.view(-1, 1).index_select(0, flatn).view(-1, K2, 1)
        out_v1_neg = out_v1.view(-1, 1).index_select(0, flatn).view(-1, K2, 1)
    else:
        seln = None
        out_v2_neg = out_v2.narrow(1, P, K)
        out_v1_neg = out_v1.narrow(1, P, K)
    out_v2 = torch.cat((out_v2_pos, out_v2_neg), 1)
    out_v1 = torch.cat((out_v1_pos, out_v1_neg), 1)

    if Z_v1 < 0:                                                                 # :368-375
        st.params[2] = out_v1.mean().detach() * n_out
        Z_v1 = st.params[2].item()
    if Z_v2 < 0:
        st.params[3] = out_v2.mean().detach() * n_out
        Z_v2 = st.params[3].item()
    out_v1 = out_v1 / Z_v1
    out_v2 = out_v2 / Z_v2

    with torch.no_grad():                                                        # :382-395
        l_pos = torch.index_select(st.memory_v1, 0, y.view(-1)) * momentum + v1 * (1 - momentum)
        st.memory_v1.index_copy_(0, y, l_pos / l_pos.pow(2).sum(1, keepdim=True).pow(0.5))
        ab_pos = torch.index_select(st.memory_v2, 0, y.view(-1)) * momentum + v2 * (1 - momentum)
        st.memory_v2.index_copy_(0, y, ab_pos / ab_pos.pow(2).sum(1, keepdim=True).pow(0.5))
    aux = {"sel_pos": sel, "sel_neg": seln, "diff_pos": diff_pos.squeeze(-1)}
    return out_v1, out_v2, aux


def contrast_loss_v2(x, P, n_data, sample_KD="False"):
    """ContrastLoss_v2.forward (CRD_loss.py:221-252): the 0-d batch mean of sample_KD == "False" (:240-244) or the [B]
    per-sample losses of sample_KD == "True" (:246-250)."""
    bsz = x.shape[0]
    N = x.size(1) - P
    m = N
    Pn = 1 / float(n_data)
    P_pos = x.narrow(1, 0, P)
    log_D1 = torch.div(P_pos, P_pos.add(m * Pn + EPS)).log()
    P_neg = x.narrow(1, P, N)
    log_D0 = torch.div(torch.full_like(P_neg, m * Pn), P_neg.add(m * Pn + EPS)).log()
    if sample_KD == "True":
        return -(log_D1.squeeze(-1) + log_D0.repeat(1, 1, P).sum(1)).sum(1) / P
    loss = -((log_D1.squeeze(-1).sum(0) + log_D0.reshape(-1, 1).repeat(1, P).sum(0)) / bsz).sum(0) / P
    return loss


def crd_loss(st, f_s, f_t, idx, contrast_idx, P2, K2, select_pos_mode="mid", mid_ranks=None,
             return_aux=False, sample_KD="False"):
    """CRDLoss.forward (CRD_loss.py:153-175); sample_KD == "True": the [B] per-sample losses (:148-149 -> :246-250)."""
    v_s = embed_forward(f_s, st.embed_s["linear.weight"], st.embed_s["linear.bias"])
    v_t = embed_forward(f_t, st.embed_t["linear.weight"], st.embed_t["linear.bias"])
    out_s, out_t, aux = contrast_memory_v3(st, v_s, v_t, idx, contrast_idx, P2, K2,
                                           select_pos_mode, mid_ranks)
    loss = contrast_loss_v2(out_s, P2, st.n_data, sample_KD) + contrast_loss_v2(out_t, P2, st.n_data, sample_KD)
    if return_aux:
        aux.update(v_s=v_s, v_t=v_t, out_s=out_s, out_t=out_t)
        return loss, aux
    return loss


def aekd_loss(main_loss, feat_s, loss_t_list, ce_grads=True):
    """AEKD_loss / GK-Refine (train_test_path_multi_distill.py:41-70).

    The reference runs one full backward per loss and keeps only d loss / d feat_s (hook :46-47);
    the value is identical to torch.autograd.grad stopped at feat_s.
    """
    losses = list(loss_t_list) + ([main_loss] if ce_grads else [])
    grads = [torch.autograd.grad(l, feat_s, retain_graph=True)[0].detach().clone() for l in losses]
    all_grads = torch.stack(grads).view(len(grads), -1)
    norm = torch.norm(all_grads, p=2, dim=1, keepdim=True)
    rel = torch.matmul(all_grads, all_grads.T) * len(loss_t_list) / torch.matmul(norm, norm.T)   # :61
    scale = rel.sum(dim=1)
    total = torch.dot(scale[:-1], torch.stack(list(loss_t_list)))                                # :68
    return scale, total
