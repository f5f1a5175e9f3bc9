"""CPU restatement of the MIA-2022 / MIA-2023 stage-2 variants of the hot path (TEST INFRASTRUCTURE).

MIA-2022:
  * momentum_aekd_loss  <- "MIA 2022/train_test_path_multi_distill_v2.py":89-132
  * crd_v3_loss         <- "MIA 2022/CL_utils/CRD_criterion_v3.py":25-81 (ContrastMemory), :168-189 (CRDLoss),
                           :192-224 (ContrastLoss with per-sample weights)
MIA-2023:
  * distill_kl_per_sample  <- "MIA 2023/stage2_unimodal_student/KD_loss.py":14-20
  * assign_sample_weights  <- ".../train_test_path_multi_distill.py":131-158
  * gk_refine_thresh       <- ".../train_test_path_multi_distill.py":81-128
  * crd_v10_loss           <- ".../CL_utils/CRD_criterion_v10.py":45-176 (pos_extra="neighbors"), :241-314
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from .losses import embed_forward, EPS


class CRDv3State:
    """ContrastMemory of CRD_criterion_v3.py:8-23: params = [K, T, Z_v1, Z_v2, momentum] (no P)."""

    def __init__(self, n_data, feat_dim=128, K=1024, T=0.07, momentum=0.5, seed=0, embed_s=None, embed_t=None):
        g = torch.Generator(device="cpu")
        g.manual_seed(seed)
        stdv = 1.0 / math.sqrt(feat_dim / 3)
        self.params = torch.tensor([K, T, -1, -1, momentum], dtype=torch.float32)
        self.memory_v1 = torch.rand(n_data, feat_dim, generator=g).mul_(2 * stdv).add_(-stdv)
        self.memory_v2 = torch.rand(n_data, feat_dim, generator=g).mul_(2 * stdv).add_(-stdv)
        self.embed_s, self.embed_t, self.n_data = embed_s, embed_t, n_data


def _bank_update(st, v1, v2, y):
    momentum = st.params[4].item()
    with torch.no_grad():
        l_pos = torch.index_select(st.memory_v1, 0, y.view(-1)) * momentum + v1 * (1 - momentum)
        st.memory_v1.index_copy_(0, y, l_pos / l_pos.pow(2).sum(1, keepdim=True).pow(0.5))
        ab_pos = torch.index_select(st.memory_v2, 0, y.view(-1)) * momentum + v2 * (1 - momentum)
        st.memory_v2.index_copy_(0, y, ab_pos / ab_pos.pow(2).sum(1, keepdim=True).pow(0.5))


def _contrast_loss_weighted(w, x, n_data):
    """ContrastLoss.forward(sample_weights, x) of CRD_criterion_v3.py:200-224 (slot 0 = the positive)."""
    bsz = x.shape[0]
    m = x.size(1) - 1
    Pn = 1 / float(n_data)
    P_pos = x.select(1, 0)
    log_D1 = torch.div(P_pos, P_pos.add(m * Pn + EPS)).log()
    P_neg = x.narrow(1, 1, m)
    log_D0 = torch.div(torch.full_like(P_neg, m * Pn), P_neg.add(m * Pn + EPS)).log()
    sample_loss = -(log_D1 + log_D0.sum(1))
    return (w * sample_loss).sum(0) / bsz


def crd_v3_loss(st, sample_weights, f_s, f_t, y, idx):
    """CRDLoss.forward of CRD_criterion_v3.py:168-189 -> tensor of shape [1] (as the reference returns)."""
    v1 = embed_forward(f_s, st.embed_s["linear.weight"], st.embed_s["linear.bias"])
    v2 = embed_forward(f_t, st.embed_t["linear.weight"], st.embed_t["linear.bias"])
    K = int(st.params[0].item()); T = st.params[1].item()
    B, D = v1.shape
    n_out = st.memory_v1.size(0)
    w1 = torch.index_select(st.memory_v1, 0, idx.view(-1)).detach().view(B, K + 1, D)
    out_v2 = torch.exp(torch.bmm(w1, v2.view(B, D, 1)) / T)
    w2 = torch.index_select(st.memory_v2, 0, idx.view(-1)).detach().view(B, K + 1, D)
    out_v1 = torch.exp(torch.bmm(w2, v1.view(B, D, 1)) / T)
    if st.params[2].item() < 0:
        st.params[2] = out_v1.mean().detach() * n_out
    if st.params[3].item() < 0:
        st.params[3] = out_v2.mean().detach() * n_out
    out_v1 = out_v1 / st.params[2].item()
    out_v2 = out_v2 / st.params[3].item()
    _bank_update(st, v1, v2, y)
    return _contrast_loss_weighted(sample_weights, out_v1, st.n_data) + \
        _contrast_loss_weighted(sample_weights, out_v2, st.n_data)


# ------------------------------------------------------------------------------------------------ stage-1 terms
def orth_loss(x1, x2):
    """OrthLoss.forward (MICCAI-2022/CL_utils/orthogonal_loss.py:18-32): rows divided by their DETACHED L2 norm + 1e-6,
    mean square of the D x D cross-correlation."""
    b = x1.size(0)
    x1 = x1.view(b, -1); x2 = x2.view(b, -1)
    a = x1 / (torch.norm(x1, p=2, dim=1, keepdim=True).detach() + 1e-6)
    c = x2 / (torch.norm(x2, p=2, dim=1, keepdim=True).detach() + 1e-6)
    return torch.mean((a.t().mm(c)).pow(2))


def crd_v0_loss(st, f_s, f_t, y, idx):
    """CRDLoss.forward of MICCAI-2022/CL_utils/CRD_criterion.py:167-189 (stage-1 trainer): the vanilla bank of
    crd_v3_loss with unit sample weights and the two-layer heads Linear-ReLU-Linear + L2 norm (:219-234).
    st.embed_s / st.embed_t hold linear.0.* and linear.2.* tensors."""
    def head(x, d):
        h = F.relu(F.linear(x.view(x.shape[0], -1), d["linear.0.weight"], d["linear.0.bias"]))
        h = F.linear(h, d["linear.2.weight"], d["linear.2.bias"])
        return h / h.pow(2).sum(1, keepdim=True).pow(0.5)
    v1, v2 = head(f_s, st.embed_s), head(f_t, st.embed_t)
    K = int(st.params[0].item()); T = st.params[1].item()
    B, D = v1.shape
    n_out = st.memory_v1.size(0)
    w1 = torch.index_select(st.memory_v1, 0, idx.view(-1)).detach().view(B, K + 1, D)
    out_v2 = torch.exp(torch.bmm(w1, v2.view(B, D, 1)) / T)
    w2 = torch.index_select(st.memory_v2, 0, idx.view(-1)).detach().view(B, K + 1, D)
    out_v1 = torch.exp(torch.bmm(w2, v1.view(B, D, 1)) / T)
    if st.params[2].item() < 0:
        st.params[2] = out_v1.mean().detach() * n_out
    if st.params[3].item() < 0:
        st.params[3] = out_v2.mean().detach() * n_out
    out_v1 = out_v1 / st.params[2].item()
    out_v2 = out_v2 / st.params[3].item()
    _bank_update(st, v1, v2, y)
    return _contrast_loss_weighted(1.0, out_v1, st.n_data) + _contrast_loss_weighted(1.0, out_v2, st.n_data)


def momentum_aekd_loss(main_loss, feat_s, loss_t_list, mo_scale, grads_m=0.9, grads_thresh="False", thresh=0.0,
                       ce_grads=True):
    """momentum_AEKD_loss (train_test_path_multi_distill_v2.py:89-132): cosine Gram WITHOUT the x len(list)
    factor of the MICCAI version, optional binarisation, EMA of the weights across iterations."""
    losses = list(loss_t_list) + ([main_loss] if ce_grads else [])
    grads = [torch.autograd.grad(l, feat_s, retain_graph=True)[0].detach().clone() for l in losses]
    all_grads = torch.stack(grads).view(len(grads), -1)
    norm = torch.norm(all_grads, p=2, dim=1, keepdim=True)
    rel = torch.matmul(all_grads, all_grads.T) / torch.matmul(norm, norm.T)
    if grads_thresh == "True":
        rel = torch.where(rel > thresh, 1.0, 0.0)
    scale = rel.sum(dim=1)
    mo_scale = scale if mo_scale is None else grads_m * mo_scale + (1 - grads_m) * scale
    total = torch.dot(mo_scale[:-1], torch.stack(list(loss_t_list)))
    return mo_scale, total


# ------------------------------------------------------------------------------------------------ MIA-2023
def distill_kl_per_sample(y_s, y_t, T=1.0):
    """DistillKL.forward of MIA-2023 KD_loss.py:14-20 -> (loss, sample_loss[B])."""
    p_s = F.log_softmax(y_s / T, dim=1)
    p_t = F.softmax(y_t / T, dim=1)
    sample_loss = torch.sum(F.kl_div(p_s, p_t, reduction="none"), dim=-1) * (T ** 2)
    return sample_loss.sum() / y_s.shape[0], sample_loss


def assign_sample_weights(pred_s, pred_t, gt, max_discrep):
    """assign_sample_weights of MIA-2023 train_test_path_multi_distill.py:131-158 (pred_* are probabilities)."""
    gt = F.one_hot(gt, 3).float()
    conf_t = torch.log(torch.sum(pred_t * gt, 1)) - torch.log(torch.max(pred_t * (1 - gt), 1)[0])
    conf_s = torch.log(torch.sum(pred_s * gt, 1)) - torch.log(torch.max(pred_s * (1 - gt), 1)[0])
    d = torch.maximum(conf_t - conf_s, torch.zeros_like(conf_t)).detach()
    return torch.minimum(d, max_discrep * torch.ones_like(conf_t))


def gk_refine_thresh(main_loss, feat_s, loss_t_list, use_grads_thresh="True", grads_thresh=0.25, ce_grads=True):
    """GK_refine_thresh (MIA-2023 train_test_path_multi_distill.py:81-128): per-sample cosine of the gradients
    (sklearn cosine_similarity semantics: zero rows -> 0), per-sample loss weights [B, n], batch-mean scale."""
    losses = [l.sum() for l in loss_t_list] + ([main_loss] if ce_grads else [])
    grads = torch.stack([torch.autograd.grad(l, feat_s, retain_graph=True)[0].detach() for l in losses])   # [n,B,D]
    B = grads.shape[1]
    g = grads.permute(1, 0, 2)                                    # [B, n, D]
    nrm = g.norm(dim=2, keepdim=True)
    gn = torch.where(nrm > 0, g / nrm.clamp_min(1e-30), torch.zeros_like(g))
    cos = torch.bmm(gn, gn.transpose(1, 2))                       # [B, n, n]
    if use_grads_thresh == "True":
        all_scale = torch.where(cos > grads_thresh, 1.0, 0.0).sum(1)
    else:
        all_scale = torch.where(cos > 0, cos, torch.zeros_like(cos)).sum(1)
    total = torch.sum(all_scale[:, :-1].transpose(0, 1) * torch.stack(list(loss_t_list))) / B
    return all_scale.mean(0), total, all_scale


class CRDv10State(CRDv3State):
    """ContrastMemory of CRD_criterion_v10.py:20-43: v3 buffers + the class label of every bank row."""

    def __init__(self, n_data, labels, **kw):
        super().__init__(n_data, **kw)
        self.labels = torch.as_tensor(labels).long()


def crd_v10_loss(st, sample_weights, f_s, f_t, batch_label, y, idx, num_pos):
    """CRDLoss.forward with pos_extra == "neighbors" (CRD_criterion_v10.py:45-176, :210-239, :281-314).
    Returns (loss, sample_loss[B])."""
    v1 = embed_forward(f_s, st.embed_s["linear.weight"], st.embed_s["linear.bias"])
    v2 = embed_forward(f_t, st.embed_t["linear.weight"], st.embed_t["linear.bias"])
    K = int(st.params[0].item()); T = st.params[1].item()
    B, D = v1.shape
    n_out = st.memory_v1.size(0)
    mask = (st.labels.view(1, -1) == batch_label.view(-1, 1)).float()            # batch_class_mask :57-60

    def side(mem):
        w = torch.index_select(mem, 0, idx.view(-1)).detach().view(B, K + 1, D)
        q = w[:, 0, :]
        cos = F.normalize(q, dim=1) @ F.normalize(mem, dim=1).T                  # sklearn cosine_similarity :72-73
        sim = mask * cos
        srt = torch.sort(sim, descending=True, dim=-1)
        nb, nbs = srt[1][:, :num_pos], srt[0][:, :num_pos]
        knn = torch.index_select(mem, 0, nb.reshape(-1)).detach().view(B, num_pos, D)
        return torch.cat((knn, w[:, 1:, :]), 1), nbs, nb

    w1, s1, nb1 = side(st.memory_v1)
    out_v2 = torch.exp(torch.bmm(w1, v2.view(B, D, 1)) / T)
    w2, s2, nb2 = side(st.memory_v2)
    out_v1 = torch.exp(torch.bmm(w2, v1.view(B, D, 1)) / T)
    if st.params[2].item() < 0:
        st.params[2] = out_v1.mean().detach() * n_out
    if st.params[3].item() < 0:
        st.params[3] = out_v2.mean().detach() * n_out
    out_v1 = out_v1 / st.params[2].item()
    out_v2 = out_v2 / st.params[3].item()
    _bank_update(st, v1, v2, y)

    def closs(x, knn_sim):                                                       # ContrastLoss_v2 :286-314
        P = num_pos
        m = x.size(1) - P
        Pn = 1 / float(st.n_data)
        P_pos = x.narrow(1, 0, P)
        log_D1 = torch.div(P_pos, P_pos.add(m * Pn + EPS)).log()
        P_neg = x.narrow(1, P, m)
        log_D0 = torch.div(torch.full_like(P_neg, m * Pn), P_neg.add(m * Pn + EPS)).log()
        sl = -((log_D1.squeeze(-1) + log_D0.sum(1).view(B, 1).repeat(1, P)) * knn_sim).sum(1) / knn_sim.sum(1)
        sl = sample_weights.view(-1) * sl
        return sl.sum(0) / B, sl

    ls, sls = closs(out_v1, s2)      # criterion_s(out_s, t_similarity)  :226-227
    lt, slt = closs(out_v2, s1)
    return ls + lt, sls + slt, dict(nb1=nb1, nb2=nb2, sim1=s1, sim2=s2)


def crd_v10_centers_loss(st, sample_weights, f_s, f_t, batch_label, y, idx, class_idx, num_pos=2):
    """CRDLoss.forward with pos_extra == "centers" and num_pos == 2 (CRD_criterion_v10.py:81-101, :118-139, :228-231,
    ContrastLoss :241-277): the positives of a query are the MEAN bank row of its class (:88-89) and its own bank row,
    the negatives its K sampled rows plus the centres of the other classes.  num_pos > 2 replaces the mean by sklearn
    KMeans centres (random initialisation: not reproducible, not restated).  Returns (loss, sample_loss[B])."""
    if num_pos != 2:
        raise NotImplementedError("num_pos > 2 uses sklearn KMeans (random init): parity unpinned")
    v1 = embed_forward(f_s, st.embed_s["linear.weight"], st.embed_s["linear.bias"])
    v2 = embed_forward(f_t, st.embed_t["linear.weight"], st.embed_t["linear.bias"])
    K = int(st.params[0].item()); T = st.params[1].item()
    B, D = v1.shape
    n_out = st.memory_v1.size(0)
    C = len(class_idx)
    onehot = F.one_hot(batch_label, num_classes=C).numpy()
    import numpy as np
    neg_labels = torch.as_tensor(np.argwhere(onehot == 0)[:, 1])                 # :63-64, ascending other classes

    def side(mem):
        centers = torch.stack([mem[torch.as_tensor(np.asarray(class_idx[c])).long()].mean(0) for c in range(C)]).view(C, 1, D)
        w = torch.index_select(mem, 0, idx.view(-1)).detach().view(B, K + 1, D)
        pos_c = torch.index_select(centers, 0, batch_label).view(B, num_pos - 1, D)
        neg_c = torch.index_select(centers, 0, neg_labels).view(B, (C - 1) * (num_pos - 1), D)
        return torch.cat((pos_c, w, neg_c), 1)

    out_v2 = torch.exp(torch.bmm(side(st.memory_v1), v2.view(B, D, 1)) / T)
    out_v1 = torch.exp(torch.bmm(side(st.memory_v2), v1.view(B, D, 1)) / T)
    if st.params[2].item() < 0:
        st.params[2] = out_v1.mean().detach() * n_out
    if st.params[3].item() < 0:
        st.params[3] = out_v2.mean().detach() * n_out
    out_v1 = out_v1 / st.params[2].item()
    out_v2 = out_v2 / st.params[3].item()
    _bank_update(st, v1, v2, y)

    def closs(x):                                                                # ContrastLoss :246-277
        P = num_pos
        m = x.size(1) - P
        Pn = 1 / float(st.n_data)
        P_pos = x.narrow(1, 0, P)
        log_D1 = torch.div(P_pos, P_pos.add(m * Pn + EPS)).log()
        P_neg = x.narrow(1, P, m)
        log_D0 = torch.div(torch.full_like(P_neg, m * Pn), P_neg.add(m * Pn + EPS)).log()
        sl = -(log_D1.squeeze(-1) + log_D0.sum(1).view(B, 1).repeat(1, P)).sum(1) / P
        sl = sample_weights.view(-1) * sl
        return sl.sum(0) / B, sl

    ls, sls = closs(out_v1)
    lt, slt = closs(out_v2)
    return ls + lt, sls + slt


# ------------------------------------------------------------------------------------------------------------------
# Row a16: t-SVD low-rank constraint of the MIA-2022 stage-1 trainer ("MIA 2022/train_test_tSVD.py").
# update_adj_tensor (:57-70) and the penalty (:413-431) are pinned by tests/golden/mia2022_tsvd.npz (produced by running
# the reference functions).  update_aux is NOT in the reference repository (`from my_utils.TSVD_update_aux import
# update_aux`, :31 - module absent): parity unpinned.  What follows the call contract of :382-391 (input [B,B,n_views]
# detached adjacency stack, threshold Lambda_global/mu, returns (aux of the same shape, TNN scalar)) is the standard
# proximal operator of the tensor nuclear norm: FFT along the view axis, singular-value soft-thresholding of every
# frontal slice, inverse FFT.  TNN is reported as (1/n_views) * sum over the frequency slices of the nuclear norm of the
# thresholded slice.  This is OUR choice of algorithm and normalisation; the HIP kernel is tested against this function.
def update_adj_tensor(feats):
    """feats: list of [B, D] -> list of row-L2-normalised Gram matrices F.normalize(f f^T)  (train_test_tSVD.py:57-70)"""
    return [torch.nn.functional.normalize(torch.mm(f, f.t())) for f in feats]


def tsvd_penalty(adj, aux, mu):
    """sum_v mu/2 * ||adj_v - aux_v||_F^2  (train_test_tSVD.py:418-431, one modality)"""
    loss = 0
    for a, x in zip(adj, aux):
        loss = loss + mu / 2.0 * (torch.norm(a - x)) ** 2
    return loss


def update_aux(adj_stack, tau):
    """adj_stack: [B, B, V] (numpy or tensor).  Returns (aux [B, B, V] float64 numpy, TNN float)."""
    import numpy as np
    x = np.asarray(adj_stack.detach().cpu().numpy() if torch.is_tensor(adj_stack) else adj_stack, dtype=np.float64)
    V = x.shape[2]
    xf = np.fft.fft(x, axis=2)
    yf = np.zeros_like(xf)
    tnn = 0.0
    for k in range(V):
        u, s, vh = np.linalg.svd(xf[:, :, k], full_matrices=False)
        s = np.maximum(s - tau, 0.0)
        tnn += s.sum()
        yf[:, :, k] = (u * s) @ vh
    return np.real(np.fft.ifft(yf, axis=2)), tnn / V
