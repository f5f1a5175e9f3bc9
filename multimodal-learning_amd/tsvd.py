"""t-SVD low-rank constraint of the MIA-2022 stage-1 trainer, reference
"MIA 2022/train_test_tSVD.py": `update_adj_tensor` (:57-70), the auxiliary-variable update called at :382-391
(`update_aux`, whose source `my_utils/TSVD_update_aux.py` is ABSENT from the reference repository), the Frobenius
penalty (:418-431) and the mu schedule (:413).

`update_adj_tensor` and the penalty are pinned against the reference (tests/golden/mia2022_tsvd.npz).  `update_aux` is
parity-unpinned: it implements the call contract with the standard tensor-nuclear-norm proximal operator (DFT along
the view axis, singular-value soft-thresholding per frequency slice, inverse DFT; csrc/tsvd.hip) and is tested against
oracle/variants.py:update_aux.  The reference returns the aux tensor through the host (`.float().cuda()`, :391); here
everything stays on the device."""
import torch

from . import ops
from ._lib import lib, check, ptr, stream


def update_adj_tensor(adj_tensor, feats):
    """adj_tensor[i] = F.normalize(feats[i] @ feats[i].T)  (train_test_tSVD.py:57-70; gradients flow to feats).
    F.normalize's eps = 1e-12 clamp never binds for a Gram row of a non-zero feature row; `Normalize` has none."""
    for i in range(len(feats)):
        f = ops._f32(feats[i])
        adj_tensor[i] = ops.L2NormFn.apply(ops.LinearFn.apply(f, f, None))
    return adj_tensor


def maxnorm_mix(a, b, wa, wb):
    """wa * a / max(a) + wb * b / max(b): the mixed-feature views of n_views = 6 / 8 (train_test_tSVD.py:305-307,
    :334-363; both operands are mean-teacher features, no gradient)."""
    a, b = ops._f32(a.detach()), ops._f32(b.detach())
    out = torch.empty_like(a)
    check(lib().ph_maxnorm_mix(ptr(a), ptr(b), ptr(out), a.numel(), float(wa), float(wb), stream()), "ph_maxnorm_mix")
    return out


def update_aux(adj, tau, print_bool=False):
    """adj: [B, B, n_views] detached adjacency stack, tau = Lambda_global / mu (a float, or a 1-element device tensor that a
    captured graph re-reads at every replay).  Returns (aux [B, B, n_views], TNN) like the call at train_test_tSVD.py:382;
    TNN is a 0-d device tensor (no host sync)."""
    B, B2, V = adj.shape
    if B != B2:
        raise ValueError("adjacency stack must be [B, B, n_views]")
    a = ops._f32(adj.detach()).permute(2, 0, 1).contiguous()          # view-major [V][B][B]
    aux = torch.empty_like(a)
    tnn = torch.empty(1, device=a.device, dtype=torch.float32)
    ws = torch.empty(lib().ph_tsvd_workspace_bytes(V, B), device=a.device, dtype=torch.uint8)
    if torch.is_tensor(tau):
        check(lib().ph_tsvd_update_aux_dev(ptr(a), ptr(aux), ptr(tnn), V, B, ptr(ops._f32(tau).reshape(1)), ptr(ws), stream()),
              "ph_tsvd_update_aux_dev")
    else:
        check(lib().ph_tsvd_update_aux(ptr(a), ptr(aux), ptr(tnn), V, B, float(tau), ptr(ws), stream()), "ph_tsvd_update_aux")
    return aux.permute(1, 2, 0), tnn[0]


class _SqDiffFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, half_mu):
        a, b = ops._f32(a), ops._f32(b.detach())
        out = torch.empty(1, device=a.device, dtype=torch.float32)
        check(lib().ph_sqdiff_sum(ptr(a), ptr(b), ptr(out), a.numel(), float(half_mu), stream()), "ph_sqdiff_sum")
        ctx.save_for_backward(a, b)
        ctx.half_mu = float(half_mu)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = ops._f32(g).reshape(1)
        d = torch.empty_like(a)
        check(lib().ph_scaled_diff(ptr(a), ptr(b), ptr(g), 2.0 * ctx.half_mu, ptr(d), a.numel(), stream()), "ph_scaled_diff")
        return d, None, None


def tsvd_penalty(adj_tensor, aux_tensor, mu):
    """sum_v mu/2 * ||adj_v - aux_v||_F^2  (train_test_tSVD.py:418-431, one modality); mu: a float, or a 1-element device
    tensor (captured graphs)."""
    loss = 0
    if torch.is_tensor(mu):
        for a, x in zip(adj_tensor, aux_tensor):
            loss = loss + _SqDiffFn.apply(a, x, 1.0)
        return loss * (mu.reshape(()) * 0.5)
    for a, x in zip(adj_tensor, aux_tensor):
        loss = loss + _SqDiffFn.apply(a, x, mu / 2.0)
    return loss


class TSVDState:
    """The per-epoch state of train_test_tSVD.py:285-413 for one modality pair: adjacency / aux tensors of n_views views
    and the mu schedule `mu = min(mu * pho, max_mu)` (:413) applied after every aux update."""

    def __init__(self, opt, batch_size, device):
        self.n_views = opt.n_views
        self.Lambda_global, self.mu, self.pho, self.max_mu = opt.Lambda_global, opt.mu, opt.pho, opt.max_mu
        self.adj = [None] * self.n_views
        self.aux = [torch.zeros(batch_size, batch_size, device=device) for _ in range(self.n_views)]
        self.tnn = None

    def step(self, feats, update=True):
        """feats: n_views tensors [B, D].  Returns the penalty of this batch (graph attached to feats)."""
        self.adj = update_adj_tensor(self.adj, feats)
        if update:
            stack = torch.stack([a.detach() for a in self.adj], dim=2)
            aux, self.tnn = update_aux(stack, self.Lambda_global / self.mu)
            self.aux = [aux[:, :, v].contiguous() for v in range(self.n_views)]
            self.mu = min(self.mu * self.pho, self.max_mu)
        return tsvd_penalty(self.adj, self.aux, self.mu)
