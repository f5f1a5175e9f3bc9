"""Drop-in for the reference's CL_utils/memory_new.py ContrastMemory_v3 (:225-397): same buffers
(`params` [K,T,Z_v1,Z_v2,momentum,P], `memory_v1`, `memory_v2`), same forward semantics
(score with the PRE-update bank, discrepancy-driven positive/negative selection, first-call Z,
momentum update).  The forward returns (out_v1, out_v2) like the reference when called directly; inside
CRDLoss the fused loss+gradient kernel is used instead (nothing [B,P+K,128]-sized is materialised)."""
import math

import numpy as np
import torch
import torch.nn as nn

from .. import ops
from .._lib import lib, check, ptr, stream, require_cuda


def crd_core(v1, v2, mem, y, idx, ranks, per_sample=False, loss_out=None, outputs_only=False):
    """The fused CRD step on normalised embeddings: scores against the PRE-update banks, pair selection, first-call Z,
    NCE loss, analytic gradients, bank momentum update.  Returns (loss, dv1, dv2) with dv = d loss / d v for a unit
    upstream gradient (per_sample: row b of dv belongs to loss[b]).  `loss_out`: optional 0-d destination of the summed
    loss.  No autograd here: _CRDCoreFn wraps it, the fused loss head of DistillStep calls it directly.
    `outputs_only`: the standalone ContrastMemory_v3.forward - instead of the loss, return (out_v1, out_v2, rows1, rows2):
    the selected scores / Z [B, P2+K2] and the gathered pre-update bank rows its backward needs."""
    v1, v2 = ops._f32(v1), ops._f32(v2)
    if getattr(mem, "_scan_neg", None) is not None:
        if outputs_only:
            raise NotImplementedError("the bank-scan form of the negatives has no [B, P+K] score tensor to return")
        return _crd_core_scan(v1, v2, mem, y, idx, per_sample, loss_out)
    B, D = v1.shape
    P, K, P2, K2 = mem.P, mem.K, mem.P2, mem.K2
    PK, S2 = P + K, P2 + K2
    dev = v1.device
    idx = require_cuda(idx).contiguous()
    y = require_cuda(y).contiguous()
    if idx.dtype != torch.int64 or y.dtype != torch.int64 or idx.shape != (B, PK):
        raise RuntimeError("contrast_idx must be int64 [B, nce_p+nce_k] and idx int64 [B]")
    out1 = torch.empty(B, PK, device=dev, dtype=torch.float32)
    out2 = torch.empty_like(out1)
    diff = torch.empty_like(out1)
    L = lib()
    st = stream()
    T = mem.T
    idx2 = getattr(mem, "_idx_bank2", None)          # MIA-2023 v10: bank-specific positive rows
    posw_s = getattr(mem, "_posw_s", None)
    posw_t = getattr(mem, "_posw_t", None)
    check(L.ph_crd_score(ptr(v1), ptr(v2), ptr(idx), ptr(idx2), ptr(mem.memory_v1), ptr(mem.memory_v2), ptr(out1),
                         ptr(out2), ptr(diff), B, PK, D, T, st), "ph_crd_score")
    sel = torch.empty(B, S2, device=dev, dtype=torch.int32)
    xs = torch.empty(B, S2, device=dev, dtype=torch.float32)
    xt = torch.empty_like(xs)
    check(L.ph_crd_select(ptr(diff), ptr(out1), ptr(out2), ptr(ranks), ptr(sel), ptr(xs), ptr(xt), B, P, K, P2, K2,
                          1 if mem.select_neg_pairs == "True" else 0,
                          1 if getattr(mem, "select_pos_pairs", False) is True else 0, st), "ph_crd_select")
    if not mem._z_set:
        sums = torch.empty(2, device=dev, dtype=torch.float32)
        check(L.ph_crd_zsum(ptr(xs), ptr(xt), ptr(sums), B * S2, st), "ph_crd_zsum")
        count = float(B * S2)
        if mem.sync is not None:
            count = mem.sync.all_reduce_z(sums, count)
        check(L.ph_crd_setz(ptr(mem.params), ptr(sums), count, float(mem.nLem), st), "ph_crd_setz")
        mem._z_set = True
        if mem.verbose:   # the reference prints Z once (memory_new.py:371,375); costs one host sync
            z = mem.params[2:4].tolist()
            print("normalization constant Z_v1 is set to {:.1f}".format(z[0]))
            print("normalization constant Z_v2 is set to {:.1f}".format(z[1]))
    if outputs_only:
        o1 = torch.empty(B, S2, device=dev, dtype=torch.float32)
        o2 = torch.empty_like(o1)
        rows1 = torch.empty(B, S2, D, device=dev, dtype=torch.float32)
        rows2 = torch.empty_like(rows1)
        check(L.ph_crd_outputs(ptr(xs), ptr(xt), ptr(sel), ptr(idx), ptr(idx2), ptr(mem.memory_v1), ptr(mem.memory_v2),
                               ptr(mem.params), ptr(o1), ptr(o2), ptr(rows1), ptr(rows2), B, PK, S2, D, st), "ph_crd_outputs")
        _crd_update(mem, y, v1, v2, D)
        mem.last = dict(sel=sel, xs=xs, xt=xt, diff=diff)
        return o1, o2, rows1, rows2
    lossp = torch.empty(B, device=dev, dtype=torch.float32)
    dv1 = torch.empty(B, D, device=dev, dtype=torch.float32)
    dv2 = torch.empty_like(dv1)
    bnorm = float(mem.batch_norm_size or B)
    if getattr(mem, "sample_KD", False):
        bnorm = 1.0      # ContrastLoss_v2's per-sample branch (CRD_loss.py:246-250) does not divide by the batch size
    # long column lists (nce_k = 4096 of the MIA trainers) are dealt to several workgroups per sample through a workspace
    lg_ws = (torch.empty(L.ph_crd_loss_grad_workspace_bytes(B), device=dev, dtype=torch.uint8) if P2 + K2 >= 1024 else None)
    check(L.ph_crd_loss_grad(ptr(xs), ptr(xt), ptr(sel), ptr(idx), ptr(idx2), ptr(posw_s), ptr(posw_t),
                             ptr(mem.memory_v1), ptr(mem.memory_v2),
                             ptr(mem.params), ptr(lossp), ptr(dv1), ptr(dv2), B, PK, P2, K2, D, float(mem.nLem),
                             1.0 / bnorm, ptr(lg_ws), st), "ph_crd_loss_grad")
    if per_sample:
        loss = lossp       # [B] per-sample losses (each already divided by the batch normaliser)
    else:
        loss = loss_out if loss_out is not None else torch.empty((), device=dev, dtype=torch.float32)
        check(L.ph_sum(ptr(lossp), ptr(loss), B, 1.0, st), "ph_sum")
    _crd_update(mem, y, v1, v2, D)
    mem.last = dict(sel=sel, xs=xs, xt=xt, diff=diff)
    return loss, dv1, dv2


def _crd_core_scan(v1, v2, mem, y, idx, per_sample, loss_out):
    """crd_core with the negatives in bank-scan form (`mem._scan_neg` = dict(idx [B, >= col0 + K] int64, col0, K), set by
    CRD_criterion_v10.CRDLoss.neighbor_columns when nce_k reaches the number of bank rows - BASELINE configs[4] read as 65 536
    negatives per query, SURVEY 8-e assumption (i)).  `idx` = the P positive columns only.  Same sums as the gathered kernels
    (reference CRD_criterion_v10.py:106-153, :300-306), taken over every bank row weighted by its multiplicity among the
    query's negatives: one read of each bank per GEMM instead of B x K gathered rows."""
    B, D = v1.shape
    P, n = mem.P, mem.nLem
    scan = mem._scan_neg
    K, col0 = int(scan["K"]), int(scan["col0"])
    neg = require_cuda(scan["idx"])
    dev = v1.device
    idx = require_cuda(idx).contiguous()
    y = require_cuda(y).contiguous()
    if idx.dtype != torch.int64 or y.dtype != torch.int64 or idx.shape != (B, P):
        raise RuntimeError("bank-scan form: the positive columns must be int64 [B, nce_p] and idx int64 [B]")
    if neg.dtype != torch.int64 or neg.dim() != 2 or neg.shape[0] != B or neg.shape[1] < col0 + K or neg.stride(1) != 1:
        raise RuntimeError("bank-scan form: the sampled negatives must be int64 [B, >= %d] with unit column stride" % (col0 + K))
    if mem.select_neg_pairs == "True" or getattr(mem, "select_pos_pairs", False) is True:
        raise NotImplementedError("bank-scan form with a ranked pair selection (it needs the per-column score list)")
    L, st = lib(), stream()
    idx2 = getattr(mem, "_idx_bank2", None)
    posw_s, posw_t = getattr(mem, "_posw_s", None), getattr(mem, "_posw_t", None)
    xs = torch.empty(B, P, device=dev, dtype=torch.float32); xt = torch.empty_like(xs); diff = torch.empty_like(xs)
    check(L.ph_crd_score(ptr(v1), ptr(v2), ptr(idx), ptr(idx2), ptr(mem.memory_v1), ptr(mem.memory_v2), ptr(xs), ptr(xt),
                         ptr(diff), B, P, D, mem.T, st), "ph_crd_score")
    sel = torch.arange(P, device=dev, dtype=torch.int32).repeat(B, 1).contiguous()
    mult = torch.empty(B, n, device=dev, dtype=torch.int32)
    check(L.ph_crd_neg_hist(ptr(neg), neg.stride(0), col0, K, B, n, ptr(mult), st), "ph_crd_neg_hist")
    S1 = torch.empty(B, n, device=dev, dtype=torch.float32); S2 = torch.empty_like(S1)
    ops.sgemm(v1, mem.memory_v2, None, S1, B, n, D, D, 1, 1, D)      # S1[b][r] = v1[b] . bank2[r]  (out_v1, Z_v1, feeds dv1)
    ops.sgemm(v2, mem.memory_v1, None, S2, B, n, D, D, 1, 1, D)
    ws = torch.empty(L.ph_crd_scan_neg_workspace_bytes(B, n), device=dev, dtype=torch.uint8)
    if not mem._z_set:
        sums = torch.empty(2, device=dev, dtype=torch.float32)
        check(L.ph_crd_zsum(ptr(xs), ptr(xt), ptr(sums), B * P, st), "ph_crd_zsum")
        check(L.ph_crd_scan_neg(ptr(S1), ptr(S2), ptr(mult), ptr(mem.params), ptr(ws), None, ptr(sums), B, n, K, 1.0, 1, st),
              "ph_crd_scan_neg (Z)")
        count = float(B * (P + K))
        if mem.sync is not None:
            count = mem.sync.all_reduce_z(sums, count)
        check(L.ph_crd_setz(ptr(mem.params), ptr(sums), count, float(n), st), "ph_crd_setz")
        mem._z_set = True
        if mem.verbose:
            z = mem.params[2:4].tolist()
            print("normalization constant Z_v1 is set to {:.1f}".format(z[0]))
            print("normalization constant Z_v2 is set to {:.1f}".format(z[1]))
    bnorm = float(mem.batch_norm_size or B)
    if getattr(mem, "sample_KD", False):
        bnorm = 1.0
    lossn = torch.empty(B, device=dev, dtype=torch.float32)
    check(L.ph_crd_scan_neg(ptr(S1), ptr(S2), ptr(mult), ptr(mem.params), ptr(ws), ptr(lossn), None, B, n, K, 1.0 / bnorm, 0, st),
          "ph_crd_scan_neg")
    lossp = torch.empty(B, device=dev, dtype=torch.float32)
    dv1 = torch.empty(B, D, device=dev, dtype=torch.float32); dv2 = torch.empty_like(dv1)
    check(L.ph_crd_loss_grad_pos(ptr(xs), ptr(xt), ptr(sel), ptr(idx), ptr(idx2), ptr(posw_s), ptr(posw_t), ptr(mem.memory_v1),
                                 ptr(mem.memory_v2), ptr(mem.params), ptr(lossp), ptr(dv1), ptr(dv2), B, P, K, D, float(n),
                                 1.0 / bnorm, st), "ph_crd_loss_grad_pos")
    # dv1 += coef1 [B, n] x bank2 [n, D] (split over the rows: fixed-order partial sums), likewise dv2
    nsplit = 128 if n >= 8192 else (32 if n >= 1024 else 1)
    part = torch.empty(nsplit * B * D, device=dev, dtype=torch.float32)
    for coef, bank, dv in ((S1, mem.memory_v2, dv1), (S2, mem.memory_v1, dv2)):
        dn = torch.empty(B, D, device=dev, dtype=torch.float32)
        if nsplit > 1:
            check(L.ph_sgemm_splitk(ptr(coef), ptr(bank), None, ptr(dn), ptr(part), nsplit, B, D, n, n, 1, D, 1, D, ops.ACT_NONE, st),
                  "ph_sgemm_splitk")
        else:
            ops.sgemm(coef, bank, None, dn, B, D, n, n, 1, D, 1)
        dv.add_(dn)
    lossp.add_(lossn)
    if per_sample:
        loss = lossp
    else:
        loss = loss_out if loss_out is not None else torch.empty((), device=dev, dtype=torch.float32)
        check(L.ph_sum(ptr(lossp), ptr(loss), B, 1.0, st), "ph_sum")
    _crd_update(mem, y, v1, v2, D)
    mem.last = dict(sel=sel, xs=xs, xt=xt, diff=diff, mult=mult)
    return loss, dv1, dv2


def _crd_update(mem, y, v1, v2, D):
    """Momentum update AFTER scoring (memory_new.py:382-395); under data parallelism every replica applies the update of
    the whole global batch."""
    if mem.sync is not None:
        yy, vv1, vv2 = mem.sync.all_gather_rows(y, v1.detach(), v2.detach())
    else:
        yy, vv1, vv2 = y, v1.detach(), v2.detach()
    check(lib().ph_crd_update(ptr(mem.memory_v1), ptr(mem.memory_v2), ptr(vv1), ptr(vv2), ptr(yy), ptr(mem.params),
                              yy.shape[0], D, stream()), "ph_crd_update")


class _CRDCoreFn(torch.autograd.Function):
    """(v1, v2) -> NCE loss (s_loss + t_loss of CRD_loss.py:172-174) with analytic gradients."""

    @staticmethod
    def forward(ctx, v1, v2, mem, y, idx, ranks, per_sample=False):
        loss, dv1, dv2 = crd_core(v1, v2, mem, y, idx, ranks, per_sample)
        ctx.save_for_backward(dv1, dv2)
        ctx.per_sample = per_sample
        return loss

    @staticmethod
    def backward(ctx, g):
        dv1, dv2 = ctx.saved_tensors
        if ctx.per_sample:
            g = g.reshape(-1, 1)      # d loss_b / d v_b is row b of dv
        return dv1 * g, dv2 * g, None, None, None, None, None


class _CRDOutputsFn(torch.autograd.Function):
    """(v1, v2) -> (out_v1, out_v2) of ContrastMemory_v3.forward (memory_new.py:249-397), each [B, P2+K2, 1]:
    out_v1 = exp(memory_v2[sel] . v1 / T) / Z_v1 depends on v1 only, out_v2 on v2 only (Z are constants, :368-379)."""

    @staticmethod
    def forward(ctx, v1, v2, mem, y, idx, ranks):
        o1, o2, rows1, rows2 = crd_core(v1, v2, mem, y, idx, ranks, outputs_only=True)
        ctx.save_for_backward(o1, o2, rows1, rows2)
        ctx.T = float(mem.T)
        return o1.unsqueeze(-1), o2.unsqueeze(-1)

    @staticmethod
    def backward(ctx, g1, g2):
        o1, o2, rows1, rows2 = ctx.saved_tensors
        B, S2 = o1.shape
        D = rows1.shape[2]
        g1 = None if g1 is None else ops._f32(g1).reshape(B, S2).contiguous()
        g2 = None if g2 is None else ops._f32(g2).reshape(B, S2).contiguous()
        dv1 = torch.empty(B, D, device=o1.device, dtype=torch.float32)
        dv2 = torch.empty_like(dv1)
        check(lib().ph_crd_outputs_bwd(ptr(g1), ptr(g2), ptr(o1), ptr(o2), ptr(rows1), ptr(rows2), ctx.T, ptr(dv1), ptr(dv2),
                                       B, S2, D, stream()), "ph_crd_outputs_bwd")
        return dv1, dv2, None, None, None, None


def draw_uniform_indices(mem, y, width):
    """`contrast_idx is None` in every ContrastMemory of the reference (memory_new.py:265-267, CRD_criterion.py:37-39,
    CRD_criterion_v3.py:37-39): `width` bank rows per sample drawn from the AliasMethod table over uniform unigrams
    (:229-231, :401-458) - with all-ones unigrams every table entry keeps probability 1, so the draw IS the uniform integer
    draw - and column 0 overwritten with y.  One launch (ph_alias_uniform_draw: counter-based, keyed by (seed, step, element));
    the device-side step counter makes captured replays draw afresh.  Distributional parity: the reference's stream is torch's
    CUDA generator.  Under data parallelism (`mem.sync`) the rank is mixed into the seed, so that W replicas draw W times as
    many distinct negatives as one (the reference's one DataParallel process draws one global batch of independent rows).  The
    step counter `mem._draw_step` is not part of the module's state_dict (its keys are the reference's); DistillStep.state_dict
    saves it (`crd_draw_steps`) so that a resumed run continues the stream."""
    y = require_cuda(y).contiguous()
    B = y.shape[0]
    if getattr(mem, "_draw_step", None) is None or mem._draw_step.device != y.device:
        mem._draw_step = torch.zeros(1, device=y.device, dtype=torch.int64)
    if getattr(mem, "_draw_seed", None) is None:
        seed = int(torch.initial_seed())
        sync = getattr(mem, "sync", None)
        if sync is not None and getattr(sync, "rank", 0):
            seed ^= (int(sync.rank) * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        mem._draw_seed = seed & 0x7FFFFFFFFFFFFFFF
    out = torch.empty(B, width, device=y.device, dtype=torch.int64)
    check(lib().ph_alias_uniform_draw(ptr(y), ptr(out), int(mem.nLem), B, width, mem._draw_seed, ptr(mem._draw_step), stream()),
          "ph_alias_uniform_draw")
    mem._draw_step += 1
    return out


class ContrastMemory_v3(nn.Module):
    def __init__(self, inputSize, outputSize, P, K, T=0.07, momentum=0.5, select_pos_pairs=True, P2=10,
                 select_neg_pairs=True, K2=512):
        super().__init__()
        self.nLem = outputSize
        self.P, self.K, self.P2, self.K2 = P, K, P2, K2
        self.T = T
        self.select_pos_pairs = select_pos_pairs
        self.select_neg_pairs = select_neg_pairs
        if select_pos_pairs is not True or select_neg_pairs not in ("True", "False", True, False):
            raise NotImplementedError("select_pos_pairs is always on in the reference (options.py:44-45)")
        if select_neg_pairs in (True, False):
            self.select_neg_pairs = "True" if select_neg_pairs else "False"
        if self.select_neg_pairs == "False":
            self.K2 = K
        self.register_buffer("params", torch.tensor([K, T, -1, -1, momentum, P], dtype=torch.float32))
        stdv = 1.0 / math.sqrt(inputSize / 3)
        self.register_buffer("memory_v1", torch.rand(outputSize, inputSize).mul_(2 * stdv).add_(-stdv))
        self.register_buffer("memory_v2", torch.rand(outputSize, inputSize).mul_(2 * stdv).add_(-stdv))
        self._z_set = False
        self.sync = None               # set by dist.attach() under data parallelism
        self.batch_norm_size = None    # global batch under data parallelism
        self.verbose = True
        self.last = None

    def _load_from_state_dict(self, *a, **k):
        super()._load_from_state_dict(*a, **k)
        self._z_set = bool((self.params[2:4] > 0).all().item())

    def draw_ranks(self, epoch, select_pos_mode):
        """The host-RNG draw of memory_new.py:311-322 (numpy global RNG, like the reference)."""
        if select_pos_mode == "hard":
            return None
        if select_pos_mode == "mid":
            r = np.random.choice(np.arange(30, 100, 1), self.P2, replace=False)
        elif select_pos_mode == "random":
            r = np.random.randint(0, self.P, self.P2)
        elif select_pos_mode == "curriculum":
            interval = 4 - np.ceil(3 * epoch)
            r = np.random.randint(50 * (interval - 1), 50 * interval, self.P2)
        else:
            raise NotImplementedError(select_pos_mode)
        if r.max() >= self.P:
            raise RuntimeError("select_pos_mode '%s' needs nce_p > %d" % (select_pos_mode, int(r.max())))
        return torch.as_tensor(np.asarray(r), dtype=torch.int32)

    def draw_indices(self, y):
        """idx == None (memory_new.py:265-267): B * (K + P) rows drawn from the AliasMethod table, column 0 := y."""
        return draw_uniform_indices(self, y, self.P + self.K)

    def _ranks(self, epoch, select_pos_mode, ranks, dev):
        if ranks is None:
            ranks = self.draw_ranks(epoch, select_pos_mode)
        if ranks is not None and not (torch.is_tensor(ranks) and ranks.is_cuda and ranks.dtype == torch.int32):
            ranks = torch.as_tensor(np.asarray(ranks), dtype=torch.int32).to(dev, non_blocking=True)
        return ranks

    def loss(self, epoch, v1, v2, y, idx, select_pos_mode="mid", ranks=None):
        """s_loss + t_loss of CRDLoss.forward (CRD_loss.py:167-174) in one fused pass (what CRDLoss calls)."""
        if idx is None:
            idx = self.draw_indices(y)
        return _CRDCoreFn.apply(v1, v2, self, y, idx, self._ranks(epoch, select_pos_mode, ranks, v1.device),
                                bool(getattr(self, "sample_KD", False)))

    def forward(self, epoch, v1, v2, y, idx=None, select_pos_mode="mid", ranks=None):
        """memory_new.py:249-397 as written: returns (out_v1, out_v2), each [B, P2+K2, 1] (selected positives first),
        differentiable w.r.t. v1 / v2; sets Z on the first call, momentum-updates the rows `y` of both banks.  The rank
        list of the `mid` / `random` / `curriculum` modes comes from numpy's global RNG exactly as in the reference
        (:311-322) unless `ranks` is given."""
        if idx is None:
            idx = self.draw_indices(y)
        return _CRDOutputsFn.apply(v1, v2, self, y, idx, self._ranks(epoch, select_pos_mode, ranks, v1.device))
