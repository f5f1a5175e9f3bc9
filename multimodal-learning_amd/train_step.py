"""Drop-in for the hot-loop functions of the reference's MICCAI-2022/train_test_path_multi_distill.py:
``update_ema_variables`` (:34-38), ``AEKD_loss`` (:41-70) and the batch body of ``train()`` (:242-330,
here ``DistillStep``), plus the fused optimiser behind ``define_optimizer`` (networks_new.py:80-90)."""
import math

import numpy as np
import torch
import torch.nn as nn

from . import ops
from ._lib import lib, check, ptr, stream


# ----------------------------------------------------------------------------------------- flat storage
class FlatParams:
    """Re-homes a list of tensors into ONE contiguous fp32 buffer (each tensor 16-B aligned) so that the
    optimiser / EMA / gradient all-reduce are single streaming kernels instead of ~70 small launches."""

    def __init__(self, tensors, with_grad=False):
        tensors = list(tensors)
        dev = tensors[0].device
        offs, off = [], 0
        for t in tensors:
            offs.append(off)
            off += (t.numel() + 3) // 4 * 4
        self.numel = off
        self.flat = torch.zeros(off, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(off, device=dev, dtype=torch.float32) if with_grad else None
        self.offsets = offs
        self.tensors = tensors
        for t, o in zip(tensors, offs):
            n = t.numel()
            self.flat[o:o + n].copy_(t.detach().reshape(-1))
            t.data = self.flat[o:o + n].view(t.shape)
            if with_grad and t.requires_grad:
                t.grad = self.grad[o:o + n].view(t.shape)
        ops.register_flat(self)       # define_reg finds the flat layout of a parameter through its storage

    def segments(self, pred):
        """Maximal [start, end) runs of consecutive tensors for which pred(t) holds."""
        segs, cur = [], None
        for t, o in zip(self.tensors, self.offsets):
            e = o + (t.numel() + 3) // 4 * 4
            if pred(t):
                if cur is not None and cur[1] == o:
                    cur[1] = e
                else:
                    cur = [o, e]
                    segs.append(cur)
            else:
                cur = None
        return [tuple(s) for s in segs]


class PinnedRing:
    """Host staging for small per-step values that are uploaded with an asynchronous copy while the device still
    works on earlier steps: a ring of pinned rows, each guarded by an event recorded after its copy (a single pinned
    buffer would be overwritten by the host, which runs steps ahead of the device, before the copy has executed)."""

    def __init__(self, shape, dtype, depth=16):
        self.buf = torch.zeros((depth,) + tuple(shape), dtype=dtype).pin_memory()
        self.events = [None] * depth
        self.k = 0

    def upload(self, dst, values):
        i = self.k % len(self.events)
        self.k += 1
        if self.events[i] is not None:
            self.events[i].synchronize()
        row = self.buf[i]
        row.copy_(torch.as_tensor(values, dtype=row.dtype).reshape(row.shape))
        dst.copy_(row, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.events[i] = ev


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam semantics (L2-in-grad weight decay, bias correction, eps outside the sqrt) as one
    HIP kernel over flat parameter / gradient / state buffers (ph_adam_ema_step)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        params = list(params)
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        if len(self.param_groups) != 1:
            raise NotImplementedError("one parameter group (the reference passes model.parameters())")
        self._flat = None
        self._step = 0
        self.ema_flat = None      # optional FlatParams of the EMA model to update in the same kernel
        self.ema_alpha = None
        self.ema_range = None
        self._hyper = None        # device float[12] {lr, 1-beta1^t, sqrt(1-beta2^t), ema_alpha, 1-ema_alpha, 0, 0, 0, beta1, 1-beta1, beta2, 1-beta2}: graph-replayable
        self._hyper_host = None
        self._prepared = False

    def _ensure_flat(self):
        if self._flat is None:
            ps = self.param_groups[0]["params"]
            if not ps[0].is_cuda:
                raise RuntimeError("FusedAdam needs parameters on the GPU (no CPU fallback)")
            self._flat = FlatParams(ps, with_grad=True)
            self._m = torch.zeros_like(self._flat.flat)
            self._v = torch.zeros_like(self._flat.flat)
            inactive = getattr(self, "_inactive", ())
            self._segs = self._flat.segments(lambda t: t.requires_grad and id(t) not in inactive)
            self._hyper = torch.zeros(12, device=ps[0].device, dtype=torch.float32)
            self._hyper_host = PinnedRing((12,), torch.float32)
            ops.bump_weight_epoch()
        return self._flat

    def prepare_step(self):
        """Host side of a step: advance t, compute the bias corrections in double (as torch.optim.Adam's Python
        scalars) and enqueue their copy to device memory.  Split from step() so that the kernel launches can be
        captured in a HIP graph while these per-step scalars keep changing."""
        self._ensure_flat()
        g = self.param_groups[0]
        self._step += 1
        b1, b2 = g["betas"]
        a = self.ema_alpha if self.ema_alpha is not None else 0.0
        # [8..11] = beta1, 1 - beta1, beta2, 1 - beta2: read by the kernel in place of its launch arguments (a captured launch
        # would keep the betas of the capture; OneCycleLR cycles beta1 - networks_new.define_scheduler)
        self._hyper_host.upload(self._hyper, [g["lr"], 1.0 - b1 ** self._step, math.sqrt(1.0 - b2 ** self._step), a, 1.0 - a,
                                              0.0, 0.0, 0.0, b1, 1.0 - b1, b2, 1.0 - b2])
        self._prepared = True

    @property
    def flat(self):
        return self._ensure_flat()

    def set_inactive(self, params):
        """Parameters that take no part in the objective (an option branch leaves a module unused): the reference's
        torch.optim optimiser skips a parameter whose .grad is None - no weight decay, no moment update - whereas the flat
        gradient buffer would hand the kernel a zero gradient and decay the weights.  Their segments leave the update."""
        self._inactive = {id(p) for p in params}
        if self._flat is not None:
            self._segs = self._flat.segments(lambda t: t.requires_grad and id(t) not in self._inactive)

    def zero_grad(self, set_to_none=False):
        """Gradients are views of one flat buffer: zeroing is a single memset and the views persist."""
        f = self._ensure_flat()
        f.grad.zero_()
        for t, o in zip(f.tensors, f.offsets):
            if t.requires_grad and (t.grad is None or t.grad.data_ptr() != f.grad.data_ptr() + 4 * o):
                t.grad = f.grad[o:o + t.numel()].view(t.shape)

    @torch.no_grad()
    def step(self, closure=None):
        f = self._ensure_flat()
        # a gradient tensor torch re-allocated (e.g. after zero_grad(set_to_none=True)) is copied back
        for t, o in zip(f.tensors, f.offsets):
            if t.requires_grad and t.grad is not None and t.grad.data_ptr() != f.grad.data_ptr() + 4 * o:
                f.grad[o:o + t.numel()].copy_(t.grad.reshape(-1))
                t.grad = f.grad[o:o + t.numel()].view(t.shape)
        g = self.param_groups[0]
        if not self._prepared:
            self.prepare_step()
        self._prepared = False
        segs = []
        for (s, e) in self._segs:      # split at the EMA boundary (student params | embed params)
            if self.ema_flat is not None and self.ema_range is not None and s < self.ema_range[1] < e:
                segs += [(s, self.ema_range[1]), (self.ema_range[1], e)]
            else:
                segs.append((s, e))
        for (s, e) in segs:
            ema = None
            if self.ema_flat is not None and self.ema_range is not None and s >= self.ema_range[0] and e <= self.ema_range[1]:
                ema = self.ema_flat.flat[s:e]
            check(lib().ph_adam_ema_step_dev(ptr(f.flat[s:e]), ptr(f.grad[s:e]), ptr(self._m[s:e]), ptr(self._v[s:e]),
                                             ptr(ema), e - s, -1.0, g["betas"][1], g["eps"],      # (beta1 < 0: betas from hyper[8..11])
                                             g["weight_decay"], ptr(self._hyper), stream()), "ph_adam_ema_step_dev")
        ops.bump_weight_epoch()

    def state_dict(self):
        """torch.optim.Adam's layout (state[i] = {step, exp_avg, exp_avg_sq} per parameter index, so the reference's
        `optimizer.load_state_dict` - train_test_path_multi_distill.py:387-402 saves / restores exactly that - accepts
        it) plus the flat buffers under "fused" for an exact, copy-only restore."""
        sd = super().state_dict()
        if self._flat is not None and self._step > 0:
            f = self._flat
            sd["state"] = {i: dict(step=torch.tensor(float(self._step)),
                                   exp_avg=self._m[o:o + t.numel()].view(t.shape).clone(),
                                   exp_avg_sq=self._v[o:o + t.numel()].view(t.shape).clone())
                           for i, (t, o) in enumerate(zip(f.tensors, f.offsets))
                           if t.requires_grad and id(t) not in getattr(self, "_inactive", ())}
        sd["fused"] = dict(step=self._step, exp_avg=None if self._flat is None else self._m.clone(),
                           exp_avg_sq=None if self._flat is None else self._v.clone())
        return sd

    def load_state_dict(self, sd):
        """Accepts this class's own state_dict or a plain torch.optim.Adam one (a checkpoint written by the reference)."""
        sd = dict(sd)                      # the caller's checkpoint object stays intact (it may be loaded again)
        fused = sd.pop("fused", None)
        per_param = sd.get("state") or {}
        sd["state"] = {}                   # the moments live in the flat buffers, not in torch's per-parameter state
        super().load_state_dict(sd)
        if fused is not None and fused["exp_avg"] is not None:
            self._ensure_flat()
            self._step = fused["step"]
            self._m.copy_(fused["exp_avg"])
            self._v.copy_(fused["exp_avg_sq"])
        elif per_param:
            f = self._ensure_flat()
            steps = set()
            self._m.zero_(); self._v.zero_()
            for i, st in per_param.items():
                t, o = f.tensors[int(i)], f.offsets[int(i)]
                if tuple(st["exp_avg"].shape) != tuple(t.shape):
                    raise ValueError("optimizer state %d has shape %s, parameter %s" % (int(i), tuple(st["exp_avg"].shape),
                                                                                       tuple(t.shape)))
                self._m[o:o + t.numel()].copy_(st["exp_avg"].reshape(-1))
                self._v[o:o + t.numel()].copy_(st["exp_avg_sq"].reshape(-1))
                steps.add(int(float(st["step"])))
            if len(steps) != 1:
                raise NotImplementedError("per-parameter step counts differ (%s): the fused kernel keeps one" % sorted(steps))
            self._step = steps.pop()


class FusedAdagrad(FusedAdam):
    """torch.optim.Adagrad as the reference builds it (networks_new.py:86-87: lr, weight_decay, initial_accumulator_value = 0.1;
    lr_decay 0, eps 1e-10) over the same flat buffers as FusedAdam, EMA copy fused (ph_adagrad_ema_step_dev).  `_v` holds the
    accumulator `sum`; there is no first moment."""

    def __init__(self, params, lr=1e-2, weight_decay=0.0, initial_accumulator_value=0.0, eps=1e-10):
        torch.optim.Optimizer.__init__(self, list(params), dict(lr=lr, lr_decay=0, eps=eps, weight_decay=weight_decay,
                                                                  initial_accumulator_value=initial_accumulator_value))
        if len(self.param_groups) != 1:
            raise NotImplementedError("one parameter group (the reference passes model.parameters())")
        self._flat = None
        self._step = 0
        self.ema_flat = self.ema_alpha = self.ema_range = None
        self._hyper = self._hyper_host = None
        self._prepared = False

    def _ensure_flat(self):
        if self._flat is None:
            super()._ensure_flat()
            self._m = None
            self._v.fill_(float(self.param_groups[0]["initial_accumulator_value"]))
        return self._flat

    def prepare_step(self):
        self._ensure_flat()
        self._step += 1
        a = self.ema_alpha if self.ema_alpha is not None else 0.0
        self._hyper_host.upload(self._hyper, [self.param_groups[0]["lr"], 1.0, 1.0, a, 1.0 - a, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0])
        self._prepared = True

    @torch.no_grad()
    def step(self, closure=None):
        f = self._ensure_flat()
        for t, o in zip(f.tensors, f.offsets):
            if t.requires_grad and t.grad is not None and t.grad.data_ptr() != f.grad.data_ptr() + 4 * o:
                f.grad[o:o + t.numel()].copy_(t.grad.reshape(-1))
                t.grad = f.grad[o:o + t.numel()].view(t.shape)
        g = self.param_groups[0]
        if not self._prepared:
            self.prepare_step()
        self._prepared = False
        for (s, e) in self._segs:
            cuts = [s, e]
            if self.ema_flat is not None and self.ema_range is not None and s < self.ema_range[1] < e:
                cuts = [s, self.ema_range[1], e]
            for a, b in zip(cuts[:-1], cuts[1:]):
                ema = None
                if self.ema_flat is not None and self.ema_range is not None and a >= self.ema_range[0] and b <= self.ema_range[1]:
                    ema = self.ema_flat.flat[a:b]
                check(lib().ph_adagrad_ema_step_dev(ptr(f.flat[a:b]), ptr(f.grad[a:b]), ptr(self._v[a:b]), ptr(ema), b - a,
                                                    g["eps"], g["weight_decay"], ptr(self._hyper), stream()),
                      "ph_adagrad_ema_step_dev")
        ops.bump_weight_epoch()

    def state_dict(self):
        """torch.optim.Adagrad's layout (state[i] = {step, sum}; like torch, filled from construction on: step 0 and
        sum = initial_accumulator_value before the first step) plus the flat accumulator under "fused".
        One difference from torch.optim.Adagrad is documented rather than mirrored: a requires_grad parameter that received NO
        gradient in a step (torch: .grad is None -> skipped) sees a zero gradient here, i.e. weight decay and the accumulator
        still advance - unless the owner declares it unused (`set_inactive`, what DistillStep does for the option branches that
        leave a criterion idle)."""
        sd = torch.optim.Optimizer.state_dict(self)
        inactive = getattr(self, "_inactive", ())
        if self._flat is not None:
            f = self._flat
            sd["state"] = {i: dict(step=torch.tensor(float(self._step)), sum=self._v[o:o + t.numel()].view(t.shape).clone())
                           for i, (t, o) in enumerate(zip(f.tensors, f.offsets)) if t.requires_grad and id(t) not in inactive}
        else:
            init = float(self.param_groups[0]["initial_accumulator_value"])
            sd["state"] = {i: dict(step=torch.tensor(0.0), sum=torch.full_like(t, init, dtype=torch.float32))
                           for i, t in enumerate(self.param_groups[0]["params"]) if t.requires_grad and id(t) not in inactive}
        sd["fused"] = dict(step=self._step, sum=None if self._flat is None else self._v.clone())
        return sd

    def load_state_dict(self, sd):
        """Accepts this class's own state_dict or a plain torch.optim.Adagrad one (a checkpoint written by the reference)."""
        sd = dict(sd)
        fused = sd.pop("fused", None)
        per_param = sd.get("state") or {}
        sd["state"] = {}
        torch.optim.Optimizer.load_state_dict(self, sd)
        f = self._ensure_flat()
        if fused is not None and fused.get("sum") is not None:
            if tuple(fused["sum"].shape) != tuple(self._v.shape):
                raise ValueError("fused accumulator has %d elements, the parameters %d" % (fused["sum"].numel(), self._v.numel()))
            self._step = fused["step"]
            self._v.copy_(fused["sum"])
        elif per_param:
            steps = set()
            for i, st in per_param.items():
                t, o = f.tensors[int(i)], f.offsets[int(i)]
                if tuple(st["sum"].shape) != tuple(t.shape):
                    raise ValueError("optimizer state %d has shape %s, parameter %s" % (int(i), tuple(st["sum"].shape), tuple(t.shape)))
                self._v[o:o + t.numel()].copy_(st["sum"].reshape(-1))
                steps.add(int(float(st["step"])))
            if len(steps) != 1:
                raise NotImplementedError("per-parameter step counts differ (%s): the fused kernel keeps one" % sorted(steps))
            self._step = steps.pop()


def update_ema_variables(model, ema_model, alpha, global_step):
    """train_test_path_multi_distill.py:34-38 - parameters only, BN buffers untouched."""
    alpha = min(1 - 1 / (global_step + 1), alpha)
    for mod in ema_model.modules():
        if hasattr(mod, "_get_packed"):
            mod._follow_epoch = True       # its weights change through raw pointers
    with torch.no_grad():
        for ema_param, param in zip(ema_model.parameters(), model.parameters()):
            check(lib().ph_ema_update(ptr(ema_param.data), ptr(param.data), param.numel(), alpha, stream()),
                  "ph_ema_update")
    ops.bump_weight_epoch()


def AEKD_loss(opt, optimizer, main_loss, feat_s, loss_t_list, sync=None):
    """GK-Refine (train_test_path_multi_distill.py:41-70): weights = row sums of the (x len(loss_t_list))
    cosine Gram matrix of d loss_i / d feat_s.  The reference obtains each gradient by a FULL backward pass
    (5 ResNet-18 backward passes thrown away except the hook value, :49-56); the identical values are the
    VJPs stopped at feat_s, so this costs a few [B,128]-sized kernels.  Returns (scale[5], total_KD_loss)."""
    losses = list(loss_t_list) + ([main_loss] if opt.CE_grads else [])
    grads = [torch.autograd.grad(l, feat_s, retain_graph=True)[0] for l in losses]
    ng = len(grads)
    G = torch.stack([g.reshape(-1) for g in grads]).contiguous()
    gram = torch.empty(ng * ng, device=G.device, dtype=torch.float32)
    check(lib().ph_gram(ptr(G), ptr(gram), ng, G.shape[1], stream()), "ph_gram")
    if sync is not None:
        sync.all_reduce_sum(gram)        # global-batch cosine under data parallelism
    scale = torch.empty(ng, device=G.device, dtype=torch.float32)
    check(lib().ph_gk_scale(ptr(gram), None, ng, 0, float(len(loss_t_list)), ptr(scale), None, stream()), "ph_gk_scale")
    losses_div_tensor = torch.stack(list(loss_t_list))
    total_KD_loss = torch.dot(scale[:-1], losses_div_tensor)      # :68 (mirrors the slicing as written)
    return scale, total_KD_loss


def momentum_AEKD_loss(opt, optimizer, main_loss, feat_s, loss_t_list, mo_scale, sync=None):
    """GK-Refine of the MIA-2022 trainer ("MIA 2022/train_test_path_multi_distill_v2.py":89-132): like AEKD_loss but
    the cosine Gram is NOT multiplied by len(loss_t_list) (:113), may be binarised with `opt.thresh` when
    opt.grads_thresh == "True" (:114-115) and the weights are an EMA over iterations with opt.grads_m (:121-124).
    `mo_scale` is None on the first call; returns (mo_scale, total_KD_loss)."""
    losses = list(loss_t_list) + ([main_loss] if opt.CE_grads else [])
    grads = [torch.autograd.grad(l, feat_s, retain_graph=True)[0] for l in losses]
    ng = len(grads)
    G = torch.stack([g.reshape(-1) for g in grads]).contiguous()
    gram = torch.empty(ng * ng, device=G.device, dtype=torch.float32)
    check(lib().ph_gram(ptr(G), ptr(gram), ng, G.shape[1], stream()), "ph_gram")
    if sync is not None:
        sync.all_reduce_sum(gram)
    if mo_scale is None:
        state = torch.zeros(ng, device=G.device, dtype=torch.float32)
        init = torch.zeros(1, device=G.device, dtype=torch.int32)
    else:
        state = mo_scale.detach().clone()
        init = torch.ones(1, device=G.device, dtype=torch.int32)
    check(lib().ph_gk_scale_momentum(ptr(gram), ng, 1 if opt.grads_thresh == "True" else 0, float(opt.thresh),
                                     float(opt.grads_m), ptr(state), ptr(init), stream()), "ph_gk_scale_momentum")
    total_KD_loss = torch.dot(state[:-1], torch.stack(list(loss_t_list)))
    return state, total_KD_loss


def _validate_opt(opt, who):
    """A drop-in must fail loudly where it diverges: option values the reference's batch body would honour and this
    package does not implement raise here instead of silently training another objective."""
    if getattr(opt, "task", "grad") != "grad":
        raise NotImplementedError("%s: task %r - the survival (Cox) branch of the trainers is out of scope (SURVEY 2.1 #8); "
                                  "only the grading task is built" % (who, opt.task))
    if getattr(opt, "reg_type", "none") not in ("none", "path", "mm", "all", "omic"):
        raise NotImplementedError("reg method [%s] is not implemented" % opt.reg_type)      # networks_new.py:106-107
    if getattr(opt, "mode", "pathomic") != "pathomic":
        raise NotImplementedError("%s: mode %r (the distillation trainers build the pathomic teacher and a path student)"
                                  % (who, opt.mode))
    if getattr(opt, "optimizer_type", "adam") not in ("adam", "adagrad"):
        raise NotImplementedError("%s: optimizer_type %r (adam and adagrad are built; adabound needs the absent `adabound` "
                                  "package, networks_new.py:82-83)" % (who, opt.optimizer_type))
    if getattr(opt, "act_type", "LSM") != "LSM":
        raise NotImplementedError("%s: act_type %r (the grading task uses the log-softmax head)" % (who, opt.act_type))
    if getattr(opt, "fusion_type", "pofusion") != "pofusion":
        raise NotImplementedError("%s: fusion_type %r (pofusion is built)" % (who, opt.fusion_type))
    if getattr(opt, "return_grad", "False") != "False":
        raise NotImplementedError("%s: return_grad needs the reference's absent my_utils.compute_gradients" % who)
    if getattr(opt, "use_vgg_features", 0):
        raise NotImplementedError("%s: use_vgg_features (pre-extracted features bypass the ResNet trunk)" % who)


# ----------------------------------------------------------------------------------------- the hot loop
class DistillStep:
    """The batch body of train() (train_test_path_multi_distill.py:242-330) over the drop-in modules:
    student fwd, EMA fwd, teacher fwd, CE + 2xKL + 2xCRD, GK-Refine, backward, fused Adam + EMA.

    No host synchronisation inside `step` (the reference does >= 7 `.item()` syncs); losses are returned
    as device scalars."""

    def __init__(self, opt, n_data, device="cuda", k=1, sync=None, models=None, variant="miccai2022",
                 train_class_idx=None):
        """variant "miccai2022": MICCAI-2022/train_test_path_multi_distill.py:242-330 (DC-Distill CRD + GK-Refine);
        variant "mia2022": "MIA 2022/train_test_path_multi_distill_v2.py":388-500 (vanilla K+1 CRD bank of
        CRD_criterion_v3 with the epoch weight, momentum GK-Refine carried over the iterations, row a17);
        variant "mia2023": "MIA 2023/stage2_unimodal_student/train_test_path_multi_distill.py":307-448 (per-sample KL
        rows, confidence-discrepancy query weights, the class-aware KNN bank of CRD_criterion_v10 - needs
        `train_class_idx`, the per-class lists of bank rows - and the per-sample GK-Refine, row a18)."""
        from .networks_new import define_net, define_optimizer, define_scheduler
        from .kd_loss import DistillKL
        if variant == "mia2022":
            from .CL_utils.CRD_criterion_v3 import CRDLoss
        elif variant == "miccai2022":
            from .CL_utils import CRDLoss
        elif variant == "mia2023":
            from .CL_utils.CRD_criterion_v10 import CRDLoss as _CRDv10
            from .mia2023 import DistillKL
            if train_class_idx is None:
                raise ValueError("variant 'mia2023' needs train_class_idx (CRD_criterion_v10.py:25)")
            CRDLoss = lambda o, n: _CRDv10(o, n, train_class_idx)    # noqa: E731
        else:
            raise ValueError("variant must be 'miccai2022', 'mia2022' or 'mia2023'")
        self.variant = variant
        _validate_opt(opt, "DistillStep")
        self._mo_state = None     # momentum GK-Refine weights (mia2022), updated in place on the device
        self.sampler = None       # optional ContrastIndexSampler: draws sample_idx when the batch carries None
        self._mo_init = None
        self.opt = opt
        self.device = torch.device(device)
        self.sync = sync
        if models is None:
            self.fix_model = define_net(opt, k).to(self.device)
            self.model = define_net(opt, k, path_only=True).to(self.device)
            self.ema_model = define_net(opt, k, path_only=True).to(self.device)
        else:
            self.fix_model, self.model, self.ema_model = models
        for p in self.fix_model.parameters():
            p.detach_(); p.requires_grad = False                                   # :170-173
        for p in self.ema_model.parameters():
            p.detach_()                                                            # :178-180
        self.criterion_div = DistillKL(opt.kd_T)                                   # :197
        self.criterion_kd = CRDLoss(opt, n_data).to(self.device)                   # :202
        self.criterion_kd_path = CRDLoss(opt, n_data).to(self.device)              # :206
        self.module_list = nn.ModuleList([self.model, self.criterion_kd.embed_s, self.criterion_kd.embed_t,
                                          self.criterion_kd_path.embed_s, self.criterion_kd_path.embed_t])
        # the other `--distill` choices of the MIA-2022 trainer (train_test_path_multi_distill_v2.py:316-342): one
        # feature-level baseline criterion, no embedding heads in the optimiser
        self.zoo_kd = None
        if variant == "mia2022" and opt.distill != "crd":
            from . import distiller_zoo as Z
            zoo = {"kd": None, "feats_KL": Z.feats_KL, "rkd": Z.RKDLoss, "pkt": Z.PKT, "similarity": Z.Similarity}
            if opt.distill not in zoo:
                raise NotImplementedError("--distill %s (built: crd, kd, feats_KL, rkd, pkt, similarity)" % opt.distill)
            if sync is not None and opt.distill in ("rkd", "pkt", "similarity"):
                raise NotImplementedError("--distill %s relates the rows of ONE batch to each other: per-replica batches "
                                          "would change the loss (single GPU only)" % opt.distill)
            self.zoo_kd = zoo[opt.distill]() if zoo[opt.distill] is not None else False
            self.module_list = nn.ModuleList([self.model])
        self.optimizer = define_optimizer(opt, self.module_list)                   # :211
        self.scheduler = define_scheduler(opt, self.optimizer)                     # :212
        # option branches that leave a CRD criterion unused (`--distill kd`: both; `--num_teachers 1`: criterion_kd_path, the
        # one call goes through criterion_kd whichever teacher is chosen, :282-285): its embedding heads stay in the
        # optimiser's parameter list as in the reference, where their .grad stays None and the update skips them
        self._branch_check()
        unused = []
        if self.zoo_kd is None:
            if opt.distill == "kd":
                unused = [self.criterion_kd, self.criterion_kd_path]
            elif opt.num_teachers == 1:
                unused = [self.criterion_kd_path]
        if unused and hasattr(self.optimizer, "set_inactive"):
            self.optimizer.set_inactive([p for c in unused for m_ in (c.embed_s, c.embed_t) for p in m_.parameters()])
        self.iter_num = opt.global_step
        self._want_graph = False
        self._static = None
        self._slots = None
        self._side_stream = torch.cuda.Stream(device=self.device) if getattr(opt, "overlap_teachers", True) else None
        # the fused teacher on a third stream (opt.teacher_streams = 1: behind the mean teacher on the second)
        self._side_stream2 = (torch.cuda.Stream(device=self.device)
                              if self._side_stream is not None and getattr(opt, "teacher_streams", 2) == 2 else None)
        self._stamps = None      # int64 [16] device tensor: set BEFORE enable_graph() to have the body mark its phases
        # side stream of the fused loss head (loss_head.py: second CRD chain, head weight gradients); opt.overlap_head = False: off
        self._head_side = (torch.cuda.Stream(device=self.device)
                           if self.device.type == "cuda" and getattr(opt, "overlap_head", True) else None)
        self.module_list.train(); self.fix_model.train()                           # :231-232 (EMA stays in train mode)
        # flat EMA storage with the student's layout -> EMA is fused into the Adam kernel
        flat = self.optimizer.flat
        n_student = len(list(self.model.parameters()))
        self.ema_flat = FlatParams(list(self.ema_model.parameters()))
        self.model._direct_grad = True     # trunk gradients are written straight into the flat .grad views
        # `opt.lambda_reg * define_reg(opt, model)` (:312-313).  The stage-2 command passes --reg_type none; with `all`
        # the L1 gradient is added into the same flat buffer, so the trunk's gradients must ACCUMULATE (autograd's
        # AccumulateGrad) instead of overwriting.  `path` / `mm` / `omic` probe attributes the ResNet student does not
        # have: the reference fails with AttributeError on its first batch, here already at construction.
        self._reg_on = opt.reg_type != "none"
        if self._reg_on:
            from .networks_new import define_reg
            define_reg(opt, self.model)
            self.model._direct_grad = False
        for mod in self.ema_model.modules():
            if hasattr(mod, "_get_packed"):
                mod._follow_epoch = True   # updated by the fused Adam+EMA kernel through raw pointers
        end = flat.offsets[n_student] if n_student < len(flat.offsets) else flat.numel
        self.optimizer.ema_flat = self.ema_flat
        self.optimizer.ema_range = (0, end)
        if sync is not None:
            sync.attach(self)
            # with a regulariser the trunk gradients reach the flat buffer later, through AccumulateGrad (_direct_grad is
            # False): the layer-3/4 slice is NOT final when the trunk backward passes layer 3, so its all-reduce must not
            # start there (it would reduce a buffer that is still being accumulated into: ADVICE r02)
            if (hasattr(sync, "begin_grad_slice") and getattr(opt, "overlap_grad_allreduce", True)
                    and self.model._direct_grad):
                # everything from layer3.0.conv1.weight to the end of the flat buffer (layers 3-4, heads, CRD embeddings)
                # is final once the trunk backward has passed layer 3: its all-reduce starts there (dist.begin_grad_slice)
                first = self.model.layer3[0].conv1.weight
                lo = next(o for t, o in zip(flat.tensors, flat.offsets) if t is first)
                # (stamp 11: the moment the first phase starts - bench.py's `comm` reports the window it has to hide in)
                self.model._grad_ready_hook = lambda: (self._stamp(11), self.sync.begin_grad_slice(self.optimizer.flat, lo))[1]

    def _head_stream(self):
        return self._head_side

    def _two_chains(self, f1, f2):
        """Two independent launch chains (the two CRD criteria: separate banks, separate heads): the second on the head's side
        stream, joined before anything reads its result; autograd replays each backward on the stream its forward used.  One
        stream under data parallelism (the criteria's all-gathers must keep one issue order on every rank)."""
        hs = self._head_side if self.sync is None else None
        if hs is None:
            return f1(), f2()
        main = torch.cuda.current_stream()
        hs.wait_stream(main)
        with torch.cuda.stream(hs):
            r2 = f2()
        r1 = f1()
        main.wait_stream(hs)
        for t in (r2 if isinstance(r2, (tuple, list)) else (r2,)):
            if torch.is_tensor(t):
                t.record_stream(main)
        return r1, r2

    def _stamp(self, k):
        """Phase marker k (ph_prof_stamp: the device wall clock when the current stream gets here; also inside a graph)."""
        if self._stamps is not None:
            check(lib().ph_prof_stamp(self._stamps.data_ptr() + 8 * k, stream()), "ph_prof_stamp")

    # ------------------------------------------------------------------ one step
    def _device_body(self, x_path, ema_x_path, x_omic, grade, index, sample_idx, bnorm, e, r1, r2):
        """Everything of the step that runs on the device (capturable in one HIP graph)."""
        opt = self.opt
        # The two teachers' forwards do not depend on the student's: they run on a second stream (a parallel branch
        # of the captured graph) so their small-kernel tails overlap the student's convolutions.
        main = torch.cuda.current_stream()
        side = self._side_stream
        # the student and the teacher read the same x_path (:249, :256): one packing pass into the trunk's input layout
        # serves both (before the streams fork, so both consumers are ordered after it)
        from .resnets import pack_shared_input
        self._stamp(0)
        pack_shared_input(x_path, (self.model, self.fix_model.path_net))
        side2 = self._side_stream2 if side is not None else None
        if side is not None:
            side.wait_stream(main)
            with torch.cuda.stream(side), torch.no_grad():
                _, ema_path_feat, ema_logit_path, _, _ = self.ema_model(x_path=ema_x_path)                  # :254
                self._stamp(8)
            if side2 is not None:
                side2.wait_stream(main)
            with torch.cuda.stream(side2 if side2 is not None else side), torch.no_grad():
                fuse_feat, _, _, _, logits, pred, _, _, _, _, _ = self.fix_model(x_path=x_path, x_omic=x_omic)  # :256
                self._stamp(9)
            for t in (ema_path_feat, ema_logit_path, fuse_feat, logits[-1]):
                t.record_stream(main)
        _, path_feat, logit_path, pred_path, _ = self.model(x_path=x_path)                                   # :249
        self._stamp(1)
        if self._stamps is not None and path_feat.requires_grad:
            path_feat.register_hook(lambda g: (self._stamp(4), g)[1])
        if side is not None:
            main.wait_stream(side)
            if side2 is not None:
                main.wait_stream(side2)
        else:
            with torch.no_grad():
                _, ema_path_feat, ema_logit_path, _, _ = self.ema_model(x_path=ema_x_path)                  # :254
                fuse_feat, _, _, _, logits, pred, _, _, _, _, _ = self.fix_model(x_path=x_path, x_omic=x_omic)  # :256
        self._stamp(2)
        if self._fused_head_ok():
            # :262-313 as one function of the student feature (loss_head.py): same values, ~50 launches instead of ~125
            from .loss_head import FusedDistillLossFn, FusedMia2023LossFn, LossHeadCtx
            Hc = LossHeadCtx(self, grade, logits[-1].detach(), ema_logit_path.detach(), fuse_feat.detach(),
                             ema_path_feat.detach(), index, sample_idx, r1, r2, bnorm, variant=self.variant,
                             e_dev=e if self.variant != "miccai2022" else None)
            loss = (FusedMia2023LossFn if self.variant == "mia2023" else FusedDistillLossFn).apply(path_feat, Hc)
            loss = self._add_reg(loss)                                                                      # :312-313
            self.optimizer.zero_grad()                                                                      # :326
            self._stamp(3)
            loss.backward()                                                                                 # :327
            # the heads' weight gradients (loss_head.py) - joined only where loss_head forks that stream (`hs = ... if
            # step.sync is None`): under data parallelism it never joins the capture, and waiting on a stream outside a
            # capture invalidates the capture (hipErrorStreamCaptureIsolation)
            if self._head_side is not None and self.sync is None:
                torch.cuda.current_stream().wait_stream(self._head_side)
            self._stamp(5)
            if self.sync is not None:
                self.sync.all_reduce_grads(self.optimizer.flat)
                self._stamp(10)       # stamp 5 -> 10 = the part of the gradient all-reduce the backward did not hide
            self.optimizer.step()                                                                           # :328 (+ :329 fused)
            self._stamp(6)
            o = Hc.out
            extra = {k: o[k] for k in ("rows_div1", "rows_kd1", "w1", "w2") if k in o}       # (the MIA-2023 body's per-sample values)
            return dict(loss=loss.detach(), loss_cls=o["loss_cls"], loss_div1=o["loss_div1"], loss_div2=o["loss_div2"],
                        loss_kd1=o["loss_kd1"], loss_kd2=o["loss_kd2"], scale=o["scale"], logit_path=o["logit_path"],
                        pred_path=o["pred_path"], path_feat=path_feat.detach(), ema_logit=ema_logit_path,
                        fuse_logit=logits[-1], fuse_feat=fuse_feat, ema_feat=ema_path_feat, **extra)
        loss_cls = ops.NLLFn.apply(pred_path, grade, bnorm)                                                 # :262
        if self.zoo_kd is None and not (opt.num_teachers == 2 and opt.distill == "crd"):
            return self._branch_tail(e, r1, grade, index, sample_idx, loss_cls, path_feat, logit_path, pred_path, ema_path_feat,
                                     ema_logit_path, fuse_feat, logits)
        if self.variant == "mia2023":
            return self._mia2023_tail(e, loss_cls, grade, index, sample_idx, path_feat, logit_path, pred_path,
                                      ema_path_feat, ema_logit_path, fuse_feat, logits)
        if self.zoo_kd is not None:
            return self._mia2022_baseline_tail(loss_cls, path_feat, logit_path, pred_path, ema_path_feat, ema_logit_path,
                                               fuse_feat, logits)
        loss_div1 = self.criterion_div(logit_path, logits[-1].detach())                                     # :264
        loss_div2 = self.criterion_div(logit_path, ema_logit_path.detach())                                 # :265
        if self.variant == "mia2022":
            # v2 trainer :436-437: the first argument is the epoch weight of the per-sample CRD loss; result shape [1]
            loss_kd1, loss_kd2 = self._two_chains(
                lambda: self.criterion_kd(e, path_feat, fuse_feat.detach(), index, sample_idx).reshape(()),
                lambda: self.criterion_kd_path(e, path_feat, ema_path_feat.detach(), index, sample_idx).reshape(()))
        else:
            loss_kd1, loss_kd2 = self._two_chains(
                lambda: self.criterion_kd(e, path_feat, fuse_feat.detach(), index, sample_idx, ranks=r1),         # :278
                lambda: self.criterion_kd_path(e, path_feat, ema_path_feat.detach(), index, sample_idx, ranks=r2))  # :279
        loss_div1 = opt.alpha * loss_div1; loss_div2 = opt.alpha * loss_div2                                # :293-294
        loss_kd1 = opt.beta * loss_kd1; loss_kd2 = opt.beta * loss_kd2                                      # :296-297
        KD_loss_list = [loss_div1, loss_div2, loss_kd1, loss_kd2]
        if opt.assign_weights == "True" and self.variant == "mia2022":
            # v2 trainer :474-477: momentum GK-Refine; the weight vector lives in ONE device buffer updated in place
            # (so that a captured HIP graph keeps reading / writing the same storage), x len(list) unless thresholded
            scale, loss_KD = self._momentum_gk(loss_cls, path_feat, KD_loss_list)
            if opt.grads_thresh == "False":
                loss_KD = loss_KD * len(KD_loss_list)
        elif opt.assign_weights == "True":
            scale, loss_KD = AEKD_loss(opt, self.optimizer, loss_cls, path_feat, KD_loss_list, self.sync)   # :304
        else:
            scale = None
            loss_KD = loss_div1 + loss_div2 + loss_kd1 + loss_kd2                                           # :309
        loss = self._add_reg(opt.lambda_nll * loss_cls + loss_KD)                                           # :312-313
        self.optimizer.zero_grad()                                                                          # :326
        loss.backward()                                                                                     # :327
        if self.sync is not None:
            self.sync.all_reduce_grads(self.optimizer.flat)
        self.optimizer.step()                                                                               # :328 (+ :329 fused)
        return dict(loss=loss.detach(), loss_cls=loss_cls.detach(), loss_div1=loss_div1.detach(),
                    loss_div2=loss_div2.detach(), loss_kd1=loss_kd1.detach(), loss_kd2=loss_kd2.detach(),
                    scale=scale, logit_path=logit_path.detach(), pred_path=pred_path.detach(),
                    path_feat=path_feat.detach(), ema_logit=ema_logit_path, fuse_logit=logits[-1],
                    fuse_feat=fuse_feat, ema_feat=ema_path_feat)

    def _mia2022_baseline_tail(self, loss_cls, path_feat, logit_path, pred_path, ema_path_feat, ema_logit_path, fuse_feat,
                               logits):
        """The `--distill kd | feats_KL | rkd | pkt | similarity` baselines of the MIA-2022 trainer
        ("MIA 2022/train_test_path_multi_distill_v2.py":419-483): KL to one or two teachers plus one feature-level
        criterion between the student feature and the fused teacher feature, summed with fixed weights alpha / beta
        (GK-Refine needs the per-loss list, which the trainer only builds for crd and kd with two teachers)."""
        opt = self.opt
        z = torch.zeros((), device=self.device)
        loss_div1 = loss_div2 = z
        if opt.num_teachers == 2:                                                                           # :419-422
            loss_div1 = self.criterion_div(logit_path, logits[-1].detach())
            loss_div2 = self.criterion_div(logit_path, ema_logit_path.detach())
        elif opt.num_teachers == 1 and opt.which_teacher == "fuse":
            loss_div1 = self.criterion_div(logit_path, logits[-1].detach())
        elif opt.num_teachers == 1 and opt.which_teacher == "self_EMA":
            loss_div2 = self.criterion_div(logit_path, ema_logit_path.detach())
        else:
            raise NotImplementedError("num_teachers %r / which_teacher %r" % (opt.num_teachers, opt.which_teacher))
        loss_div = loss_div1 + loss_div2
        loss_kd = z if self.zoo_kd is False else self.zoo_kd(path_feat, fuse_feat.detach()).reshape(())     # :430-457
        scale = None
        if opt.assign_weights == "True":
            if not (opt.distill == "kd" and opt.num_teachers == 2):
                raise NotImplementedError("assign_weights with --distill %s: the trainer builds KD_loss_list only for crd and "
                                          "kd with two teachers (:463-468)" % opt.distill)
            KD_loss_list = [opt.alpha * loss_div1, opt.alpha * loss_div2]
            scale, loss_KD = self._momentum_gk(loss_cls, path_feat, KD_loss_list)                           # :474
            if opt.grads_thresh == "False":
                loss_KD = loss_KD * len(KD_loss_list)
        else:
            loss_KD = opt.alpha * loss_div + opt.beta * loss_kd                                             # :482
        loss = self._add_reg(opt.lambda_nll * loss_cls + loss_KD)                                           # :485-486
        self.optimizer.zero_grad()
        loss.backward()
        if self.sync is not None:
            self.sync.all_reduce_grads(self.optimizer.flat)
        self.optimizer.step()
        return dict(loss=loss.detach(), loss_cls=loss_cls.detach(), loss_div1=loss_div1.detach(), loss_div2=loss_div2.detach(),
                    loss_kd1=(opt.beta * loss_kd).detach(), loss_kd2=z, scale=scale, logit_path=logit_path.detach(),
                    pred_path=pred_path.detach(), path_feat=path_feat.detach(), ema_logit=ema_logit_path,
                    fuse_logit=logits[-1], fuse_feat=fuse_feat, ema_feat=ema_path_feat)

    def _branch_check(self):
        """Option combinations the reference's batch body itself cannot run fail here, with its own exception type."""
        opt = self.opt
        if self.zoo_kd is not None:
            return
        if not (opt.num_teachers == 2 or (opt.num_teachers == 1 and opt.which_teacher in ("fuse", "self_EMA"))):
            # :263-271 define loss_div in three branches only; anything else reaches `loss_KD = opt.alpha * loss_div` unbound
            raise UnboundLocalError("local variable 'loss_div' referenced before assignment (num_teachers %r / which_teacher "
                                    "%r: train_test_path_multi_distill.py:263-271)" % (opt.num_teachers, opt.which_teacher))
        if opt.distill not in ("crd", "kd"):
            raise NotImplementedError(opt.distill)                                                          # :289-290
        if opt.assign_weights == "True" and opt.num_teachers != 2:
            # :293-301 build KD_loss_list under `if opt.num_teachers == 2` only; :304 then reads it
            raise UnboundLocalError("local variable 'KD_loss_list' referenced before assignment (--assign_weights True needs "
                                    "--num_teachers 2: train_test_path_multi_distill.py:293-304)")

    def _branch_tail(self, e, r1, grade, index, sample_idx, loss_cls, path_feat, logit_path, pred_path, ema_path_feat,
                     ema_logit_path, fuse_feat, logits):
        """The batch body's non-default option branches, all three trainers (MICCAI-2022 train_test_path_multi_distill.py:263-309,
        "MIA 2022/train_test_path_multi_distill_v2.py":419-482, "MIA 2023/stage2_unimodal_student/
        train_test_path_multi_distill.py":348-427): `--num_teachers 1` with `--which_teacher fuse | self_EMA` (ONE KL term and ONE
        CRD call - through criterion_kd whichever teacher is chosen - summed with the fixed weights alpha / beta) and
        `--distill kd` (no CRD term; with two teachers and `--assign_weights True` GK-Refine over the two KL terms).  Generic
        autograd path; every value stays on the device."""
        opt, v = self.opt, self.variant
        z = torch.zeros((), device=self.device)
        two = opt.num_teachers == 2
        fuse_on = two or opt.which_teacher == "fuse"
        ema_on = two or opt.which_teacher == "self_EMA"
        rows = v == "mia2023"                   # (its DistillKL / CRDLoss also return the per-sample rows)
        loss_div1 = loss_div2 = loss_kd1 = loss_kd2 = z
        rows_div1 = rows_div2 = rows_kd1 = rows_kd2 = None
        if fuse_on:
            r = self.criterion_div(logit_path, logits[-1].detach())
            loss_div1, rows_div1 = r if rows else (r, None)
        if ema_on:
            r = self.criterion_div(logit_path, ema_logit_path.detach())
            loss_div2, rows_div2 = r if rows else (r, None)
        loss_div = loss_div1 + loss_div2
        if opt.distill == "crd":
            if rows:
                from .mia2023 import assign_sample_weights
                w1 = assign_sample_weights(logit_path, logits[-1], grade, opt.discrep_scale, opt.max_discrep, from_logits=True)
                w2 = assign_sample_weights(logit_path, ema_logit_path, grade, opt.discrep_scale, opt.max_discrep, from_logits=True)
                w1 = (1.0 + e * w1).view(-1, 1); w2 = (1.0 + e * w2).view(-1, 1)
            # one teacher: the call goes through criterion_kd with that teacher's feature (MICCAI :282-285)
            t_feat, w = (fuse_feat, w1 if rows else None) if fuse_on else (ema_path_feat, w2 if rows else None)
            if v == "miccai2022":
                loss_kd1 = self.criterion_kd(e, path_feat, t_feat.detach(), index, sample_idx, ranks=r1)
            elif v == "mia2022":
                loss_kd1 = self.criterion_kd(e, path_feat, t_feat.detach(), index, sample_idx).reshape(())
            else:
                loss_kd1, rows_kd1 = self.criterion_kd(w, path_feat, t_feat.detach(), grade, index, sample_idx)
        loss_kd = loss_kd1 + loss_kd2
        scale = None
        if opt.assign_weights == "True":       # (two teachers, --distill kd: _branch_check)
            if v == "miccai2022":
                KD_loss_list = [opt.alpha * loss_div1, opt.alpha * loss_div2]                               # :293-300
                scale, loss_KD = AEKD_loss(opt, self.optimizer, loss_cls, path_feat, KD_loss_list, self.sync)
            elif v == "mia2022":
                KD_loss_list = [opt.alpha * loss_div1, opt.alpha * loss_div2]
                scale, loss_KD = self._momentum_gk(loss_cls, path_feat, KD_loss_list)
                if opt.grads_thresh == "False":
                    loss_KD = loss_KD * len(KD_loss_list)
            else:
                from .mia2023 import GK_refine_thresh
                if getattr(opt, "loss_weighting", "GK_refine") != "GK_refine":
                    raise NotImplementedError("loss_weighting '%s' (the shipped command uses GK_refine)" % opt.loss_weighting)
                KD_loss_list = [rows_div1, rows_div2]                                                       # (:414: unscaled rows)
                scale, loss_KD = GK_refine_thresh(opt, self.optimizer, loss_cls, path_feat, KD_loss_list,
                                                  batch_norm_size=self.criterion_div.batch_norm_size, sync=self.sync)
        else:
            loss_KD = opt.alpha * loss_div + opt.beta * loss_kd                                             # :309
        loss = self._add_reg(opt.lambda_nll * loss_cls + loss_KD)                                           # :312-313
        self.optimizer.zero_grad()
        loss.backward()
        if self.sync is not None:
            self.sync.all_reduce_grads(self.optimizer.flat)
        self.optimizer.step()
        dt = lambda t: t.detach() if torch.is_tensor(t) else t      # noqa: E731
        return dict(loss=loss.detach(), loss_cls=loss_cls.detach(), loss_div1=dt(opt.alpha * loss_div1),
                    loss_div2=dt(opt.alpha * loss_div2), loss_kd1=dt(opt.beta * loss_kd1), loss_kd2=dt(opt.beta * loss_kd2),
                    scale=scale, logit_path=logit_path.detach(), pred_path=pred_path.detach(), path_feat=path_feat.detach(),
                    ema_logit=ema_logit_path, fuse_logit=logits[-1], fuse_feat=fuse_feat, ema_feat=ema_path_feat)

    def _add_reg(self, loss):
        """+ opt.lambda_reg * define_reg(opt, model) (train_test_path_multi_distill.py:312-313)."""
        if not self._reg_on:
            return loss
        from .networks_new import define_reg
        # replicas SUM their losses / gradients (global-batch normalisers): the batch-independent L1 term is split evenly
        # so that the sum carries it once, as the reference's one DataParallel process does
        w = self.sync.world_size if self.sync is not None else 1
        return loss + (self.opt.lambda_reg / w) * define_reg(self.opt, self.model)

    def _fused_head_ok(self):
        """The closed-form loss head covers the shipped MICCAI and MIA-2022 stage-2 commands: two teachers, CRD, GK-Refine
        (plain / momentum) with the CE gradient, a log-softmax grading head.  `opt.fused_loss_head = False` selects the
        generic autograd path."""
        opt = self.opt
        if getattr(opt, "sample_KD", "False") != "False":
            # per-sample CRD rows ([B]-shaped loss_kd): the reference's body then fails inside AEKD_loss's `loss_t.backward()`
            # ("grad can be implicitly created only for scalar outputs") - the generic path fails the same way, the closed
            # form would silently compute something else
            return False
        if self._reg_on:
            # the fused head WRITES fc_new2's gradients (accumulate=False) after L1RegFn.backward has added
            # lambda_reg * sgn(W) into the same flat buffer: the L1 term on fc_new2 would be lost
            return False
        if self.variant != "miccai2022" and (self.zoo_kd is not None or not torch.is_tensor(getattr(self, "_e_dev", None))):
            return False
        if self.variant == "mia2023" and (getattr(opt, "loss_weighting", "GK_refine") != "GK_refine"
                                          or getattr(opt, "pos_extra", "neighbors") != "neighbors"):
            return False
        return (self.variant in ("miccai2022", "mia2022", "mia2023") and getattr(opt, "fused_loss_head", True)
                and opt.assign_weights == "True"
                and bool(opt.CE_grads) and opt.num_teachers == 2 and opt.distill == "crd"
                and isinstance(getattr(self.model, "fc_new2", None), nn.Linear)
                and isinstance(getattr(self.model, "act", None), nn.LogSoftmax))

    def _mia2023_tail(self, rw, loss_cls, grade, index, sample_idx, path_feat, logit_path, pred_path, ema_path_feat,
                      ema_logit_path, fuse_feat, logits):
        """Losses, backward and update of the MIA-2023 batch body
        ("MIA 2023/stage2_unimodal_student/train_test_path_multi_distill.py":348-448).  `rw` is a device scalar:
        0 while epoch < opt.start_reweight (query weights are all ones, :373-375), 1 afterwards (1 + discrepancy)."""
        from .mia2023 import assign_sample_weights, GK_refine_thresh
        opt = self.opt
        loss_div1, rows_div1 = self.criterion_div(logit_path, logits[-1].detach())                          # :349
        loss_div2, rows_div2 = self.criterion_div(logit_path, ema_logit_path.detach())                      # :350
        # :361-364 - log-probability margins: the softmax normaliser cancels, so the kernel reads the logits
        w1 = assign_sample_weights(logit_path, logits[-1], grade, opt.discrep_scale, opt.max_discrep, from_logits=True)
        w2 = assign_sample_weights(logit_path, ema_logit_path, grade, opt.discrep_scale, opt.max_discrep, from_logits=True)
        w1 = (1.0 + rw * w1).view(-1, 1); w2 = (1.0 + rw * w2).view(-1, 1)                                  # :373-382
        (loss_kd1, rows_kd1), (loss_kd2, rows_kd2) = self._two_chains(
            lambda: self.criterion_kd(w1, path_feat, fuse_feat.detach(), grade, index, sample_idx),            # :386
            lambda: self.criterion_kd_path(w2, path_feat, ema_path_feat.detach(), grade, index, sample_idx))    # :388
        KD_loss_list = [opt.alpha * rows_div1, opt.alpha * rows_div2, opt.beta * rows_kd1, opt.beta * rows_kd2]  # :407-414
        if opt.assign_weights == "True":
            if getattr(opt, "loss_weighting", "GK_refine") != "GK_refine":
                raise NotImplementedError("loss_weighting '%s' (the shipped command uses GK_refine)" % opt.loss_weighting)
            scale, loss_KD = GK_refine_thresh(opt, self.optimizer, loss_cls, path_feat, KD_loss_list,        # :422
                                              batch_norm_size=self.criterion_div.batch_norm_size, sync=self.sync)
        else:
            scale = None
            loss_KD = opt.alpha * (loss_div1 + loss_div2) + opt.beta * (loss_kd1 + loss_kd2)                # :427
        loss = self._add_reg(opt.lambda_nll * loss_cls + loss_KD)                                           # :430-431
        self.optimizer.zero_grad()
        loss.backward()
        if self.sync is not None:
            self.sync.all_reduce_grads(self.optimizer.flat)
        self.optimizer.step()                                                                               # :446-447 fused
        return dict(loss=loss.detach(), loss_cls=loss_cls.detach(), loss_div1=loss_div1.detach(),
                    loss_div2=loss_div2.detach(), loss_kd1=loss_kd1.detach(), loss_kd2=loss_kd2.detach(),
                    rows_div1=rows_div1.detach(), rows_kd1=rows_kd1.detach(), w1=w1.detach(), w2=w2.detach(),
                    scale=scale, logit_path=logit_path.detach(), pred_path=pred_path.detach(),
                    path_feat=path_feat.detach(), ema_logit=ema_logit_path, fuse_logit=logits[-1],
                    fuse_feat=fuse_feat, ema_feat=ema_path_feat)

    def _momentum_gk(self, main_loss, feat_s, loss_t_list):
        """momentum_AEKD_loss ("MIA 2022/train_test_path_multi_distill_v2.py":89-132) with persistent device state."""
        opt = self.opt
        losses = list(loss_t_list) + ([main_loss] if opt.CE_grads else [])
        grads = [torch.autograd.grad(l, feat_s, retain_graph=True)[0] for l in losses]
        ng = len(grads)
        G = torch.stack([g.reshape(-1) for g in grads]).contiguous()
        gram = torch.empty(ng * ng, device=G.device, dtype=torch.float32)
        check(lib().ph_gram(ptr(G), ptr(gram), ng, G.shape[1], stream()), "ph_gram")
        if self.sync is not None:
            self.sync.all_reduce_sum(gram)
        if self._mo_state is None:
            self._mo_state = torch.zeros(ng, device=G.device, dtype=torch.float32)
            self._mo_init = torch.zeros(1, device=G.device, dtype=torch.int32)
        check(lib().ph_gk_scale_momentum(ptr(gram), ng, 1 if opt.grads_thresh == "True" else 0, float(opt.thresh),
                                         float(opt.grads_m), ptr(self._mo_state), ptr(self._mo_init), stream()),
              "ph_gk_scale_momentum")
        self._mo_init.fill_(1)
        scale = self._mo_state.clone()       # a value snapshot for the caller; no graph through the weights
        total_KD_loss = torch.dot(scale[:-1], torch.stack([l.reshape(()) for l in loss_t_list]))
        return scale, total_KD_loss

    def _draw_ranks(self, epoch, ranks):
        """Host-RNG rank draws of memory_new.py:311 for the two CRD calls (kd1 then kd2), as device int32."""
        e = epoch / self.opt.niter_decay
        if self.variant != "miccai2022":
            return e, [None, None]         # the vanilla / KNN banks have no rank-drawn pair selection
        out = []
        # the host RNG advances once per CRD CALL (memory_new.py:311): two calls in the shipped command, one with
        # --num_teachers 1 (through criterion_kd), none with --distill kd
        ncalls = 0 if self.opt.distill != "crd" else (2 if self.opt.num_teachers == 2 else 1)
        for i, crd in enumerate((self.criterion_kd, self.criterion_kd_path)):
            if i >= ncalls:
                out.append(None)
                continue
            r = ranks[i] if ranks is not None else crd.contrast.draw_ranks(e, crd.select_pos_mode)
            out.append(None if r is None else torch.as_tensor(np.asarray(r), dtype=torch.int32))
        return e, out

    def enable_graph(self):
        """Replay the device part of the step from ONE captured HIP graph (the step is ~700 kernel launches;
        eager dispatch makes it host-bound).  Call after at least two eager steps (first-call CRD Z, lazy plans);
        shapes must stay fixed afterwards."""
        self._want_graph = True

    def step(self, batch, epoch=0, ranks=None):
        opt = self.opt
        # :231-232 re-asserted per call: the reference sets train mode at every epoch because its test() leaves the
        # networks in eval mode (evaluate.test does the same here); the EMA model is never put in eval mode there
        self.module_list.train(); self.fix_model.train()
        self._capture_pre = None
        if batch is None:
            # on-device input pipeline (augment.ResidentTileLoader, `step.loader = loader`): the batch is produced from the
            # resident tile store.  In graph mode the loader's launches are captured IN FRONT of the step's own, so a
            # replay is the whole iteration - shuffle indices, both augmented views, contrast indices, step - and the host
            # launches nothing in between (eager launches between replays cost 1-4 ms per step on this runtime).
            loader = getattr(self, "loader", None)
            if loader is None:
                raise ValueError("step(None) needs step.loader (augment.ResidentTileLoader)")
            in_graph = getattr(self, "_want_graph", False) and self.iter_num - opt.global_step >= 2
            if getattr(self, "_loader_bt", None) is None:
                self._loader_bt = loader.next(opt.batch_size)          # allocates the resident input set
            elif not in_graph:
                loader.next(into=self._loader_bt)
            batch = self._loader_bt
            if in_graph:
                self._capture_pre = lambda: loader.next(into=self._loader_bt)
        (x_path, ema_x_path), x_grph, x_omic, censor, survtime, grade, index, sample_idx = batch
        dev = self.device
        if sample_idx is None:
            # the loader's contrast-index draw (data_loaders_MT.py:229-249) on the device: set `step.sampler` to a
            # ContrastIndexSampler built from the training labels; one launch per step, outside the captured graph
            if getattr(self, "sampler", None) is None:
                raise ValueError("the batch carries no sample_idx and no step.sampler (ContrastIndexSampler) is set")
            buf = getattr(self, "_sample_idx_buf", None)
            if buf is None or buf.shape[0] != index.shape[0]:
                buf = self._sample_idx_buf = torch.empty(index.shape[0], self.sampler.width, device=dev, dtype=torch.int64)
            sample_idx = self.sampler(index, grade, out=buf)
        bnorm = float(x_path.shape[0] * (self.sync.world_size if self.sync is not None else 1))
        self.criterion_div.batch_norm_size = bnorm
        self.criterion_kd.contrast.batch_norm_size = bnorm
        self.criterion_kd_path.contrast.batch_norm_size = bnorm
        e, rk = self._draw_ranks(epoch, ranks)
        if self.variant == "mia2022":
            # the epoch weight multiplies the CRD loss on the device: one persistent scalar so a captured graph
            # reads the current value instead of baking the capture-time epoch in
            if getattr(self, "_e_dev", None) is None:
                self._e_dev = torch.zeros(1, device=dev, dtype=torch.float32)
            self._e_dev.fill_(float(e))
            e = self._e_dev
        elif self.variant == "mia2023":
            if getattr(self, "_e_dev", None) is None:
                self._e_dev = torch.zeros(1, device=dev, dtype=torch.float32)
            self._e_dev.fill_(0.0 if epoch < opt.start_reweight else 1.0)         # the re-weighting switch (:373)
            e = self._e_dev
        self.optimizer.ema_alpha = min(1 - 1 / (self.iter_num + 1), opt.ema_decay)                          # :36
        use_graph = getattr(self, "_want_graph", False) and self.iter_num - opt.global_step >= 2
        if not use_graph:
            x_path = x_path.to(dev, non_blocking=True)                             # ONE H2D copy (reference: three)
            ema_x_path = ema_x_path.to(dev, non_blocking=True)
            x_omic = x_omic.to(dev, non_blocking=True)
            grade = grade.to(dev, non_blocking=True)
            index = index.to(dev, non_blocking=True)
            sample_idx = sample_idx.to(dev, non_blocking=True)
            r1, r2 = [None if r is None else r.to(dev, non_blocking=True) for r in rk]
            self.optimizer.prepare_step()
            out = self._device_body(x_path, ema_x_path, x_omic, grade, index, sample_idx, bnorm, e, r1, r2)
            self.iter_num += 1
            return out
        # ---- graph path: per-step scalars in device memory, inputs at fixed addresses.  A captured graph reads its
        # inputs from the addresses it was captured with.  If the caller hands over DEVICE tensors (a loader's ring of
        # resident batch buffers), up to two such input sets are adopted as they are - one captured graph each, sharing
        # one memory pool - and replayed without any copy; anything else (host tensors, more than two sets) is staged
        # into the first set's buffers by device-to-device / host-to-device copies (2 x 76 us for two 201 MB views).
        given = dict(zip(self._IN_NAMES, (x_path, ema_x_path, x_omic, grade, index, sample_idx)))
        st = self._ensure_slot(given, rk, bnorm, e)
        if st is None:                       # capture failed: eager from now on
            out = self._device_body(*[given[k].to(dev) for k in self._IN_NAMES], bnorm, e,
                                    *[None if r is None else r.to(dev) for r in rk])
            self.iter_num += 1
            return out
        self._static = st
        for name in self._IN_NAMES:
            if given[name].data_ptr() != st[name].data_ptr():
                st[name].copy_(given[name], non_blocking=True)
        for j, (dst, src) in enumerate(zip(st["r"], rk)):
            if dst is not None:
                ring = st.setdefault("r_ring", {}).get(j)
                if ring is None:
                    ring = st["r_ring"][j] = PinnedRing(tuple(dst.shape), torch.int32)
                ring.upload(dst, src)
        self.optimizer.prepare_step()
        self.optimizer._prepared = False
        st["graph"].replay()
        self.iter_num += 1
        return st["out"]

    _IN_NAMES = ("x_path", "ema_x_path", "x_omic", "grade", "index", "sample_idx")

    def _ensure_slot(self, given, r_like, bnorm, e):
        """The input set (static buffers + captured graph) that serves `given`; captures its graph if needed."""
        dev = self.device
        names = self._IN_NAMES
        on_dev = all(t.is_cuda and t.is_contiguous() for t in given.values())
        ptrs = tuple(t.data_ptr() for t in given.values()) if on_dev else None
        shapes = (tuple(given["x_path"].shape), tuple(given["x_omic"].shape), tuple(given["sample_idx"].shape))
        slots = getattr(self, "_slots", None)
        if slots is None or (slots and slots[0]["key"] != shapes):
            slots = self._slots = []
        st = next((q for q in slots if ptrs is not None and q["ptrs"] == ptrs), None)
        if st is None and (not slots or (on_dev and len(slots) < 2)):
            if on_dev:
                bufs = dict(given)                    # adopted: the caller refills these tensors in place
            else:
                bufs = {k: torch.empty(t.shape, device=dev, dtype=t.dtype) for k, t in given.items()}
            st = dict(key=shapes, ptrs=tuple(bufs[k].data_ptr() for k in names), graph=None, out=None,
                      r=[None if r is None else torch.empty(tuple(r.shape), device=dev, dtype=torch.int32) for r in r_like],
                      **bufs)
            slots.append(st)
        if st is None:
            st = slots[0]
        if st["graph"] is None:
            lib().ph_prof_enable(0)          # event timing is an eager-mode facility
            g = torch.cuda.CUDAGraph()
            torch.cuda.synchronize()
            pool = next((q["graph"].pool() for q in slots if q["graph"] is not None), None)
            was_prepared = self.optimizer._prepared
            try:
                with torch.cuda.graph(g, pool=pool, capture_error_mode=_capture_mode(self.sync)):
                    self.optimizer._prepared = True       # the step scalars are read from device memory at replay
                    if getattr(self, "_capture_pre", None) is not None:
                        self._capture_pre()               # the on-device input pipeline fills this input set first
                    st["out"] = self._device_body(st["x_path"], st["ema_x_path"], st["x_omic"], st["grade"],
                                                  st["index"], st["sample_idx"], bnorm, e, st["r"][0], st["r"][1])
                st["graph"] = g
                # the graph holds raw pointers into the trunk workspaces it was captured with: keep them alive with it
                st["ws_refs"] = [ws for net in (self.model, self.ema_model, getattr(self.fix_model, "path_net", None))
                                 if net is not None and hasattr(net, "pinned_workspaces") for ws in net.pinned_workspaces()]
                self.optimizer._prepared = was_prepared   # capture does not execute anything
            except Exception as exc:     # e.g. a collective that cannot be captured on this stack: stay eager
                import warnings
                warnings.warn("HIP graph capture of the distill step failed (%r); continuing with eager launches" % (exc,))
                torch.cuda.synchronize()
                try:      # on this HIP runtime a failed capture can leave its streams in capture state for good
                    torch.ones(1).to(self.device)
                except Exception as exc2:
                    raise RuntimeError("HIP graph capture of the distill step failed (%r) and the runtime did not leave capture "
                                       "mode (%r): restart the process without enable_graph()" % (exc, exc2)) from exc
                self._want_graph = False
                self._static = None
                self._slots = None
                self.optimizer._prepared = was_prepared
                return None
        return st

    def precapture(self, batch, epoch=0):
        """Capture the graph for another resident input set without running a step (e.g. the second buffer of a
        loader's two-deep ring, so that its first use does not pay the capture).  Needs one graph step done before."""
        slots = getattr(self, "_slots", None)
        if not slots or slots[0]["graph"] is None:
            raise RuntimeError("precapture: run a graph-mode step first")
        (x_path, ema_x_path), _, x_omic, _, _, grade, index, sample_idx = batch
        given = dict(zip(self._IN_NAMES, (x_path, ema_x_path, x_omic, grade, index, sample_idx)))
        bnorm = float(x_path.shape[0] * (self.sync.world_size if self.sync is not None else 1))
        e = getattr(self, "_e_dev", None) if self.variant != "miccai2022" else epoch / self.opt.niter_decay
        return self._ensure_slot(given, slots[0]["r"], bnorm, e) is not None

    # ------------------------------------------------------------------ checkpoint / resume
    def state_dict(self):
        """Everything a bit-identical resume needs.  The reference saves `model_state_dict` + `optimizer_state_dict`
        only (train_test_path_multi_distill.py:387-402) and silently loses the CRD banks, their normalisation
        constants Z, the embed heads' optimiser state, the EMA model and the step counter; the first two keys are kept
        reference-compatible, the rest is added."""
        rng = {}
        for name, mod in self.fix_model.named_modules():
            if hasattr(mod, "rng_step"):
                rng[name] = mod.rng_step.clone()
        return dict(model_state_dict=self.model.state_dict(), optimizer_state_dict=self.optimizer.state_dict(),
                    ema_model_state_dict=self.ema_model.state_dict(), teacher_state_dict=self.fix_model.state_dict(),
                    crd_kd_state_dict=self.criterion_kd.state_dict(),
                    crd_kd_path_state_dict=self.criterion_kd_path.state_dict(),
                    scheduler_state_dict=self.scheduler.state_dict(), iter_num=self.iter_num,
                    teacher_rng_steps=rng,
                    gk_momentum_scale=None if self._mo_state is None else self._mo_state.clone(),
                    # draw counters of the on-device input pipeline and the host RNG behind the CRD rank draws
                    # (memory_new.py:311): without them a resumed run with step(None) / step.sampler draws other batches
                    input_rng=self._input_rng_state(),
                    # step counters / seeds of the `contrast_idx is None` draws (CL_utils/memory_new.draw_uniform_indices)
                    crd_draw_steps=[(None if getattr(c.contrast, "_draw_step", None) is None else c.contrast._draw_step.clone(),
                                     getattr(c.contrast, "_draw_seed", None))
                                    for c in (self.criterion_kd, self.criterion_kd_path)])

    def _input_rng_state(self, load=None):
        """Device-side draw counters of step.sampler / step.loader (ContrastIndexSampler.step, DeviceAugment.step,
        ResidentTileLoader.batch_no) and numpy's global RNG state; `load` restores a dict this method returned."""
        loader = getattr(self, "loader", None)
        cells = dict(sampler=getattr(getattr(self, "sampler", None), "step", None),
                     loader_sampler=getattr(getattr(loader, "sampler", None), "step", None),
                     loader_aug=getattr(getattr(loader, "aug", None), "step", None),
                     loader_batch_no=getattr(loader, "batch_no", None))
        if load is None:
            out = {k: (None if t is None else t.clone()) for k, t in cells.items()}
            # numpy's MT19937 state as tensors / plain scalars: `torch.load(weights_only=True)` (the default since
            # torch 2.6) rejects numpy globals, so the raw get_state() tuple would make the checkpoint unloadable
            name, key, pos, has_gauss, cached = np.random.get_state()
            out["numpy"] = dict(name=str(name), key=torch.from_numpy(key.astype(np.int64)), pos=int(pos),
                                has_gauss=int(has_gauss), cached_gaussian=float(cached))
            return out
        for k, t in cells.items():
            v = load.get(k)
            if v is None:
                continue
            if t is None:
                if k == "loader_batch_no" and loader is not None:
                    loader.batch_no = v.to(self.device).clone()
                    continue
                raise RuntimeError("checkpoint carries the draw counter %r but this step has no such component" % k)
            t.copy_(v)
        st = load.get("numpy")
        if st is not None:
            # NOTE: restores the PROCESS-GLOBAL numpy RNG (the reference draws its CRD ranks from it, memory_new.py:311)
            if isinstance(st, dict):
                st = (st["name"], np.asarray(st["key"].cpu().numpy(), dtype=np.uint32), int(st["pos"]), int(st["has_gauss"]),
                      float(st["cached_gaussian"]))
            np.random.set_state(st)

    def load_state_dict(self, sd):
        self.model.load_state_dict(sd["model_state_dict"])
        self.ema_model.load_state_dict(sd["ema_model_state_dict"])
        self.fix_model.load_state_dict(sd["teacher_state_dict"])
        for crd, key in ((self.criterion_kd, "crd_kd_state_dict"), (self.criterion_kd_path, "crd_kd_path_state_dict")):
            crd.load_state_dict(sd[key])
            crd.contrast._z_set = bool((crd.contrast.params[2:4] > 0).all().item())
        self.optimizer.load_state_dict(sd["optimizer_state_dict"])
        self.scheduler.load_state_dict(sd["scheduler_state_dict"])
        self.iter_num = sd["iter_num"]
        mo = sd.get("gk_momentum_scale")
        if mo is not None:
            if self._mo_state is None:
                self._mo_state = torch.zeros_like(mo, device=self.device)
                self._mo_init = torch.ones(1, device=self.device, dtype=torch.int32)
            self._mo_state.copy_(mo); self._mo_init.fill_(1)
        for name, mod in self.fix_model.named_modules():
            if hasattr(mod, "rng_step") and name in sd.get("teacher_rng_steps", {}):
                mod.rng_step.copy_(sd["teacher_rng_steps"][name])
        if sd.get("input_rng") is not None:
            self._input_rng_state(load=sd["input_rng"])
        for c, rec in zip((self.criterion_kd, self.criterion_kd_path), sd.get("crd_draw_steps") or ()):
            if rec[0] is not None:
                c.contrast._draw_step = rec[0].to(self.device).clone()
                c.contrast._draw_seed = rec[1]
        ops.bump_weight_epoch()      # packed MFMA weight images are rebuilt from the loaded parameters
        self._static = None          # a captured graph keeps pointing at valid buffers, but is rebuilt to be safe
        self._slots = None

    def static_inputs(self):
        """The graph path's resident input buffers (fill them in place to skip the device-to-device copy)."""
        st = getattr(self, "_static", None)
        return None if st is None else {k: st[k] for k in ("x_path", "ema_x_path", "x_omic", "grade", "index", "sample_idx")}


def _capture_mode(sync):
    """Stream-capture error mode of the step graphs.  Under data parallelism the process group's watchdog thread keeps polling
    the completion events of earlier collectives (hipEventQuery); in the default "global" mode that call from ANOTHER thread is
    illegal while a capture runs and takes the process down (seen intermittently with `bench.py --force-dist`): "thread_local"
    restricts the check to the capturing thread.  Kernels enqueued on the capturing stream by other threads (autograd's
    backward worker) are captured either way."""
    return "thread_local" if sync is not None else "global"


class _GatherRowsFn(torch.autograd.Function):
    """All-gather of feature rows with a local backward: forward returns the rows of every replica (rank order),
    backward returns the gradient of THIS replica's rows.  Every replica evaluates the same function of the gathered
    rows (the t-SVD penalty), so the partial derivative with respect to its own rows is complete as it stands; the
    gradient all-reduce then adds the replicas' parameter gradients."""

    @staticmethod
    def forward(ctx, x, sync):
        ctx.lo, ctx.n = sync.rank * x.shape[0], x.shape[0]
        return sync.all_gather_cat(x.detach().contiguous())

    @staticmethod
    def backward(ctx, g):
        return g[ctx.lo:ctx.lo + ctx.n].contiguous(), None


class TeacherStage1Step:
    """The batch body of the stage-1 mean-teacher trainer (MICCAI-2022/train_test_MT.py:121-230) for the
    grading task: student PathomicNet forward/backward, EMA PathomicNet forward, three-branch NLL (:208-212),
    `pred_KD_loss` consistency (CL_utils/KD_losses.py:12-36; --num_teachers 1/2/3, :180-201), the optional vanilla
    CRD term on the fused features (--CRD_distill 1, CL_utils/CRD_criterion.py) and orthogonality term (--orth_loss
    True, CL_utils/orthogonal_loss.py), Adam, EMA update.  (--SP_distill names a class the reference never imports.)"""

    def __init__(self, opt, device="cuda", k=1, models=None, sync=None):
        """`sync`: a dist.ReplicaSync for data parallelism (one process per GPU, `opt.batch_size` tiles each).  Loss
        normalisers use the global batch, the vanilla CRD bank is updated identically on every replica, the
        cross-correlation of the orthogonality loss is all-reduced, and the t-SVD adjacency tensors are built over the
        global batch from all-gathered feature views (every replica computes the same auxiliary tensors; the penalty's
        gradient reaches this replica's rows only, the gradient all-reduce adds the replicas up)."""
        from .networks_new import define_net, define_optimizer, define_scheduler
        self.opt = opt
        self.device = torch.device(device)
        self.sync = sync
        _validate_opt(opt, "TeacherStage1Step")
        if models is None:
            self.model = define_net(opt, k).to(self.device)
            self.ema_model = define_net(opt, k).to(self.device)
        else:
            self.model, self.ema_model = models
        for p in self.ema_model.parameters():
            p.detach_()                                                             # train_test_MT.py:74-76
        # optional terms: vanilla CRD between the student's and the EMA teacher's fused features (:74-76,157-165; the
        # reference builds three criteria, appends all six embed heads to the optimiser and uses the `fuse` one) and the
        # orthogonality loss between the path and omic features (:79,216-218)
        self.crd_on = getattr(opt, "CRD_distill", 0) == 1
        self.orth_on = getattr(opt, "orth_loss", "False") == "True"
        if getattr(opt, "SP_distill", 0) == 1:
            raise NotImplementedError("--SP_distill: the reference's stage-1 trainer cannot run it either (its `Similarity` "
                                      "import is commented out, train_test_MT.py:27,169)")
        module_list = nn.ModuleList([self.model])
        if self.crd_on:
            from .CL_utils.CRD_criterion import CRDLoss as CRDLossV0
            self.CRD_criterion_path = CRDLossV0(opt).to(self.device)
            self.CRD_criterion_omic = CRDLossV0(opt).to(self.device)
            self.CRD_criterion_fuse = CRDLossV0(opt).to(self.device)
            for c in (self.CRD_criterion_path, self.CRD_criterion_omic, self.CRD_criterion_fuse):
                c.contrast.verbose = False
            # :84-90 appends all six heads to the optimiser, but only the `fuse` pair ever receives a gradient and
            # torch's Adam skips parameters whose .grad is None: the other four never change.  The flat fused Adam has
            # no "None" gradients (it would apply weight decay to them), so they are simply left out.
            module_list.append(self.CRD_criterion_fuse.embed_s); module_list.append(self.CRD_criterion_fuse.embed_t)
        if self.orth_on:
            from .CL_utils.orthogonal_loss import OrthLoss
            self.Orth_loss = OrthLoss()
        # t-SVD low-rank constraint of the MIA-2022 stage-1 trainer ("MIA 2022/train_test_tSVD.py":161-181,299-431):
        # adjacency tensors over n_views feature views per modality, auxiliary tensors from the tensor-nuclear-norm prox
        self.tsvd_on = getattr(opt, "tSVD_loss", "False") == "True"
        if self.tsvd_on:
            if opt.n_views not in (2, 4, 6, 8):
                raise ValueError("n_views %d (train_test_tSVD.py:309-363 builds 2, 4, 6 or 8 views)" % opt.n_views)
            if opt.tSVD_mode not in ("path", "omic", "pathomic"):
                raise ValueError(opt.tSVD_mode)
            self.mu = float(opt.mu)                                                              # :161 (never reset per epoch)
            gb = opt.batch_size * (sync.world_size if sync is not None else 1)
            z = lambda: [torch.zeros(gb, gb, device=self.device) for _ in range(opt.n_views)]
            self.adj_tensor1, self.aux_tensor1, self.adj_tensor2, self.aux_tensor2 = z(), z(), z(), z()   # :165-177
            self.path_TNN = self.omic_TNN = None
            self._batch_idx = 0
        self.optimizer = define_optimizer(opt, module_list if self.crd_on else self.model)      # :86-94
        self.scheduler = define_scheduler(opt, self.optimizer)
        self.iter_num = opt.global_step
        self.model.train(); self.ema_model.train()
        self.ema_flat = FlatParams(list(self.ema_model.parameters()))
        for mod in self.ema_model.modules():
            if hasattr(mod, "_get_packed"):
                mod._follow_epoch = True
        self.optimizer.ema_flat = self.ema_flat
        n_model = len(list(self.model.parameters()))
        flat = self.optimizer.flat
        self.optimizer.ema_range = (0, flat.offsets[n_model] if n_model < len(flat.offsets) else flat.numel)
        if sync is not None:
            crds = (self.CRD_criterion_path, self.CRD_criterion_omic, self.CRD_criterion_fuse) if self.crd_on else ()
            sync.attach_parts(crds, (flat.flat, self.ema_flat.flat), (self.model, self.ema_model))
            if self.orth_on:
                self.Orth_loss.sync = sync

    @staticmethod
    def pred_KD_loss(p_s, p_t, bnorm=None):
        """KD_losses.py:27-29 (grading, sample_KD False): sum(kl_div(p_s, exp(p_t))) / B on log-probabilities - the KL
        kernel at T = 1 (log_softmax of a log-probability vector is the vector itself).  `bnorm`: the global batch
        under data parallelism."""
        return ops.KLFn.apply(p_s, p_t.detach(), 1.0, float(bnorm or p_s.shape[0]))

    def start_epoch(self):
        """The t-SVD auxiliary update runs every opt.aux_iter batches of an epoch (:375): reset the batch counter."""
        self._batch_idx = 0

    def enable_graph(self):
        """Replay the device part of the step from captured HIP graphs (one per resident input set and per kind of step -
        with / without the t-SVD auxiliary update).  Call after at least two eager steps; shapes must stay fixed.  The
        per-step scalars that the eager path bakes into its launches (the CRD weight that drops at epoch 15, the t-SVD
        threshold Lambda / mu and the penalty's mu, Adam's bias corrections, the EMA rate) are read from device memory.
        Steps with the superpixel-masking terms (MIA-2023 stage 1) stay eager: their attention masks are built with host
        control flow."""
        self._want_graph = True

    def step(self, batch, epoch=0, batch_idx=None):
        opt = self.opt
        self.model.train()              # train_test_MT.py:110 `module_list.train()` per epoch (its test() leaves eval mode)
        if epoch >= 15:
            opt.CRD_weight = 0.01                                                                # train_test_MT.py:118-119
        views, x_grph, x_omic, censor, survtime, grade, index, sample_idx = batch
        sp_mask = x_path_m_v1 = x_path_m_v2 = None
        if len(views) == 6:      # the MIA-2023 loader (train_test_MT_SP_Masking.py:185): superpixel maps and two masked views
            x_path, sp_mask, ema_x_path, _, x_path_m_v1, x_path_m_v2 = views
        else:
            x_path, ema_x_path = views
        dev = self.device
        B = float(x_path.shape[0] * (self.sync.world_size if self.sync is not None else 1))   # global batch
        self.optimizer.ema_alpha = min(1 - 1 / (self.iter_num + 1), opt.ema_decay)
        masking_on = bool(getattr(opt, "masking", 0)) and epoch > opt.start_epoch
        # t-SVD schedule on the host: whether this batch updates the auxiliary tensors (:375), the threshold it uses and the
        # mu of the penalty, which is the value AFTER the update's `mu = min(mu * pho, max_mu)` (:413)
        do_aux, tau, mu_pen, bidx = False, 0.0, 0.0, 0
        if self.tsvd_on:
            bidx = self._batch_idx if batch_idx is None else batch_idx
            do_aux = bidx % opt.aux_iter == 0
            tau = opt.Lambda_global / self.mu
            mu_pen = min(self.mu * opt.pho, opt.max_mu) if do_aux else self.mu
        given = dict(x_path=x_path, ema_x_path=ema_x_path, x_omic=x_omic, grade=grade, index=index, sample_idx=sample_idx)
        use_graph = (getattr(self, "_want_graph", False) and not masking_on and self.iter_num - opt.global_step >= 2
                     and all(torch.is_tensor(t) for t in given.values()))
        out = None
        if use_graph:
            out = self._graph_step(given, B, do_aux, float(getattr(opt, "CRD_weight", 1.0)), tau, mu_pen)
        if out is None:
            loss_masking = None
            x_path, ema_x_path = x_path.to(dev, non_blocking=True), ema_x_path.to(dev, non_blocking=True)
            x_omic, grade = x_omic.to(dev, non_blocking=True), grade.to(dev, non_blocking=True)
            if masking_on:                                                                       # MIA-2023 :198-220
                # superpixel attention: mask the Path_K superpixels / Omic_K genes the fused prediction is most sensitive to
                # and ask the two masked views to agree with the mean teacher's predictions on the loader's masked views
                from . import superpixel as SPX
                if sp_mask is None:
                    raise ValueError("opt.masking needs the 6-view loader tuple (x_path, sp_mask, ema_x_path, ema_sp_mask, "
                                     "x_path_m_v1, x_path_m_v2)")
                for mod in self.model.modules():
                    if hasattr(mod, "_get_workspace"):
                        mod._multi_forward = True     # three taped forwards of the same trunk before one backward
                pm, om = SPX.superpixel_attention_mask(opt, self.optimizer, self.model, x_path, x_grph, x_omic, sp_mask, grade,
                                                       dev, getattr(opt, "num_superpixels_max", None))
                pred_m1 = self.model(x_path=SPX.apply_mask(x_path, pm), x_omic=x_omic)[5]            # :204-205
                pred_m2 = self.model(x_path=x_path, x_omic=SPX.apply_mask(x_omic, om))[5]            # :207-208
                with torch.no_grad():
                    ema_m1 = self.ema_model(x_path=x_path_m_v1.to(dev), x_omic=x_omic)[5]             # :211-215
                    ema_m2 = self.ema_model(x_path=x_path_m_v2.to(dev), x_omic=x_omic)[5]
                loss_masking = self.pred_KD_loss(pred_m1, ema_m1, B) + self.pred_KD_loss(pred_m2, ema_m2, B)   # :217-220
            out = self._device_body(x_path, ema_x_path, x_omic, grade, index.to(dev) if self.crd_on else index,
                                    sample_idx.to(dev) if self.crd_on else sample_idx, B, do_aux, getattr(opt, "CRD_weight", 1.0), tau, mu_pen,
                                    loss_masking)
        if self.tsvd_on:
            if do_aux:
                self.mu = mu_pen                                                                 # :413
            self._batch_idx = bidx + 1
        self.iter_num += 1
        return out

    _IN_NAMES = ("x_path", "ema_x_path", "x_omic", "grade", "index", "sample_idx")

    def _graph_step(self, given, B, do_aux, crd_w, tau, mu_pen):
        """One step from a captured graph; None if no graph can serve it (capture failed: eager from now on).  Device-resident,
        contiguous inputs are adopted as they are (up to two input sets, one graph each per kind of step, sharing one memory
        pool); anything else is copied into the first set's buffers."""
        dev = self.device
        names = self._IN_NAMES
        if getattr(self, "_g_scal", None) is None:
            self._g_scal = torch.zeros(3, device=dev, dtype=torch.float32)       # CRD weight | tau | mu of the penalty
            self._g_ring = PinnedRing((3,), torch.float32)
            self._g_sets = []
        on_dev = all(t.is_cuda and t.is_contiguous() for t in given.values())
        ptrs = tuple(given[k].data_ptr() for k in names) if on_dev else None
        shapes = tuple(tuple(given[k].shape) for k in names)
        if self._g_sets and self._g_sets[0]["shapes"] != shapes:
            self._g_sets = []
        st = next((q for q in self._g_sets if q["adopted"] and ptrs is not None and q["ptrs"] == ptrs), None)
        if st is None and on_dev and sum(q["adopted"] for q in self._g_sets) < 2:
            bufs = dict(given)
            st = dict(shapes=shapes, ptrs=ptrs, bufs=bufs, graphs={}, adopted=True)
            self._g_sets.append(st)
        if st is None:
            # host tensors, or a third resident set: staged into PRIVATE buffers - never into an adopted set, whose tensors
            # belong to the caller (a loader's ring buffer must not be overwritten behind its back)
            st = next((q for q in self._g_sets if not q["adopted"]), None)
            if st is None:
                bufs = {k: torch.empty(given[k].shape, device=dev, dtype=given[k].dtype) for k in names}
                st = dict(shapes=shapes, ptrs=tuple(bufs[k].data_ptr() for k in names), bufs=bufs, graphs={}, adopted=False)
                self._g_sets.append(st)
            for k in names:
                st["bufs"][k].copy_(given[k], non_blocking=True)
        self._g_ring.upload(self._g_scal, [crd_w, tau, mu_pen])      # (pinned rows: no host sync per replayed step)
        if do_aux not in st["graphs"]:
            g = torch.cuda.CUDAGraph()
            torch.cuda.synchronize()
            pool = next((gr.pool() for q in self._g_sets for gr, _, _ in q["graphs"].values()), None)
            was_prepared = self.optimizer._prepared
            try:
                bf = st["bufs"]
                with torch.cuda.graph(g, pool=pool, capture_error_mode=_capture_mode(self.sync)):
                    self.optimizer._prepared = True       # the step scalars are read from device memory at replay
                    out = self._device_body(bf["x_path"], bf["ema_x_path"], bf["x_omic"], bf["grade"], bf["index"],
                                            bf["sample_idx"], B, do_aux, self._g_scal[0], self._g_scal[1], self._g_scal[2], None)
                # the graph holds raw pointers into the trunk workspaces it was captured with: keep them alive with it
                refs = [ws for net in (self.model, self.ema_model) for mod in net.modules() if hasattr(mod, "pinned_workspaces")
                        for ws in mod.pinned_workspaces()]
                st["graphs"][do_aux] = (g, out, refs)
                self.optimizer._prepared = was_prepared
            except Exception as exc:
                import warnings
                warnings.warn("HIP graph capture of the stage-1 step failed (%r); continuing with eager launches" % (exc,))
                torch.cuda.synchronize()
                try:      # on this HIP runtime a failed capture can leave its streams in capture state for good
                    torch.ones(1).to(dev)
                except Exception as exc2:
                    raise RuntimeError("HIP graph capture of the stage-1 step failed (%r) and the runtime did not leave capture "
                                       "mode (%r): restart the process without enable_graph()" % (exc, exc2)) from exc
                self._want_graph = False
                self._g_sets = []
                self.optimizer._prepared = was_prepared
                return None
        g, out, _ = st["graphs"][do_aux]
        self.optimizer.prepare_step()
        self.optimizer._prepared = False
        g.replay()
        return out

    def _device_body(self, x_path, ema_x_path, x_omic, grade, index, sample_idx, B, do_aux, crd_w, tau, mu_pen, loss_masking):
        """Everything of the step that runs on the device (capturable in one HIP graph).  `crd_w`, `tau`, `mu_pen`: floats on
        the eager path, 1-element device tensors under capture."""
        opt = self.opt
        dev = self.device
        if loss_masking is None:
            loss_masking = torch.zeros((), device=dev)
        # the mean teacher's forward (no_grad, :143-145) does not depend on the student's (:137): a second stream, joined
        # before the losses, as in DistillStep._device_body
        main = torch.cuda.current_stream()
        if getattr(self, "_ema_side", None) is None:
            self._ema_side = torch.cuda.Stream(device=dev)
        eside = self._ema_side
        eside.wait_stream(main)
        with torch.cuda.stream(eside), torch.no_grad():
            ema_fuse_feat, ema_path_feat, ema_omic_feat, _, _, ema_pred, ema_pred_path, ema_pred_omic, _, _, _ = self.ema_model(
                x_path=ema_x_path, x_omic=x_omic)                                                # :143-145
            for t in (ema_fuse_feat, ema_path_feat, ema_omic_feat, ema_pred, ema_pred_path, ema_pred_omic):
                t.record_stream(main)
        fuse_feat, path_feat, omic_feat, _, _, pred, pred_path, pred_omic, _, _, _ = self.model(
            x_path=x_path, x_omic=x_omic)                                                        # :137
        main.wait_stream(eside)
        loss_CRD = torch.zeros((), device=dev)
        if self.crd_on:                                                                          # :157-165
            self.CRD_criterion_fuse.contrast.batch_norm_size = B
            loss_CRD = crd_w * self.CRD_criterion_fuse(
                fuse_feat, ema_fuse_feat.detach(), index, sample_idx).reshape(())
        kd = lambda p_s, p_t: self.pred_KD_loss(p_s, p_t, B)      # noqa: E731
        if getattr(opt, "pred_distill", 1) == 1:
            nt = opt.num_teachers
            kd_fuse = kd(pred, ema_pred)
            if nt == 1:
                kd_path, kd_omic = kd(pred_path, ema_pred_path), kd(pred_omic, ema_pred_omic)
            elif nt == 2:
                kd_path = (kd(pred_path, ema_pred_path) + kd(pred_path, ema_pred)) / 2.0
                kd_omic = (kd(pred_omic, ema_pred_omic) + kd(pred_omic, ema_pred)) / 2.0
            elif nt == 3:
                kd_path = (kd(pred_path, ema_pred_path) + kd(pred_path, ema_pred) + kd(pred_path, ema_pred_omic)) / 3.0
                kd_omic = (kd(pred_omic, ema_pred_omic) + kd(pred_omic, ema_pred) + kd(pred_omic, ema_pred_path)) / 3.0
            else:
                raise NotImplementedError("num_teachers in {1,2,3}")
            loss_pred_KD = getattr(opt, "KD_weight", 1.0) * (kd_fuse + kd_path + kd_omic)       # :203
        else:
            loss_pred_KD = torch.zeros((), device=dev)
        nll = lambda p: ops.NLLFn.apply(p, grade, B)
        loss_nll = nll(pred_path) + nll(pred_omic) + nll(pred)                                  # :208-212
        loss = opt.lambda_nll * loss_nll + loss_CRD + loss_pred_KD + loss_masking               # :217-218; MIA-2023 :303-304
        loss_reg = torch.zeros((), device=dev)
        if opt.reg_type != "none":      # :209 - the shipped stage-1 command keeps the default `omic` (options.py:132)
            from .networks_new import define_reg
            loss_reg = define_reg(opt, self.model)
            # split over the replicas, whose losses / gradients are summed (see DistillStep._add_reg)
            loss = loss + (opt.lambda_reg / (self.sync.world_size if self.sync is not None else 1)) * loss_reg
        loss_orth = torch.zeros((), device=dev)
        if self.orth_on:
            loss_orth = self.Orth_loss(path_feat, omic_feat)                                    # :216-218
            loss = loss + loss_orth
        loss_tsvd = torch.zeros((), device=dev)
        if self.tsvd_on:                                                                        # train_test_tSVD.py:299-431
            from . import tsvd as T
            if self.sync is not None:
                # the adjacency tensors span the global batch: gather every feature view (rank-ordered rows); the
                # backward of the gather hands this replica's rows their gradient
                fuse_feat, path_feat, omic_feat = (_GatherRowsFn.apply(t, self.sync) for t in (fuse_feat, path_feat, omic_feat))
                ema_fuse_feat, ema_path_feat, ema_omic_feat = (self.sync.all_gather_cat(t) for t in
                                                               (ema_fuse_feat, ema_path_feat, ema_omic_feat))
            if opt.n_views == 2:
                feats1 = [path_feat, ema_path_feat]                                             # :320-322
                feats2 = [omic_feat, ema_omic_feat]
            else:
                feats1 = [fuse_feat.detach(), ema_fuse_feat, path_feat, ema_path_feat]          # :311-313
                feats2 = [fuse_feat.detach(), ema_fuse_feat, omic_feat, ema_omic_feat]
                # n_views 6 / 8 add mixtures of the max-normalised mean-teacher features (:305-307, :341-363)
                for wa in (0.9, 0.8, 0.7, 0.6)[:opt.n_views - 4]:
                    feats1.append(T.maxnorm_mix(ema_path_feat, ema_omic_feat, wa, 1.0 - wa))
                    feats2.append(T.maxnorm_mix(ema_omic_feat, ema_path_feat, wa, 1.0 - wa))
            self.adj_tensor1 = T.update_adj_tensor(self.adj_tensor1, feats1)                    # :365-366
            self.adj_tensor2 = T.update_adj_tensor(self.adj_tensor2, feats2)
            if do_aux:
                # `if opt.tSVD_mode == "path" or "pathomic":` (:378, :398) is always true: both tensors are updated.  The two
                # proximal updates are independent and each is a launch of only n_views / 2 + 1 workgroups (one per Fourier
                # slice, ~5 ms at B = 128): the second runs on a side stream beside the first.  The results land in the
                # PERSISTENT auxiliary tensors (captured graphs of other input sets / of the steps without an update read them)
                main = torch.cuda.current_stream()
                if getattr(self, "_tsvd_side", None) is None:
                    self._tsvd_side = torch.cuda.Stream(device=dev)
                side = self._tsvd_side
                side.wait_stream(main)
                tnns = {}
                for adj, auxs, name, strm in ((self.adj_tensor2, self.aux_tensor2, "2", side),
                                              (self.adj_tensor1, self.aux_tensor1, "1", main)):
                    with torch.cuda.stream(strm):
                        stack = torch.stack([a.detach() for a in adj], dim=2)
                        aux, tnns[name] = T.update_aux(stack, tau)                              # :382, :402
                        for v in range(opt.n_views):
                            auxs[v].copy_(aux[:, :, v])
                        if strm is side:
                            stack.record_stream(main); aux.record_stream(main)
                main.wait_stream(side)
                if torch.is_tensor(tnns["2"]):
                    tnns["2"].record_stream(main)
                self.path_TNN, self.omic_TNN = tnns["1"], tnns["2"]
            if opt.tSVD_mode in ("path", "pathomic"):                                           # :418-431
                loss_tsvd = loss_tsvd + T.tsvd_penalty(self.adj_tensor1, self.aux_tensor1, mu_pen)
            if opt.tSVD_mode in ("omic", "pathomic"):
                loss_tsvd = loss_tsvd + T.tsvd_penalty(self.adj_tensor2, self.aux_tensor2, mu_pen)
            loss = loss + loss_tsvd
        self.optimizer.zero_grad()
        loss.backward()
        if self.sync is not None:
            self.sync.all_reduce_grads(self.optimizer.flat)
        self.optimizer.step()                                                                   # + EMA (:229) fused
        if self.tsvd_on:
            # the adjacency tensors are kept as VALUES: with their grad_fn they would keep this step's autograd graph - and the
            # AccumulateGrad nodes of every parameter, bound to the stream of this step - alive into the next one (a capture of
            # the next step then runs them on the wrong stream: the runtime crashes in hipStreamEndCapture)
            self.adj_tensor1 = [a.detach() for a in self.adj_tensor1]
            self.adj_tensor2 = [a.detach() for a in self.adj_tensor2]
        return dict(loss=loss.detach(), loss_nll=loss_nll.detach(), loss_pred_KD=loss_pred_KD.detach(),
                    loss_CRD=loss_CRD.detach(), loss_orth=loss_orth.detach(), loss_tsvd=loss_tsvd.detach(),
                    loss_pred_KD_masking=loss_masking.detach(), loss_reg=loss_reg.detach(),
                    pred=pred.detach(), pred_path=pred_path.detach(), pred_omic=pred_omic.detach())
