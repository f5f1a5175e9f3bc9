"""Fused loss head of the MICCAI-2022 stage-2 step (train_test_path_multi_distill.py:262-313 + AEKD_loss :41-70).

The generic path builds the five loss terms as separate autograd graphs and lets GK-Refine differentiate each of them
w.r.t. the student feature (five small backward passes), then back-propagates the weighted sum a sixth time: ~125
kernels of 3-6 us, a third of them torch scalar glue, 1.3 ms of a 14 ms step.  Everything in that block is a function of
ONE differentiable tensor - the student feature [B,128] - through three tiny linear maps (fc_new2 and the two CRD
student heads), so the per-loss gradients are computed once, in closed form, with the same kernels:

    d KL_k / d feat   = kl_bwd(logits, teacher_k) @ W2            (k = fused teacher, EMA teacher)
    d CE   / d feat   = log_softmax_bwd(nll_bwd) @ W2
    d CRD_k / d feat  = l2norm_bwd(d CRD_k / d v1) @ W_embed_s,k  (d CRD / d v is produced by the CRD loss kernel)

GK-Refine's weights are the row sums of the cosine Gram of these five vectors (scale-invariant, so the alpha / beta
factors are applied afterwards); the gradient of the final loss is their weighted sum - no second pass through the
losses.  Values equal the generic path's up to fp32 summation order (tests/test_gpu_step.py).

The MIA-2022 stage-2 body ("MIA 2022/train_test_path_multi_distill_v2.py":419-483, `LossHeadCtx.variant == "mia2022"`) is the
same function of the student feature with two differences: its CRD criterion is the plain v3 bank (one positive, no
discrepancy selection) whose loss carries the epoch weight e (a device scalar), and GK-Refine keeps an exponential moving
average of the weights across iterations (`step._mo_state`, the trainer's order [div1, div2, kd1, kd2, CE]) with an optional
binarisation - ph_gk_finish_momentum."""
import torch

from . import ops
from ._lib import lib, check, ptr, stream
from .CL_utils.memory_new import crd_core

_ORDER_INT = (0, 1, 4, 2, 3)      # external [div1, div2, kd1, kd2, CE] -> internal [div1, div2, CE, kd1, kd2] position


class LossHeadCtx:
    """Non-differentiable inputs of one step (plain container)."""

    def __init__(self, step, grade, t_logit, ema_logit, fuse_feat, ema_feat, index, sample_idx, r1, r2, bnorm,
                 variant="miccai2022", e_dev=None):
        self.step, self.grade, self.t_logit, self.ema_logit = step, grade, t_logit, ema_logit
        self.fuse_feat, self.ema_feat, self.index, self.sample_idx = fuse_feat, ema_feat, index, sample_idx
        self.r1, self.r2, self.bnorm = r1, r2, bnorm
        self.variant, self.e_dev = variant, e_dev
        self.out = None


def _const(step, name, values, dtype=torch.float32):
    cache = step.__dict__.setdefault("_lh_const", {})
    key = (name, tuple(values))
    if key not in cache:
        cache[key] = torch.tensor(list(values), device=step.device, dtype=dtype)
    return cache[key]


def _grad_dst(p):
    if p.grad is None:
        p.grad = torch.zeros_like(p)
    return p.grad


class FusedDistillLossFn(torch.autograd.Function):
    """loss = lambda_nll * CE + sum_i scale_i * KD_i as a function of the student feature; TERMINAL: backward ignores the
    incoming gradient value (it is the 1.0 of loss.backward()) and writes the gradients of fc_new2 and of the four CRD
    projection heads straight into their .grad buffers (zeroed by optimizer.zero_grad() just before)."""

    @staticmethod
    def forward(ctx, feat, H):
        step = H.step
        opt, model = step.opt, step.model
        L = lib()
        st = stream()
        feat = ops._f32(feat).contiguous()
        B, D = feat.shape
        dev = feat.device
        W2, b2 = model.fc_new2.weight, model.fc_new2.bias
        Cc = W2.shape[0]
        inv = 1.0 / H.bnorm
        T = float(opt.kd_T)
        logits = ops.linear_fwd(feat, W2, b2)
        pred = torch.empty_like(logits)
        Lb = torch.empty(5, device=dev, dtype=torch.float32)      # unscaled losses, internal order [div1, div2, CE, kd1, kd2]
        dl = torch.empty(3, B, Cc, device=dev, dtype=torch.float32)   # d loss / d logits for div1, div2, CE
        yt1, yt2, grade = ops._f32(H.t_logit).contiguous(), ops._f32(H.ema_logit).contiguous(), H.grade.contiguous()
        check(L.ph_logit_losses(ptr(logits), ptr(yt1), ptr(yt2), ptr(grade), ptr(pred), ptr(Lb), ptr(dl), B, Cc, T, inv, st),
              "ph_logit_losses")
        G = torch.empty(5, B, D, device=dev, dtype=torch.float32)
        ops.sgemm(dl, W2, None, G, 3 * B, D, Cc, Cc, 1, D, 1)          # rows 0..2: d{div1, div2, CE}/d feat = dlogit @ W2
        crd_saved = [None, None]

        def crd_chain(k, crd, tf, rk):
            stc = stream()
            tf = ops._f32(tf).contiguous()
            es, et = crd.embed_s.linear, crd.embed_t.linear
            v1 = torch.empty(B, es.weight.shape[0], device=dev, dtype=torch.float32); n1 = torch.empty(B, device=dev)
            v2 = torch.empty_like(v1); n2 = torch.empty_like(n1)
            pre_s = ops.linear_fwd(feat, es.weight, es.bias)
            check(L.ph_l2norm_fwd(ptr(pre_s), ptr(v1), ptr(n1), B, v1.shape[1], stc), "ph_l2norm_fwd")
            pre_t = ops.linear_fwd(tf, et.weight, et.bias)
            check(L.ph_l2norm_fwd(ptr(pre_t), ptr(v2), ptr(n2), B, v2.shape[1], stc), "ph_l2norm_fwd")
            _, dv1, dv2 = crd_core(v1, v2, crd.contrast, H.index, H.sample_idx, rk, False, loss_out=Lb[3 + k])
            dps = torch.empty_like(v1)
            check(L.ph_l2norm_bwd(ptr(dv1), ptr(v1), ptr(n1), ptr(dps), B, v1.shape[1], stc), "ph_l2norm_bwd")
            Do = es.weight.shape[0]
            ops.sgemm(dps, es.weight, None, G[3 + k], B, D, Do, Do, 1, D, 1)      # d CRD_k / d feat
            crd_saved[k] = (dps, dv2, v2, n2, tf)
            return (tf, v1, n1, v2, n2, pre_s, pre_t, dv1, dv2, dps)

        chains = ((step.criterion_kd, H.fuse_feat, H.r1), (step.criterion_kd_path, H.ema_feat, H.r2))
        # the two CRD terms are independent chains of ~10 short launches (two banks, two pairs of heads): the second runs on a
        # side stream beside the first (not under data parallelism: its all-gathers must keep one issue order on every rank)
        hs = step._head_stream() if step.sync is None and hasattr(step, "_head_stream") else None
        if hs is not None:
            main_s = torch.cuda.current_stream()
            hs.wait_stream(main_s)
            with torch.cuda.stream(hs):
                made = crd_chain(1, *chains[1])
                for t in made:
                    t.record_stream(main_s)
            crd_chain(0, *chains[0])
            main_s.wait_stream(hs)
        else:
            for k, (crd, tf, rk) in enumerate(chains):
                crd_chain(k, crd, tf, rk)
        gram = torch.empty(25, device=dev, dtype=torch.float32)
        check(L.ph_gram(ptr(G), ptr(gram), 5, B * D, st), "ph_gram")
        if step.sync is not None:
            step.sync.all_reduce_sum(gram)
        a, b, lam = float(opt.alpha), float(opt.beta), float(opt.lambda_nll)
        fin = torch.empty(21, device=dev, dtype=torch.float32)    # scale_int[5] | w[5] | total | scaled[5] | scale_ext[5]
        if H.variant == "mia2022":
            if step._mo_state is None:       # persistent across iterations (and in the checkpoint): the trainer's mo_scale
                step._mo_state = torch.zeros(5, device=dev, dtype=torch.float32)
                step._mo_init = torch.zeros(1, device=dev, dtype=torch.int32)
            thr_on = opt.grads_thresh == "True"
            check(L.ph_gk_finish_momentum(ptr(gram), ptr(Lb), a, b, ptr(H.e_dev), lam, 1.0 if thr_on else 4.0,   # :474-477
                                          1 if thr_on else 0, float(opt.thresh), float(opt.grads_m), ptr(step._mo_state),
                                          ptr(step._mo_init), ptr(fin[5:10]), ptr(fin[10:11]), ptr(fin[11:16]),
                                          ptr(fin[16:21]), st), "ph_gk_finish_momentum")
        else:
            check(L.ph_gk_finish(ptr(gram), ptr(Lb), ptr(_const(step, "coef", (a, a, 0, b, b))),
                                 ptr(_const(step, "lam", (0, 0, lam, 0, 0))), ptr(_const(step, "logc", (a, a, 1, b, b))), 4.0,
                                 ptr(fin[0:5]), ptr(fin[5:10]), ptr(fin[10:11]), ptr(fin[11:16]), ptr(fin[16:21]), st),
                  "ph_gk_finish")                                     # x len(KD list) = 4 (:61)
        w, total, scaled = fin[5:10], fin[10], fin[11:16]
        H.out = dict(loss_cls=scaled[2], loss_div1=scaled[0], loss_div2=scaled[1], loss_kd1=scaled[3], loss_kd2=scaled[4],
                     scale=fin[16:21], logit_path=logits, pred_path=pred)
        ctx.H, ctx.feat, ctx.dl, ctx.G, ctx.w, ctx.crd_saved = H, feat, dl, G, w, crd_saved
        return total

    @staticmethod
    def backward(ctx, g):
        H, feat, dl, G, w = ctx.H, ctx.feat, ctx.dl, ctx.G, ctx.w
        step = H.step
        model = step.model
        L = lib()
        st = stream()
        B, D = feat.shape
        dev = feat.device
        Cc = dl.shape[2]
        dfeat = torch.empty(B, D, device=dev, dtype=torch.float32)
        ops.sgemm(w, G, None, dfeat, 1, B * D, 5, 5, 1, B * D, 1)                       # sum_i w_i G_i
        dlt = torch.empty(B, Cc, device=dev, dtype=torch.float32)
        ops.sgemm(w, dl, None, dlt, 1, B * Cc, 3, 3, 1, B * Cc, 1)                      # w[0:3] . {dl_div1, dl_div2, dl_CE}
        # Only dfeat feeds the trunk backward: the gradients of fc_new2 and of the four CRD heads (14 short launches) go to the
        # side stream, beside the trunk backward; DistillStep joins it before the optimiser step
        # (not under data parallelism: the all-reduce of the late gradient slice starts inside the trunk backward and reads them)
        hs = step._head_stream() if step.sync is None and hasattr(step, "_head_stream") else None
        main_s = torch.cuda.current_stream()
        if hs is not None:
            hs.wait_stream(main_s)
        with torch.cuda.stream(hs if hs is not None else main_s):
            stb = stream()
            ones = ops._ones(B, dev)
            ops.sgemm(dlt, feat, None, _grad_dst(model.fc_new2.weight), Cc, D, B, 1, Cc, D, 1)      # dW2 = dlogit^T feat
            ops.sgemm(ones, dlt, None, _grad_dst(model.fc_new2.bias), 1, Cc, B, 0, 1, Cc, 1)
            for k, (crd, (dps, dv2, v2, n2, tf)) in enumerate(zip((step.criterion_kd, step.criterion_kd_path), ctx.crd_saved)):
                wk = w[3 + k:4 + k]
                es, et = crd.embed_s.linear, crd.embed_t.linear
                Do = es.weight.shape[0]
                gs = dps * wk
                ops.sgemm(gs, feat, None, _grad_dst(es.weight), Do, D, B, 1, Do, D, 1)
                ops.sgemm(ones, gs, None, _grad_dst(es.bias), 1, Do, B, 0, 1, Do, 1)
                dpt = torch.empty_like(v2)
                check(L.ph_l2norm_bwd(ptr(dv2), ptr(v2), ptr(n2), ptr(dpt), B, v2.shape[1], stb), "ph_l2norm_bwd")
                gt = dpt * wk
                Dt = tf.shape[1]
                ops.sgemm(gt, tf, None, _grad_dst(et.weight), Do, Dt, B, 1, Do, Dt, 1)
                ops.sgemm(ones, gt, None, _grad_dst(et.bias), 1, Do, B, 0, 1, Do, 1)
            if hs is not None:
                for t in (dlt, feat, w, dl):
                    t.record_stream(hs)
                for dps, dv2, v2, n2, tf in ctx.crd_saved:
                    for t in (dps, dv2, v2, n2, tf):
                        t.record_stream(hs)
        ctx.crd_saved = None
        return dfeat, None


class FusedMia2023LossFn(torch.autograd.Function):
    """The MIA-2023 stage-2 body ("MIA 2023/stage2_unimodal_student/train_test_path_multi_distill.py":348-448) as one function
    of the student feature.  Everything there is per sample: KL rows, CRD rows of the v10 criterion (KNN positives) under the
    discrepancy weights w = 1 + rw * discrepancy, and GK_refine_thresh's weights s[b, i] = sum_j f(cos(g_i[b], g_j[b])) over
    the five per-sample gradient rows (:81-128).  With g_i[b] in closed form (as in FusedDistillLossFn) the loss is
        lambda * CE + (1 / B) sum_b ( alpha s[b,0] KL1[b] + alpha s[b,1] KL2[b] + beta s[b,2] kd1[b] + beta s[b,3] kd2[b] )
    and its gradient with respect to feature row b is the same combination of the rows g_i[b] (s is a constant, :120-126).
    TERMINAL like FusedDistillLossFn: backward writes the heads' gradients into their .grad buffers."""

    @staticmethod
    def forward(ctx, feat, H):
        from .mia2023 import assign_sample_weights
        step = H.step
        opt, model = step.opt, step.model
        L = lib()
        st = stream()
        feat = ops._f32(feat).contiguous()
        B, D = feat.shape
        dev = feat.device
        W2, b2 = model.fc_new2.weight, model.fc_new2.bias
        Cc = W2.shape[0]
        inv = 1.0 / H.bnorm
        T = float(opt.kd_T)
        a, b, lam = float(opt.alpha), float(opt.beta), float(opt.lambda_nll)
        logits = ops.linear_fwd(feat, W2, b2)
        pred = torch.empty_like(logits)
        Lb = torch.empty(3, device=dev, dtype=torch.float32)           # batch means: KL to the fused teacher, to the EMA teacher, CE
        dl = torch.empty(3, B, Cc, device=dev, dtype=torch.float32)    # d (those means) / d logits
        yts = (ops._f32(H.t_logit).contiguous(), ops._f32(H.ema_logit).contiguous())
        grade = H.grade.contiguous()
        check(L.ph_logit_losses(ptr(logits), ptr(yts[0]), ptr(yts[1]), ptr(grade), ptr(pred), ptr(Lb), ptr(dl), B, Cc, T, inv, st),
              "ph_logit_losses")
        rows_div = torch.empty(2, B, device=dev, dtype=torch.float32)   # per-sample KL (:349-350)
        for k in range(2):
            check(L.ph_kl_rows_fwd(ptr(logits), ptr(yts[k]), ptr(rows_div[k]), B, Cc, T, st), "ph_kl_rows_fwd")
        # :361-382 - discrepancy weights of the two CRD terms (constants of the graph); rw = H.e_dev: 0 before opt.start_reweight
        ws = [1.0 + H.e_dev * assign_sample_weights(logits, yt, grade, opt.discrep_scale, opt.max_discrep, from_logits=True)
              for yt in yts]
        G = torch.empty(5, B, D, device=dev, dtype=torch.float32)       # the trainer's order [div1, div2, kd1, kd2, CE]
        ops.sgemm(dl[0:2], W2, None, G[0:2], 2 * B, D, Cc, Cc, 1, D, 1)
        ops.sgemm(dl[2], W2, None, G[4], B, D, Cc, Cc, 1, D, 1)
        lossp = [None, None]
        crd_saved = [None, None]

        def crd_chain(k, crd, tf):
            stc = stream()
            tf = ops._f32(tf).contiguous()
            es, et = crd.embed_s.linear, crd.embed_t.linear
            v1 = torch.empty(B, es.weight.shape[0], device=dev, dtype=torch.float32); n1 = torch.empty(B, device=dev)
            v2 = torch.empty_like(v1); n2 = torch.empty_like(n1)
            pre_s = ops.linear_fwd(feat, es.weight, es.bias)
            check(L.ph_l2norm_fwd(ptr(pre_s), ptr(v1), ptr(n1), B, v1.shape[1], stc), "ph_l2norm_fwd")
            pre_t = ops.linear_fwd(tf, et.weight, et.bias)
            check(L.ph_l2norm_fwd(ptr(pre_t), ptr(v2), ptr(n2), B, v2.shape[1], stc), "ph_l2norm_fwd")
            idx1, knn = crd.neighbor_columns(B, v1.shape[1], grade, H.sample_idx)
            rows, dv1, dv2 = crd_core(v1, v2, crd.contrast, H.index, idx1, None, True)     # rows: (s + t) losses / bsz per sample
            crd.contrast.last.update(knn)
            dps = torch.empty_like(v1)
            check(L.ph_l2norm_bwd(ptr(dv1), ptr(v1), ptr(n1), ptr(dps), B, v1.shape[1], stc), "ph_l2norm_bwd")
            Do = es.weight.shape[0]
            ops.sgemm(dps, es.weight, None, G[2 + k], B, D, Do, Do, 1, D, 1)              # d rows[b] / d feat[b]
            lossp[k] = rows
            crd_saved[k] = (dps, dv2, v2, n2, tf)
            return (tf, v1, n1, v2, n2, pre_s, pre_t, dv1, dv2, dps, rows, idx1) + tuple(knn.values())

        chains = ((step.criterion_kd, H.fuse_feat), (step.criterion_kd_path, H.ema_feat))
        hs = step._head_stream() if step.sync is None and hasattr(step, "_head_stream") else None
        if hs is not None:
            main_s = torch.cuda.current_stream()
            hs.wait_stream(main_s)
            with torch.cuda.stream(hs):
                made = crd_chain(1, *chains[1])
                for t in made:
                    if torch.is_tensor(t):
                        t.record_stream(main_s)
            crd_chain(0, *chains[0])
            main_s.wait_stream(hs)
        else:
            for k, (crd, tf) in enumerate(chains):
                crd_chain(k, crd, tf)
        s = torch.empty(B, 5, device=dev, dtype=torch.float32)
        check(L.ph_gk_rows(ptr(G), 5, B, D, 1 if opt.use_grads_thresh == "True" else 0, float(opt.grads_thresh), ptr(s), st),
              "ph_gk_rows")
        # coefficient of gradient row (i, b) in d loss / d feat[b]; G rows 0, 1, 4 are gradients of batch MEANS (x 1 / B already),
        # rows 2, 3 of the per-sample CRD rows, which crd_core has divided by B
        Cf = torch.empty(5, B, device=dev, dtype=torch.float32)
        Cf[0] = a * s[:, 0]; Cf[1] = a * s[:, 1]
        Cf[2] = b * s[:, 2] * ws[0]; Cf[3] = b * s[:, 3] * ws[1]
        Cf[4] = lam
        total = lam * Lb[2] + inv * (Cf[0] * rows_div[0] + Cf[1] * rows_div[1]).sum() + (Cf[2] * lossp[0] + Cf[3] * lossp[1]).sum()
        mean_scale = s.mean(0)
        if step.sync is not None:
            mean_scale = step.sync.all_reduce_sum(mean_scale) / step.sync.world_size
        H.out = dict(loss_cls=Lb[2], loss_div1=Lb[0], loss_div2=Lb[1],
                     loss_kd1=(ws[0] * lossp[0]).sum(), loss_kd2=(ws[1] * lossp[1]).sum(),
                     rows_div1=rows_div[0], rows_kd1=ws[0] * lossp[0] * H.bnorm, w1=ws[0].view(-1, 1), w2=ws[1].view(-1, 1),
                     scale=mean_scale, logit_path=logits, pred_path=pred)
        ctx.H, ctx.feat, ctx.dl, ctx.G, ctx.Cf, ctx.crd_saved = H, feat, dl, G, Cf, crd_saved
        return total

    @staticmethod
    def backward(ctx, g):
        H, feat, dl, G, Cf = ctx.H, ctx.feat, ctx.dl, ctx.G, ctx.Cf
        step = H.step
        model = step.model
        L = lib()
        B, D = feat.shape
        dev = feat.device
        Cc = dl.shape[2]
        dfeat = (Cf.unsqueeze(-1) * G).sum(0)
        dlt = Cf[0].unsqueeze(-1) * dl[0] + Cf[1].unsqueeze(-1) * dl[1] + Cf[4].unsqueeze(-1) * dl[2]
        hs = step._head_stream() if step.sync is None and hasattr(step, "_head_stream") else None
        main_s = torch.cuda.current_stream()
        if hs is not None:
            hs.wait_stream(main_s)
        with torch.cuda.stream(hs if hs is not None else main_s):
            stb = stream()
            ones = ops._ones(B, dev)
            ops.sgemm(dlt, feat, None, _grad_dst(model.fc_new2.weight), Cc, D, B, 1, Cc, D, 1)      # dW2 = dlogit^T feat
            ops.sgemm(ones, dlt, None, _grad_dst(model.fc_new2.bias), 1, Cc, B, 0, 1, Cc, 1)
            made = [dlt]
            for k, (crd, (dps, dv2, v2, n2, tf)) in enumerate(zip((step.criterion_kd, step.criterion_kd_path), ctx.crd_saved)):
                wk = Cf[2 + k].unsqueeze(-1)
                es, et = crd.embed_s.linear, crd.embed_t.linear
                Do = es.weight.shape[0]
                gs = dps * wk
                ops.sgemm(gs, feat, None, _grad_dst(es.weight), Do, D, B, 1, Do, D, 1)
                ops.sgemm(ones, gs, None, _grad_dst(es.bias), 1, Do, B, 0, 1, Do, 1)
                dpt = torch.empty_like(v2)
                check(L.ph_l2norm_bwd(ptr(dv2), ptr(v2), ptr(n2), ptr(dpt), B, v2.shape[1], stb), "ph_l2norm_bwd")
                gt = dpt * wk
                Dt = tf.shape[1]
                ops.sgemm(gt, tf, None, _grad_dst(et.weight), Do, Dt, B, 1, Do, Dt, 1)
                ops.sgemm(ones, gt, None, _grad_dst(et.bias), 1, Do, B, 0, 1, Do, 1)
                made += [gs, dpt, gt]
            if hs is not None:
                for t in [feat, Cf, dl] + made:
                    t.record_stream(hs)
                for dps, dv2, v2, n2, tf in ctx.crd_saved:
                    for t in (dps, dv2, v2, n2, tf):
                        t.record_stream(hs)
        ctx.crd_saved = None
        return dfeat, None
