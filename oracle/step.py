"""CPU restatement of one teacher+student distillation step (TEST INFRASTRUCTURE).

Mirrors the batch body of /root/reference/MICCAI-2022/train_test_path_multi_distill.py:242-330
with the README stage-2 flags (MICCAI-2022/README.md:30-33): ``--distill crd -a 1 -b 0.02
--nce_p2 20 --num_teachers 2 --CE_grads True --reg_type none --beta1 0.9 --select_pos_mode mid
--assign_weights True``.

Two execution modes give identical numbers:
  * ``faithful=True``  - the reference's 3 forward + 6 backward ResNet passes (AEKD_loss runs a full
    backward per loss);
  * ``faithful=False`` - 3 forward + 1 backward (VJPs stop at feat_s).
"""
import copy
import math
from collections import OrderedDict
from types import SimpleNamespace

import numpy as np
import torch

from .weights import make_state_dict, student_shapes, teacher_shapes, embed_shapes
from .nets import resnet_forward, pathomic_forward
from .losses import CRDState, distill_kl, crd_loss, aekd_loss, nll_loss


def default_opt(**kw):
    """The attributes of the reference's `opt` that the hot path reads (options.py:8-164) at the
    README stage-2 values."""
    o = SimpleNamespace(
        mode="pathomic", task="grad", act_type="LSM", init_type="max", init_gain=0.02, gpu_ids=[],
        path_dim=128, omic_dim=128, mmhid=128, label_dim=3, input_size_omic=320,
        dropout_rate=0.0, fusion_type="pofusion", skip=0, use_bilinear=1, path_gate=1, omic_gate=1,
        path_scale=1, omic_scale=1, return_grad="False", cut_fuse_grad=True,
        distill="crd", alpha=1.0, beta=0.02, kd_T=1.0, num_teachers=2, which_teacher="fuse",
        CE_grads=True, assign_weights="True", reg_type="none", sample_KD="False",
        s_dim=128, t_dim=128, feat_dim=128, nce_p=300, nce_p2=20, nce_k=700, nce_k2=512,
        nce_t=0.07, nce_m=0.5, select_pos_pairs=True, select_neg_pairs="True",
        select_pos_mode="mid", n_data=1024,
        optimizer_type="adam", lr=5e-4, beta1=0.9, beta2=0.999, weight_decay=4e-4,
        lr_policy="linear", niter=0, niter_decay=30, epoch_count=1,
        lambda_cox=1.0, lambda_nll=1.0, lambda_reg=3e-4, ema_decay=0.99, global_step=0,
        batch_size=16)
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def synthetic_batch(B, H, n_data=1024, omic=320, P=300, K=700, seed=0):
    """Synthetic batch of SURVEY.md section 8-d in the loader's tuple layout
    (data_loaders_MT.py:256): column 0 of sample_idx is the sample's own index (:238-239)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(1234 + seed)
    x_path = torch.rand(B, 3, H, H, generator=g) * 2 - 1
    ema_x_path = x_path + 0.01 * torch.randn(B, 3, H, H, generator=g)
    x_omic = torch.randn(B, omic, generator=g)
    grade = torch.randint(0, 3, (B,), generator=g)
    index = torch.randperm(n_data, generator=g)[:B]
    sample_idx = torch.randint(0, n_data, (B, P + K), generator=g)
    sample_idx[:, 0] = index
    return dict(x_path=x_path, ema_x_path=ema_x_path, x_omic=x_omic, grade=grade, index=index,
                sample_idx=sample_idx)


class DistillOracle:
    """State + one step of the stage-2 distillation loop, CPU fp32."""

    def __init__(self, opt=None, seed=0, n_data=1024):
        self.opt = opt or default_opt()
        o = self.opt
        self.n_data = n_data
        self.student = make_state_dict(student_shapes(o.path_dim, o.label_dim), seed + 1)
        # create_model(ema=True) builds a second, independently initialised net (:176-184)
        self.ema = make_state_dict(student_shapes(o.path_dim, o.label_dim), seed + 2)
        self.teacher = make_state_dict(teacher_shapes(o.input_size_omic, o.path_dim, o.omic_dim,
                                                      o.mmhid, o.label_dim), seed + 3)
        self.crd = []
        for i in range(2):   # criterion_kd (fuse teacher), criterion_kd_path (EMA teacher) :202-208
            es = make_state_dict(embed_shapes(o.s_dim, o.feat_dim), seed + 10 + 2 * i)
            et = make_state_dict(embed_shapes(o.t_dim, o.feat_dim), seed + 11 + 2 * i)
            self.crd.append(CRDState(n_data, o.feat_dim, o.nce_p, o.nce_k, o.nce_t, o.nce_m,
                                     seed=seed + 20 + i, embed_s=es, embed_t=et))
        self.iter_num = o.global_step
        self.adam_t = 0
        self._m = {}
        self._v = {}
        self.lr = o.lr

    # ---- optimiser: torch.optim.Adam with L2-in-grad weight decay (networks_new.py:85) ----
    def trainable(self):
        """(name, tensor) in the order define_optimizer sees them: model, embed_s, embed_t, ... (:192-211)."""
        out = []
        for k, t in self.student.items():
            if t.dtype.is_floating_point and not k.endswith(("running_mean", "running_var",
                                                               "output_range", "output_shift")):
                out.append(("student." + k, t))
        for i, c in enumerate(self.crd):
            for nm, d in (("embed_s", c.embed_s), ("embed_t", c.embed_t)):
                for k, t in d.items():
                    out.append((f"crd{i}.{nm}.{k}", t))
        return out

    def _adam(self, grads):
        o = self.opt
        self.adam_t += 1
        t = self.adam_t
        bc1 = 1 - o.beta1 ** t
        bc2 = 1 - o.beta2 ** t
        for (name, p), g in zip(self.trainable(), grads):
            if g is None:
                continue
            g = g + o.weight_decay * p
            m = self._m.setdefault(name, torch.zeros_like(p))
            v = self._v.setdefault(name, torch.zeros_like(p))
            m.mul_(o.beta1).add_(g, alpha=1 - o.beta1)
            v.mul_(o.beta2).addcmul_(g, g, value=1 - o.beta2)
            denom = (v.sqrt() / math.sqrt(bc2)).add_(1e-8)
            p.addcdiv_(m, denom, value=-self.lr / bc1)

    def _ema_update(self):
        """update_ema_variables (:34-38): parameters only, BN buffers untouched."""
        alpha = min(1 - 1 / (self.iter_num + 1), self.opt.ema_decay)
        for k, t in self.student.items():
            if t.dtype.is_floating_point and not k.endswith(("running_mean", "running_var")):
                self.ema[k].mul_(alpha).add_(t, alpha=1 - alpha)

    def step(self, batch, mid_ranks=None, faithful=False, gen=None):
        """One batch of :249-330.  `mid_ranks` = the two np.random.choice draws of
        memory_new.py:311 (kd1 then kd2); drawn from np.random here when None."""
        o = self.opt
        names = [n for n, _ in self.trainable()]
        params = [p for _, p in self.trainable()]
        for p in params:
            p.requires_grad_(True)
        grade = batch["grade"]
        # student (:249), EMA (:254), teacher (:256) forwards - all with train-mode BN
        _, path_feat, logit_path, pred_path, _ = resnet_forward(batch["x_path"], self.student)
        with torch.no_grad():
            _, ema_feat, ema_logit, _, _ = resnet_forward(batch["ema_x_path"], self.ema)
            t_out = pathomic_forward(batch["x_path"], batch["x_omic"], self.teacher,
                                     o.dropout_rate, gen)
            fuse_feat, logits = t_out[0], t_out[4]
        loss_cls = nll_loss(pred_path, grade)                                        # :262
        two = o.num_teachers == 2
        if not (two or (o.num_teachers == 1 and o.which_teacher in ("fuse", "self_EMA"))):
            raise UnboundLocalError("loss_div (:263-271 define it for 2 teachers or 1 teacher fuse / self_EMA)")
        fuse_on, ema_on = two or o.which_teacher == "fuse", two or o.which_teacher == "self_EMA"
        zero = torch.zeros(())
        loss_div1 = distill_kl(logit_path, logits[-1].detach(), o.kd_T) if fuse_on else zero     # :264 / :268
        loss_div2 = distill_kl(logit_path, ema_logit.detach(), o.kd_T) if ema_on else zero       # :265 / :270
        loss_kd1 = loss_kd2 = zero
        aux1 = aux2 = None
        if o.distill == "crd":
            # one np.random.choice draw per CRD call (memory_new.py:311): two calls with two teachers, one otherwise
            ncalls = 2 if two else 1
            if mid_ranks is None:
                mid_ranks = [np.random.choice(np.arange(30, 100, 1), o.nce_p2, replace=False) for _ in range(ncalls)]
            # one teacher: the call goes through criterion_kd (self.crd[0]) whichever teacher is chosen (:282-285)
            t1 = fuse_feat if fuse_on else ema_feat
            loss_kd1, aux1 = crd_loss(self.crd[0], path_feat, t1.detach(), batch["index"],
                                      batch["sample_idx"], o.nce_p2, o.nce_k2, o.select_pos_mode,
                                      mid_ranks[0], return_aux=True)                     # :278
            if two:
                loss_kd2, aux2 = crd_loss(self.crd[1], path_feat, ema_feat.detach(), batch["index"],
                                          batch["sample_idx"], o.nce_p2, o.nce_k2, o.select_pos_mode,
                                          mid_ranks[1], return_aux=True)                 # :279
        elif o.distill != "kd":
            raise NotImplementedError(o.distill)                                         # :289-290
        if two and o.distill == "crd":
            kd_list = [o.alpha * loss_div1, o.alpha * loss_div2, o.beta * loss_kd1, o.beta * loss_kd2]  # :293-298
        elif two:
            kd_list = [o.alpha * loss_div1, o.alpha * loss_div2]                         # :299-300
        else:
            kd_list = None
        if o.assign_weights != "True":
            scale = None
            loss_KD = o.alpha * (loss_div1 + loss_div2) + o.beta * (loss_kd1 + loss_kd2)  # :309
        elif kd_list is None:
            raise UnboundLocalError("KD_loss_list (:293-304: built for two teachers only)")
        elif faithful:
            # the reference's AEKD_loss: one FULL backward per loss, only the hook value is kept
            gl = []
            for l in kd_list + [loss_cls]:
                gs = torch.autograd.grad(l, [path_feat] + params, retain_graph=True, allow_unused=True)
                gl.append(gs[0].detach().clone())
            all_g = torch.stack(gl).view(len(gl), -1)
            nrm = torch.norm(all_g, p=2, dim=1, keepdim=True)
            rel = all_g @ all_g.T * len(kd_list) / (nrm @ nrm.T)
            scale = rel.sum(1)
            loss_KD = torch.dot(scale[:-1], torch.stack(kd_list))
        else:
            scale, loss_KD = aekd_loss(loss_cls, path_feat, kd_list, o.CE_grads)     # :304
        loss = o.lambda_nll * loss_cls + loss_KD                                     # :313 (reg none, no cox)
        grads = torch.autograd.grad(loss, params, allow_unused=True)                 # :326-327
        for p in params:
            p.requires_grad_(False)
        with torch.no_grad():
            self._adam(grads)                                                        # :328
            self._ema_update()                                                       # :329
        self.iter_num += 1
        return dict(logit_path=logit_path.detach(), pred_path=pred_path.detach(),
                    path_feat=path_feat.detach(), ema_feat=ema_feat, ema_logit=ema_logit,
                    fuse_feat=fuse_feat, fuse_logit=logits[-1],
                    loss_cls=loss_cls.detach(), loss_div1=loss_div1.detach(), loss_div2=loss_div2.detach(),
                    loss_kd1=loss_kd1.detach(), loss_kd2=loss_kd2.detach(), scale=None if scale is None else scale.detach(),
                    loss_KD=loss_KD.detach(), loss=loss.detach(),
                    grads=OrderedDict((n, g) for n, g in zip(names, grads)),
                    aux1=aux1, aux2=aux2)
