#!/usr/bin/env python3
"""Golden vectors for the MIA-2023 stage-2 batch body, produced by importing and RUNNING the
reference's own modules from "/root/reference/MIA 2023/stage2_unimodal_student": networks_new.define_net /
define_optimizer, KD_loss.DistillKL (per-sample rows), CL_utils.CRD_criterion_v10.CRDLoss, and the trainer's
GK_refine_thresh / assign_sample_weights / update_ema_variables (train_test_path_multi_distill.py; the trainer module
itself imports packages that are absent here, so those functions are compiled from the file where it lies).  The loop
below makes the same calls, in the same order, as the trainer's batch body (:318-448).  Build container only.
Writes tests/golden/mia2023_step_b8_h64{,_fp64}.npz."""
import contextlib
import io
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
REF = "/root/reference/MIA 2023/stage2_unimodal_student"


def main():
    from make_golden import install_shims, npz
    install_shims()
    sys.path.insert(0, REF)
    os.chdir(REF)
    tmp = tempfile.mkdtemp()
    sys.argv = ["x", "--distill", "crd", "-a", "1", "-b", "0.02", "--num_teachers", "2", "--CE_grads",
                "--model_name", "golden", "--fixed_model", "t", "--reg_type", "none", "--beta1", "0.9",
                "--assign_weights", "True", "--cut_fuse_grad", "--input_size_omic", "320", "--dropout_rate", "0",
                "--gpu_ids", "-1", "--checkpoints_dir", tmp, "--nce_k", "256", "--nce_p", "6",
                "--neg_mode", "all_others", "--start_reweight", "1", "--pos_extra", "neighbors", "--max_discrep", "1",
                "--grads_thresh", "0.25", "--use_grads_thresh", "True", "--batch_size", "8"]
    with contextlib.redirect_stdout(io.StringIO()):
        import options_new as options        # the options file the MIA-2023 trainer scripts import
        opt = options.parse_args()
        import networks_new as NN
        from KD_loss import DistillKL
        import importlib
        crdv10 = importlib.import_module("CL_utils.CRD_criterion_v10")
    src = open(os.path.join(REF, "train_test_path_multi_distill.py")).read()
    from sklearn.metrics.pairwise import cosine_similarity
    ns = {"torch": torch, "np": np, "F": torch.nn.functional, "Variable": torch.autograd.Variable,
          "cosine_similarity": cosine_similarity}
    for a, b in (("def update_ema_variables", "def GK_refine("), ("def GK_refine_thresh", "def intra_inter_similarity")):
        s0 = src.index(a); s1 = src.index(b, s0)
        exec(compile(src[s0:s1], a + "<reference>", "exec"), ns)     # runs the reference's own function text
    GK_refine_thresh, assign_sample_weights = ns["GK_refine_thresh"], ns["assign_sample_weights"]
    update_ema_variables = ns["update_ema_variables"]
    F = torch.nn.functional

    from oracle import weights as W
    from oracle.step import synthetic_batch
    from oracle.variants import CRDv10State
    B, H, n_data, K = 8, 64, 1024, opt.nce_k
    labels = torch.randint(0, 3, (n_data,), generator=torch.Generator().manual_seed(11))
    class_idx = [np.nonzero((labels == c).numpy())[0] for c in range(3)]

    def run(dt, rec, warm=None, collect=False):
        with contextlib.redirect_stdout(io.StringIO()):
            student = NN.define_net(opt, 1, path_only=True)
            ema = NN.define_net(opt, 1, path_only=True)
            teacher = NN.define_net(opt, 1)
        student.load_state_dict(W.make_state_dict(W.student_shapes(), 1))
        ema.load_state_dict(W.make_state_dict(W.student_shapes(), 2))
        teacher.load_state_dict(W.make_state_dict(W.teacher_shapes(320), 3))
        for p in ema.parameters():
            p.detach_()
        for p in teacher.parameters():
            p.detach_(); p.requires_grad = False
        crds = []
        for i in range(2):
            torch.manual_seed(20 + i)
            with contextlib.redirect_stdout(io.StringIO()):
                c = crdv10.CRDLoss(opt, n_data, class_idx)
            c.embed_s.load_state_dict(W.make_state_dict(W.embed_shapes(), 10 + 2 * i))
            c.embed_t.load_state_dict(W.make_state_dict(W.embed_shapes(), 11 + 2 * i))
            st = CRDv10State(n_data, labels, K=K, seed=20 + i)
            c.contrast.memory_v1.copy_(st.memory_v1); c.contrast.memory_v2.copy_(st.memory_v2)
            crds.append(c)
        ml = torch.nn.ModuleList([student, crds[0].embed_s, crds[0].embed_t, crds[1].embed_s, crds[1].embed_t])
        optimizer = NN.define_optimizer(opt, ml)
        import _warm
        wnames, wparams = _warm.param_names(student), list(ml.parameters())
        if warm is not None:
            _warm.set_torch_adam(optimizer, wnames, wparams, warm)
            NN.define_scheduler(opt, optimizer)     # the trainer's LambdaLR applies its epoch-0 factor on creation
            rec["lr"] = optimizer.param_groups[0]["lr"]
        kl = DistillKL(opt.kd_T)
        for mod in (student, ema, teacher, crds[0], crds[1]):
            mod.to(dt)
        ml.train(); teacher.train()
        iter_num = _warm.T0 if warm is not None else 0
        epochs = [0, 1, 2]               # start_reweight = 1: step 0 runs with unit query weights, steps 1-2 re-weighted
        for it in range(3):
            bt = synthetic_batch(B, H, n_data=n_data, P=1, K=K, seed=400 + it)
            bt["grade"] = labels[bt["index"]]            # the bank's class table and the batch labels agree
            bt = {k: (v.to(dt) if v.dtype.is_floating_point else v) for k, v in bt.items()}
            epoch, grade = epochs[it], bt["grade"]
            _, path_feat, logit_path, pred_path, _ = student(x_path=bt["x_path"])
            with torch.no_grad():
                _, ema_path_feat, ema_logit_path, _, _ = ema(x_path=bt["ema_x_path"])
                fuse_feat, _, _, _, logits, pred, _, _, _, _, _ = teacher(x_path=bt["x_path"], x_omic=bt["x_omic"])
            loss_cls = F.nll_loss(pred_path, grade)
            loss_div1, sample_loss_div1 = kl(logit_path, logits[-1].detach())
            loss_div2, sample_loss_div2 = kl(logit_path, ema_logit_path.detach())
            w1 = assign_sample_weights(F.softmax(logit_path, 1), F.softmax(logits[-1], 1), grade, opt.discrep_scale,
                                       opt.max_discrep)
            w2 = assign_sample_weights(F.softmax(logit_path, 1), F.softmax(ema_logit_path, 1), grade, opt.discrep_scale,
                                       opt.max_discrep)
            if epoch < opt.start_reweight:
                w1 = torch.ones_like(w1); w2 = torch.ones_like(w2)
            else:
                w1 += 1; w2 += 1
            w1 = w1.view(-1, 1); w2 = w2.view(-1, 1)
            with contextlib.redirect_stdout(io.StringIO()):
                loss_kd1, sample_loss_kd1 = crds[0](w1, path_feat, fuse_feat.detach(), grade, bt["index"], bt["sample_idx"])
                loss_kd2, sample_loss_kd2 = crds[1](w2, path_feat, ema_path_feat.detach(), grade, bt["index"],
                                                    bt["sample_idx"])
            kd_list = [opt.alpha * sample_loss_div1, opt.alpha * sample_loss_div2, opt.beta * sample_loss_kd1,
                       opt.beta * sample_loss_kd2]
            scale, loss_KD = GK_refine_thresh(opt, optimizer, loss_cls, path_feat, kd_list)
            loss = opt.lambda_nll * loss_cls + loss_KD        # reg_type none: define_reg contributes 0
            optimizer.zero_grad()
            loss.backward()
            if collect:
                return _warm.grad_scales(wnames, wparams, opt.weight_decay)
            if it == 0:
                rec.update(g0_conv1=student.conv1.weight.grad.clone(), g0_fc2_w=student.fc_new2.weight.grad.clone(),
                           g0_embed_s0=crds[0].embed_s.linear.weight.grad.clone(),
                           g0_embed_t1=crds[1].embed_t.linear.weight.grad.clone())
            optimizer.step()
            update_ema_variables(student, ema, opt.ema_decay, iter_num)
            iter_num += 1
            sd = student.state_dict(); esd = ema.state_dict()
            rec.update({f"epoch{it}": epoch, f"logit_path{it}": logit_path, f"path_feat{it}": path_feat,
                        f"ema_logit{it}": ema_logit_path, f"fuse_logit{it}": logits[-1], f"loss_cls{it}": loss_cls,
                        f"loss_div1_{it}": loss_div1, f"loss_div2_{it}": loss_div2, f"loss_kd1_{it}": loss_kd1,
                        f"loss_kd2_{it}": loss_kd2, f"scale{it}": scale.clone(), f"loss_KD{it}": loss_KD, f"loss{it}": loss,
                        f"w1_{it}": w1, f"w2_{it}": w2, f"rows_div1_{it}": sample_loss_div1, f"rows_kd1_{it}": sample_loss_kd1,
                        f"p_fc2_{it}": sd["fc_new2.weight"].clone(), f"ema_fc2_{it}": esd["fc_new2.weight"].clone(),
                        f"bank0_v1_rows{it}": crds[0].contrast.memory_v1[bt["index"]].clone(),
                        f"bank1_v2_rows{it}": crds[1].contrast.memory_v2[bt["index"]].clone(),
                        f"params0_{it}": crds[0].contrast.params.clone()})
            print(str(dt), "step", it, "loss", float(loss), "scale", scale.tolist())

    rec = dict(B=B, H=H, n_data=n_data, K=K, num_pos=opt.nce_p, labels=labels, alpha=opt.alpha, beta=opt.beta,
               start_reweight=opt.start_reweight, max_discrep=opt.max_discrep, grads_thresh=opt.grads_thresh)
    rec_meta = dict(rec)
    run(torch.float32, rec)
    np.savez_compressed(os.path.join(HERE, "mia2023_step_b8_h64.npz"), **npz(rec))
    # the same three steps from a mid-training optimiser state (tests/golden/_warm.py): every step comparable at 1e-3
    import _warm
    scales = run(torch.float32, {}, collect=True)
    recw = dict(rec_meta)
    recw.update(_warm.pack_scales(scales)); recw["t0"] = _warm.T0
    run(torch.float32, recw, warm=scales)
    recw = {k: v for k, v in recw.items() if not k.startswith(("path_feat", "g0_conv1"))}
    np.savez_compressed(os.path.join(HERE, "mia2023_step_warm_b8_h64.npz"), **npz(recw))
    # the same calls in double precision: the noise floor that steps >= 1 are judged against (Adam's first updates
    # are sign-like and amplify fp32 rounding; see tests/golden/make_fp64_truth.py)
    rec64 = {}
    run(torch.float64, rec64)
    keep = ("logit_path", "path_feat", "ema_logit", "loss", "scale", "bank0", "bank1", "g0_conv1", "p_fc2", "ema_fc2",
            "w1_", "w2_", "rows_")
    rec64 = {k: v for k, v in rec64.items() if k.startswith(keep)}
    np.savez_compressed(os.path.join(HERE, "mia2023_step_b8_h64_fp64.npz"), **npz(rec64))
    print("written mia2023_step_b8_h64{,_fp64}.npz")


if __name__ == "__main__":
    main()
