#!/usr/bin/env python3
"""Golden vectors for the NON-DEFAULT option branches of the MIA-2022 and MIA-2023 stage-2 batch bodies
("MIA 2022/train_test_path_multi_distill_v2.py":419-482, "MIA 2023/stage2_unimodal_student/train_test_path_multi_distill.py":
348-427): `--num_teachers 1` with the fused or the mean teacher (ONE KL term, ONE CRD call through criterion_kd, fixed weights
alpha / beta) and - MIA-2023 - `--distill kd` with two teachers under the per-sample GK-Refine.  Produced by importing and
RUNNING each reference's own modules (the two trees carry modules of the same names: one child process per tree); the loop makes
the same calls under the same `if` / `elif` conditions as the trainer's batch body.  Two steps per branch from a mid-training
optimiser state (tests/golden/_warm.py).  Build container only.  Writes tests/golden/branches_mia2022.npz, branches_mia2023.npz."""
import contextlib
import io
import os
import subprocess
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

BR22 = [("t1_fuse_crd", 1, "fuse", "crd", "False"), ("t1_ema_crd", 1, "self_EMA", "crd", "False")]
BR23 = [("t1_fuse_crd", 1, "fuse", "crd", "False"), ("t1_ema_crd", 1, "self_EMA", "crd", "False"),
        ("t1_fuse_kd", 1, "fuse", "kd", "False"), ("t2_kd_gk", 2, "fuse", "kd", "True")]


def common_setup(REF, argv, options_mod):
    from make_golden import install_shims
    install_shims()
    sys.path.insert(0, REF)
    os.chdir(REF)
    sys.argv = argv
    with contextlib.redirect_stdout(io.StringIO()):
        options = __import__(options_mod)
        opt = options.parse_args()
        import networks_new as NN
        from KD_loss import DistillKL
    return opt, NN, DistillKL


def nets(NN, opt, W):
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
    return student, ema, teacher


def record(rec, pre, it, logit_path, loss_cls, loss_div, loss_kd, loss_KD, loss, scale, student, ema, crds, index, g, wnames):
    sd = student.state_dict(); esd = ema.state_dict()
    rec.update({pre + f"logit_path{it}": logit_path, pre + f"loss_cls{it}": loss_cls, pre + f"loss_div{it}": loss_div,
                pre + f"loss_kd{it}": torch.as_tensor(float(loss_kd)), pre + f"loss_KD{it}": loss_KD, pre + f"loss{it}": loss,
                pre + f"p_fc2_{it}": sd["fc_new2.weight"].clone(), pre + f"ema_fc2_{it}": esd["fc_new2.weight"].clone(),
                pre + f"g_fc2_{it}": g["student.fc_new2.weight"],
                pre + f"g_l4_1_conv2_abs{it}": g["student.layer4.1.conv2.weight"].abs().sum(),
                pre + f"embed_s0_{it}": crds[0].embed_s.linear.weight.detach()[:8].clone(),
                pre + f"embed_t1_{it}": crds[1].embed_t.linear.weight.detach()[:8].clone(),
                pre + f"bank0_v1_rows{it}": crds[0].contrast.memory_v1[index].clone(),
                pre + f"bank1_v2_rows{it}": crds[1].contrast.memory_v2[index].clone(),
                pre + f"params0_{it}": crds[0].contrast.params.clone(), pre + f"params1_{it}": crds[1].contrast.params.clone(),
                pre + f"no_grad{it}": np.array([n for n in wnames if g[n] is None])})
    if scale is not None:
        rec[pre + f"scale{it}"] = scale


def gen_mia2022():
    REF = "/root/reference/MIA 2022"
    tmp = tempfile.mkdtemp()
    argv = ["x", "--distill", "crd", "-a", "1", "-b", "0.02", "--num_teachers", "2", "--CE_grads", "--model_name", "golden",
            "--fixed_model", "t", "--reg_type", "none", "--beta1", "0.9", "--assign_weights", "True", "--cut_fuse_grad",
            "--input_size_omic", "320", "--dropout_rate", "0", "--gpu_ids", "-1", "--checkpoints_dir", tmp, "--nce_k", "512",
            "--grads_m", "0.9", "--grads_thresh", "False"]
    opt, NN, DistillKL = common_setup(REF, argv, "options")
    import importlib
    crdv3 = importlib.import_module("CL_utils.CRD_criterion_v3")
    src = open(os.path.join(REF, "train_test_path_multi_distill_v2.py")).read()
    ns = {"torch": torch, "Variable": torch.autograd.Variable}
    for a, b in (("def momentum_AEKD_loss", "def AEKD_loss"), ("def update_ema_variables", "def train(")):
        s0 = src.index(a); s1 = src.index(b, s0)
        exec(compile(src[s0:s1], a + "<reference>", "exec"), ns)
    momentum_AEKD_loss, update_ema_variables = ns["momentum_AEKD_loss"], ns["update_ema_variables"]
    from oracle import weights as W
    from oracle.step import synthetic_batch
    from oracle.variants import CRDv3State
    from make_golden import npz
    import _warm
    B, H, n_data, K = 8, 64, 1024, opt.nce_k

    def run(br, rec, warm=None, collect=False):
        name, opt.num_teachers, opt.which_teacher, opt.distill, opt.assign_weights = br
        student, ema, teacher = nets(NN, opt, W)
        crds = []
        for i in range(2):
            torch.manual_seed(20 + i)
            with contextlib.redirect_stdout(io.StringIO()):
                c = crdv3.CRDLoss(opt, n_data)
            c.embed_s.load_state_dict(W.make_state_dict(W.embed_shapes(), 10 + 2 * i))
            c.embed_t.load_state_dict(W.make_state_dict(W.embed_shapes(), 11 + 2 * i))
            st = CRDv3State(n_data, K=K, seed=20 + i)
            c.contrast.memory_v1.copy_(st.memory_v1); c.contrast.memory_v2.copy_(st.memory_v2)
            crds.append(c)
        criterion_kd, criterion_kd_path = crds
        ml = torch.nn.ModuleList([student, crds[0].embed_s, crds[0].embed_t, crds[1].embed_s, crds[1].embed_t])
        optimizer = NN.define_optimizer(opt, ml)
        wnames, wparams = _warm.param_names(student), list(ml.parameters())
        if warm is not None:
            _warm.set_torch_adam(optimizer, wnames, wparams, warm)
        NN.define_scheduler(opt, optimizer)
        rec["lr"] = optimizer.param_groups[0]["lr"]
        criterion_div = DistillKL(opt.kd_T)
        ml.train(); teacher.train()
        scale, iter_num = None, (_warm.T0 if warm is not None else 0)
        for it in range(2):
            epoch = 3 + 4 * it
            bt = synthetic_batch(B, H, n_data=n_data, P=1, K=K, seed=600 + it)
            index, sample_idx, grade = bt["index"], bt["sample_idx"], bt["grade"]
            _, path_feat, logit_path, pred_path, _ = student(x_path=bt["x_path"])
            with torch.no_grad():
                _, ema_path_feat, ema_logit_path, _, _ = ema(x_path=bt["ema_x_path"])
                fuse_feat, _, _, _, logits, pred, _, _, _, _, _ = teacher(x_path=bt["x_path"], x_omic=bt["x_omic"])
            loss_cls = torch.nn.functional.nll_loss(pred_path, grade)
            # ---- :419-426
            if opt.num_teachers == 2:
                loss_div1 = criterion_div(logit_path, logits[-1].detach())
                loss_div2 = criterion_div(logit_path, ema_logit_path.detach())
                loss_div = loss_div1 + loss_div2
            elif opt.num_teachers == 1 and opt.which_teacher == "fuse":
                loss_div = criterion_div(logit_path, logits[-1].detach())
            elif opt.num_teachers == 1 and opt.which_teacher == "self_EMA":
                loss_div = criterion_div(logit_path, ema_logit_path.detach())
            # ---- :429-444
            with contextlib.redirect_stdout(io.StringIO()):
                if opt.distill == "kd":
                    loss_kd = 0
                elif opt.distill == "crd":
                    if opt.num_teachers == 2:
                        loss_kd1 = criterion_kd(epoch / opt.niter_decay, path_feat, fuse_feat.detach(), index, sample_idx)
                        loss_kd2 = criterion_kd_path(epoch / opt.niter_decay, path_feat, ema_path_feat.detach(), index, sample_idx)
                        loss_kd = loss_kd1 + loss_kd2
                    elif opt.num_teachers == 1 and opt.which_teacher == "fuse":
                        loss_kd = criterion_kd(epoch / opt.niter_decay, path_feat, fuse_feat.detach(), index, sample_idx)
                    elif opt.num_teachers == 1 and opt.which_teacher == "self_EMA":
                        loss_kd = criterion_kd(epoch / opt.niter_decay, path_feat, ema_path_feat.detach(), index, sample_idx)
            # ---- :461-482
            if opt.num_teachers == 2:
                loss_div1 = opt.alpha * loss_div1; loss_div2 = opt.alpha * loss_div2
                if opt.distill == "crd":
                    # (the criterion returns shape [1]; torch.stack inside momentum_AEKD_loss needs equal shapes - the same
                    # harness reshape as tests/golden/make_golden_mia2022_step.py)
                    loss_kd1 = opt.beta * loss_kd1.reshape(()); loss_kd2 = opt.beta * loss_kd2.reshape(())
                    KD_loss_list = [loss_div1, loss_div2, loss_kd1, loss_kd2]
                elif opt.distill == "kd":
                    KD_loss_list = [loss_div1, loss_div2]
            sc = None
            if opt.assign_weights == "True":
                scale, loss_KD = momentum_AEKD_loss(opt, optimizer, loss_cls, path_feat, KD_loss_list, scale)
                if opt.grads_thresh == "False":
                    loss_KD = loss_KD * len(KD_loss_list)
                sc = scale.detach().clone()
            else:
                loss_KD = opt.alpha * loss_div + opt.beta * loss_kd
            loss_KD = loss_KD.reshape(())
            loss = opt.lambda_nll * loss_cls + opt.lambda_reg * NN.define_reg(opt, student) + loss_KD
            optimizer.zero_grad()
            loss.backward()
            if collect:
                return _warm.grad_scales(wnames, wparams, opt.weight_decay)
            g = {n: (None if p.grad is None else p.grad.clone()) for n, p in zip(wnames, wparams)}
            optimizer.step()
            update_ema_variables(student, ema, opt.ema_decay, iter_num)
            iter_num += 1
            if scale is not None:
                scale = scale.detach()
            rec[name + f".epoch{it}"] = epoch
            record(rec, name + ".", it, logit_path, loss_cls, loss_div, float(torch.as_tensor(loss_kd).reshape(-1)[0]) if torch.is_tensor(loss_kd) else loss_kd,
                   loss_KD, loss, sc, student, ema, crds, index, g, wnames)
            print("mia2022", name, "step", it, "loss", float(loss), "loss_KD", float(loss_KD))

    rec = dict(B=B, H=H, n_data=n_data, K=K, grads_m=opt.grads_m, niter_decay=opt.niter_decay, t0=_warm.T0,
               names=np.array([b[0] for b in BR22]))
    scales = run(("scales", 2, "fuse", "crd", "True"), {}, collect=True)
    rec.update(_warm.pack_scales(scales))
    for br in BR22:
        run(br, rec, warm=scales)
    np.savez_compressed(os.path.join(HERE, "branches_mia2022.npz"), **npz(rec))


def gen_mia2023():
    REF = "/root/reference/MIA 2023/stage2_unimodal_student"
    tmp = tempfile.mkdtemp()
    argv = ["x", "--distill", "crd", "-a", "1", "-b", "0.02", "--num_teachers", "2", "--CE_grads", "--model_name", "golden",
            "--fixed_model", "t", "--reg_type", "none", "--beta1", "0.9", "--assign_weights", "True", "--cut_fuse_grad",
            "--input_size_omic", "320", "--dropout_rate", "0", "--gpu_ids", "-1", "--checkpoints_dir", tmp, "--nce_k", "256",
            "--nce_p", "6", "--neg_mode", "all_others", "--start_reweight", "1", "--pos_extra", "neighbors", "--max_discrep", "1",
            "--grads_thresh", "0.25", "--use_grads_thresh", "True", "--batch_size", "8"]
    opt, NN, DistillKL = common_setup(REF, argv, "options_new")
    import importlib
    crdv10 = importlib.import_module("CL_utils.CRD_criterion_v10")
    src = open(os.path.join(REF, "train_test_path_multi_distill.py")).read()
    from sklearn.metrics.pairwise import cosine_similarity
    F = torch.nn.functional
    ns = {"torch": torch, "np": np, "F": F, "Variable": torch.autograd.Variable, "cosine_similarity": cosine_similarity}
    for a, b in (("def update_ema_variables", "def GK_refine("), ("def GK_refine_thresh", "def intra_inter_similarity")):
        s0 = src.index(a); s1 = src.index(b, s0)
        exec(compile(src[s0:s1], a + "<reference>", "exec"), ns)
    GK_refine_thresh, assign_sample_weights = ns["GK_refine_thresh"], ns["assign_sample_weights"]
    update_ema_variables = ns["update_ema_variables"]
    from oracle import weights as W
    from oracle.step import synthetic_batch
    from oracle.variants import CRDv10State
    from make_golden import npz
    import _warm
    B, H, n_data, K = 8, 64, 1024, opt.nce_k
    labels = torch.randint(0, 3, (n_data,), generator=torch.Generator().manual_seed(11))
    class_idx = [np.nonzero((labels == c).numpy())[0] for c in range(3)]

    def run(br, rec, warm=None, collect=False):
        name, opt.num_teachers, opt.which_teacher, opt.distill, opt.assign_weights = br
        student, ema, teacher = nets(NN, opt, W)
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
        criterion_kd, criterion_kd_path = crds
        ml = torch.nn.ModuleList([student, crds[0].embed_s, crds[0].embed_t, crds[1].embed_s, crds[1].embed_t])
        optimizer = NN.define_optimizer(opt, ml)
        wnames, wparams = _warm.param_names(student), list(ml.parameters())
        if warm is not None:
            _warm.set_torch_adam(optimizer, wnames, wparams, warm)
        NN.define_scheduler(opt, optimizer)
        rec["lr"] = optimizer.param_groups[0]["lr"]
        criterion_div = DistillKL(opt.kd_T)
        ml.train(); teacher.train()
        iter_num = _warm.T0 if warm is not None else 0
        for it in range(2):
            epoch = it                       # start_reweight = 1: step 0 with unit query weights, step 1 re-weighted
            bt = synthetic_batch(B, H, n_data=n_data, P=1, K=K, seed=700 + it)
            bt["grade"] = labels[bt["index"]]
            index, sample_idx, grade = bt["index"], bt["sample_idx"], bt["grade"]
            _, path_feat, logit_path, pred_path, _ = student(x_path=bt["x_path"])
            with torch.no_grad():
                _, ema_path_feat, ema_logit_path, _, _ = ema(x_path=bt["ema_x_path"])
                fuse_feat, _, _, _, logits, pred, _, _, _, _, _ = teacher(x_path=bt["x_path"], x_omic=bt["x_omic"])
            loss_cls = F.nll_loss(pred_path, grade)
            # ---- :348-355
            if opt.num_teachers == 2:
                loss_div1, sample_loss_div1 = criterion_div(logit_path, logits[-1].detach())
                loss_div2, sample_loss_div2 = criterion_div(logit_path, ema_logit_path.detach())
                loss_div = loss_div1 + loss_div2
            elif opt.num_teachers == 1 and opt.which_teacher == "fuse":
                loss_div, sample_loss_div = criterion_div(logit_path, logits[-1].detach())
            elif opt.num_teachers == 1 and opt.which_teacher == "self_EMA":
                loss_div, sample_loss_div = criterion_div(logit_path, ema_logit_path.detach())
            teacher1_sample_weights = assign_sample_weights(F.softmax(logit_path, 1), F.softmax(logits[-1], 1), grade,
                                                            opt.discrep_scale, opt.max_discrep)
            teacher2_sample_weights = assign_sample_weights(F.softmax(logit_path, 1), F.softmax(ema_logit_path, 1), grade,
                                                            opt.discrep_scale, opt.max_discrep)
            # ---- :366-398
            with contextlib.redirect_stdout(io.StringIO()):
                if opt.distill == "kd":
                    loss_kd = 0
                elif opt.distill == "crd":
                    if epoch < opt.start_reweight:
                        teacher1_sample_weights = torch.ones_like(teacher1_sample_weights)
                        teacher2_sample_weights = torch.ones_like(teacher2_sample_weights)
                    else:
                        teacher1_sample_weights += 1
                        teacher2_sample_weights += 1
                    teacher1_sample_weights = teacher1_sample_weights.view(-1, 1)
                    teacher2_sample_weights = teacher2_sample_weights.view(-1, 1)
                    if opt.num_teachers == 2:
                        loss_kd1, sample_loss_kd1 = criterion_kd(teacher1_sample_weights, path_feat, fuse_feat.detach(), grade, index, sample_idx)
                        loss_kd2, sample_loss_kd2 = criterion_kd_path(teacher2_sample_weights, path_feat, ema_path_feat.detach(), grade, index, sample_idx)
                        loss_kd = loss_kd1 + loss_kd2
                    elif opt.num_teachers == 1 and opt.which_teacher == "fuse":
                        loss_kd, sample_loss_kd = criterion_kd(teacher1_sample_weights, path_feat, fuse_feat.detach(), grade, index, sample_idx)
                    elif opt.num_teachers == 1 and opt.which_teacher == "self_EMA":
                        loss_kd, sample_loss_kd = criterion_kd(teacher2_sample_weights, path_feat, ema_path_feat.detach(), grade, index, sample_idx)
            # ---- :403-427
            if opt.num_teachers == 2:
                loss_div1 = opt.alpha * sample_loss_div1
                loss_div2 = opt.alpha * sample_loss_div2
                if opt.distill == "crd":
                    loss_kd1 = opt.beta * sample_loss_kd1; loss_kd2 = opt.beta * sample_loss_kd2
                    KD_loss_list = [loss_div1, loss_div2, loss_kd1, loss_kd2]
                elif opt.distill == "kd":
                    KD_loss_list = [sample_loss_div1, sample_loss_div2]
            scale = None
            if opt.assign_weights == "True":
                if opt.loss_weighting == "GK_refine":
                    scale, loss_KD = GK_refine_thresh(opt, optimizer, loss_cls, path_feat, KD_loss_list)
            else:
                loss_KD = opt.alpha * loss_div + opt.beta * loss_kd
            loss = opt.lambda_nll * loss_cls + opt.lambda_reg * NN.define_reg(opt, student) + loss_KD
            optimizer.zero_grad()
            loss.backward()
            if collect:
                return _warm.grad_scales(wnames, wparams, opt.weight_decay)
            g = {n: (None if p.grad is None else p.grad.clone()) for n, p in zip(wnames, wparams)}
            optimizer.step()
            update_ema_variables(student, ema, opt.ema_decay, iter_num)
            iter_num += 1
            rec[name + f".epoch{it}"] = epoch
            record(rec, name + ".", it, logit_path, loss_cls, loss_div, loss_kd, loss_KD, loss, scale, student, ema, crds, index, g, wnames)
            print("mia2023", name, "step", it, "loss", float(loss), "loss_KD", float(loss_KD),
                  "scale", None if scale is None else scale.tolist())

    rec = dict(B=B, H=H, n_data=n_data, K=K, num_pos=opt.nce_p, labels=labels, t0=_warm.T0, start_reweight=opt.start_reweight,
               max_discrep=opt.max_discrep, grads_thresh=opt.grads_thresh, names=np.array([b[0] for b in BR23]))
    scales = run(("scales", 2, "fuse", "crd", "True"), {}, collect=True)
    rec.update(_warm.pack_scales(scales))
    for br in BR23:
        run(br, rec, warm=scales)
    np.savez_compressed(os.path.join(HERE, "branches_mia2023.npz"), **npz(rec))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "mia2022":
        gen_mia2022()
    elif len(sys.argv) > 1 and sys.argv[1] == "mia2023":
        gen_mia2023()
    else:
        for which in ("mia2022", "mia2023"):
            subprocess.check_call([sys.executable, os.path.abspath(__file__), which])
        print("written branches_mia2022.npz, branches_mia2023.npz")
