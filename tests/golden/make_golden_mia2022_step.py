#!/usr/bin/env python3
"""Golden vectors for the MIA-2022 stage-2 batch body, produced by importing and RUNNING the
reference's own modules from "/root/reference/MIA 2022": networks_new.define_net / define_optimizer, KD_loss.DistillKL,
CL_utils.CRD_criterion_v3.CRDLoss, and the trainer's momentum_AEKD_loss / update_ema_variables
(train_test_path_multi_distill_v2.py; the trainer module itself imports dgl / torchvision, which are absent here, so
those two functions are compiled from the file where it lies).  The loop below makes the same calls, in the same order,
as the trainer's batch body (:397-507).  Build container only.  Writes tests/golden/mia2022_step_b8_h64.npz."""
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
REF = "/root/reference/MIA 2022"


def main():
    from make_golden import install_shims, npz
    install_shims()
    sys.path.insert(0, REF)
    os.chdir(REF)
    tmp = tempfile.mkdtemp()
    sys.argv = ["x", "--distill", "crd", "-a", "1", "-b", "0.02", "--num_teachers", "2", "--CE_grads",
                "--model_name", "golden", "--fixed_model", "t", "--reg_type", "none", "--beta1", "0.9",
                "--assign_weights", "True", "--cut_fuse_grad", "--input_size_omic", "320", "--dropout_rate", "0",
                "--gpu_ids", "-1", "--checkpoints_dir", tmp, "--nce_k", "512", "--grads_m", "0.9",
                "--grads_thresh", "False"]
    with contextlib.redirect_stdout(io.StringIO()):
        import options
        opt = options.parse_args()
        import networks_new as NN
        from KD_loss import DistillKL
        import importlib
        crdv3 = importlib.import_module("CL_utils.CRD_criterion_v3")
    src = open(os.path.join(REF, "train_test_path_multi_distill_v2.py")).read()
    ns = {"torch": torch, "Variable": torch.autograd.Variable}
    for a, b in (("def momentum_AEKD_loss", "def AEKD_loss"), ("def update_ema_variables", "def train(")):
        s0 = src.index(a); s1 = src.index(b, s0)
        exec(compile(src[s0:s1], a + "<reference>", "exec"), ns)     # runs the reference's own function text
    momentum_AEKD_loss, update_ema_variables = ns["momentum_AEKD_loss"], ns["update_ema_variables"]

    from oracle import weights as W
    from oracle.step import synthetic_batch
    from oracle.variants import CRDv3State
    B, H, n_data, K = 8, 64, 1024, opt.nce_k

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
                c = crdv3.CRDLoss(opt, n_data)
            c.embed_s.load_state_dict(W.make_state_dict(W.embed_shapes(), 10 + 2 * i))
            c.embed_t.load_state_dict(W.make_state_dict(W.embed_shapes(), 11 + 2 * i))
            st = CRDv3State(n_data, K=K, seed=20 + i)
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
        scale, iter_num = None, (_warm.T0 if warm is not None else 0)
        epochs = [3, 3, 7]               # the epoch weight changes between steps (the CRD loss is multiplied by it)
        for it in range(3):
            bt = synthetic_batch(B, H, n_data=n_data, P=1, K=K, seed=300 + it)
            bt = {k: (v.to(dt) if v.dtype.is_floating_point else v) for k, v in bt.items()}
            epoch = epochs[it]
            _, path_feat, logit_path, pred_path, _ = student(x_path=bt["x_path"])
            with torch.no_grad():
                _, ema_path_feat, ema_logit_path, _, _ = ema(x_path=bt["ema_x_path"])
                fuse_feat, _, _, _, logits, pred, _, _, _, _, _ = teacher(x_path=bt["x_path"], x_omic=bt["x_omic"])
            loss_cls = torch.nn.functional.nll_loss(pred_path, bt["grade"])
            loss_div1 = kl(logit_path, logits[-1].detach())
            loss_div2 = kl(logit_path, ema_logit_path.detach())
            with contextlib.redirect_stdout(io.StringIO()):
                loss_kd1 = crds[0](epoch / opt.niter_decay, path_feat, fuse_feat.detach(), bt["index"], bt["sample_idx"])
                loss_kd2 = crds[1](epoch / opt.niter_decay, path_feat, ema_path_feat.detach(), bt["index"], bt["sample_idx"])
            loss_div1 = opt.alpha * loss_div1; loss_div2 = opt.alpha * loss_div2
            loss_kd1 = opt.beta * loss_kd1.reshape(()); loss_kd2 = opt.beta * loss_kd2.reshape(())
            kd_list = [loss_div1, loss_div2, loss_kd1, loss_kd2]
            scale, loss_KD = momentum_AEKD_loss(opt, optimizer, loss_cls, path_feat, kd_list, scale)
            if opt.grads_thresh == "False":
                loss_KD = loss_KD * len(kd_list)
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
            scale = scale.detach()
            sd = student.state_dict(); esd = ema.state_dict()
            rec.update({f"epoch{it}": epoch, f"logit_path{it}": logit_path, f"path_feat{it}": path_feat,
                        f"ema_logit{it}": ema_logit_path, f"fuse_logit{it}": logits[-1], f"loss_cls{it}": loss_cls,
                        f"loss_div1_{it}": loss_div1, f"loss_div2_{it}": loss_div2, f"loss_kd1_{it}": loss_kd1,
                        f"loss_kd2_{it}": loss_kd2, f"scale{it}": scale.clone(), f"loss_KD{it}": loss_KD, f"loss{it}": loss,
                        f"p_fc2_{it}": sd["fc_new2.weight"].clone(), f"ema_fc2_{it}": esd["fc_new2.weight"].clone(),
                        f"bank0_v1_rows{it}": crds[0].contrast.memory_v1[bt["index"]].clone(),
                        f"bank1_v2_rows{it}": crds[1].contrast.memory_v2[bt["index"]].clone(),
                        f"params0_{it}": crds[0].contrast.params.clone()})
            print(str(dt), "step", it, "loss", float(loss), "scale", scale.tolist())

    rec = dict(B=B, H=H, n_data=n_data, K=K, grads_m=opt.grads_m, niter_decay=opt.niter_decay, alpha=opt.alpha,
               beta=opt.beta)
    rec_meta = dict(rec)
    run(torch.float32, rec)
    np.savez_compressed(os.path.join(HERE, "mia2022_step_b8_h64.npz"), **npz(rec))
    # the same three steps from a mid-training optimiser state (tests/golden/_warm.py): every step comparable at 1e-3
    import _warm
    scales = run(torch.float32, {}, collect=True)
    recw = dict(rec_meta)
    recw.update(_warm.pack_scales(scales)); recw["t0"] = _warm.T0
    run(torch.float32, recw, warm=scales)
    recw = {k: v for k, v in recw.items() if not k.startswith(("path_feat", "g0_conv1"))}
    np.savez_compressed(os.path.join(HERE, "mia2022_step_warm_b8_h64.npz"), **npz(recw))
    # the same calls in double precision: the noise floor that steps >= 1 are judged against (Adam's first updates
    # are sign-like and amplify fp32 rounding; see tests/golden/make_fp64_truth.py)
    rec64 = {}
    run(torch.float64, rec64)
    keep = ("logit_path", "path_feat", "ema_logit", "loss", "scale", "bank0", "bank1", "g0_conv1", "p_fc2", "ema_fc2")
    rec64 = {k: v for k, v in rec64.items() if k.startswith(keep)}
    np.savez_compressed(os.path.join(HERE, "mia2022_step_b8_h64_fp64.npz"), **npz(rec64))
    print("written mia2022_step_b8_h64{,_fp64}.npz")


if __name__ == "__main__":
    main()
