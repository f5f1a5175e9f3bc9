#!/usr/bin/env python3
"""Golden vectors for the L1 weight regulariser `opt.lambda_reg * define_reg(opt, model)` of the trainers
(MICCAI-2022/train_test_MT.py:209-217, networks_new.py:93-108, utils.py:60-198), produced by running the reference's own
functions.  Build container only.  Writes tests/golden/stage1_reg_b4_h64.npz:

* the value of define_reg on the reference's PathomicNet for --reg_type omic (the default the shipped stage-1 command
  keeps, options.py:132), mm and all, and which reg types raise AttributeError (path on PathomicNet - no `.linear`;
  path / mm / omic on the ResNet student - no `__hasattr__`);
* two steps of the stage-1 mean-teacher batch body (the recipe of make_golden_stage1.py) with the DEFAULT --reg_type
  and --lambda_reg: losses, predictions, the regulariser's value and the gradient of two omic_net tensors before the
  first update (the L1 term is 3e-4 * sgn(W) there), updated weights.

The reference unwraps DataParallel with `model.module`; on CPU (`--gpu_ids -1`) init_net does not wrap, so the
generator gives the bare network a `.module` alias to itself, which is what DataParallel would provide."""
import contextlib
import io
import os
import sys
import tempfile

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
REF = "/root/reference/MICCAI-2022"


def main():
    from make_golden import install_shims, npz
    install_shims()
    sys.path.insert(0, REF)
    os.chdir(REF)
    from oracle import weights as W
    from oracle.step import synthetic_batch
    # the stage-1 README command (README.md:26-27) passes neither --reg_type nor --lambda_reg: defaults apply
    sys.argv = ["x", "--model_name", "golden", "--beta1", "0.9", "--input_size_omic", "320", "--dropout_rate", "0",
                "--gpu_ids", "-1", "--checkpoints_dir", tempfile.mkdtemp(), "--num_teachers", "2"]
    with contextlib.redirect_stdout(io.StringIO()):
        import options
        opt = options.parse_args()
        import networks_new as NN
        from CL_utils.KD_losses import pred_KD_loss
    assert opt.reg_type == "omic" and abs(opt.lambda_reg - 3e-4) < 1e-12, (opt.reg_type, opt.lambda_reg)
    opt.cut_fuse_grad = False

    def net(path_only=False):
        with contextlib.redirect_stdout(io.StringIO()):
            n = NN.define_net(opt, 1, path_only=path_only)
        n.__dict__["module"] = n          # the DataParallel unwrap the regularisers perform
        return n

    rec = dict(B=4, H=64, weight_seed=3, lr=opt.lr, weight_decay=opt.weight_decay, ema_decay=opt.ema_decay,
               lambda_reg=opt.lambda_reg, KD_weight=opt.KD_weight)
    # ---- values and error behaviour of define_reg
    model = net()
    model.load_state_dict(W.make_state_dict(W.teacher_shapes(320), 3))
    student = net(path_only=True)
    student.load_state_dict(W.make_state_dict(W.student_shapes(), 1))
    for rt in ("omic", "mm", "all"):
        opt.reg_type = rt
        rec["reg_teacher_" + rt] = NN.define_reg(opt, model).detach()
    opt.reg_type = "all"
    rec["reg_student_all"] = NN.define_reg(opt, student).detach()
    raises = []
    for who, m_ in (("teacher", model), ("student", student)):
        for rt in ("path", "mm", "omic"):
            opt.reg_type = rt
            try:
                NN.define_reg(opt, m_)
            except AttributeError:
                raises.append(who + ":" + rt)
    rec["raises_attribute_error"] = np.array(raises)
    opt.reg_type = "omic"

    # ---- two steps of train_test_MT.py:121-230 with the default regulariser
    ema = net()
    ema.load_state_dict(W.make_state_dict(W.teacher_shapes(320), 3))
    for p in ema.parameters():
        p.detach_()
    optimizer = NN.define_optimizer(opt, model)
    model.train(); ema.train()

    def update_ema_variables(model, ema_model, alpha, global_step):     # train_test_MT.py:34-38
        alpha = min(1 - 1 / (global_step + 1), alpha)
        for ema_param, param in zip(ema_model.parameters(), model.parameters()):
            ema_param.data.mul_(alpha).add_(param.data, alpha=1 - alpha)

    iter_num = 0
    watch = ("omic_net.encoder.0.0.weight", "omic_net.classifier.0.bias", "fusion.linear_h1.0.weight", "classifier.0.weight")
    for it in range(2):
        bt = synthetic_batch(4, 64, seed=20 + it)
        out = model(x_path=bt["x_path"], x_omic=bt["x_omic"])
        pred, pred_path, pred_omic = out[5], out[6], out[7]
        with torch.no_grad():
            eo = ema(x_path=bt["ema_x_path"], x_omic=bt["x_omic"])
        ema_pred, ema_pred_path, ema_pred_omic = eo[5], eo[6], eo[7]
        kd_fuse = pred_KD_loss(opt, pred, ema_pred)
        kd_path = (pred_KD_loss(opt, pred_path, ema_pred_path) + pred_KD_loss(opt, pred_path, ema_pred)) / 2.0
        kd_omic = (pred_KD_loss(opt, pred_omic, ema_pred_omic) + pred_KD_loss(opt, pred_omic, ema_pred)) / 2.0
        loss_kd = opt.KD_weight * (kd_fuse + kd_path + kd_omic)
        g = bt["grade"]
        loss_nll = F.nll_loss(pred_path, g) + F.nll_loss(pred_omic, g) + F.nll_loss(pred, g)
        loss_reg = NN.define_reg(opt, model)                                            # :209
        loss = opt.lambda_nll * loss_nll + opt.lambda_reg * loss_reg + loss_kd         # :217-218 (grading task: loss_cox 0)
        optimizer.zero_grad()
        loss.backward()
        if it == 0:
            named = dict(model.named_parameters())
            for k in watch:
                rec["g0_" + k] = named[k].grad.clone()
        optimizer.step()
        update_ema_variables(model, ema, opt.ema_decay, iter_num)
        iter_num += 1
        rec[f"loss{it}"] = loss; rec[f"loss_nll{it}"] = loss_nll; rec[f"loss_kd{it}"] = loss_kd; rec[f"loss_reg{it}"] = loss_reg
        rec[f"pred{it}"] = pred; rec[f"pred_path{it}"] = pred_path; rec[f"pred_omic{it}"] = pred_omic
        sd = model.state_dict()
        for k in watch:
            rec[f"w{it}_{k}"] = sd[k].clone()
    np.savez_compressed(os.path.join(HERE, "stage1_reg_b4_h64.npz"), **npz(rec))
    print("wrote stage1_reg_b4_h64.npz", [round(float(rec[f"loss{i}"]), 5) for i in range(2)],
          "reg", float(rec["loss_reg0"]), "raises", raises)


if __name__ == "__main__":
    main()
