#!/usr/bin/env python3
"""Golden vectors for the stage-1 mean-teacher batch body, produced by running the reference's own
modules (MICCAI-2022: networks_new.define_net / define_optimizer, CL_utils.KD_losses.pred_KD_loss,
train_test_MT.update_ema_variables) in the order of train_test_MT.py:121-230 for two steps (grading task, dropout 0,
pred_distill on, CRD/SP/orth off, num_teachers 2).  Build container only.  Writes tests/golden/stage1_b4_h64.npz."""
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
    from make_golden import install_shims, ref_opt, npz
    install_shims()
    sys.path.insert(0, REF)
    os.chdir(REF)
    from oracle import weights as W
    from oracle.step import synthetic_batch
    opt = ref_opt(tempfile.mkdtemp())
    opt.cut_fuse_grad = False          # the stage-1 trainer lets the fused loss train both encoders
    opt.num_teachers = 2
    with contextlib.redirect_stdout(io.StringIO()):
        import networks_new as NN
        from CL_utils.KD_losses import pred_KD_loss
        model = NN.define_net(opt, 1)
        ema = NN.define_net(opt, 1)
    model.load_state_dict(W.make_state_dict(W.teacher_shapes(320), 3))
    ema.load_state_dict(W.make_state_dict(W.teacher_shapes(320), 3))
    for p in ema.parameters():
        p.detach_()
    optimizer = NN.define_optimizer(opt, model)
    model.train(); ema.train()

    def update_ema_variables(model, ema_model, alpha, global_step):     # train_test_MT.py:34-38 (same as stage 2)
        alpha = min(1 - 1 / (global_step + 1), alpha)
        for ema_param, param in zip(ema_model.parameters(), model.parameters()):
            ema_param.data.mul_(alpha).add_(param.data, alpha=1 - alpha)

    rec = dict(B=4, H=64, weight_seed=3, lr=opt.lr, weight_decay=opt.weight_decay, ema_decay=opt.ema_decay)
    iter_num = 0
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
        loss = opt.lambda_nll * loss_nll + loss_kd
        optimizer.zero_grad()
        loss.backward()
        optimizer.step()
        update_ema_variables(model, ema, opt.ema_decay, iter_num)
        iter_num += 1
        rec[f"loss{it}"] = loss; rec[f"loss_nll{it}"] = loss_nll; rec[f"loss_kd{it}"] = loss_kd
        rec[f"pred{it}"] = pred; rec[f"pred_path{it}"] = pred_path; rec[f"pred_omic{it}"] = pred_omic
        sd, esd = model.state_dict(), ema.state_dict()
        for k in ("omic_net.encoder.0.0.weight", "omic_net.classifier.0.weight", "fusion.linear_h1.0.weight",
                  "fusion.linear_o2.0.bias", "fusion.encoder2.0.weight", "classifier.0.weight", "path_net.conv1.weight",
                  "path_net.layer3.0.downsample.1.weight"):
            rec[f"w{it}_{k}"] = sd[k].clone()
            rec[f"e{it}_{k}"] = esd[k].clone()
    rec["KD_weight"] = opt.KD_weight
    np.savez_compressed(os.path.join(HERE, "stage1_b4_h64.npz"), **npz(rec))
    print("wrote stage1_b4_h64.npz", [round(float(rec[f"loss{i}"]), 5) for i in range(2)])


if __name__ == "__main__":
    main()
