#!/usr/bin/env python3
"""Golden vectors for the backward of the multi-modal teacher (stage-1 training needs the gradients of
MaxNet / BilinearFusion / PathomicNet), produced by running the reference's own PathomicNet
(MICCAI-2022/networks_new.py, fusion.py) forward + backward on CPU: the three-branch NLL of train_test_MT.py:208-212
(`loss_nll_path + loss_nll_omic + loss_nll_fuse`), dropout 0, both with and without --cut_fuse_grad.
Build container only.  Writes tests/golden/teacher_bwd_b4_h64.npz."""
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
    rec = dict(B=4, H=64, batch_seed=7, weight_seed=3)
    bt = synthetic_batch(4, 64, seed=7)
    for tag, extra in (("cut", ()), ("nocut", ("--no_cut",))):
        opt = ref_opt(tempfile.mkdtemp())
        if tag == "nocut":
            opt.cut_fuse_grad = False
        with contextlib.redirect_stdout(io.StringIO()):
            import networks_new as NN
            net = NN.define_net(opt, 1)
        net.load_state_dict(W.make_state_dict(W.teacher_shapes(320), 3))
        net.train()
        x_omic = bt["x_omic"].clone().requires_grad_(True)
        out = net(x_path=bt["x_path"], x_omic=x_omic)
        features, path_vec, omic_vec, f3, logits, pred, pred_path, pred_omic = out[:8]
        grade = bt["grade"]
        loss = F.nll_loss(pred_path, grade) + F.nll_loss(pred_omic, grade) + F.nll_loss(pred, grade)
        loss.backward()
        rec[f"{tag}_loss"] = loss
        rec[f"{tag}_pred"] = pred
        rec[f"{tag}_dx_omic"] = x_omic.grad
        for k, p in net.named_parameters():
            if p.grad is not None and (k.startswith("omic_net.") or k.startswith("fusion.") or k.startswith("classifier.")
                                       or k in ("path_net.conv1.weight", "path_net.layer4.1.conv2.weight",
                                                "path_net.layer1.0.bn1.weight") or k.startswith("path_net.fc")
                                       or k.startswith("path_net.classifier")):
                g = p.grad.reshape(-1)
                if g.numel() > 65536:   # big tensors: a strided sample of 4096 elements + the l2 norm (small fixtures)
                    stride = g.numel() // 4096
                    rec[f"{tag}_gs_{k}"] = g[::stride][:4096].clone()
                    rec[f"{tag}_gn_{k}"] = g.double().norm()
                else:
                    rec[f"{tag}_g_{k}"] = p.grad
    np.savez_compressed(os.path.join(HERE, "teacher_bwd_b4_h64.npz"), **npz(rec))
    print("wrote teacher_bwd_b4_h64.npz with", len(rec), "entries")


if __name__ == "__main__":
    main()
