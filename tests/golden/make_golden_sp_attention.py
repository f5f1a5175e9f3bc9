#!/usr/bin/env python3
"""Golden vectors for `superpixel_attention_mask` end to end: the reference's own function
("MIA 2023/stage1_multi_modal_teacher/train_test_MT_SP_Masking.py":42-102), compiled from the file where it lies, run on
the reference's PathomicNet (MICCAI-2022/networks_new.py, seeded weights, no --cut_fuse_grad so that the fused prediction
depends on both inputs) with a synthetic batch and label map.  Also records the two input gradients, computed by the
same statements, for a toleranced check.  Build container only.  Writes tests/golden/sp_attention_b4_h64.npz."""
import contextlib
import io
import os
import sys
import tempfile
import types

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
REF = "/root/reference/MICCAI-2022"
SRC = "/root/reference/MIA 2023/stage1_multi_modal_teacher/train_test_MT_SP_Masking.py"


def main():
    from make_golden import install_shims, npz
    from make_golden_superpixel import label_map
    install_shims()
    sys.path.insert(0, REF)
    os.chdir(REF)
    sys.argv = ["x", "--model_name", "golden", "--reg_type", "none", "--input_size_omic", "320", "--dropout_rate", "0.25",
                "--gpu_ids", "-1", "--checkpoints_dir", tempfile.mkdtemp()]
    with contextlib.redirect_stdout(io.StringIO()):
        import options
        opt = options.parse_args()
        import networks_new as NN
    assert not opt.cut_fuse_grad
    src = open(SRC).read()
    s0 = src.index("def superpixel_attention_mask("); s1 = src.index("def train(", s0)
    ns = {"torch": torch, "F": F, "Variable": torch.autograd.Variable}
    exec(compile(src[s0:s1], "superpixel_attention_mask<reference>", "exec"), ns)
    from oracle import weights as W
    from oracle.step import synthetic_batch
    with contextlib.redirect_stdout(io.StringIO()):
        model = NN.define_net(opt, 1)
    model.load_state_dict(W.make_state_dict(W.teacher_shapes(320), 3))
    B, H = 4, 64
    bt = synthetic_batch(B, H, seed=620)
    gen = torch.Generator().manual_seed(31)
    sp_mask = label_map(B, H, H, 6, gen)
    o = types.SimpleNamespace(Path_K=5, Omic_K=8)
    model.train()
    pm, om = ns["superpixel_attention_mask"](o, None, model, bt["x_path"], torch.zeros(B), bt["x_omic"], sp_mask, bt["grade"],
                                             torch.device("cpu"))
    assert model.training
    # the input gradients, by the same statements (:62-75)
    model.eval()
    xp = bt["x_path"].clone().requires_grad_(True); xo = bt["x_omic"].clone().requires_grad_(True)
    pred = model(x_path=xp, x_grph=torch.zeros(B), x_omic=xo)[5]
    cost = F.nll_loss(pred, bt["grade"])
    gp, go = torch.autograd.grad(cost, [xp, xo])
    rec = dict(B=B, H=H, Path_K=5, Omic_K=8, sp_mask=sp_mask, path_mask=pm, omic_mask=om, x_path_grad=gp, x_omic_grad=go,
               cost=cost.detach(), dropout_rate=0.25)
    np.savez_compressed(os.path.join(HERE, "sp_attention_b4_h64.npz"), **npz(rec))
    print("wrote sp_attention_b4_h64.npz: mask pixels", int(pm.sum()), "omic", int(om.sum()), "cost", float(cost))


if __name__ == "__main__":
    main()
