#!/usr/bin/env python3
"""Eval-mode golden vectors (reference test(), train_test_path_multi_distill.py:409-431: fix_model.eval(); model.eval()):
student and teacher forwards with running-statistics BatchNorm, produced by running the reference.  Build container only."""
import contextlib
import io
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)


def main():
    from make_golden import install_shims, ref_opt, npz, REF
    install_shims()
    sys.path.insert(0, REF)
    os.chdir(REF)
    opt = ref_opt(tempfile.mkdtemp())
    with contextlib.redirect_stdout(io.StringIO()):
        import networks_new as NN
    from oracle import weights as W
    from oracle.step import synthetic_batch
    with contextlib.redirect_stdout(io.StringIO()):
        student = NN.define_net(opt, 1, path_only=True)
        teacher = NN.define_net(opt, 1)
    student.load_state_dict(W.make_state_dict(W.student_shapes(), 1))
    teacher.load_state_dict(W.make_state_dict(W.teacher_shapes(320), 3))
    student.eval(); teacher.eval()
    bt = synthetic_batch(4, 96, seed=9)
    with torch.no_grad():
        f3, feat, hazard, pred, _ = student(x_path=bt["x_path"])
        t = teacher(x_path=bt["x_path"], x_omic=bt["x_omic"])
    rec = dict(B=4, H=96, batch_seed=9, f3=f3, feat=feat, hazard=hazard, pred=pred, t_fuse=t[0], t_path_vec=t[1],
               t_omic_vec=t[2], t_h_fuse=t[4][2], t_pred=t[5], t_pred_path=t[6], t_pred_omic=t[7])
    np.savez_compressed(os.path.join(HERE, "modules_eval_b4_h96.npz"), **npz(rec))
    print("written modules_eval_b4_h96.npz")


if __name__ == "__main__":
    main()
