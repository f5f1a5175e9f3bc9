#!/usr/bin/env python3
"""Golden vectors for the evaluation loop: the reference's own `test()` with its helpers
`compute_accuracy` / `grading_metrics` (MICCAI-2022/train_test_path_multi_distill.py:409-526), compiled from the file
where it lies (the trainer module as a whole imports loaders and plotting packages that are absent here) and run over a
three-batch synthetic loader on the reference's own networks in eval mode.  Build container only.
Writes tests/golden/eval_loop_b6_h64.npz."""
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


class Loader(list):
    """The two things test() asks of a DataLoader: iteration / len() over batches, and len(loader.dataset)."""
    dataset = None


def batches(nb=3, B=6, H=64):
    from oracle.step import synthetic_batch
    out = Loader()
    for i in range(nb):
        bt = synthetic_batch(B, H, seed=500 + i)
        z = torch.zeros(B)
        out.append((bt["x_path"], z, bt["x_omic"], z, z, bt["grade"]))
    out.dataset = range(nb * B)
    return out


def main():
    from make_golden import install_shims, npz
    install_shims()
    sys.path.insert(0, REF)
    os.chdir(REF)
    sys.argv = ["x", "--model_name", "golden", "--reg_type", "none", "--input_size_omic", "320", "--dropout_rate", "0.25",
                "--gpu_ids", "-1", "--checkpoints_dir", tempfile.mkdtemp(), "--cut_fuse_grad"]
    with contextlib.redirect_stdout(io.StringIO()):
        import options
        opt = options.parse_args()
        import networks_new as NN
    from sklearn.preprocessing import LabelBinarizer
    from sklearn.metrics import roc_auc_score, average_precision_score, f1_score
    src = open(os.path.join(REF, "train_test_path_multi_distill.py")).read()
    ns = {"torch": torch, "np": np, "F": F, "LabelBinarizer": LabelBinarizer, "roc_auc_score": roc_auc_score,
          "average_precision_score": average_precision_score, "f1_score": f1_score, "define_reg": NN.define_reg}
    s0 = src.index("def test(opt, fix_model, model, test_loader, device):"); s1 = src.index("def test_model(")
    exec(compile(src[s0:s1], "test<reference>", "exec"), ns)          # test, compute_accuracy, grading_metrics
    from oracle import weights as W
    with contextlib.redirect_stdout(io.StringIO()):
        student = NN.define_net(opt, 1, path_only=True)
        teacher = NN.define_net(opt, 1)
    student.load_state_dict(W.make_state_dict(W.student_shapes(), 1))
    teacher.load_state_dict(W.make_state_dict(W.teacher_shapes(320), 3))
    loader = batches()
    with contextlib.redirect_stdout(io.StringIO()):
        loss_test, cindex, pvalue, surv_acc, grad_path_test, metrics, pred_test, grads_test, feats_test = ns["test"](
            opt, teacher, student, loader, torch.device("cpu"))
    assert cindex is None and pvalue is None and surv_acc is None
    rec = dict(loss_test=loss_test, grad_path_test=grad_path_test, metrics=np.asarray(metrics, dtype=np.float64),
               probs_all=pred_test[5], probs_path=pred_test[6], gt_all=pred_test[8], feat_path_all=feats_test[1],
               nb=len(loader), B=6, H=64)
    np.savez_compressed(os.path.join(HERE, "eval_loop_b6_h64.npz"), **npz(rec))
    print("wrote eval_loop_b6_h64.npz loss", loss_test, "acc", grad_path_test, "metrics", metrics)


if __name__ == "__main__":
    main()
