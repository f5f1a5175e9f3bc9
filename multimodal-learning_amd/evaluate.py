"""Evaluation loop of the stage-2 trainer, reference MICCAI-2022/train_test_path_multi_distill.py:
`test()` (:409-501), `grading_metrics` (:516-526).

Both networks run their eval-mode forward through the C-ABI (running-statistics BatchNorm, dropout off, no tape).  The
reference copies logits, features and a loss scalar to the host after EVERY batch (three `.cpu()` / `.item()` syncs per
batch); here the per-batch results stay on the device and leave it once, after the last batch.  The ranking metrics
(ROC-AUC, average precision, F1) are host-side sklearn calls on ~N x 3 numbers, exactly as in the reference."""
import numpy as np
import torch

from . import ops
from .networks_new import define_reg


def grading_metrics(y_label, y_pred, avg="micro"):
    """The reference's four ranking numbers of the grading task (train_test_path_multi_distill.py:516-526): ROC-AUC and
    average precision of the one-hot labels against the predicted scores, the F1 of the arg-max grades, and the F1 of the
    third class (grade IV) alone.  (Its per-batch `compute_accuracy` helper, :504-513, has no counterpart: `test()` below keeps
    the per-batch predictions on the device and counts once.)"""
    from sklearn import metrics as skm
    true_grade, pred_grade = np.argmax(y_label, axis=1), np.argmax(y_pred, axis=1)
    per_class_f1 = skm.f1_score(true_grade, pred_grade, average=None)
    return (skm.roc_auc_score(y_label, y_pred, average=avg), skm.average_precision_score(y_label, y_pred, average=avg),
            skm.f1_score(true_grade, pred_grade, average=avg), per_class_f1[2])


def test_model(opt, model, test_loader, device):
    """:530-611 - `test()` for the student alone (no frozen teacher): same 9-tuple, `probs_all` is None."""
    return test(opt, None, model, test_loader, device)


def test(opt, fix_model, model, test_loader, device):
    """:409-501 for the grading task.  Returns the reference's 9-tuple
    (loss_test, cindex_path, pvalue_test, surv_acc_test, grad_path_test, all_grad_metrics, pred_test, grads_test,
    feats_test); the survival entries are None.  `fix_model=None` gives `test_model()` (:530-611)."""
    if opt.task != "grad":
        raise NotImplementedError("evaluation of the survival (Cox) task is out of scope")
    from sklearn.preprocessing import LabelBinarizer
    if fix_model is not None:
        fix_model.eval()
    model.eval()
    dev = torch.device(device)
    preds_path, preds_fuse, feats, grades, losses = [], [], [], [], []
    with torch.no_grad():
        loss_reg = define_reg(opt, model)
        for x_path, x_grph, x_omic, censor, survtime, grade in test_loader:
            x_path = x_path.to(dev, non_blocking=True)
            x_omic = x_omic.to(dev, non_blocking=True)
            grade = grade.to(dev, non_blocking=True)
            _, feat_path, _, pred_path, _ = model(x_path=x_path, x_grph=x_grph, x_omic=x_omic)            # :427
            pred = None
            if fix_model is not None:
                _, _, _, _, _, pred, _, _, _, _, _ = fix_model(x_path=x_path, x_grph=x_grph, x_omic=x_omic)  # :431
            loss_nll = ops.NLLFn.apply(pred_path, grade, float(pred_path.shape[0]))                        # :439
            losses.append((opt.lambda_nll * loss_nll + opt.lambda_reg * loss_reg).reshape(1))              # :441
            preds_path.append(pred_path); preds_fuse.append(pred); feats.append(feat_path); grades.append(grade)
    nb = len(losses)
    loss_test = float(torch.cat(losses).sum().item()) / len(test_loader)                                   # :442, :465
    probs_path = torch.cat(preds_path).cpu().numpy()
    probs_all = torch.cat(preds_fuse).cpu().numpy() if fix_model is not None else None
    feat_path_all = torch.cat(feats).cpu().numpy()
    gt = torch.cat(grades).cpu().numpy().reshape(-1)
    gt_all = gt.astype(np.float64)                                     # np.concatenate onto np.array([]) (:444)
    grad_path_test = float((probs_path.argmax(axis=1) == gt).sum()) / len(test_loader.dataset)             # :458, :472
    enc = LabelBinarizer()
    enc.fit(gt_all)
    grad_gt = enc.transform(gt_all)                                                                        # :476-478
    if fix_model is not None:
        rocauc_fuse, ap_fuse, f1_micro_fuse, f1_gradeIV_fuse = grading_metrics(grad_gt, probs_all)         # :481
        print("fixed fuse branch:", rocauc_fuse, ap_fuse, f1_micro_fuse, f1_gradeIV_fuse)
    rocauc_path, ap_path, f1_micro_path, f1_gradeIV_path = grading_metrics(grad_gt, probs_path)
    print("Path branch:", rocauc_path, ap_path, f1_micro_path, f1_gradeIV_path)
    all_grad_metrics = [rocauc_path, ap_path, f1_micro_path, f1_gradeIV_path]
    e = np.array([])
    pred_test = [e, e, e, e, e, probs_all, probs_path, None, gt_all]                                       # :491-492
    grads_test = [None, None, None]
    feats_test = [None, feat_path_all, None, gt_all]
    del nb
    return loss_test, None, None, None, grad_path_test, all_grad_metrics, pred_test, grads_test, feats_test
