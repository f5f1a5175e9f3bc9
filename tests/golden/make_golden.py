#!/usr/bin/env python3
"""Generate golden vectors by importing and RUNNING the reference on CPU.

Runs only in the build container (needs /root/reference); the GPU box gets the committed
``tests/golden/*.npz``.  Nothing from the reference is copied: the reference modules are imported
from where they lie, fed weights from ``oracle.weights.make_state_dict`` (a seed recipe, so no
45 MB weight blobs), executed, and only inputs-recipes + outputs are saved.

Shims: stub modules for packages absent here (lifelines, imblearn, seaborn,
pylab, torch_geometric), CUDA->CPU identity for ``.cuda()`` / ``torch.cuda.FloatTensor``, and a
``torch.load`` wrapper returning {} for the hard-coded ImageNet checkpoint path (resnets.py:281,
loaded strict=False).

Usage:  python tests/golden/make_golden.py            # writes tests/golden/*.npz
"""
import os
import sys
import types
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/MICCAI-2022"
sys.path.insert(0, ROOT)


def install_shims():
    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Any:
        def __init__(self, *a, **k): pass
        def __call__(self, *a, **k): return self
        def __getattr__(self, k): return _Any()

    stub("lifelines", KaplanMeierFitter=_Any, CoxPHFitter=_Any)
    stub("lifelines.utils", concordance_index=_Any(), k_fold_cross_validation=_Any())
    stub("lifelines.datasets", load_regression_dataset=_Any())
    stub("lifelines.statistics", logrank_test=_Any())
    stub("imblearn"); stub("imblearn.over_sampling", RandomOverSampler=_Any)
    stub("seaborn"); stub("pylab")
    tg = stub("torch_geometric"); tgd = stub("torch_geometric.data", Batch=_Any, Data=_Any, DataLoader=_Any)
    stub("torch_geometric.data.data", Data=_Any)
    tg.data = tgd
    import scipy
    scipy.interp = np.interp
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    torch.cuda.FloatTensor = torch.FloatTensor
    torch.cuda.set_device = lambda *a, **k: None
    _load = torch.load

    def load(f, *a, **k):
        if isinstance(f, str) and "pretrained_resnet" in f:
            return {}
        return _load(f, *a, **k)
    torch.load = load


def ref_opt(tmp, extra=()):
    sys.argv = ["x", "--distill", "crd", "-a", "1", "-b", "0.02", "--nce_p2", "20", "--num_teachers", "2",
                "--CE_grads", "--model_name", "golden", "--fixed_model", "t", "--reg_type", "none",
                "--beta1", "0.9", "--select_pos_mode", "mid", "--assign_weights", "True",
                "--cut_fuse_grad", "--input_size_omic", "320", "--dropout_rate", "0", "--gpu_ids", "-1",
                "--checkpoints_dir", tmp] + list(extra)
    import options
    import io, contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        return options.parse_args()


def npz(d):
    out = {}
    for k, v in d.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    return out


def main():
    install_shims()
    sys.path.insert(0, REF)
    os.chdir(REF)   # options/print path is relative-safe; nothing is written under REF (tmp ckpt dir)
    tmp = tempfile.mkdtemp()
    import io, contextlib
    opt = ref_opt(tmp)
    with contextlib.redirect_stdout(io.StringIO()):
        import networks_new as NN
        from CL_utils.CRD_loss import CRDLoss
        from KD_loss import DistillKL
        import train_test_path_multi_distill as TT
    from oracle import weights as W
    from oracle.step import synthetic_batch

    def build(seed):
        with contextlib.redirect_stdout(io.StringIO()):
            student = NN.define_net(opt, 1, path_only=True)
            ema = NN.define_net(opt, 1, path_only=True)
            teacher = NN.define_net(opt, 1)
        student.load_state_dict(W.make_state_dict(W.student_shapes(), seed + 1))
        ema.load_state_dict(W.make_state_dict(W.student_shapes(), seed + 2))
        teacher.load_state_dict(W.make_state_dict(W.teacher_shapes(320), seed + 3))
        for p in ema.parameters():
            p.detach_()
        for p in teacher.parameters():
            p.detach_(); p.requires_grad = False
        return student, ema, teacher

    # key-set / shape contract check: our recipe layout == the reference's state_dict layout
    student, ema, teacher = build(0)
    for mod, shp in ((student, W.student_shapes()), (teacher, W.teacher_shapes(320))):
        sd = mod.state_dict()
        assert list(sd.keys()) == list(shp.keys()), (set(sd) ^ set(shp))
        for k in sd:
            assert tuple(sd[k].shape) == tuple(shp[k]), k

    # ---------------------------------------------------------------- (i) module-level vectors
    B, H = 4, 64
    bt = synthetic_batch(B, H, seed=7)
    student.train(); teacher.train()
    x = bt["x_path"].clone().requires_grad_(True)
    f3, feat, hazard, pred, _ = student(x_path=x)
    lossm = (feat * torch.linspace(0.5, 1.5, 128)).sum() + (hazard * torch.tensor([1.0, -2.0, 0.5])).sum() \
        + 0.1 * f3.sum()
    lossm.backward()
    sdg = {k: p.grad for k, p in student.named_parameters() if p.grad is not None}
    mod = dict(seed=0, B=B, H=H, batch_seed=7, f3=f3, feat=feat, hazard=hazard, pred=pred, dx=x.grad,
               g_conv1=sdg["conv1.weight"], g_bn1_w=sdg["bn1.weight"], g_bn1_b=sdg["bn1.bias"],
               g_l1_0_conv1=sdg["layer1.0.conv1.weight"], g_l2_0_ds=sdg["layer2.0.downsample.0.weight"],
               g_l4_1_bn2_w=sdg["layer4.1.bn2.weight"], g_fc1_w=sdg["fc_new1.0.weight"],
               g_fc2_w=sdg["fc_new2.weight"], g_fc2_b=sdg["fc_new2.bias"],
               g_l3_1_conv2_sum=sdg["layer3.1.conv2.weight"].sum(), g_l3_1_conv2_abs=sdg["layer3.1.conv2.weight"].abs().sum(),
               rm_bn1=student.state_dict()["bn1.running_mean"], rv_bn1=student.state_dict()["bn1.running_var"],
               rm_l4=student.state_dict()["layer4.1.bn2.running_mean"], rv_l4=student.state_dict()["layer4.1.bn2.running_var"])
    with torch.no_grad():
        t = teacher(x_path=bt["x_path"], x_omic=bt["x_omic"])
        mod.update(t_fuse=t[0], t_path_vec=t[1], t_omic_vec=t[2], t_f3=t[3], t_h_path=t[4][0],
                   t_h_omic=t[4][1], t_h_fuse=t[4][2], t_pred=t[5], t_pred_path=t[6], t_pred_omic=t[7])
        om = teacher.omic_net(x_omic=bt["x_omic"])
        mod.update(omic_feat=om[0], omic_out=om[1], omic_pred=om[2])
        v1 = torch.randn(B, 128, generator=torch.Generator().manual_seed(5))
        v2 = torch.randn(B, 128, generator=torch.Generator().manual_seed(6))
        mod.update(fus_in1=v1, fus_in2=v2, fus_out=teacher.fusion(v1, v2))
    ys = torch.randn(B, 3, generator=torch.Generator().manual_seed(8)).requires_grad_(True)
    yt = torch.randn(B, 3, generator=torch.Generator().manual_seed(9))
    for T in (1.0, 4.0):
        kl = DistillKL(T)(ys, yt)
        g, = torch.autograd.grad(kl, ys)
        mod.update({f"kl_ys": ys, "kl_yt": yt, f"kl_T{int(T)}": kl, f"kl_T{int(T)}_g": g})
    np.savez_compressed(os.path.join(HERE, "modules_b4_h64.npz"), **npz(mod))

    # ---------------------------------------------------------------- (ii) CRD loss vectors
    n_data = 1024
    for mode in ("mid", "hard"):
        opt.select_pos_mode = mode
        torch.manual_seed(20)
        with contextlib.redirect_stdout(io.StringIO()):
            crd = CRDLoss(opt, n_data)
        crd.embed_s.load_state_dict(W.make_state_dict(W.embed_shapes(), 10))
        crd.embed_t.load_state_dict(W.make_state_dict(W.embed_shapes(), 11))
        g = torch.Generator().manual_seed(31)
        Bc = 8
        from oracle.losses import CRDState
        st0 = CRDState(n_data, seed=20)           # bank recipe shared with the oracle (no bank blobs)
        crd.contrast.memory_v1.copy_(st0.memory_v1); crd.contrast.memory_v2.copy_(st0.memory_v2)
        rec = dict(mode=mode, bank_seed=20)
        ranks_all = []
        _choice = np.random.choice

        def rec_choice(*a, **k):
            r = _choice(*a, **k); ranks_all.append(np.asarray(r)); return r
        np.random.choice = rec_choice
        np.random.seed(2019)
        for it in range(2):   # two calls: Z is set on the first, frozen on the second
            f_s = torch.randn(Bc, 128, generator=g).relu_().requires_grad_(True)
            f_t = torch.randn(Bc, 128, generator=g).relu_()
            index = torch.randperm(n_data, generator=g)[:Bc]
            sidx = torch.randint(0, n_data, (Bc, 1000), generator=g); sidx[:, 0] = index
            with contextlib.redirect_stdout(io.StringIO()):
                loss = crd(0.1, f_s, f_t, index, sidx)
            gs = torch.autograd.grad(loss, [f_s, crd.embed_s.linear.weight, crd.embed_t.linear.weight,
                                            crd.embed_s.linear.bias])
            rec.update({f"f_s{it}": f_s, f"f_t{it}": f_t, f"index{it}": index, f"sidx{it}": sidx,
                        f"loss{it}": loss, f"g_fs{it}": gs[0], f"g_ws{it}": gs[1], f"g_wt{it}": gs[2],
                        f"g_bs{it}": gs[3], f"params{it}": crd.contrast.params.clone(),
                        f"bank_v1_rows{it}": crd.contrast.memory_v1[index].clone(),
                        f"bank_v2_rows{it}": crd.contrast.memory_v2[index].clone()})
        np.random.choice = _choice
        rec["ranks"] = np.stack(ranks_all) if ranks_all else np.zeros((0, 20), dtype=np.int64)
        np.savez_compressed(os.path.join(HERE, f"crd_{mode}.npz"), **npz(rec))
    opt.select_pos_mode = "mid"

    # ---------------------------------------------------------------- (iii) full step B=16 / 224
    B, H = 16, 224
    student, ema, teacher = build(0)
    student.train(); teacher.train()
    torch.manual_seed(20)
    crds = []
    for i in range(2):
        torch.manual_seed(20 + i)
        with contextlib.redirect_stdout(io.StringIO()):
            c = CRDLoss(opt, n_data)
        c.embed_s.load_state_dict(W.make_state_dict(W.embed_shapes(), 10 + 2 * i))
        c.embed_t.load_state_dict(W.make_state_dict(W.embed_shapes(), 11 + 2 * i))
        sti = CRDState(n_data, seed=20 + i)
        c.contrast.memory_v1.copy_(sti.memory_v1); c.contrast.memory_v2.copy_(sti.memory_v2)
        crds.append(c)
    ml = torch.nn.ModuleList([student, crds[0].embed_s, crds[0].embed_t, crds[1].embed_s, crds[1].embed_t])
    optimizer = NN.define_optimizer(opt, ml)
    kl = DistillKL(opt.kd_T)
    step_rec = dict(B=B, H=H, n_data=n_data, seed=0)
    ranks_all = []
    _choice = np.random.choice

    def rec_choice2(*a, **k):
        r = _choice(*a, **k); ranks_all.append(np.asarray(r)); return r
    np.random.choice = rec_choice2
    np.random.seed(2019)
    iter_num = 0
    for it in range(3):
        bt = synthetic_batch(B, H, seed=100 + it)
        # ---- the reference's batch body, train_test_path_multi_distill.py:249-330, verbatim calls
        _, path_feat, logit_path, pred_path, _ = student(x_path=bt["x_path"])
        with torch.no_grad():
            _, ema_path_feat, ema_logit_path, _, _ = ema(x_path=bt["ema_x_path"])
            fuse_feat, _, _, _, logits, pred, _, _, _, _, _ = teacher(x_path=bt["x_path"], x_omic=bt["x_omic"])
        loss_cls = torch.nn.functional.nll_loss(pred_path, bt["grade"])
        loss_div1 = kl(logit_path, logits[-1].detach())
        loss_div2 = kl(logit_path, ema_logit_path.detach())
        with contextlib.redirect_stdout(io.StringIO()):
            loss_kd1 = crds[0](it / opt.niter_decay, path_feat, fuse_feat.detach(), bt["index"], bt["sample_idx"])
            loss_kd2 = crds[1](it / opt.niter_decay, path_feat, ema_path_feat.detach(), bt["index"], bt["sample_idx"])
        kd_list = [opt.alpha * loss_div1, opt.alpha * loss_div2, opt.beta * loss_kd1, opt.beta * loss_kd2]
        scale, loss_KD = TT.AEKD_loss(opt, optimizer, loss_cls, path_feat, kd_list)
        loss = opt.lambda_nll * loss_cls + loss_KD
        optimizer.zero_grad()
        loss.backward()
        if it == 0:
            step_rec.update(g0_conv1=student.conv1.weight.grad.clone(),
                            g0_fc2_w=student.fc_new2.weight.grad.clone(),
                            g0_l4_1_conv2_abs=student.layer4[1].conv2.weight.grad.abs().sum(),
                            g0_embed_s0=crds[0].embed_s.linear.weight.grad.clone(),
                            g0_embed_t1=crds[1].embed_t.linear.weight.grad.clone())
        optimizer.step()
        TT.update_ema_variables(student, ema, opt.ema_decay, iter_num)
        iter_num += 1
        sd = student.state_dict(); esd = ema.state_dict()
        step_rec.update({f"logit_path{it}": logit_path, f"path_feat{it}": path_feat,
                         f"ema_logit{it}": ema_logit_path, f"fuse_logit{it}": logits[-1],
                         f"fuse_feat{it}": fuse_feat,
                         f"loss_cls{it}": loss_cls, f"loss_div1_{it}": loss_div1, f"loss_div2_{it}": loss_div2,
                         f"loss_kd1_{it}": loss_kd1, f"loss_kd2_{it}": loss_kd2, f"scale{it}": scale,
                         f"loss_KD{it}": loss_KD, f"loss{it}": loss,
                         f"p_conv1_{it}": sd["conv1.weight"].clone(), f"p_fc2_{it}": sd["fc_new2.weight"].clone(),
                         f"p_abs_sum{it}": sum(v.double().abs().sum() for k, v in sd.items() if v.dtype.is_floating_point),
                         f"ema_abs_sum{it}": sum(v.double().abs().sum() for k, v in esd.items() if v.dtype.is_floating_point),
                         f"ema_fc2_{it}": esd["fc_new2.weight"].clone(),
                         f"bank0_v1_rows{it}": crds[0].contrast.memory_v1[bt["index"]].clone(),
                         f"bank1_v2_rows{it}": crds[1].contrast.memory_v2[bt["index"]].clone(),
                         f"params0_{it}": crds[0].contrast.params.clone(),
                         f"params1_{it}": crds[1].contrast.params.clone()})
        print("step", it, "loss", float(loss), "scale", scale.tolist())
    np.random.choice = _choice
    step_rec["ranks"] = np.stack(ranks_all)
    np.savez_compressed(os.path.join(HERE, "step_b16_h224.npz"), **npz(step_rec))
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
