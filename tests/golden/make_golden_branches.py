#!/usr/bin/env python3
"""Golden vectors for the NON-DEFAULT option branches of the MICCAI-2022 stage-2 batch body
(/root/reference/MICCAI-2022/train_test_path_multi_distill.py:262-313): `--num_teachers 1` with `--which_teacher fuse` /
`self_EMA`, `--distill kd`, `--assign_weights False`, produced by importing and RUNNING the reference's own modules
(networks_new.define_net / define_optimizer / define_scheduler, KD_loss.DistillKL, CL_utils.CRD_loss.CRDLoss, the trainer's
AEKD_loss / update_ema_variables).  The loop below makes the same calls, in the same order and under the same `if` / `elif`
conditions, as the trainer's batch body.  Two steps per branch from a mid-training optimiser state (tests/golden/_warm.py), so
that both steps are comparable at the north-star tolerance.  Also: the learning rates of every `--lr_policy` the reference's
define_scheduler builds (networks_new.py:111-129) over 12 epochs.  Build container only.
Writes tests/golden/branches_b8_h64.npz and tests/golden/lr_policies.npz."""
import contextlib
import io
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
REF = "/root/reference/MICCAI-2022"

# (name, num_teachers, which_teacher, distill, assign_weights)
BRANCHES = [("t1_fuse_crd", 1, "fuse", "crd", "False"), ("t1_ema_crd", 1, "self_EMA", "crd", "False"),
            ("t1_fuse_kd", 1, "fuse", "kd", "False"), ("t1_ema_kd", 1, "self_EMA", "kd", "False"),
            ("t2_kd_gk", 2, "fuse", "kd", "True"), ("t2_kd_sum", 2, "fuse", "kd", "False"),
            ("t2_crd_sum", 2, "fuse", "crd", "False")]


def main():
    from make_golden import install_shims, npz, ref_opt
    install_shims()
    sys.path.insert(0, REF)
    os.chdir(REF)
    tmp = tempfile.mkdtemp()
    opt = ref_opt(tmp)
    with contextlib.redirect_stdout(io.StringIO()):
        import networks_new as NN
        from CL_utils.CRD_loss import CRDLoss
        from KD_loss import DistillKL
        import train_test_path_multi_distill as TT
    from oracle import weights as W
    from oracle.step import synthetic_batch
    from oracle.losses import CRDState
    import _warm
    B, H, n_data = 8, 64, 1024

    def run(br, rec, warm=None, collect=False, dt=torch.float32):
        name, opt.num_teachers, opt.which_teacher, opt.distill, opt.assign_weights = br
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
                c = CRDLoss(opt, n_data)
            c.embed_s.load_state_dict(W.make_state_dict(W.embed_shapes(), 10 + 2 * i))
            c.embed_t.load_state_dict(W.make_state_dict(W.embed_shapes(), 11 + 2 * i))
            st = CRDState(n_data, seed=20 + i)
            c.contrast.memory_v1.copy_(st.memory_v1); c.contrast.memory_v2.copy_(st.memory_v2)
            crds.append(c)
        criterion_kd, criterion_kd_path = crds
        for mod in (student, ema, teacher, crds[0], crds[1]):      # (before the optimiser: its state takes the parameters' type)
            mod.to(dt)
        ml = torch.nn.ModuleList([student, crds[0].embed_s, crds[0].embed_t, crds[1].embed_s, crds[1].embed_t])   # :192-208
        optimizer = NN.define_optimizer(opt, ml)
        wnames, wparams = _warm.param_names(student), list(ml.parameters())
        if warm is not None:
            _warm.set_torch_adam(optimizer, wnames, wparams, warm)
        NN.define_scheduler(opt, optimizer)     # :212 - the trainer's LambdaLR applies its epoch-0 factor on creation
        rec["lr"] = optimizer.param_groups[0]["lr"]
        criterion_div = DistillKL(opt.kd_T)
        ml.train(); teacher.train()
        ranks_all = []
        _choice = np.random.choice

        def rec_choice(*a, **k):
            r = _choice(*a, **k); ranks_all.append(np.asarray(r)); return r
        np.random.choice = rec_choice
        np.random.seed(2019)
        iter_num = _warm.T0 if warm is not None else 0
        try:
            for it in range(2):
                epoch = 3 + it
                bt = synthetic_batch(B, H, n_data=n_data, seed=500 + it)
                bt = {k: (v.to(dt) if v.dtype.is_floating_point else v) for k, v in bt.items()}
                index, sample_idx, grade = bt["index"], bt["sample_idx"], bt["grade"]
                _, path_feat, logit_path, pred_path, _ = student(x_path=bt["x_path"])                       # :249
                with torch.no_grad():
                    _, ema_path_feat, ema_logit_path, _, _ = ema(x_path=bt["ema_x_path"])                  # :254
                    fuse_feat, _, _, _, logits, pred, _, _, _, _, _ = teacher(x_path=bt["x_path"], x_omic=bt["x_omic"])
                loss_cls = torch.nn.functional.nll_loss(pred_path, grade)                                   # :262
                # ---- :263-271
                if opt.num_teachers == 2:
                    loss_div1 = criterion_div(logit_path, logits[-1].detach())
                    loss_div2 = criterion_div(logit_path, ema_logit_path.detach())
                    loss_div = loss_div1 + loss_div2
                elif opt.num_teachers == 1 and opt.which_teacher == "fuse":
                    loss_div = criterion_div(logit_path, logits[-1].detach())
                elif opt.num_teachers == 1 and opt.which_teacher == "self_EMA":
                    loss_div = criterion_div(logit_path, ema_logit_path.detach())
                # ---- :274-288
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
                # ---- :293-301
                if opt.num_teachers == 2:
                    loss_div1 = opt.alpha * loss_div1
                    loss_div2 = opt.alpha * loss_div2
                    if opt.distill == "crd":
                        loss_kd1 = opt.beta * loss_kd1
                        loss_kd2 = opt.beta * loss_kd2
                        KD_loss_list = [loss_div1, loss_div2, loss_kd1, loss_kd2]
                    elif opt.distill == "kd":
                        KD_loss_list = [loss_div1, loss_div2]
                # ---- :303-309
                scale = None
                if opt.assign_weights == "True":
                    scale, loss_KD = TT.AEKD_loss(opt, optimizer, loss_cls, path_feat, KD_loss_list)
                else:
                    loss_KD = opt.alpha * loss_div + opt.beta * loss_kd
                loss_reg = NN.define_reg(opt, student)
                loss = opt.lambda_nll * loss_cls + opt.lambda_reg * loss_reg + loss_KD                      # :313
                optimizer.zero_grad()
                loss.backward()
                if collect:
                    return _warm.grad_scales(wnames, wparams, opt.weight_decay)
                g = {n: (None if p.grad is None else p.grad.clone()) for n, p in zip(wnames, wparams)}
                optimizer.step()
                TT.update_ema_variables(student, ema, opt.ema_decay, iter_num)
                iter_num += 1
                sd = student.state_dict(); esd = ema.state_dict()
                pre = f"{name}."
                rec.update({pre + f"logit_path{it}": logit_path, pre + f"loss_cls{it}": loss_cls, pre + f"loss_div{it}": loss_div,
                            pre + f"loss_kd{it}": torch.as_tensor(float(loss_kd)), pre + f"loss_KD{it}": loss_KD,
                            pre + f"loss{it}": loss, pre + f"p_fc2_{it}": sd["fc_new2.weight"].clone(),
                            pre + f"ema_fc2_{it}": esd["fc_new2.weight"].clone(),
                            pre + f"p_abs_sum{it}": sum(v.double().abs().sum() for k, v in sd.items() if v.dtype.is_floating_point),
                            pre + f"g_fc2_{it}": g["student.fc_new2.weight"],
                            pre + f"g_l4_1_conv2_abs{it}": g["student.layer4.1.conv2.weight"].abs().sum(),
                            pre + f"embed_s0_{it}": crds[0].embed_s.linear.weight.detach()[:8].clone(),      # (rows 0-7)
                            pre + f"embed_t1_{it}": crds[1].embed_t.linear.weight.detach()[:8].clone(),
                            pre + f"bank0_v1_rows{it}": crds[0].contrast.memory_v1[index].clone(),
                            pre + f"bank1_v2_rows{it}": crds[1].contrast.memory_v2[index].clone(),
                            pre + f"params0_{it}": crds[0].contrast.params.clone(),
                            pre + f"params1_{it}": crds[1].contrast.params.clone()})
                # which parameters the reference's optimiser did NOT touch (grad None -> torch.optim.Adam skips them)
                rec[pre + f"no_grad{it}"] = np.array([n for n in wnames if g[n] is None])
                if scale is not None:
                    rec[pre + f"scale{it}"] = scale
                print(name, "step", it, "loss", float(loss), "loss_KD", float(loss_KD),
                      "scale", None if scale is None else scale.tolist())
        finally:
            np.random.choice = _choice
        rec[name + ".ranks"] = np.stack(ranks_all) if ranks_all else np.zeros((0, 20), dtype=np.int64)

    rec = dict(B=B, H=H, n_data=n_data, t0=_warm.T0, names=np.array([b[0] for b in BRANCHES]))
    # one set of moment scales for every branch (from the default branch's first backward): the state recipe, not a result
    scales = run(("scales", 2, "fuse", "crd", "True"), {}, collect=True)
    rec.update(_warm.pack_scales(scales))
    for br in BRANCHES:
        run(br, rec, warm=scales)
    # the reference's batch body with --num_teachers 1 and --assign_weights True reaches AEKD_loss with KD_loss_list unbound
    try:
        run(("t1_fuse_crd_gk", 1, "fuse", "crd", "True"), {}, warm=scales)
        rec["t1_gk_error"] = np.array("none")
    except Exception as exc:      # UnboundLocalError here (a NameError inside train(), where the name is function-local)
        rec["t1_gk_error"] = np.array(type(exc).__name__)
        print("num_teachers 1 + assign_weights True ->", type(exc).__name__, exc)
    np.savez_compressed(os.path.join(HERE, "branches_b8_h64.npz"), **npz(rec))
    # the same calls in double precision: the reference's own fp32 distance to the truth after one update (the rows of step 1
    # are judged against it where a run that is closer to the truth than the reference's fp32 one cannot match the golden)
    rec64 = {}
    for br in BRANCHES:
        run(br, rec64, warm=scales, dt=torch.float64)
    keep = ("logit_path", "loss", "scale", "g_fc2", "p_fc2")
    rec64 = {k: v for k, v in rec64.items() if k.split(".", 1)[-1].startswith(keep)}
    np.savez_compressed(os.path.join(HERE, "branches_b8_h64_fp64.npz"), **npz(rec64))

    # ---- define_scheduler (networks_new.py:111-129): learning rate per epoch under every policy it builds
    lrs = {}
    for policy in ("linear", "exp", "step", "plateau", "cosine", "onecycle"):
        opt.lr_policy = policy
        opt.niter, opt.niter_decay, opt.lr_decay_iters = 4, 8, 3
        lin = torch.nn.Linear(2, 2)
        o = torch.optim.Adam(lin.parameters(), lr=opt.lr, betas=(opt.beta1, opt.beta2), weight_decay=opt.weight_decay)
        sch = NN.define_scheduler(opt, o)
        seq, b1 = [], []
        for ep in range(12):
            seq.append(o.param_groups[0]["lr"]); b1.append(o.param_groups[0]["betas"][0])
            o.step()
            if policy == "plateau":
                sch.step(1.0 if ep < 3 else 2.0)      # (a metric that stops improving)
            else:
                sch.step()
        lrs[policy + ".lr"] = np.array(seq); lrs[policy + ".beta1"] = np.array(b1)
        print(policy, ["%.3e" % v for v in seq])
    lrs.update(lr=opt.lr, niter=4, niter_decay=8, lr_decay_iters=3, epoch_count=opt.epoch_count, beta1=opt.beta1, beta2=opt.beta2)
    np.savez_compressed(os.path.join(HERE, "lr_policies.npz"), **lrs)
    print("written branches_b8_h64.npz, lr_policies.npz")


if __name__ == "__main__":
    main()
