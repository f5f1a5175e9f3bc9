"""The L1 weight regulariser `opt.lambda_reg * define_reg(opt, model)` (reference networks_new.py:93-108,
utils.py:60-198; train_test_MT.py:209-217, train_test_path_multi_distill.py:312-313) against vectors produced by the
reference's own functions (tests/golden/make_golden_stage1_reg.py), and the option checks of the step classes."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _nets(opt):
    import multimodal_learning_amd as m
    from oracle import weights as W
    teacher = m.define_net(opt, 1)
    teacher.load_state_dict(W.make_state_dict(W.teacher_shapes(320), 3))
    student = m.define_net(opt, 1, path_only=True)
    student.load_state_dict(W.make_state_dict(W.student_shapes(), 1))
    return teacher.cuda(), student.cuda()


def test_define_reg_values_and_error_behaviour(golden_dir):
    import multimodal_learning_amd as m
    g = np.load(os.path.join(golden_dir, "stage1_reg_b4_h64.npz"))
    opt = m.stage2_opt(dropout_rate=0.0)
    teacher, student = _nets(opt)
    for rt in ("omic", "mm", "all"):
        opt.reg_type = rt
        v = m.define_reg(opt, teacher)
        ref = float(g["reg_teacher_" + rt])
        assert abs(v.item() - ref) <= 2e-6 * ref, (rt, v.item(), ref)
    opt.reg_type = "all"
    ref = float(g["reg_student_all"])
    assert abs(m.define_reg(opt, student).item() - ref) <= 2e-6 * ref
    opt.reg_type = "none"
    assert m.define_reg(opt, teacher) == 0
    # the reference's helpers raise AttributeError where the attribute walk finds nothing to walk (same cases here)
    raised = []
    for who, net in (("teacher", teacher), ("student", student)):
        for rt in ("path", "mm", "omic"):
            opt.reg_type = rt
            try:
                m.define_reg(opt, net)
            except AttributeError:
                raised.append(who + ":" + rt)
    assert raised == list(g["raises_attribute_error"]), raised
    opt.reg_type = "bogus"
    with pytest.raises(NotImplementedError):
        m.define_reg(opt, teacher)


def test_define_reg_gradient_is_sign_of_weight():
    """Backward of the term: upstream * sgn(W) - checked on free tensors (returned to autograd) and on parameters that
    live in a flat buffer (accumulated in place into the flat gradient)."""
    import multimodal_learning_amd as m
    from multimodal_learning_amd import ops
    from multimodal_learning_amd.train_step import FlatParams
    torch.manual_seed(0)
    a = torch.randn(37, 5, device="cuda").requires_grad_(True)
    b = torch.randn(11, device="cuda").requires_grad_(True)
    with torch.no_grad():
        a[0, 0] = 0.0
    loss = 0.25 * ops.l1_norm_sum([a, b])
    loss.backward()
    assert abs(loss.item() - 0.25 * (a.abs().sum() + b.abs().sum()).item()) < 1e-4
    assert torch.equal(a.grad, 0.25 * torch.sign(a.detach())) and torch.equal(b.grad, 0.25 * torch.sign(b.detach()))
    ps = [torch.nn.Parameter(torch.randn(s, device="cuda")) for s in ((7, 3), (5,), (2, 2, 2))]
    fp = FlatParams(ps, with_grad=True)
    fp.grad.fill_(1.0)
    (3.0 * ops.l1_norm_sum(ps)).backward()
    for p in ps:
        assert torch.equal(p.grad, 1.0 + 3.0 * torch.sign(p.detach())), p.shape


def test_stage1_step_with_default_regulariser_vs_reference_golden(golden_dir):
    """Two stage-1 steps with the option values the shipped stage-1 command runs with (README.md:26-27: no --reg_type,
    so `omic`, lambda_reg 3e-4): step 0 at 1e-3 against the reference (losses, the regulariser's value, predictions,
    the gradient of omic_net tensors - where the L1 term is lambda_reg * sgn(W)); step 1 after Adam's sign-like first
    update at the looser post-update bound of the stage-1 test."""
    import multimodal_learning_amd as m
    from oracle import weights as W
    from oracle.step import synthetic_batch
    g = np.load(os.path.join(golden_dir, "stage1_reg_b4_h64.npz"))
    m.set_precision("bf16x6")
    try:
        opt = m.stage2_opt(dropout_rate=0.0, batch_size=4, cut_fuse_grad=False, num_teachers=2, reg_type="omic",
                           lambda_reg=float(g["lambda_reg"]))
        opt.pred_distill, opt.KD_weight = 1, float(g["KD_weight"])
        opt.lr, opt.weight_decay, opt.ema_decay = float(g["lr"]), float(g["weight_decay"]), float(g["ema_decay"])
        model = m.define_net(opt, 1); ema = m.define_net(opt, 1)
        sd = W.make_state_dict(W.teacher_shapes(320), 3)
        model.load_state_dict(sd); ema.load_state_dict(sd)
        st = m.TeacherStage1Step(opt, device="cuda", models=(model.cuda(), ema.cuda()))
        grads0 = {}
        named = dict(st.model.named_parameters())
        watch = [k[3:] for k in g.files if k.startswith("g0_")]
        # capture the pre-update gradients of step 0: the optimiser's step() is the first reader after backward
        orig_step = st.optimizer.step

        def spy(*a, **k):
            if not grads0:
                for name in watch:
                    grads0[name] = named[name].grad.detach().clone()
            return orig_step(*a, **k)
        st.optimizer.step = spy
        for it in range(2):
            bt = synthetic_batch(4, 64, seed=20 + it)
            z = torch.zeros(4)
            batch = ((bt["x_path"], bt["ema_x_path"]), z, bt["x_omic"], z, z, bt["grade"], bt["index"], bt["sample_idx"])
            out = st.step(batch)
            tol = 1e-3 if it == 0 else 5e-2
            for k in ("loss", "loss_nll"):
                assert abs(out[k].item() - float(g[f"{k}{it}"])) <= tol * abs(float(g[f"{k}{it}"])), (it, k, out[k].item())
            # the regulariser is a smooth function of the weights: lr * sign-like updates move it by < 1e-3 relative
            assert abs(out["loss_reg"].item() - float(g[f"loss_reg{it}"])) <= 1e-3 * float(g[f"loss_reg{it}"]), it
            for k in ("pred", "pred_path", "pred_omic"):
                assert np.abs(out[k].cpu().numpy() - g[f"{k}{it}"]).max() <= tol * 10, (it, k)
        for name in watch:
            ref = g["g0_" + name]
            err = np.abs(grads0[name].cpu().numpy() - ref).max()
            assert err <= 1e-3 * max(np.abs(ref).max(), 1e-3), (name, err)
    finally:
        m.set_precision("bf16")


def test_distill_step_all_regulariser_adds_sign_gradient():
    """DistillStep with --reg_type all: the loss grows by lambda_reg * sum|W| over the student's parameters and the
    gradient that reaches Adam by lambda_reg * sgn(W) (the trunk then accumulates instead of overwriting its flat
    gradient views).  Compared against the same step with --reg_type none on identical inputs, parity mode."""
    import multimodal_learning_amd as m
    from oracle.step import default_opt, synthetic_batch
    from tests.test_gpu_step import _mk_step, _tuple
    m.set_precision("bf16x6")
    try:
        res = {}
        for rt in ("none", "all"):
            opt = default_opt(nce_p=120, nce_k=200, nce_p2=20, nce_k2=128)
            opt.reg_type = rt
            step = _mk_step(opt, 256, seed=0)
            grads = {}
            orig = step.optimizer.step

            def spy(*a, _g=grads, _s=step, _o=orig, **k):
                _g["flat"] = _s.optimizer.flat.grad.detach().clone()
                return _o(*a, **k)
            step.optimizer.step = spy
            bt = synthetic_batch(4, 64, n_data=256, P=opt.nce_p, K=opt.nce_k, seed=0)
            ranks = [np.random.RandomState(5).choice(np.arange(30, 100), opt.nce_p2, replace=False) for _ in range(2)]
            w_before = step.optimizer.flat.flat.detach().clone()
            out = step.step(_tuple(bt), ranks=ranks)
            res[rt] = (out["loss"].item(), grads["flat"], w_before, step)
        (l0, g0, w0, s0), (l1, g1, w1, s1) = res["none"], res["all"]
        assert torch.equal(w0, w1)
        n_student = len(list(s1.model.parameters()))
        flat = s1.optimizer.flat
        end = flat.offsets[n_student] if n_student < len(flat.offsets) else flat.numel
        reg = w0[:end].abs().sum().item()
        assert abs((l1 - l0) - opt.lambda_reg * reg) <= 1e-3 * abs(l1), (l0, l1, opt.lambda_reg * reg)
        # gradient difference: lambda_reg * sgn(W) on the requires-grad student tensors, zero on the embedding heads
        mask = torch.zeros_like(w0, dtype=torch.bool)
        for t, o in zip(flat.tensors[:n_student], flat.offsets[:n_student]):
            if t.requires_grad:
                mask[o:o + t.numel()] = True
        d = g1 - g0
        want = torch.where(mask, opt.lambda_reg * torch.sign(w0), torch.zeros_like(w0))
        scale = g0.abs().max().item()
        assert (d - want)[:end].abs().max().item() <= 1e-5 * max(scale, 1.0) + 1e-7
        assert (d[end:]).abs().max().item() <= 1e-5 * max(scale, 1.0)
        # per tensor, absolute, well below lambda_reg (3e-4): a lost L1 term on ONE small tensor (fc_new2: 3 x 128 + 3
        # elements, whose gradient the fused loss head writes itself) hides under a whole-buffer relative bound
        for name, t in s1.model.named_parameters():
            if not t.requires_grad:
                continue
            o = next(o for tt, o in zip(flat.tensors, flat.offsets) if tt is t)
            err = (d[o:o + t.numel()] - want[o:o + t.numel()]).abs().max().item()
            gmax = g0[o:o + t.numel()].abs().max().item()
            assert err <= 2e-5 + 2e-5 * gmax, (name, err, gmax)
        assert not s1._fused_head_ok() and s0._fused_head_ok()
    finally:
        m.set_precision("bf16")


def test_steps_reject_options_they_do_not_implement():
    import multimodal_learning_amd as m
    for bad in (dict(task="surv"), dict(reg_type="l2"), dict(optimizer_type="adabound"), dict(act_type="Sigmoid"),
                dict(fusion_type="concat"), dict(return_grad="True")):
        opt = m.stage2_opt(**bad)
        with pytest.raises(NotImplementedError):
            m.TeacherStage1Step(opt, device="cuda")
        with pytest.raises(NotImplementedError):
            m.DistillStep(opt, 64, device="cuda")
    # the ResNet student has no `__hasattr__`: path / mm / omic fail in the reference at the first batch
    for rt in ("path", "mm", "omic"):
        with pytest.raises(AttributeError):
            m.DistillStep(m.stage2_opt(reg_type=rt), 64, device="cuda")


def test_distill_step_with_the_adagrad_optimizer_runs_and_updates():
    """`--optimizer_type adagrad` (networks_new.py:86-87) through the whole distillation step: FusedAdagrad drives the student
    and the EMA copy (the update rule itself is pinned against torch.optim.Adagrad in tests/test_gpu_optim.py)."""
    import numpy as np
    import multimodal_learning_amd as m
    from oracle.step import synthetic_batch
    torch.manual_seed(0); np.random.seed(2019)
    opt = m.stage2_opt(optimizer_type="adagrad", dropout_rate=0.0, batch_size=4, nce_p=120, nce_k=200, nce_p2=20, nce_k2=128)
    step = m.DistillStep(opt, 256, device="cuda")
    assert type(step.optimizer).__name__ == "FusedAdagrad"
    for c in (step.criterion_kd, step.criterion_kd_path):
        c.contrast.verbose = False
    before = step.model.fc_new2.weight.detach().clone()
    ema_before = step.ema_model.fc_new2.weight.detach().clone()
    for it in range(2):
        bt = synthetic_batch(4, 64, n_data=256, P=opt.nce_p, K=opt.nce_k, seed=it)
        out = step.step(((bt["x_path"], bt["ema_x_path"]), torch.zeros(4), bt["x_omic"], torch.zeros(4), torch.zeros(4), bt["grade"],
                         bt["index"], bt["sample_idx"]))
        assert torch.isfinite(out["loss"]).item()
    assert (step.model.fc_new2.weight - before).abs().max().item() > 0
    assert (step.ema_model.fc_new2.weight - ema_before).abs().max().item() > 0
    sd = step.optimizer.state_dict()
    first = sd["state"][min(sd["state"])]      # (index 0 / 1 are the student's frozen output_range / output_shift: no state)
    assert "sum" in first and int(first["step"]) == 2 and len(sd["state"]) > 60
