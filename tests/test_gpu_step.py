"""Parity of the whole distillation step (train_test_path_multi_distill.py:249-330) on the GPU.
  * parity mode vs golden vectors produced by running the REFERENCE for 3 steps at BASELINE config 1
    (B=16, 224x224, 320-d omic): logits and losses within 1e-3 (north-star tolerance), GK-Refine scale,
    gradients, updated parameters, EMA, CRD banks and Z.
  * perf mode (bf16) vs the oracle with like-for-like operand rounding.
  * determinism: same inputs twice -> bit-identical loss."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _mk_step(opt, n_data, seed=0, verbose=False):
    import multimodal_learning_amd as m
    from oracle import weights as W
    from oracle.losses import CRDState
    step = m.DistillStep(opt, n_data, device="cuda")
    step.model.load_state_dict(W.make_state_dict(W.student_shapes(), seed + 1))
    step.ema_model.load_state_dict(W.make_state_dict(W.student_shapes(), seed + 2))
    step.fix_model.load_state_dict(W.make_state_dict(W.teacher_shapes(320), seed + 3))
    for i, crd in enumerate((step.criterion_kd, step.criterion_kd_path)):
        crd.embed_s.load_state_dict(W.make_state_dict(W.embed_shapes(), seed + 10 + 2 * i))
        crd.embed_t.load_state_dict(W.make_state_dict(W.embed_shapes(), seed + 11 + 2 * i))
        st = CRDState(n_data, opt.feat_dim, opt.nce_p, opt.nce_k, seed=seed + 20 + i)
        crd.contrast.memory_v1.copy_(st.memory_v1); crd.contrast.memory_v2.copy_(st.memory_v2)
        crd.contrast.verbose = verbose
    return step


def grad_row(R, ref, truth, got, atol, rtol, what):
    """A gradient against the reference golden OR the fp64 truth: passes when within atol + rtol * max|ref| of the golden, or
    when its distance to the truth does not exceed twice the reference's own distance to the truth (ReLU-boundary elements:
    the reference's fp32 run put one on the other side than the truth, this run possibly another one).  Both distances go into the report."""
    from tests.gpu_util import maxerr
    e_ref, mx = maxerr(ref, got)
    e_tru, _ = maxerr(truth, got)
    d_ref, _ = maxerr(truth, ref)
    tol = atol + rtol * mx
    ok = e_ref <= tol or e_tru <= max(tol, 2.0 * d_ref)      # (boundary elements are discrete events: the reference's run holds its
    #                                                           own, another fp32-grade run another - twice its distance)
    R.rows.append((what + " [golden %.2e | truth %.2e | golden-truth %.2e]" % (e_ref, e_tru, d_ref), 0.0 if ok else e_ref, mx, tol))


def _tuple(bt):
    B = bt["x_path"].shape[0]
    return ((bt["x_path"], bt["ema_x_path"]), torch.zeros(B), bt["x_omic"], torch.zeros(B), torch.zeros(B),
            bt["grade"], bt["index"], bt["sample_idx"])


@pytest.mark.parametrize("pmode", ["bf16x6", "fp16x3", "fp16x3/x1"])
def test_three_steps_vs_reference_golden(golden_dir, pmode):
    """Step 0 (before any parameter update): everything within 1e-3 of the REFERENCE's golden values.
    Steps 1-2 (after Adam updates): the reference's own fp32 run sits 1e-2..6e-2 from the fp64 truth on
    logits/loss (tests/golden/make_fp64_truth.py) because Adam's sign-like first steps amplify rounding, so the
    assertion is made against that noise floor: |HIP - truth| <= 6 x |reference_fp32 - truth| (+ the step's
    trajectory noise scale) + 1e-3.  That bound cannot see a 1e-2 bug after step 0: the REAL post-update check (every
    quantity at 1e-3, steps 0 and 1) is test_two_steps_from_mid_training_state_vs_reference_golden below, which starts
    both sides from a state where Adam is not sign-like; this test keeps steps 1-2 only as a cold-start regime check."""
    import multimodal_learning_amd as m
    from oracle.step import default_opt, synthetic_batch
    from tests.gpu_util import assert_close, maxerr, Report
    g = np.load(os.path.join(golden_dir, "step_b16_h224.npz"))
    t64 = np.load(os.path.join(golden_dir, "step_b16_h224_fp64.npz"))
    m.set_precision(pmode)
    try:
        step = _mk_step(default_opt(), int(g["n_data"]), seed=int(g["seed"]))
        R = Report("3 distill steps, parity mode vs REFERENCE golden (B=16, 224x224)")
        assert_close = R.close

        def ref_d(key, scale=1.0):
            return maxerr(np.asarray(t64[key]) * scale, np.asarray(g[key]) * scale)[0]

        def floor(key, got, what, scale=1.0, extra=0.0):
            """distance to the fp64 truth, bounded by 6x the reference's own distance (+ the step's
            trajectory noise scale `extra` for scalars, whose own reference distance can be accidentally tiny)"""
            err, mx = maxerr(np.asarray(t64[key]) * scale, got)
            R.rows.append((what + " [vs fp64 truth]", err, mx, 6 * ref_d(key, scale) + extra + 1e-3))

        for it in range(3):
            bt = synthetic_batch(int(g["B"]), int(g["H"]), seed=100 + it)
            out = step.step(_tuple(bt), epoch=it, ranks=[g["ranks"][2 * it], g["ranks"][2 * it + 1]])
            sd = step.model.state_dict(); esd = step.ema_model.state_dict()
            idx = bt["index"].cuda()
            if it == 0:
                P = dict(step.model.named_parameters())
                assert_close(g["g0_fc2_w"], P["fc_new2.weight"].grad, 1e-5, 1e-3, "grad fc2")
                floor("g0_conv1", P["conv1.weight"].grad, "grad conv1")
                assert_close(g["g0_l4_1_conv2_abs"], P["layer4.1.conv2.weight"].grad.abs().sum(), 1e-4, 1e-3, "grad l4.1.conv2 |.|_1")
                assert_close(g["g0_embed_s0"], step.criterion_kd.embed_s.linear.weight.grad, 1e-7, 1e-3, "grad embed_s")
                assert_close(g["g0_embed_t1"], step.criterion_kd_path.embed_t.linear.weight.grad, 1e-7, 1e-3, "grad embed_t")
                tol = 1e-3
                assert_close(g["logit_path0"], out["logit_path"], tol, 0, "logit_path step 0")
                assert_close(g["ema_logit0"], out["ema_logit"], tol, 0, "ema logit step 0")
                assert_close(g["fuse_logit0"], out["fuse_logit"], tol, 0, "teacher logit step 0")
                assert_close(g["path_feat0"], out["path_feat"], tol, 0, "path_feat step 0")
                assert_close(g["loss_cls0"], out["loss_cls"], tol, 0, "loss_cls step 0")
                assert_close(g["loss0"], out["loss"], tol, 0, "loss step 0")
                assert_close(g["loss_div1_0"], out["loss_div1"], tol, 0, "div1 step 0")
                assert_close(g["loss_div2_0"], out["loss_div2"], tol, 0, "div2 step 0")
                # golden stores the unscaled CRD losses; DistillStep returns beta-scaled ones (:296-297)
                assert_close(float(g["loss_kd1_0"]) * 0.02, out["loss_kd1"], tol, 0, "kd1 step 0")
                assert_close(float(g["loss_kd2_0"]) * 0.02, out["loss_kd2"], tol, 0, "kd2 step 0")
                assert_close(g["scale0"], out["scale"], tol, 0, "GK scale step 0")
                assert_close(g["p_fc2_0"], sd["fc_new2.weight"], 1e-4, 0, "Adam-updated fc2 step 0")
                assert_close(g["ema_fc2_0"], esd["fc_new2.weight"], 1e-4, 0, "EMA fc2 step 0")
                assert_close(g["params0_0"], step.criterion_kd.contrast.params, 1e-2, 1e-5, "CRD params / Z")
                assert_close(g["bank0_v1_rows0"], step.criterion_kd.contrast.memory_v1[idx], 1e-4, 0, "bank0 rows step 0")
                assert_close(g["bank1_v2_rows0"], step.criterion_kd_path.contrast.memory_v2[idx], 1e-4, 0, "bank1 rows step 0")
            else:
                assert_close(g[f"fuse_logit{it}"], out["fuse_logit"], 1e-3, 0, f"teacher logit step {it}")   # frozen net
                # trajectory noise scale of this step = the reference's own worst vector distance to the truth
                nz = 6 * max(ref_d(f"logit_path{it}"), ref_d(f"ema_logit{it}"), ref_d(f"path_feat{it}"))
                floor(f"logit_path{it}", out["logit_path"], f"logit_path step {it}", extra=nz)
                floor(f"ema_logit{it}", out["ema_logit"], f"ema logit step {it}", extra=nz)
                floor(f"path_feat{it}", out["path_feat"], f"path_feat step {it}", extra=nz)
                floor(f"loss_cls{it}", out["loss_cls"], f"loss_cls step {it}", extra=nz)
                floor(f"loss{it}", out["loss"], f"loss step {it}", extra=nz)
                floor(f"loss_div1_{it}", out["loss_div1"], f"div1 step {it}", extra=nz)
                floor(f"loss_kd1_{it}", out["loss_kd1"], f"kd1 step {it}", 0.02, extra=nz)
                floor(f"loss_kd2_{it}", out["loss_kd2"], f"kd2 step {it}", 0.02, extra=nz)
                floor(f"scale{it}", out["scale"], f"GK scale step {it}", extra=nz)
                floor(f"bank0_v1_rows{it}", step.criterion_kd.contrast.memory_v1[idx], f"bank0 rows step {it}")
                floor(f"bank1_v2_rows{it}", step.criterion_kd_path.contrast.memory_v2[idx], f"bank1 rows step {it}")
                assert_close(g[f"p_fc2_{it}"], sd["fc_new2.weight"], 1.5e-3, 0, f"Adam-updated fc2 step {it}")   # <= 3 Adam steps of lr 5e-4
                assert_close(g[f"params0_{it}"], step.criterion_kd.contrast.params, 1e-2, 1e-5, "CRD params / Z frozen")
        R.finish()
    finally:
        m.set_precision("bf16")


def test_perf_mode_step_vs_rounded_oracle_and_determinism():
    import multimodal_learning_amd as m
    import oracle
    from oracle.step import DistillOracle, default_opt, synthetic_batch
    from tests.gpu_util import assert_close, Report
    m.set_precision("bf16")
    opt = default_opt()
    bt = synthetic_batch(8, 128, seed=11)
    ranks = [np.random.RandomState(i).choice(np.arange(30, 100), 20, replace=False) for i in range(2)]
    losses = []
    for rep in range(2):
        step = _mk_step(opt, 1024, seed=0)
        out = step.step(_tuple(bt), ranks=ranks)
        losses.append((out["loss"].item(), out["logit_path"].cpu().clone()))
    assert losses[0][0] == losses[1][0] and torch.equal(losses[0][1], losses[1][1]), "step is not deterministic"
    # noise-floor check (see test_student_forward_perf_mode_noise_floor): HIP bf16 vs fp32 oracle is no worse
    # than the oracle's own bf16 emulation vs fp32 oracle
    ref32 = DistillOracle(opt, seed=0, n_data=1024).step(bt, mid_ranks=ranks)
    with oracle.Rounding.use("bf16"):
        refemu = DistillOracle(opt, seed=0, n_data=1024).step(bt, mid_ranks=ranks)
    e_gpu = (losses[0][1] - ref32["logit_path"]).abs()
    e_emu = (refemu["logit_path"] - ref32["logit_path"]).abs()
    dl_gpu = abs(out["loss"].item() - ref32["loss"].item()); dl_emu = abs(refemu["loss"].item() - ref32["loss"].item())
    print(f"\nperf-mode step: |dlogit| HIP max {e_gpu.max():.4f} mean {e_gpu.mean():.4f} | emu max {e_emu.max():.4f} "
          f"mean {e_emu.mean():.4f} ; |dloss| HIP {dl_gpu:.4f} emu {dl_emu:.4f} (loss {ref32['loss'].item():.4f})")
    assert e_gpu.mean() <= 2.0 * e_emu.mean() + 1e-3
    assert e_gpu.max() <= 3.0 * e_emu.max() + 1e-3


def test_checkpoint_resume_is_bit_identical():
    """Row f-3 (resume): save after two steps, rebuild a fresh DistillStep from the state dict, and the third step must be
    bitwise the same as the uninterrupted run - including the CRD banks / Z and the teacher's dropout stream, which the
    reference's own checkpoint (train_test_path_multi_distill.py:387-402) does not carry."""
    import copy
    import multimodal_learning_amd as m
    from bench import make_batch
    m.set_precision("bf16")
    opt = m.stage2_opt(dropout_rate=0.1, batch_size=8)
    n_data = 256
    torch.manual_seed(0); np.random.seed(7)
    a = m.DistillStep(opt, n_data, device="cuda")
    for crd in (a.criterion_kd, a.criterion_kd_path):
        crd.contrast.verbose = False
    batches = [make_batch(8, 64, n_data, opt, "cuda", seed=i) for i in range(3)]
    ranks = [[list(range(30, 50)), list(range(40, 60))] for _ in range(3)]
    for i in range(2):
        a.step(batches[i], epoch=1, ranks=ranks[i])
    # through torch.save / torch.load with the weights_only=True default of torch >= 2.6 (the numpy RNG
    # state used to be a raw get_state() tuple, which that loader rejects)
    import io
    buf = io.BytesIO()
    torch.save(a.state_dict(), buf)
    buf.seek(0)
    sd = torch.load(buf, weights_only=True)
    # the third step draws its CRD rank lists from numpy's global RNG (memory_new.py:311) - part of the saved state
    out_a = a.step(batches[2], epoch=1, ranks=None)
    torch.manual_seed(123); np.random.seed(99)             # a differently initialised object ...
    b = m.DistillStep(m.stage2_opt(dropout_rate=0.1, batch_size=8), n_data, device="cuda")
    for crd in (b.criterion_kd, b.criterion_kd_path):
        crd.contrast.verbose = False
    b.step(batches[0], epoch=1, ranks=ranks[0])            # ... that has even taken a step of its own
    b.load_state_dict(sd)
    out_b = b.step(batches[2], epoch=1, ranks=None)
    for k in ("loss", "loss_cls", "loss_div1", "loss_div2", "loss_kd1", "loss_kd2"):
        assert torch.equal(out_a[k], out_b[k]), k
    assert torch.equal(a.optimizer.flat.flat, b.optimizer.flat.flat)
    assert torch.equal(a.ema_flat.flat, b.ema_flat.flat)
    for ca, cb in ((a.criterion_kd, b.criterion_kd), (a.criterion_kd_path, b.criterion_kd_path)):
        assert torch.equal(ca.contrast.memory_v1, cb.contrast.memory_v1)
        assert torch.equal(ca.contrast.params, cb.contrast.params)


@pytest.mark.parametrize("pmode", ["bf16x6", "fp16x3/x1"])
def test_stage1_teacher_step_vs_reference_golden(golden_dir, pmode):
    """Row f-1: two stage-1 mean-teacher steps (train_test_MT.py:121-230, grading task, num_teachers 2) in parity mode
    against the fixture produced by the reference's modules.  Step 0 (identical weights on both sides) is asserted
    tightly; step 1 runs on Adam-updated weights and is asserted at the post-update noise floor."""
    import multimodal_learning_amd as m
    from oracle import weights as W
    from oracle.step import synthetic_batch
    g = np.load(os.path.join(golden_dir, "stage1_b4_h64.npz"))
    m.set_precision(pmode)
    try:
        opt = m.stage2_opt(dropout_rate=0.0, batch_size=4, cut_fuse_grad=False, num_teachers=2)
        opt.pred_distill, opt.KD_weight = 1, float(g["KD_weight"])
        opt.lr, opt.weight_decay, opt.ema_decay = float(g["lr"]), float(g["weight_decay"]), float(g["ema_decay"])
        model = m.define_net(opt, 1); ema = m.define_net(opt, 1)
        sd = W.make_state_dict(W.teacher_shapes(320), 3)
        model.load_state_dict(sd); ema.load_state_dict(sd)
        st = m.TeacherStage1Step(opt, device="cuda", models=(model.cuda(), ema.cuda()))
        for it in range(2):
            bt = synthetic_batch(4, 64, seed=20 + it)
            z = torch.zeros(4)
            batch = ((bt["x_path"], bt["ema_x_path"]), z, bt["x_omic"], z, z, bt["grade"], bt["index"], bt["sample_idx"])
            out = st.step(batch)
            tol = 1e-3 if it == 0 else 5e-2
            for k in ("loss", "loss_nll"):
                assert abs(out[k].item() - float(g[f"{k}{it}"])) <= tol * abs(float(g[f"{k}{it}"])), (it, k, out[k].item())
            # (the consistency term is the difference of two nearly equal networks: after the sign-like first Adam step it is
            # the most update-noise-sensitive quantity of the step)
            kd_tol = tol if it == 0 else 0.25
            assert abs(out["loss_pred_KD"].item() - float(g[f"loss_kd{it}"])) <= kd_tol * max(abs(float(g[f"loss_kd{it}"])), 1e-2)
            for k in ("pred", "pred_path", "pred_omic"):
                assert np.abs(out[k].cpu().numpy() - g[f"{k}{it}"]).max() <= tol * 10, (it, k)
            if it == 0:   # weights after the first Adam step and the EMA copies
                msd, esd = st.model.state_dict(), st.ema_model.state_dict()
                for key in g.files:
                    if key.startswith("w0_"):
                        name = key[3:]
                        ref = g[key]
                        # Adam's first step moves every weight by ~lr * sign(g): compare the update, not the weight
                        upd_ref = ref - sd[name].numpy()
                        upd = msd[name].cpu().numpy() - sd[name].numpy()
                        frac_bad = float((np.abs(upd - upd_ref) > 0.2 * float(g["lr"])).mean())
                        assert frac_bad < 0.02, (name, frac_bad)
                        # alpha = 0 at the first step: the EMA copy equals the updated student (same rare sign flips)
                        ebad = float((np.abs(esd[name].cpu().numpy() - g["e0_" + name]) > 0.2 * float(g["lr"])).mean())
                        assert ebad < 0.02, (name, ebad)
                        assert np.abs(esd[name].cpu().numpy() - msd[name].cpu().numpy()).max() <= 1e-7, name
    finally:
        m.set_precision("bf16")


@pytest.mark.parametrize("pmode", ["bf16x6", "fp16x3/x1"])
def test_mia2022_variant_steps_vs_reference_golden(golden_dir, pmode):
    """SURVEY row a17 end to end: DistillStep(variant="mia2022") = the batch body of
    "MIA 2022/train_test_path_multi_distill_v2.py":397-507 (CRD_criterion_v3 bank weighted by epoch/niter_decay,
    momentum GK-Refine carried over the iterations, x len(KD_loss_list)) against vectors produced by running the
    reference's own modules (tests/golden/make_golden_mia2022_step.py).  Step 0 is compared at 1e-3; steps 1-2 come
    after Adam's sign-like first updates, which amplify fp32 rounding, and are judged like
    test_three_steps_vs_reference_golden: |HIP - fp64 run| <= 6 x |reference fp32 - fp64 run| + noise scale + 1e-3."""
    import multimodal_learning_amd as m
    from oracle import weights as W
    from oracle.step import default_opt, synthetic_batch
    from oracle.variants import CRDv3State
    from tests.gpu_util import Report, maxerr
    g = np.load(os.path.join(golden_dir, "mia2022_step_b8_h64.npz"))
    B, H, n_data, K = int(g["B"]), int(g["H"]), int(g["n_data"]), int(g["K"])
    opt = default_opt(nce_k=K, grads_m=float(g["grads_m"]), grads_thresh="False", thresh=0.1,
                      niter_decay=int(g["niter_decay"]))
    m.set_precision(pmode)      # (fp16x3/x1: the tolerance-compliant arithmetic of the bench, same tolerances)
    try:
        step = m.DistillStep(opt, n_data, device="cuda", variant="mia2022")
        step.model.load_state_dict(W.make_state_dict(W.student_shapes(), 1))
        step.ema_model.load_state_dict(W.make_state_dict(W.student_shapes(), 2))
        step.fix_model.load_state_dict(W.make_state_dict(W.teacher_shapes(320), 3))
        for i, crd in enumerate((step.criterion_kd, step.criterion_kd_path)):
            crd.embed_s.load_state_dict(W.make_state_dict(W.embed_shapes(), 10 + 2 * i))
            crd.embed_t.load_state_dict(W.make_state_dict(W.embed_shapes(), 11 + 2 * i))
            st = CRDv3State(n_data, K=K, seed=20 + i)
            crd.contrast.memory_v1.copy_(st.memory_v1); crd.contrast.memory_v2.copy_(st.memory_v2)
            crd.contrast.verbose = False
        R = Report("3 MIA-2022 distill steps, parity mode vs REFERENCE golden (B=8, 64x64)")
        t64 = np.load(os.path.join(golden_dir, "mia2022_step_b8_h64_fp64.npz"))

        def ref_d(key):
            return maxerr(np.asarray(t64[key]), np.asarray(g[key]))[0]

        def floor(key, got, what, extra=0.0, mult=6):
            """distance to the fp64 run of the same reference calls, bounded by 6x the reference's own fp32 distance"""
            err, mx = maxerr(np.asarray(t64[key]), got)
            R.rows.append((what + " [vs fp64 truth]", err, mx, mult * ref_d(key) + extra + 1e-3))
        for it in range(3):
            bt = synthetic_batch(B, H, n_data=n_data, P=1, K=K, seed=300 + it)
            out = step.step(_tuple(bt), epoch=int(g[f"epoch{it}"]))
            sd = step.model.state_dict(); esd = step.ema_model.state_dict()
            idx = bt["index"].cuda()
            names = (("logit_path", "logit_path"), ("path_feat", "path_feat"), ("ema_logit", "ema_logit"),
                     ("loss_cls", "loss_cls"), ("loss_div1_", "loss_div1"), ("loss_div2_", "loss_div2"),
                     ("loss_kd1_", "loss_kd1"), ("loss_kd2_", "loss_kd2"), ("scale", "scale"), ("loss", "loss"))
            R.close(g[f"fuse_logit{it}"], out["fuse_logit"], 1e-3, 0, f"teacher logit step {it}")      # frozen net
            R.close(g[f"params0_{it}"], step.criterion_kd.contrast.params, 1e-2, 1e-5, "CRD v3 params / Z")
            if it == 0:
                P = dict(step.model.named_parameters())
                R.close(g["g0_fc2_w"], P["fc_new2.weight"].grad, 1e-5, 1e-3, "grad fc2")
                # (fp16x3/x1: the backward's convolutions see 11-bit operands; through 17 layers the first layer's gradient sits
                # 1.0 % from the truth in max norm where the reference's own fp32 run sits 0.15 %: 8x instead of 6x)
                floor("g0_conv1", P["conv1.weight"].grad, "grad conv1", mult=8 if pmode == "fp16x3/x1" else 6)
                R.close(g["g0_embed_s0"], step.criterion_kd.embed_s.linear.weight.grad, 1e-7, 1e-3, "grad embed_s")
                R.close(g["g0_embed_t1"], step.criterion_kd_path.embed_t.linear.weight.grad, 1e-7, 1e-3, "grad embed_t")
                for key, name in names:
                    R.close(g[f"{key}{it}"], out[name], 1e-3, 0, f"{name} step 0")
                R.close(g["p_fc2_0"], sd["fc_new2.weight"], 1e-4, 0, "Adam-updated fc2 step 0")
                R.close(g["ema_fc2_0"], esd["fc_new2.weight"], 1e-4, 0, "EMA fc2 step 0")
                R.close(g["bank0_v1_rows0"], step.criterion_kd.contrast.memory_v1[idx], 1e-4, 0, "bank0 rows step 0")
                R.close(g["bank1_v2_rows0"], step.criterion_kd_path.contrast.memory_v2[idx], 1e-4, 0, "bank1 rows step 0")
            else:
                nz = 6 * max(ref_d(f"logit_path{it}"), ref_d(f"ema_logit{it}"), ref_d(f"path_feat{it}"))
                for key, name in names:
                    floor(f"{key}{it}", out[name], f"{name} step {it}", extra=nz)
                floor(f"bank0_v1_rows{it}", step.criterion_kd.contrast.memory_v1[idx], f"bank0 rows step {it}")
                floor(f"bank1_v2_rows{it}", step.criterion_kd_path.contrast.memory_v2[idx], f"bank1 rows step {it}")
                R.close(g[f"p_fc2_{it}"], sd["fc_new2.weight"], 1.5e-3, 0, f"Adam-updated fc2 step {it}")
                R.close(g[f"ema_fc2_{it}"], esd["fc_new2.weight"], 1.5e-3, 0, f"EMA fc2 step {it}")
        R.finish()
        # the momentum weights and the epoch weight live in persistent device buffers: graph replay must follow them
        step.enable_graph()
        e_before = step._mo_state.clone()
        for it in range(3, 6):
            bt = synthetic_batch(B, H, n_data=n_data, P=1, K=K, seed=300 + it)
            out = step.step(_tuple(bt), epoch=it)
        torch.cuda.synchronize()
        assert step._static is not None and step._static["graph"] is not None, "graph path was not taken"
        assert not torch.equal(e_before, step._mo_state) and torch.isfinite(out["loss"]).item()
        assert abs(float(step._e_dev) - 5 / opt.niter_decay) < 1e-7
        sd1 = step.state_dict()
        assert torch.equal(sd1["gk_momentum_scale"], step._mo_state)
    finally:
        m.set_precision("bf16")


@pytest.mark.parametrize("pmode", ["bf16x6", "fp16x3/x1"])
def test_mia2023_variant_steps_vs_reference_golden(golden_dir, pmode):
    """SURVEY row a18 end to end: DistillStep(variant="mia2023") = the batch body of
    "MIA 2023/stage2_unimodal_student/train_test_path_multi_distill.py":318-448 (per-sample KL rows, confidence
    discrepancy query weights switched on at opt.start_reweight, CRD_criterion_v10 KNN bank, per-sample GK-Refine)
    against vectors produced by running the reference's own modules (tests/golden/make_golden_mia2023_step.py).
    Step 0 at 1e-3; steps 1-2 against the fp64 run of the same calls, bounded by 6x the reference's own fp32 distance."""
    import multimodal_learning_amd as m
    from oracle import weights as W
    from oracle.step import default_opt, synthetic_batch
    from oracle.variants import CRDv10State
    from tests.gpu_util import Report, maxerr
    g = np.load(os.path.join(golden_dir, "mia2023_step_b8_h64.npz"))
    t64 = np.load(os.path.join(golden_dir, "mia2023_step_b8_h64_fp64.npz"))
    B, H, n_data, K = int(g["B"]), int(g["H"]), int(g["n_data"]), int(g["K"])
    labels = torch.as_tensor(g["labels"])
    class_idx = [np.nonzero((labels == c).numpy())[0] for c in range(3)]
    opt = default_opt(nce_k=K, nce_p=int(g["num_pos"]), pos_extra="neighbors", neg_mode="all_others",
                      start_reweight=int(g["start_reweight"]), discrep_scale=1, max_discrep=float(g["max_discrep"]),
                      use_grads_thresh="True", grads_thresh=float(g["grads_thresh"]), loss_weighting="GK_refine",
                      batch_size=B)
    m.set_precision(pmode)      # (fp16x3/x1: the tolerance-compliant arithmetic of the bench, same tolerances)
    try:
        step = m.DistillStep(opt, n_data, device="cuda", variant="mia2023", train_class_idx=class_idx)
        step.model.load_state_dict(W.make_state_dict(W.student_shapes(), 1))
        step.ema_model.load_state_dict(W.make_state_dict(W.student_shapes(), 2))
        step.fix_model.load_state_dict(W.make_state_dict(W.teacher_shapes(320), 3))
        for i, crd in enumerate((step.criterion_kd, step.criterion_kd_path)):
            crd.embed_s.load_state_dict(W.make_state_dict(W.embed_shapes(), 10 + 2 * i))
            crd.embed_t.load_state_dict(W.make_state_dict(W.embed_shapes(), 11 + 2 * i))
            st = CRDv10State(n_data, labels, K=K, seed=20 + i)
            crd.contrast.memory_v1.copy_(st.memory_v1); crd.contrast.memory_v2.copy_(st.memory_v2)
            crd.contrast.verbose = False
        R = Report("3 MIA-2023 distill steps, parity mode vs REFERENCE golden (B=8, 64x64)")

        def ref_d(key):
            return maxerr(np.asarray(t64[key]), np.asarray(g[key]))[0]

        def floor(key, got, what, extra=0.0):
            err, mx = maxerr(np.asarray(t64[key]), got)
            R.rows.append((what + " [vs fp64 truth]", err, mx, 6 * ref_d(key) + extra + 1e-3))
        names = (("logit_path", "logit_path"), ("path_feat", "path_feat"), ("ema_logit", "ema_logit"),
                 ("loss_cls", "loss_cls"), ("loss_div1_", "loss_div1"), ("loss_div2_", "loss_div2"),
                 ("loss_kd1_", "loss_kd1"), ("loss_kd2_", "loss_kd2"), ("scale", "scale"), ("loss", "loss"),
                 ("w1_", "w1"), ("w2_", "w2"), ("rows_div1_", "rows_div1"), ("rows_kd1_", "rows_kd1"))
        for it in range(3):
            bt = synthetic_batch(B, H, n_data=n_data, P=1, K=K, seed=400 + it)
            bt["grade"] = labels[bt["index"]]
            out = step.step(_tuple(bt), epoch=it)
            sd = step.model.state_dict(); esd = step.ema_model.state_dict()
            idx = bt["index"].cuda()
            R.close(g[f"fuse_logit{it}"], out["fuse_logit"], 1e-3, 0, f"teacher logit step {it}")
            if it == 0:
                P = dict(step.model.named_parameters())
                R.close(g["g0_fc2_w"], P["fc_new2.weight"].grad, 1e-5, 1e-3, "grad fc2")
                floor("g0_conv1", P["conv1.weight"].grad, "grad conv1")
                R.close(g["g0_embed_s0"], step.criterion_kd.embed_s.linear.weight.grad, 1e-7, 1e-3, "grad embed_s")
                R.close(g["g0_embed_t1"], step.criterion_kd_path.embed_t.linear.weight.grad, 1e-7, 1e-3, "grad embed_t")
                for key, name in names:
                    R.close(g[f"{key}0"], out[name], 1e-3, 0, f"{name} step 0")
                R.close(g["p_fc2_0"], sd["fc_new2.weight"], 1e-4, 0, "Adam-updated fc2 step 0")
                R.close(g["ema_fc2_0"], esd["fc_new2.weight"], 1e-4, 0, "EMA fc2 step 0")
                R.close(g["bank0_v1_rows0"], step.criterion_kd.contrast.memory_v1[idx], 1e-4, 0, "bank0 rows step 0")
                R.close(g["bank1_v2_rows0"], step.criterion_kd_path.contrast.memory_v2[idx], 1e-4, 0, "bank1 rows step 0")
            else:
                nz = 6 * max(ref_d(f"logit_path{it}"), ref_d(f"ema_logit{it}"), ref_d(f"path_feat{it}"))
                for key, name in names:
                    floor(f"{key}{it}", out[name], f"{name} step {it}", extra=nz)
                floor(f"bank0_v1_rows{it}", step.criterion_kd.contrast.memory_v1[idx], f"bank0 rows step {it}")
                floor(f"bank1_v2_rows{it}", step.criterion_kd_path.contrast.memory_v2[idx], f"bank1 rows step {it}")
                R.close(g[f"p_fc2_{it}"], sd["fc_new2.weight"], 1.5e-3, 0, f"Adam-updated fc2 step {it}")
                R.close(g[f"ema_fc2_{it}"], esd["fc_new2.weight"], 1.5e-3, 0, f"EMA fc2 step {it}")
        R.finish()
        assert float(out["w1"].max()) > 1.0, "re-weighting must be on from opt.start_reweight"
        step.enable_graph()
        for it in range(3, 6):
            bt = synthetic_batch(B, H, n_data=n_data, P=1, K=K, seed=400 + it)
            bt["grade"] = labels[bt["index"]]
            out = step.step(_tuple(bt), epoch=0 if it == 5 else it)
        torch.cuda.synchronize()
        assert step._static is not None and step._static["graph"] is not None, "graph path was not taken"
        assert torch.isfinite(out["loss"]).item()
        assert float(out["w1"].max()) == 1.0, "the re-weighting switch is a device scalar: replay must follow the epoch"
    finally:
        m.set_precision("bf16")


def test_graph_replay_on_resident_input_sets_matches_staged_inputs():
    """Device-resident batches are adopted in place as the captured graphs' input sets (two sets, one graph each,
    DistillStep._ensure_slot / precapture): the trajectory must be bit-identical to feeding the same batches from the
    host through the staging copies, and no staging copy may be left (the adopted tensors ARE the static buffers)."""
    import multimodal_learning_amd as m
    from oracle.step import default_opt, synthetic_batch
    m.set_precision("bf16")
    opt = default_opt()
    bts = [synthetic_batch(8, 64, seed=40 + i) for i in range(2)]
    rng = np.random.RandomState(3)
    ranks = [[rng.choice(np.arange(30, 100), 20, replace=False) for _ in range(2)] for _ in range(7)]

    def run(resident):
        step = _mk_step(opt, 1024, seed=0)
        step.enable_graph()
        if resident:
            feeds = [tuple(tuple(u.cuda() for u in t) if isinstance(t, tuple) else t.cuda() for t in _tuple(bt)) for bt in bts]
        else:
            feeds = [_tuple(bt) for bt in bts]
        for i in range(7):
            if resident and i == 3:
                assert step.precapture(feeds[1], epoch=1)
                assert len(step._slots) == 2 and all(q["graph"] is not None for q in step._slots)
            out = step.step(feeds[i % 2], epoch=1, ranks=ranks[i])
        torch.cuda.synchronize()
        if resident:
            assert step._static["x_path"].data_ptr() == feeds[0][0][0].data_ptr(), "inputs were staged, not adopted"
            assert len(step._slots) == 2
        else:
            assert len(step._slots) == 1
        return out["loss"].item(), step.model.state_dict()["fc_new2.weight"].clone(), \
            step.ema_model.state_dict()["conv1.weight"].clone(), step.criterion_kd.contrast.memory_v1.clone()

    a, b = run(False), run(True)
    assert a[0] == b[0], (a[0], b[0])
    for x, y in zip(a[1:], b[1:]):
        assert torch.equal(x, y)


def test_stage1_step_with_crd_and_orth_terms_vs_reference_golden(golden_dir):
    """Row f-1 with --CRD_distill 1 --orth_loss True: two steps of train_test_MT.py:121-230 (vanilla CRD between the
    student's and the EMA teacher's fused features, six projection heads in the optimiser, orthogonality loss between
    the path and omic features) against the reference's own modules (tests/golden/make_golden_stage1_terms.py)."""
    import multimodal_learning_amd as m
    from oracle import weights as W
    from oracle.step import synthetic_batch
    from oracle.variants import CRDv3State
    from tests.test_oracle_variants import _embed2_state
    from tests.gpu_util import Report
    g = np.load(os.path.join(golden_dir, "stage1_terms_step_b4_h64.npz"))
    K, n_data = int(g["K"]), int(g["n_data"])
    m.set_precision("bf16x6")
    try:
        opt = m.stage2_opt(dropout_rate=0.0, batch_size=4, cut_fuse_grad=False, num_teachers=2)
        opt.pred_distill, opt.KD_weight = 1, float(g["KD_weight"])
        opt.CRD_distill, opt.CRD_weight, opt.orth_loss, opt.SP_distill = 1, float(g["CRD_weight"]), "True", 0
        opt.nce_k, opt.n_data = K, n_data
        model = m.define_net(opt, 1); ema = m.define_net(opt, 1)
        sd = W.make_state_dict(W.teacher_shapes(320), 3)
        model.load_state_dict(sd); ema.load_state_dict(sd)
        st = m.TeacherStage1Step(opt, device="cuda", models=(model.cuda(), ema.cuda()))
        for i, c in enumerate((st.CRD_criterion_path, st.CRD_criterion_omic, st.CRD_criterion_fuse)):
            c.embed_s.load_state_dict(_embed2_state(90 + 2 * i)); c.embed_t.load_state_dict(_embed2_state(91 + 2 * i))
            bank = CRDv3State(n_data, K=K, seed=100 + i)
            c.contrast.memory_v1.copy_(bank.memory_v1); c.contrast.memory_v2.copy_(bank.memory_v2)
        ops_lr = float(opt.lr)
        R = Report("2 stage-1 steps with CRD + orthogonality terms vs REFERENCE golden (B=4, 64x64)")
        for it in range(2):
            bt = synthetic_batch(4, 64, n_data=n_data, P=1, K=K, seed=60 + it)
            z = torch.zeros(4)
            batch = ((bt["x_path"], bt["ema_x_path"]), z, bt["x_omic"], z, z, bt["grade"], bt["index"], bt["sample_idx"])
            out = st.step(batch, epoch=it)
            rt = 1e-3 if it == 0 else 5e-2
            sc = lambda k: np.asarray(g[k]).reshape(())      # (the reference's CRD term has shape [1])
            R.close(sc(f"loss{it}"), out["loss"].reshape(()), 0, rt, f"loss step {it}")
            R.close(sc(f"loss_nll{it}"), out["loss_nll"].reshape(()), 0, rt, f"loss_nll step {it}")
            R.close(sc(f"loss_CRD{it}"), out["loss_CRD"].reshape(()), 0, rt, f"loss_CRD step {it}")
            R.close(sc(f"loss_orth{it}"), out["loss_orth"].reshape(()), 1e-6 if it == 0 else 1e-4, rt, f"loss_orth step {it}")
            R.close(g[f"pred{it}"], out["pred"], 10 * rt, 0, f"pred step {it}")
            R.close(g[f"bank_v1_rows{it}"], st.CRD_criterion_fuse.contrast.memory_v1[bt["index"].cuda()],
                    1e-4 if it == 0 else 2e-2, 0, f"fuse bank rows step {it}")
            if it == 0:
                msd = st.model.state_dict()
                for key in g.files:
                    if key.startswith("w0_") and not key.startswith("w0_embed"):
                        name = key[3:]
                        upd_ref = g[key] - sd[name].numpy()
                        upd = msd[name].cpu().numpy() - sd[name].numpy()
                        frac_bad = float((np.abs(upd - upd_ref) > 0.2 * ops_lr).mean())
                        assert frac_bad < 0.02, (name, frac_bad)
                # the projection heads are in the optimiser: the used one follows its gradient, the unused ones only decay
                e0 = _embed2_state(94)["linear.2.weight"].numpy()
                upd_ref = g["w0_embed_s_fuse"] - e0
                upd = st.CRD_criterion_fuse.embed_s.linear[2].weight.detach().cpu().numpy() - e0
                assert float((np.abs(upd - upd_ref) > 0.2 * ops_lr).mean()) < 0.02
                R.close(g["w0_embed_s_path"], st.CRD_criterion_path.embed_s.linear[0].weight, 2e-6, 0, "unused head after step 0")
        R.finish()
    finally:
        m.set_precision("bf16")


@pytest.mark.parametrize("aux_iter", [1, 2])
def test_tsvd_stage1_step_graph_replay_equals_eager(aux_iter):
    """TeacherStage1Step.enable_graph(): the stage-1 step with the t-SVD constraint replayed from captured HIP graphs (one per
    kind of step: with / without the auxiliary update; the threshold Lambda / mu, the penalty's mu and the CRD weight read from
    device memory) against the same steps launched eagerly - same weights, same batches, two resident input sets in turn.
    Perf arithmetic on both sides (the graph changes launch mechanics, not kernels): losses and updated weights agree to
    rounding of mu in fp32 (the eager path passes it as a double-precision host scalar)."""
    import copy
    import multimodal_learning_amd as m
    from oracle import weights as W
    from oracle.step import synthetic_batch
    B = 8
    opt = m.stage2_opt(dropout_rate=0.0, batch_size=B, cut_fuse_grad=False, num_teachers=2)
    opt.pred_distill, opt.KD_weight, opt.CRD_distill, opt.SP_distill, opt.orth_loss = 1, 1.0, 0, 0, "False"
    opt.tSVD_loss, opt.tSVD_mode, opt.n_views, opt.aux_iter = "True", "pathomic", 4, aux_iter
    opt.mu, opt.pho, opt.max_mu, opt.Lambda_global = 1e-3, 1.3, 10.0, 0.05
    bts = []
    for i in range(2):
        bt = synthetic_batch(B, 64, seed=90 + i)
        z = torch.zeros(B).cuda()
        bts.append(((bt["x_path"].cuda(), bt["ema_x_path"].cuda()), z, bt["x_omic"].cuda(), z, z, bt["grade"].cuda(), bt["index"].cuda(),
                    bt["sample_idx"].cuda()))
    res = {}
    for graph in (False, True):
        model = m.define_net(opt, 1); ema = m.define_net(opt, 1)
        model.load_state_dict(W.make_state_dict(W.teacher_shapes(320), 3)); ema.load_state_dict(W.make_state_dict(W.teacher_shapes(320), 4))
        st = m.TeacherStage1Step(copy.copy(opt), device="cuda", models=(model.cuda(), ema.cuda()))
        if graph:
            st.enable_graph()
        st.start_epoch()
        losses = []
        for it in range(8):
            out = st.step(bts[(it // 2) % 2])      # (both kinds of step on both input sets)
            losses.append({k: float(out[k]) for k in ("loss", "loss_nll", "loss_tsvd", "loss_pred_KD")})
        if graph:
            assert st._g_sets and len(st._g_sets) == 2 and all(len(q["graphs"]) == (1 if aux_iter == 1 else 2) for q in st._g_sets), \
                "steps 2.. must have been replayed from graphs (two input sets)"
        res[graph] = dict(losses=losses, w=st.model.state_dict()["path_net.fc_new2.weight"].clone() if "path_net.fc_new2.weight" in st.model.state_dict()
                          else next(iter(st.model.parameters())).detach().clone(), mu=st.mu, aux=st.aux_tensor1[2].clone())
    for it in range(8):
        for k, v in res[False]["losses"][it].items():
            g = res[True]["losses"][it][k]
            assert abs(g - v) <= 2e-3 * max(1.0, abs(v)), (it, k, g, v)
    assert res[True]["mu"] == res[False]["mu"]
    assert (res[True]["aux"] - res[False]["aux"]).abs().max().item() <= 2e-3 * max(1.0, res[False]["aux"].abs().max().item())
    assert (res[True]["w"] - res[False]["w"]).abs().max().item() <= 5e-3 * max(1e-3, res[False]["w"].abs().max().item())


@pytest.mark.parametrize("pmode", ["bf16x6", "fp16x3/x1"])
@pytest.mark.parametrize("fixture", ["tsvd_step_b8_h64.npz", "tsvd_step_b8_h64_v8.npz"])
def test_tsvd_stage1_step_vs_reference_trainer_logic(golden_dir, fixture, pmode):
    """Row a16 end to end (BASELINE cfg 4's computation at a small size): two stage-1 steps with the t-SVD constraint
    ("MIA 2022/train_test_tSVD.py":199-470: adjacency tensors over 4 views per modality, auxiliary update at every batch,
    mu schedule, Frobenius penalty) against the reference's modules driven in the trainer's order; the absent
    `update_aux` is our proximal operator on both sides (parity unpinned for that one function).  The
    second fixture has n_views = 8: four more views mixed from the max-normalised mean-teacher features (:341-363)."""
    import multimodal_learning_amd as m
    from oracle import weights as W
    from oracle.step import synthetic_batch
    from tests.gpu_util import Report
    g = np.load(os.path.join(golden_dir, fixture))
    B = int(g["B"])
    m.set_precision(pmode)
    try:
        opt = m.stage2_opt(dropout_rate=0.0, batch_size=B, cut_fuse_grad=bool(int(g["cut_fuse_grad"])), num_teachers=2)
        opt.pred_distill, opt.KD_weight, opt.CRD_distill, opt.SP_distill, opt.orth_loss = 1, float(g["KD_weight"]), 0, 0, "False"
        opt.tSVD_loss, opt.tSVD_mode, opt.n_views, opt.aux_iter = "True", "pathomic", int(g["n_views"]), 1
        opt.mu, opt.pho, opt.max_mu, opt.Lambda_global = float(g["mu"]), float(g["pho"]), float(g["max_mu"]), float(g["Lambda_global"])
        opt.lr = float(g["lr"])
        model = m.define_net(opt, 1); ema = m.define_net(opt, 1)
        sd = W.make_state_dict(W.teacher_shapes(320), 3)
        model.load_state_dict(sd); ema.load_state_dict(W.make_state_dict(W.teacher_shapes(320), 4))
        st = m.TeacherStage1Step(opt, device="cuda", models=(model.cuda(), ema.cuda()))
        R = Report("2 stage-1 steps with the t-SVD constraint vs the reference trainer's logic (B=8, 64x64)")
        sc = lambda k: np.asarray(g[k]).reshape(())
        for it in range(2):
            bt = synthetic_batch(B, 64, seed=70 + it)
            z = torch.zeros(B)
            out = st.step(((bt["x_path"], bt["ema_x_path"]), z, bt["x_omic"], z, z, bt["grade"], bt["index"], bt["sample_idx"]))
            rt = 1e-3 if it == 0 else 5e-2
            R.close(sc(f"loss{it}"), out["loss"].reshape(()), 0, rt, f"loss step {it}")
            R.close(sc(f"loss_nll{it}"), out["loss_nll"].reshape(()), 0, rt, f"loss_nll step {it}")
            R.close(sc(f"loss_tsvd{it}"), out["loss_tsvd"].reshape(()), 1e-5, 2 * rt, f"loss_tsvd step {it}")
            R.close(sc(f"mu{it}"), torch.tensor(st.mu), 1e-9, 1e-6, f"mu after step {it}")
            R.close(g[f"adj1_2_{it}"], st.adj_tensor1[2], 1e-4 if it == 0 else 2e-2, 0, f"adjacency path view 2 step {it}")
            R.close(g[f"adj2_3_{it}"], st.adj_tensor2[3], 1e-4 if it == 0 else 2e-2, 0, f"adjacency omic view 3 step {it}")
            R.close(g[f"aux1_2_{it}"], st.aux_tensor1[2], 2e-4 if it == 0 else 2e-2, 0, f"aux path view 2 step {it}")
            R.close(g[f"aux2_0_{it}"], st.aux_tensor2[0], 2e-4 if it == 0 else 2e-2, 0, f"aux omic view 0 step {it}")
            R.close(sc(f"path_TNN{it}"), st.path_TNN.reshape(()), 1e-3 if it == 0 else 5e-2, 1e-3, f"path TNN step {it}")
            R.close(g[f"adj1_last_{it}"], st.adj_tensor1[-1], 1e-4 if it == 0 else 2e-2, 0, f"adjacency path last view step {it}")
            R.close(g[f"aux2_last_{it}"], st.aux_tensor2[-1], 2e-4 if it == 0 else 2e-2, 0, f"aux omic last view step {it}")
            if it == 0:
                msd = st.model.state_dict()
                for key in g.files:
                    if key.startswith("w0_"):
                        name = key[3:]
                        upd_ref = g[key] - sd[name].numpy()
                        upd = msd[name].cpu().numpy() - sd[name].numpy()
                        assert float((np.abs(upd - upd_ref) > 0.2 * opt.lr).mean()) < 0.02, name
        R.finish()
    finally:
        m.set_precision("bf16")


def test_tsvd_stage1_step_at_config4_batch():
    """BASELINE config 4's batch (128) through the stage-1 step with the t-SVD constraint: the auxiliary tensors the step
    leaves behind are the oracle's proximal operator of the adjacency tensors it computed (one-sided Jacobi kernel,
    64 < B <= 128), and the step is finite."""
    import multimodal_learning_amd as m
    from oracle import weights as W
    from oracle import variants as OV
    from oracle.step import synthetic_batch
    B = 128
    opt = m.stage2_opt(dropout_rate=0.0, batch_size=B, cut_fuse_grad=True, num_teachers=2)
    opt.pred_distill, opt.KD_weight, opt.CRD_distill, opt.SP_distill, opt.orth_loss = 1, 1.0, 0, 0, "False"
    opt.tSVD_loss, opt.tSVD_mode, opt.n_views, opt.aux_iter = "True", "pathomic", 4, 1
    opt.mu, opt.pho, opt.max_mu, opt.Lambda_global = 0.01, 1.5, 1.0, 0.05
    model = m.define_net(opt, 1); ema = m.define_net(opt, 1)
    model.load_state_dict(W.make_state_dict(W.teacher_shapes(320), 3)); ema.load_state_dict(W.make_state_dict(W.teacher_shapes(320), 4))
    st = m.TeacherStage1Step(opt, device="cuda", models=(model.cuda(), ema.cuda()))
    bt = synthetic_batch(B, 64, seed=5)
    z = torch.zeros(B)
    out = st.step(((bt["x_path"], bt["ema_x_path"]), z, bt["x_omic"], z, z, bt["grade"], bt["index"], bt["sample_idx"]))
    assert torch.isfinite(out["loss"]).all() and torch.isfinite(out["loss_tsvd"]).all()
    for adj, aux, tnn in ((st.adj_tensor1, st.aux_tensor1, st.path_TNN), (st.adj_tensor2, st.aux_tensor2, st.omic_TNN)):
        stack = torch.stack([a.detach().cpu() for a in adj], dim=2)
        ref, tnn_ref = OV.update_aux(stack, opt.Lambda_global / opt.mu)
        got = torch.stack([a.cpu() for a in aux], dim=2).numpy()
        assert np.abs(got - ref).max() <= 2e-5 * max(np.abs(ref).max(), 1.0)
        assert abs(float(tnn) - tnn_ref) <= 1e-4 * max(abs(tnn_ref), 1.0)


def test_fused_loss_head_equals_generic_autograd_path():
    """loss_head.py computes the five per-loss gradients, the GK-Refine weights, the total loss and every parameter
    gradient of the loss block in closed form; the generic path builds five autograd graphs (AEKD_loss).  Same inputs,
    parity mode: losses / weights within 1e-5 relative, every parameter gradient and the post-Adam weights alike."""
    import multimodal_learning_amd as m
    from oracle.step import default_opt, synthetic_batch
    m.set_precision("bf16x6")
    try:
        res = {}
        ranks = [np.random.RandomState(i).choice(np.arange(30, 100), 20, replace=False) for i in range(2)]
        bt = synthetic_batch(8, 64, seed=21)
        for fused in (True, False):
            opt = default_opt()
            opt.fused_loss_head = fused
            step = _mk_step(opt, 1024, seed=0)
            assert step._fused_head_ok() == fused
            out = step.step(_tuple(bt), ranks=ranks)
            P = dict(step.module_list.named_parameters())
            res[fused] = dict(out={k: out[k].detach().float().clone() for k in ("loss", "loss_cls", "loss_div1", "loss_div2",
                                                                             "loss_kd1", "loss_kd2", "scale", "logit_path")},
                              grads={k: p.grad.detach().clone() for k, p in P.items() if p.grad is not None},
                              w={k: p.detach().clone() for k, p in P.items()},
                              bank=step.criterion_kd.contrast.memory_v1.clone())
        a, b = res[True], res[False]
        for k in a["out"]:
            d = (a["out"][k] - b["out"][k]).abs().max().item()
            assert d <= 1e-5 * max(1.0, b["out"][k].abs().max().item()), (k, d)
        assert set(a["grads"]) == set(b["grads"])
        worst = ("", 0.0)
        for k in b["grads"]:
            ga, gb = a["grads"][k], b["grads"][k]
            # (absolute floor: a bias in front of a train-mode BatchNorm has a mathematically zero gradient - noise only)
            d = max((ga - gb).abs().max().item() - 1e-6, 0.0) / (gb.abs().max().item() + 1e-12)
            if d > worst[1]:
                worst = (k, d)
            assert d <= 2e-4, (k, d)
        assert torch.equal(a["bank"], b["bank"])
        print(f"\nfused vs generic loss head: worst relative gradient difference {worst[1]:.2e} at {worst[0]}")
    finally:
        m.set_precision("bf16")


@pytest.mark.parametrize("grads_thresh", ["False", "True"])
def test_fused_loss_head_equals_generic_autograd_path_mia2022(grads_thresh):
    """The same for the MIA-2022 body (loss_head.py, variant mia2022: v3 bank with the epoch weight on the device, momentum
    GK-Refine through ph_gk_finish_momentum) against its generic autograd path (DistillStep._momentum_gk), two consecutive
    steps so that the moving average of the weights is exercised, plain and binarised (`grads_thresh`) cosine sums."""
    import multimodal_learning_amd as m
    from oracle import weights as W
    from oracle.step import default_opt, synthetic_batch
    from oracle.variants import CRDv3State
    n_data, K = 256, 64
    m.set_precision("bf16x6")
    try:
        res = {}
        for fused in (True, False):
            opt = default_opt(nce_k=K, grads_m=0.9, grads_thresh=grads_thresh, thresh=0.1, niter_decay=10)
            opt.fused_loss_head = fused
            step = m.DistillStep(opt, n_data, device="cuda", variant="mia2022")
            step.model.load_state_dict(W.make_state_dict(W.student_shapes(), 1))
            step.ema_model.load_state_dict(W.make_state_dict(W.student_shapes(), 2))
            step.fix_model.load_state_dict(W.make_state_dict(W.teacher_shapes(320), 3))
            for i, crd in enumerate((step.criterion_kd, step.criterion_kd_path)):
                crd.embed_s.load_state_dict(W.make_state_dict(W.embed_shapes(), 10 + 2 * i))
                crd.embed_t.load_state_dict(W.make_state_dict(W.embed_shapes(), 11 + 2 * i))
                st = CRDv3State(n_data, K=K, seed=20 + i)
                crd.contrast.memory_v1.copy_(st.memory_v1); crd.contrast.memory_v2.copy_(st.memory_v2)
                crd.contrast.verbose = False
            rec = []
            for it in range(2):
                bt = synthetic_batch(8, 64, n_data=n_data, P=1, K=K, seed=400 + it)
                out = step.step(_tuple(bt), epoch=3 + it)
                if it == 0:
                    assert step._fused_head_ok() == fused
                P = dict(step.module_list.named_parameters())
                rec.append(dict(out={k: out[k].detach().float().clone() for k in ("loss", "loss_cls", "loss_div1", "loss_div2", "loss_kd1",
                                                                               "loss_kd2", "scale", "logit_path")},
                                grads={k: p.grad.detach().clone() for k, p in P.items() if p.grad is not None},
                                mo=step._mo_state.clone()))
            res[fused] = rec
        for it in range(2):
            a, b = res[True][it], res[False][it]
            # (step 1 starts from Adam's first, sign-like update of two gradient sets that differ in the last bits: +-lr on
            # near-zero entries, 1e-3 on the logits - it checks the moving average of the weights, not rounding)
            tol = 1e-5 if it == 0 else 1e-2
            for k in a["out"]:
                d = (a["out"][k] - b["out"][k]).abs().max().item()
                assert d <= tol * max(1.0, b["out"][k].abs().max().item()), (it, k, d)
            assert (a["mo"] - b["mo"]).abs().max().item() <= tol * max(1.0, b["mo"].abs().max().item()), (it, a["mo"], b["mo"])
            if it == 0:
                assert set(a["grads"]) == set(b["grads"])
                for k in b["grads"]:
                    ga, gb = a["grads"][k], b["grads"][k]
                    d = max((ga - gb).abs().max().item() - 1e-6, 0.0) / (gb.abs().max().item() + 1e-12)
                    assert d <= 2e-4, (k, d)
    finally:
        m.set_precision("bf16")


@pytest.mark.parametrize("use_thresh", ["True", "False"])
def test_fused_loss_head_equals_generic_autograd_path_mia2023(use_thresh):
    """The same for the MIA-2023 body (loss_head.FusedMia2023LossFn: per-sample KL / CRD rows under the discrepancy weights,
    per-sample GK-Refine weights from the closed-form gradient rows) against its generic autograd path
    (DistillStep._mia2023_tail), with the re-weighting switched on, binarised and ReLU-ed cosine sums."""
    import multimodal_learning_amd as m
    from oracle import weights as W
    from oracle.step import default_opt, synthetic_batch
    from oracle.variants import CRDv10State
    n_data, K, B = 512, 64, 8
    labels = torch.arange(n_data) % 3
    class_idx = [np.nonzero((labels == c).numpy())[0] for c in range(3)]
    m.set_precision("bf16x6")
    try:
        res = {}
        for fused in (True, False):
            opt = default_opt(nce_k=K, nce_p=4, pos_extra="neighbors", neg_mode="all_others", start_reweight=0, discrep_scale=1,
                              max_discrep=2.0, use_grads_thresh=use_thresh, grads_thresh=0.1, loss_weighting="GK_refine",
                              batch_size=B)
            opt.fused_loss_head = fused
            step = m.DistillStep(opt, n_data, device="cuda", variant="mia2023", train_class_idx=class_idx)
            step.model.load_state_dict(W.make_state_dict(W.student_shapes(), 1))
            step.ema_model.load_state_dict(W.make_state_dict(W.student_shapes(), 2))
            step.fix_model.load_state_dict(W.make_state_dict(W.teacher_shapes(320), 3))
            for i, crd in enumerate((step.criterion_kd, step.criterion_kd_path)):
                crd.embed_s.load_state_dict(W.make_state_dict(W.embed_shapes(), 10 + 2 * i))
                crd.embed_t.load_state_dict(W.make_state_dict(W.embed_shapes(), 11 + 2 * i))
                st = CRDv10State(n_data, labels, K=K, seed=20 + i)
                crd.contrast.memory_v1.copy_(st.memory_v1); crd.contrast.memory_v2.copy_(st.memory_v2)
                crd.contrast.verbose = False
            bt = synthetic_batch(B, 64, n_data=n_data, P=1, K=K, seed=500)
            bt["grade"] = labels[bt["index"]].long()
            out = step.step(_tuple(bt), epoch=2)
            assert step._fused_head_ok() == fused
            P = dict(step.module_list.named_parameters())
            res[fused] = dict(out={k: out[k].detach().float().clone() for k in ("loss", "loss_cls", "loss_div1", "loss_div2", "loss_kd1",
                                                                             "loss_kd2", "scale", "logit_path", "rows_div1", "rows_kd1",
                                                                             "w1", "w2")},
                              grads={k: p.grad.detach().clone() for k, p in P.items() if p.grad is not None},
                              bank=step.criterion_kd.contrast.memory_v1.clone())
        a, b = res[True], res[False]
        assert float(b["out"]["w1"].max()) > 1.0
        for k in a["out"]:
            d = (a["out"][k].reshape(-1) - b["out"][k].reshape(-1)).abs().max().item()
            assert d <= 1e-5 * max(1.0, b["out"][k].abs().max().item()), (k, d)
        assert set(a["grads"]) == set(b["grads"])
        for k in b["grads"]:
            ga, gb = a["grads"][k], b["grads"][k]
            d = max((ga - gb).abs().max().item() - 1e-6, 0.0) / (gb.abs().max().item() + 1e-12)
            assert d <= 2e-4, (k, d)
        assert torch.equal(a["bank"], b["bank"])
    finally:
        m.set_precision("bf16")


def test_stage1_step_with_superpixel_masking_terms():
    """MIA-2023 stage-1 batch body (train_test_MT_SP_Masking.py:185-330): with opt.masking the step adds the two
    masked-view consistency terms.  Every component is pinned against the reference on its own (the attention masks
    end to end, pred_KD_loss, the network forwards); here the wiring: the term equals an independent evaluation of
    :198-220 through the module-level API on a copy of the model, three taped forwards of one trunk back-propagate
    (gradients differ from the step without masking), epoch <= start_epoch switches the term off."""
    import copy
    import multimodal_learning_amd as m
    from oracle import weights as W
    from oracle.step import synthetic_batch
    from tests.golden.make_golden_superpixel import label_map
    B, H = 4, 64
    m.set_precision("bf16x6")
    try:
        def build():
            opt = m.stage2_opt(dropout_rate=0.0, batch_size=B, cut_fuse_grad=False, num_teachers=2)
            opt.pred_distill, opt.KD_weight, opt.CRD_distill, opt.SP_distill, opt.orth_loss = 1, 1.0, 0, 0, "False"
            opt.masking, opt.start_epoch, opt.Path_K, opt.Omic_K = 1, 1, 5, 8
            model = m.define_net(opt, 1); ema = m.define_net(opt, 1)
            model.load_state_dict(W.make_state_dict(W.teacher_shapes(320), 3)); ema.load_state_dict(W.make_state_dict(W.teacher_shapes(320), 4))
            return m.TeacherStage1Step(opt, device="cuda", models=(model.cuda(), ema.cuda())), opt
        bt = synthetic_batch(B, H, seed=630)
        gen = torch.Generator().manual_seed(41)
        sp = label_map(B, H, H, 6, gen)
        v1 = bt["x_path"] * (torch.rand(B, 1, H, H, generator=gen) > 0.2)
        v2 = bt["x_path"] * (torch.rand(B, 1, H, H, generator=gen) > 0.2)
        z = torch.zeros(B)
        batch = ((bt["x_path"], sp, bt["ema_x_path"], sp, v1, v2), z, bt["x_omic"], z, z, bt["grade"], bt["index"], bt["sample_idx"])
        st, opt = build()
        # independent evaluation of :198-220 on a copy (the step updates running statistics and weights)
        ref_model, ref_ema = copy.deepcopy(st.model), copy.deepcopy(st.ema_model)
        xp, xo = bt["x_path"].cuda(), bt["x_omic"].cuda()
        pm, om = m.superpixel.superpixel_attention_mask(opt, None, ref_model, xp, z, xo, sp, bt["grade"], "cuda")
        with torch.no_grad():
            p1 = ref_model(x_path=m.superpixel.apply_mask(xp, pm), x_omic=xo)[5]
            p2 = ref_model(x_path=xp, x_omic=m.superpixel.apply_mask(xo, om))[5]
            e1 = ref_ema(x_path=v1.cuda(), x_omic=xo)[5]; e2 = ref_ema(x_path=v2.cuda(), x_omic=xo)[5]
            expect = m.TeacherStage1Step.pred_KD_loss(p1, e1) + m.TeacherStage1Step.pred_KD_loss(p2, e2)
        assert float(pm.sum()) > 0 and float(om.sum()) == B * 8
        out = st.step(batch, epoch=2)
        assert torch.isfinite(out["loss"]) and float(out["loss_pred_KD_masking"]) > 0
        assert abs(float(out["loss_pred_KD_masking"]) - float(expect)) <= 1e-4 * max(abs(float(expect)), 1e-3)
        g_mask = st.optimizer.flat.grad.clone()
        st0, _ = build()
        out0 = st0.step(batch, epoch=1)                      # epoch <= start_epoch: term off
        assert float(out0["loss_pred_KD_masking"]) == 0.0
        assert abs(float(out["loss"]) - float(out0["loss"]) - float(out["loss_pred_KD_masking"])) <= 1e-3 * abs(float(out["loss"]))
        g0 = st0.optimizer.flat.grad
        assert torch.isfinite(g_mask).all() and float((g_mask - g0).abs().max()) > 1e-6 * float(g0.abs().max())
    finally:
        m.set_precision("bf16")


@pytest.mark.parametrize("distill", ["kd", "feats_KL", "rkd", "pkt", "similarity"])
def test_mia2022_distill_baselines(distill):
    """The other `--distill` choices of the MIA-2022 trainer (train_test_path_multi_distill_v2.py:419-486) through
    DistillStep(variant="mia2022"): loss = lambda * CE + alpha * (KL terms) + beta * criterion(student feature, fused teacher
    feature), the criterion being the reference-pinned zoo module; one and two teachers; `kd` also with GK-Refine."""
    import multimodal_learning_amd as m
    from multimodal_learning_amd import distiller_zoo as Z
    from oracle import weights as W
    from oracle.step import default_opt, synthetic_batch
    B, H, n_data = 8, 64, 256
    crit = {"kd": None, "feats_KL": Z.feats_KL, "rkd": Z.RKDLoss, "pkt": Z.PKT, "similarity": Z.Similarity}[distill]
    m.set_precision("bf16x6")
    try:
        for nt, which, aw in ((2, "fuse", "False"), (1, "self_EMA", "False")) + (((2, "fuse", "True"),) if distill == "kd" else ()):
            opt = default_opt(batch_size=B, nce_k=64, distill=distill, num_teachers=nt, which_teacher=which, assign_weights=aw,
                              grads_m=0.9, grads_thresh="False", thresh=0.1, alpha=1.0, beta=0.5)
            step = m.DistillStep(opt, n_data, device="cuda", variant="mia2022")
            step.model.load_state_dict(W.make_state_dict(W.student_shapes(), 1))
            step.ema_model.load_state_dict(W.make_state_dict(W.student_shapes(), 2))
            step.fix_model.load_state_dict(W.make_state_dict(W.teacher_shapes(320), 3))
            assert len(step.module_list) == 1                      # no embedding heads in the optimiser
            bt = synthetic_batch(B, H, n_data=n_data, P=1, K=64, seed=700)
            z = torch.zeros(B)
            out = step.step(((bt["x_path"], bt["ema_x_path"]), z, bt["x_omic"], z, z, bt["grade"], bt["index"], bt["sample_idx"]), epoch=2)
            kd = float(crit()(out["path_feat"], out["fuse_feat"]).reshape(())) if crit is not None else 0.0
            assert abs(float(out["loss_kd1"]) - opt.beta * kd) <= 1e-5 * max(abs(opt.beta * kd), 1e-4)
            if nt == 1:
                assert float(out["loss_div1"]) == 0.0 and float(out["loss_div2"]) > 0.0
            if aw == "False":
                expect = float(out["loss_cls"]) + opt.alpha * (float(out["loss_div1"]) + float(out["loss_div2"])) + opt.beta * kd
                assert abs(float(out["loss"]) - expect) <= 1e-5 * abs(expect), (distill, nt)
                assert out["scale"] is None
            else:
                assert out["scale"].shape == (3,) and torch.isfinite(out["scale"]).all()
            g = step.optimizer.flat.grad
            assert torch.isfinite(g).all() and float(g.abs().max()) > 0
    finally:
        m.set_precision("bf16")


@pytest.mark.parametrize("mode,gtol", [("bf16x6", 1.0), ("bf16x6/x3", 1.0), ("bf16x3", 8.0), ("fp16x3", 1.0), ("fp16x3/x1", 1.0)])
def test_two_steps_from_mid_training_state_vs_reference_golden(golden_dir, mode, gtol):
    """Post-update parity as a REAL check.  tests/golden/make_golden_midstate.py runs the reference
    for two steps from a mid-training state: Adam step count 7 with non-zero moments (per-tensor recipe scaled by the
    reference's own gradient scale), an EMA model of its own, iter_num 7, CRD constants Z already set.  There the Adam
    update is a smooth function of the gradient (not sign(g) as in the first steps from zero moments), so everything is
    asserted at the north-star tolerance: step 0's pre-update quantities AND its update (parameters, EMA, Adam moments,
    bank rows at ~1e-6), and - the point - step 1's logits / losses / GK-Refine weights, which depend on every piece of
    the update path.  The optimiser state enters through FusedAdam.load_state_dict in torch.optim.Adam's own layout
    (what a checkpoint written by the reference holds).

    Three arithmetics (the cheapest one that meets 1e-3): `bf16x6` (six split-bf16 products everywhere),
    `bf16x6/x3` (six in every forward, the three leading ones in the backward's dgrad / wgrad kernels) - held to the SAME
    tolerances, gradients and Adam moments included - and `bf16x3` (three everywhere): its logits and six loss terms are
    held to the same 1e-3, but the GK-Refine weights are cosines of DIFFERENCES of student and teacher probabilities and
    amplify its 3e-4 logit error to 5e-3 relative, which the total loss inherits (mode-specific tolerances below:
    weights 1e-2, total loss 5e-3, gradients / moments 8 x the parity mode's, touched bank rows 1e-4)."""
    import multimodal_learning_amd as m
    from oracle import weights as W
    from oracle.step import default_opt, synthetic_batch
    from tests.gpu_util import Report, maxerr
    g = np.load(os.path.join(golden_dir, "midstate_b8_h96.npz"))
    # fp64 truth of step 0's gradients (tests/golden/make_midstate_fp64.py).  A ReLU network's gradient is discontinuous in
    # its pre-activations: at B = 8 / 96 x 96 ONE activation within ~1e-6 of zero decides ~0.5 % of the first layers' weight
    # gradient, every batch of this size holds a few such elements, and any two fp32-grade evaluations can put one on
    # different sides - the reference's own fp32 run sits 4.9e-3 (conv1) / 5.6e-3 (layer2.0.conv1) of the gradient's scale
    # from the truth for exactly that reason.  A gradient row (and the first moment it feeds, which carries 0.1 x of it)
    # therefore passes when it is within tolerance of the REFERENCE golden, or no farther from the fp64 truth than the
    # reference itself is; both distances are printed.
    t64 = np.load(os.path.join(golden_dir, "midstate_b8_h96_fp64.npz"))
    seed, n_data, t0 = int(g["seed"]), int(g["n_data"]), int(g["t0"])
    m.set_precision(mode)
    try:
        step = _mk_step(default_opt(), n_data, seed=seed)
        # ---- the mid-training state
        names = ["student." + k for k, _ in step.model.named_parameters()]
        for i in range(2):
            names += [f"crd{i}.embed_s.linear.weight", f"crd{i}.embed_s.linear.bias", f"crd{i}.embed_t.linear.weight",
                      f"crd{i}.embed_t.linear.bias"]
        params = list(step.module_list.parameters())
        assert len(names) == len(params)
        scales = dict(zip([str(s) for s in g["scale_names"]], g["scale_values"]))
        trainable = [(n, tuple(p.shape)) for n, p in zip(names, params) if p.requires_grad]
        mom = W.adam_moments(trainable, scales, seed + 30)
        sd = step.optimizer.state_dict()
        sd.pop("fused")
        sd["state"] = {i: dict(step=torch.tensor(float(t0)), exp_avg=mom[n][0], exp_avg_sq=mom[n][1])
                       for i, (n, p) in enumerate(zip(names, params)) if p.requires_grad}
        step.optimizer.load_state_dict(sd)
        assert step.optimizer._step == t0
        assert abs(step.optimizer.param_groups[0]["lr"] - float(g["lr"])) < 1e-12      # LambdaLR's epoch-0 factor on both sides
        step.iter_num = t0
        for crd, key in ((step.criterion_kd, "Z0"), (step.criterion_kd_path, "Z1")):
            crd.contrast.params[2:4] = torch.as_tensor(g[key]).cuda()
            crd.contrast._z_set = True
        R = Report(f"2 distill steps from a mid-training state, arithmetic {mode} vs REFERENCE golden (B=8, 96x96)")
        cut = lambda t: t.detach().reshape(-1)[:4096]      # noqa: E731
        watch = ("conv1.weight", "layer2.0.conv1.weight", "layer4.1.bn2.weight", "fc_new1.0.weight", "fc_new2.weight",
                 "fc_new2.bias")
        named = dict(step.model.named_parameters())
        enamed = dict(step.ema_model.named_parameters())
        flat = step.optimizer.flat
        off = {id(t): o for t, o in zip(flat.tensors, flat.offsets)}
        grads0 = {}
        orig = step.optimizer.step

        def spy(*a, **k):
            if not grads0:
                for kk in watch:
                    grads0[kk] = named[kk].grad.detach().clone()
                grads0["embed_s0"] = step.criterion_kd.embed_s.linear.weight.grad.detach().clone()
            return orig(*a, **k)
        step.optimizer.step = spy
        for it in range(2):
            bt = synthetic_batch(int(g["B"]), int(g["H"]), n_data=n_data, seed=310 + it)
            out = step.step(_tuple(bt), epoch=int(g["epoch"]), ranks=[g["ranks"][2 * it], g["ranks"][2 * it + 1]])
            for k, key in (("logit_path", "logit_path"), ("ema_logit", "ema_logit"), ("fuse_logit", "fuse_logit")):
                R.close(g[f"{key}{it}"], out[k], 1e-3, 0, f"{k} step {it}")
            for k in ("loss_cls", "loss_div1", "loss_div2", "loss_kd1", "loss_kd2", "loss"):
                ref = g[f"{k}{it}"] * (1.0 if k in ("loss_cls", "loss") else (step.opt.alpha if "div" in k else step.opt.beta))
                R.close(ref, out[k], 1e-3, 5e-3 if (k == "loss" and mode == "bf16x3") else 1e-4, f"{k} step {it}")
            R.close(g[f"scale{it}"], out["scale"], 2e-3, 1e-2 if mode == "bf16x3" else 1e-3, f"GK-Refine scale step {it}")
            for k in watch:
                if it == 0:
                    grad_row(R, g["g0_" + k], t64["g0_" + k], cut(grads0[k]), 1e-6, 2e-3 * gtol, f"grad {k}")
                R.close(g[f"p{it}_{k}"], cut(named[k]), 5e-6 * gtol, 0, f"param {k} after step {it}")
                R.close(g[f"e{it}_{k}"], cut(enamed[k]), 5e-6 * gtol, 0, f"EMA {k} after step {it}")
                o = off[id(named[k])]
                n = named[k].numel()
                # (exp_avg carries 0.1 x the fresh gradient: the step-1 gradient of the first layers is the most sensitive
                # quantity of the net - 18 train-mode BatchNorms back-propagated through weights that differ by ~4e-6 - and
                # sits at 4e-3 of its scale, the oracle's CPU run at 2e-3; everything downstream of it stays at 1e-3)
                # (the first moment carries (1 - beta1) = 0.1 x the fresh gradient: + 0.1 x the reference's own gradient distance
                # to the fp64 truth, see above)
                slack = 0.2 * maxerr(np.asarray(t64["g0_" + k]), np.asarray(g["g0_" + k]))[0]
                R.close(g[f"m{it}_{k}"], cut(step.optimizer._m[o:o + n]), 1e-7 + slack, (2e-3 if it == 0 else 6e-3) * gtol,
                        f"exp_avg {k} after step {it}")
                R.close(g[f"v{it}_{k}"], cut(step.optimizer._v[o:o + n]), 1e-10, 2e-3 * gtol, f"exp_avg_sq {k} after step {it}")
            if it == 0:
                R.close(g["g0_embed_s0"], cut(grads0["embed_s0"]), 1e-6, 2e-3 * gtol, "grad embed_s0")
            sdm, sde = step.model.state_dict(), step.ema_model.state_dict()
            R.close(g[f"p_abs_sum{it}"], sum(v.double().abs().sum() for v in sdm.values() if v.dtype.is_floating_point), 0, 2e-6,
                    f"sum|student| after step {it}")
            R.close(g[f"ema_abs_sum{it}"], sum(v.double().abs().sum() for v in sde.values() if v.dtype.is_floating_point), 0, 2e-6,
                    f"sum|EMA| after step {it}")
            R.close(g[f"rm_bn1_{it}"], sdm["bn1.running_mean"], 1e-5, 1e-4, f"bn1 running_mean after step {it}")
            R.close(g[f"rv_l4_{it}"], sdm["layer4.1.bn2.running_var"], 1e-5, 1e-3, f"layer4.1.bn2 running_var after step {it}")
            ix = bt["index"].cuda()
            # (fp16x3: this run's boundary element - see above - moves the step-0 update of the first layers by ~1e-7, which
            # step 1's bank rows see at ~2e-5; the other step-1 quantities keep the parity tolerances)
            btol = 1e-4 if mode == "bf16x3" else (3e-5 if (mode.startswith("fp16x3") and it == 1) else 1e-5)
            R.close(g[f"bank0_v1_rows{it}"], step.criterion_kd.contrast.memory_v1[ix], btol, 0, f"bank0 v1 rows step {it}")
            R.close(g[f"bank1_v2_rows{it}"], step.criterion_kd_path.contrast.memory_v2[ix], btol, 0, f"bank1 v2 rows step {it}")
            R.close(g[f"params0_{it}"], step.criterion_kd.contrast.params, 1e-3, 1e-6, f"CRD params / Z step {it}")
        R.finish()
    finally:
        m.set_precision("bf16")


@pytest.mark.parametrize("variant", ["mia2022", "mia2023"])
def test_variant_steps_from_mid_training_state_vs_reference_golden(golden_dir, variant):
    """The MIA-2022 / MIA-2023 batch bodies (rows a17 / a18) for THREE steps at the north-star tolerance: same fixtures as
    the cold-start tests above but with Adam started from a mid-training state (tests/golden/_warm.py: step count 7,
    moments of the size of the reference's own gradients) - so steps 1 and 2, which carry the momentum GK-Refine weights
    (mia2022), the re-weighting switch and the KNN bank (mia2023), the bank updates, Adam and the EMA across iterations,
    are asserted at 1e-3 instead of against a noise floor."""
    import sys
    import multimodal_learning_amd as m
    from oracle import weights as W
    from oracle.step import default_opt, synthetic_batch
    from oracle.variants import CRDv3State, CRDv10State
    from tests.gpu_util import Report
    sys.path.insert(0, golden_dir)
    import _warm
    g = np.load(os.path.join(golden_dir, f"{variant}_step_warm_b8_h64.npz"))
    B, H, n_data, K = int(g["B"]), int(g["H"]), int(g["n_data"]), int(g["K"])
    kw = {}
    if variant == "mia2022":
        opt = default_opt(nce_k=K, grads_m=float(g["grads_m"]), grads_thresh="False", thresh=0.1, niter_decay=int(g["niter_decay"]))
        epochs, seed0 = [3, 3, 7], 300
    else:
        labels = torch.as_tensor(g["labels"])
        kw["train_class_idx"] = [np.nonzero((labels == c).numpy())[0] for c in range(3)]
        opt = default_opt(nce_k=K, nce_p=int(g["num_pos"]), pos_extra="neighbors", neg_mode="all_others",
                          start_reweight=int(g["start_reweight"]), discrep_scale=1, max_discrep=float(g["max_discrep"]),
                          use_grads_thresh="True", grads_thresh=float(g["grads_thresh"]), loss_weighting="GK_refine", batch_size=B)
        epochs, seed0 = [0, 1, 2], 400
    m.set_precision("bf16x6")
    try:
        step = m.DistillStep(opt, n_data, device="cuda", variant=variant, **kw)
        step.model.load_state_dict(W.make_state_dict(W.student_shapes(), 1))
        step.ema_model.load_state_dict(W.make_state_dict(W.student_shapes(), 2))
        step.fix_model.load_state_dict(W.make_state_dict(W.teacher_shapes(320), 3))
        for i, crd in enumerate((step.criterion_kd, step.criterion_kd_path)):
            crd.embed_s.load_state_dict(W.make_state_dict(W.embed_shapes(), 10 + 2 * i))
            crd.embed_t.load_state_dict(W.make_state_dict(W.embed_shapes(), 11 + 2 * i))
            st = CRDv3State(n_data, K=K, seed=20 + i) if variant == "mia2022" else CRDv10State(n_data, labels, K=K, seed=20 + i)
            crd.contrast.memory_v1.copy_(st.memory_v1); crd.contrast.memory_v2.copy_(st.memory_v2)
            crd.contrast.verbose = False
        names = _warm.param_names(step.model)
        params = list(step.module_list.parameters())
        assert len(names) == len(params)
        _warm.load_fused_adam(step.optimizer, names, params, _warm.unpack_scales(g))
        step.iter_num = int(g["t0"])
        assert abs(step.optimizer.param_groups[0]["lr"] - float(g["lr"])) < 1e-12
        R = Report(f"3 {variant} distill steps from a mid-training optimiser state, parity mode vs REFERENCE golden (B=8, 64x64)")
        keys = [("logit_path", "logit_path"), ("ema_logit", "ema_logit"), ("fuse_logit", "fuse_logit"), ("loss_cls", "loss_cls"),
                ("loss_div1_", "loss_div1"), ("loss_div2_", "loss_div2"), ("loss_kd1_", "loss_kd1"), ("loss_kd2_", "loss_kd2"),
                ("loss", "loss")]
        if variant == "mia2023":
            keys += [("w1_", "w1"), ("w2_", "w2"), ("rows_div1_", "rows_div1"), ("rows_kd1_", "rows_kd1")]
        for it in range(3):
            bt = synthetic_batch(B, H, n_data=n_data, P=1, K=K, seed=seed0 + it)
            if variant == "mia2023":
                bt["grade"] = labels[bt["index"]]
            out = step.step(_tuple(bt), epoch=epochs[it])
            sd, esd = step.model.state_dict(), step.ema_model.state_dict()
            idx = bt["index"].cuda()
            for key, name in keys:
                R.close(g[f"{key}{it}"], out[name], 1e-3, 1e-4, f"{name} step {it}")
            # mia2023: the per-sample GK weights are 0/1 decisions on cosines (thresholded): compared as decisions
            R.close(g[f"scale{it}"], out["scale"], 2e-3 if variant == "mia2022" else 1e-6, 1e-3 if variant == "mia2022" else 0,
                    f"GK-Refine weights step {it}")
            R.close(g[f"p_fc2_{it}"], sd["fc_new2.weight"], 5e-6, 0, f"Adam-updated fc2 step {it}")
            R.close(g[f"ema_fc2_{it}"], esd["fc_new2.weight"], 5e-6, 0, f"EMA fc2 step {it}")
            R.close(g[f"bank0_v1_rows{it}"], step.criterion_kd.contrast.memory_v1[idx], 1e-4, 0, f"bank0 rows step {it}")
            R.close(g[f"bank1_v2_rows{it}"], step.criterion_kd_path.contrast.memory_v2[idx], 1e-4, 0, f"bank1 rows step {it}")
            R.close(g[f"params0_{it}"], step.criterion_kd.contrast.params, 1e-2, 1e-5, f"CRD params / Z step {it}")
        R.finish()
    finally:
        m.set_precision("bf16")


@pytest.mark.gpu
def test_multi_stream_step_equals_one_stream_step_and_phase_markers_are_ordered():
    """The replayed step forks streams (two teacher forwards, the second CRD chain and the heads' weight gradients, the
    trunk's weight gradients).  Scheduling must not change values: against the same step on ONE stream (`overlap_teachers`
    / `overlap_head` off, `_no_bwd_overlap`) the first steps agree bitwise (the weight-gradient slabs are summed from half as
    many chunks beside the BatchNorm passes: later steps see last-bit differences), and two multi-stream runs are bitwise
    identical (a missing dependency would show as run-to-run noise).  The ph_prof_stamp markers inside the graph come back
    in dependency order."""
    import multimodal_learning_amd as m
    from oracle.step import default_opt, synthetic_batch
    m.set_precision("bf16")
    bts = [synthetic_batch(8, 96, seed=50 + i) for i in range(2)]
    rng = np.random.RandomState(5)
    ranks = [[rng.choice(np.arange(30, 100), 20, replace=False) for _ in range(2)] for _ in range(6)]

    def run(streams):
        opt = default_opt()
        opt.overlap_teachers, opt.overlap_head = streams, streams
        step = _mk_step(opt, 1024, seed=0)
        step.model._no_bwd_overlap = not streams
        if streams:
            step._stamps = torch.zeros(16, dtype=torch.int64, device="cuda")
        step.enable_graph()
        losses = []
        for i in range(6):
            out = step.step(_tuple(bts[i % 2]), epoch=1, ranks=ranks[i])
            losses.append(out["loss"].item())
        torch.cuda.synchronize()
        sd = {k: v.clone() for k, v in step.model.state_dict().items()}
        return losses, sd, step.criterion_kd.contrast.memory_v1.clone(), step

    la, sa, ma, _ = run(False)
    lb, sb, mb, step = run(True)
    lc, sc, mc, _ = run(True)
    assert step._slots and step._slots[0]["graph"] is not None, "the multi-stream step was not captured"
    # no race: the multi-stream trajectory is bitwise reproducible
    assert lb == lc and torch.equal(mb, mc)
    for k in sb:
        assert torch.equal(sb[k], sc[k]), k
    # against one stream: the first two steps bitwise (same forward and loss head; Adam's first update is +-lr whatever the
    # last bits of a gradient), then the two trajectories drift apart the way any two summation orders do on this
    # untrained bf16 network (cf. test_three_steps_vs_reference_golden: cold-start steps are chaotic)
    assert la[:2] == lb[:2], (la, lb)
    # (from the third step on only a sanity bound: the weight gradients of step 1 differ in their last bits - other chunking - and on
    # this cold-start trajectory that is worth anything from 1e-4 to several per cent of the third loss depending on which kernels round
    # where; round 6's stride-2 forward kernel moved it from 3e-4 to 7e-2 with both arms bitwise reproducible and equal to each other
    # through step 1)
    assert all(np.isfinite(la)) and all(np.isfinite(lb)) and abs(la[2] - lb[2]) <= 0.25 * abs(la[2]), (la, lb)
    t = step._stamps.cpu().numpy().astype(np.int64)
    assert (t[[0, 1, 2, 3, 4, 5, 6, 8, 9]] > 0).all()
    assert t[0] <= t[1] <= t[2] <= t[3] <= t[4] <= t[5] <= t[6]      # pack, student fwd, join, head fwd, head bwd, bwd, Adam
    assert t[0] <= t[8] <= t[2] and t[0] <= t[9] <= t[2]              # both teachers finish before the join


BRANCH_NAMES = ["t1_fuse_crd", "t1_ema_crd", "t1_fuse_kd", "t1_ema_kd", "t2_kd_gk", "t2_kd_sum", "t2_crd_sum"]


@pytest.mark.parametrize("name", BRANCH_NAMES)
@pytest.mark.parametrize("mode", ["bf16x6", "fp16x3/x1"])
def test_option_branches_vs_reference_golden(golden_dir, name, mode):
    """The headline trainer's non-default option branches (train_test_path_multi_distill.py:263-309) through
    DistillStep: `--num_teachers 1` with the fused or the mean teacher, `--distill kd` (with GK-Refine over the two KL terms, or
    fixed weights), `--assign_weights False` - against the REFERENCE run under those options (tests/golden/make_golden_branches.py:
    two steps from a mid-training Adam state).  Step 1 is replayed from a captured graph in the tolerance-compliant arithmetic
    (the branch bodies hold no host synchronisation).  Also: the parameters of a criterion the branch leaves unused do not move
    (the reference's optimiser skips parameters whose .grad is None - no weight decay either)."""
    import sys
    import multimodal_learning_amd as m
    from oracle.step import synthetic_batch
    from tests.gpu_util import Report
    from tests.test_oracle_golden import branch_opt, BRANCH_OPTS
    sys.path.insert(0, golden_dir)
    import _warm
    g = np.load(os.path.join(golden_dir, "branches_b8_h64.npz"))
    n_data, t0 = int(g["n_data"]), int(g["t0"])
    m.set_precision(mode)
    try:
        opt = branch_opt(name, batch_size=int(g["B"]))
        step = _mk_step(opt, n_data, seed=0)
        names = _warm.param_names(step.model)
        params = list(step.module_list.parameters())
        assert len(names) == len(params)
        _warm.load_fused_adam(step.optimizer, names, params, _warm.unpack_scales(g))
        assert step.optimizer._step == t0
        assert abs(step.optimizer.param_groups[0]["lr"] - float(g["lr"])) < 1e-12      # LambdaLR's epoch-0 factor on both sides
        step.iter_num = t0
        e_s0 = step.criterion_kd.embed_s.linear.weight.detach().clone()
        e_t1 = step.criterion_kd_path.embed_t.linear.weight.detach().clone()
        ranks = g[name + ".ranks"]
        per = 0 if len(ranks) == 0 else len(ranks) // 2
        R = Report(f"option branch {name} ({BRANCH_OPTS[name]}), arithmetic {mode} vs REFERENCE golden (B=8, 64x64)")
        pre = name + "."
        grads = {}
        orig = step.optimizer.step

        def spy(*a, **k):
            grads["fc2"] = step.model.fc_new2.weight.grad.detach().clone()
            grads["l4"] = step.model.layer4[1].conv2.weight.grad.detach().abs().sum()
            return orig(*a, **k)
        step.optimizer.step = spy
        for it in range(2):
            bt = synthetic_batch(int(g["B"]), int(g["H"]), n_data=n_data, seed=500 + it)
            rk = list(ranks[per * it: per * (it + 1)]) if per else None
            out = step.step(_tuple(bt), epoch=3 + it, ranks=rk)
            idx = bt["index"].cuda()
            R.close(g[pre + f"logit_path{it}"], out["logit_path"], 1e-3, 0, f"logit_path step {it}")
            R.close(g[pre + f"loss_cls{it}"], out["loss_cls"], 1e-3, 1e-4, f"loss_cls step {it}")
            R.close(float(g[pre + f"loss_div{it}"]) * opt.alpha, out["loss_div1"] + out["loss_div2"], 1e-3, 1e-4, f"loss_div step {it}")
            R.close(float(g[pre + f"loss_kd{it}"]) * opt.beta, out["loss_kd1"] + out["loss_kd2"], 1e-3, 1e-4, f"loss_kd step {it}")
            R.close(g[pre + f"loss{it}"], out["loss"], 1e-3, 1e-4, f"loss step {it}")
            if pre + f"scale{it}" in g:
                R.close(g[pre + f"scale{it}"], out["scale"], 2e-3, 1e-3, f"GK-Refine scale step {it}")
            else:
                assert out["scale"] is None
            R.close(g[pre + f"g_fc2_{it}"], grads["fc2"], 1e-6, 2e-3, f"grad fc2 step {it}")
            R.close(g[pre + f"g_l4_1_conv2_abs{it}"], grads["l4"], 1e-4, 3e-3, f"grad l4.1.conv2 |.|_1 step {it}")
            # (lr = 5e-4: a wrong update is ~1e-4; step 1's update inherits step 1's 6e-4 logit distance)
            ptol = 5e-6 if it == 0 else 2e-5
            R.close(g[pre + f"p_fc2_{it}"], step.model.fc_new2.weight, ptol, 0, f"param fc2 after step {it}")
            R.close(g[pre + f"ema_fc2_{it}"], step.ema_model.fc_new2.weight, ptol, 0, f"EMA fc2 after step {it}")
            R.close(g[pre + f"embed_s0_{it}"], step.criterion_kd.embed_s.linear.weight[:8], ptol, 0, f"embed_s (kd) after step {it}")
            R.close(g[pre + f"embed_t1_{it}"], step.criterion_kd_path.embed_t.linear.weight[:8], ptol, 0, f"embed_t (kd_path) after step {it}")
            R.close(g[pre + f"bank0_v1_rows{it}"], step.criterion_kd.contrast.memory_v1[idx], 1e-4, 0, f"bank0 rows step {it}")
            R.close(g[pre + f"bank1_v2_rows{it}"], step.criterion_kd_path.contrast.memory_v2[idx], 1e-4, 0, f"bank1 rows step {it}")
            R.close(g[pre + f"params0_{it}"], step.criterion_kd.contrast.params, 1e-2, 1e-5, f"CRD params / Z (kd) step {it}")
            R.close(g[pre + f"params1_{it}"], step.criterion_kd_path.contrast.params, 1e-2, 1e-5, f"CRD params / Z (kd_path) step {it}")
            skipped = set(str(s) for s in g[pre + f"no_grad{it}"])
            if "crd0.embed_s.linear.weight" in skipped:
                assert torch.equal(step.criterion_kd.embed_s.linear.weight, e_s0)
            if "crd1.embed_t.linear.weight" in skipped:
                assert torch.equal(step.criterion_kd_path.embed_t.linear.weight, e_t1)
        R.finish()
    finally:
        m.set_precision("bf16")


def test_option_branch_replays_from_a_captured_graph():
    """A branch body (one teacher, CRD through criterion_kd) captured in a HIP graph gives the eager steps' numbers."""
    import multimodal_learning_amd as m
    from oracle.step import default_opt, synthetic_batch
    m.set_precision("bf16")
    opt = default_opt(num_teachers=1, which_teacher="self_EMA", assign_weights="False", batch_size=8)
    outs = []
    for graph in (False, True):
        torch.manual_seed(3)
        step = _mk_step(opt, 1024, seed=0)
        if graph:
            step.enable_graph()
        rk = [np.arange(30, 50)]
        losses = []
        for it in range(5):
            bt = synthetic_batch(8, 64, seed=40 + it % 2)
            losses.append(step.step(_tuple({k: v.cuda() for k, v in bt.items()}), epoch=1, ranks=rk)["loss"].clone())
        torch.cuda.synchronize()
        if graph:
            assert step._want_graph and step._slots and step._slots[0]["graph"] is not None
        outs.append(torch.stack(losses).cpu())
    assert torch.equal(outs[0], outs[1]), (outs[0], outs[1])


def test_one_teacher_with_gk_refine_raises_like_the_reference():
    """`--num_teachers 1 --assign_weights True`: KD_loss_list is unbound in the reference's batch body (:293-304)."""
    import multimodal_learning_amd as m
    from oracle.step import default_opt
    with pytest.raises(UnboundLocalError):
        m.DistillStep(default_opt(num_teachers=1, which_teacher="fuse", assign_weights="True"), 64, device="cuda")
    with pytest.raises(UnboundLocalError):
        m.DistillStep(default_opt(num_teachers=1, which_teacher="nobody", assign_weights="False"), 64, device="cuda")


MIA_BRANCHES = [("mia2022", "t1_fuse_crd"), ("mia2022", "t1_ema_crd"), ("mia2023", "t1_fuse_crd"), ("mia2023", "t1_ema_crd"),
                ("mia2023", "t1_fuse_kd"), ("mia2023", "t2_kd_gk")]
MIA_BRANCH_OPTS = {"t1_fuse_crd": (1, "fuse", "crd", "False"), "t1_ema_crd": (1, "self_EMA", "crd", "False"),
                   "t1_fuse_kd": (1, "fuse", "kd", "False"), "t2_kd_gk": (2, "fuse", "kd", "True")}


@pytest.mark.parametrize("variant,name", MIA_BRANCHES)
def test_mia_option_branches_vs_reference_golden(golden_dir, variant, name):
    """The one-teacher and `--distill kd` branches of the MIA-2022 / MIA-2023 batch bodies ("MIA 2022/
    train_test_path_multi_distill_v2.py":419-482, "MIA 2023/stage2_unimodal_student/train_test_path_multi_distill.py":348-427)
    through DistillStep(variant=...) against each REFERENCE run under those options (tests/golden/make_golden_branches_mia.py:
    two steps from a mid-training Adam state): logits, loss terms, per-sample GK-Refine weights (MIA-2023 kd), gradients, updated
    parameters, EMA, bank rows - and the unused criterion's heads stay untouched."""
    import sys
    import multimodal_learning_amd as m
    from oracle import weights as W
    from oracle.step import default_opt, synthetic_batch
    from oracle.variants import CRDv3State, CRDv10State
    from tests.gpu_util import Report
    sys.path.insert(0, golden_dir)
    import _warm
    g = np.load(os.path.join(golden_dir, "branches_%s.npz" % variant))
    B, H, n_data, K, t0 = int(g["B"]), int(g["H"]), int(g["n_data"]), int(g["K"]), int(g["t0"])
    nt, wt, distill, aw = MIA_BRANCH_OPTS[name]
    kw = dict(num_teachers=nt, which_teacher=wt, distill=distill, assign_weights=aw, nce_k=K, batch_size=B)
    if variant == "mia2022":
        opt = default_opt(grads_m=float(g["grads_m"]), grads_thresh="False", thresh=0.1, niter_decay=int(g["niter_decay"]), **kw)
        extra = {}
    else:
        labels = torch.as_tensor(g["labels"])
        class_idx = [np.nonzero((labels == c).numpy())[0] for c in range(3)]
        opt = default_opt(nce_p=int(g["num_pos"]), pos_extra="neighbors", neg_mode="all_others",
                          start_reweight=int(g["start_reweight"]), discrep_scale=1, max_discrep=float(g["max_discrep"]),
                          use_grads_thresh="True", grads_thresh=float(g["grads_thresh"]), loss_weighting="GK_refine", **kw)
        extra = dict(train_class_idx=class_idx)
    m.set_precision("bf16x6")
    try:
        step = m.DistillStep(opt, n_data, device="cuda", variant=variant, **extra)
        step.model.load_state_dict(W.make_state_dict(W.student_shapes(), 1))
        step.ema_model.load_state_dict(W.make_state_dict(W.student_shapes(), 2))
        step.fix_model.load_state_dict(W.make_state_dict(W.teacher_shapes(320), 3))
        for i, crd in enumerate((step.criterion_kd, step.criterion_kd_path)):
            crd.embed_s.load_state_dict(W.make_state_dict(W.embed_shapes(), 10 + 2 * i))
            crd.embed_t.load_state_dict(W.make_state_dict(W.embed_shapes(), 11 + 2 * i))
            st = CRDv3State(n_data, K=K, seed=20 + i) if variant == "mia2022" else CRDv10State(n_data, labels, K=K, seed=20 + i)
            crd.contrast.memory_v1.copy_(st.memory_v1); crd.contrast.memory_v2.copy_(st.memory_v2)
            crd.contrast.verbose = False
        names = _warm.param_names(step.model)
        params = list(step.module_list.parameters())
        assert len(names) == len(params)
        _warm.load_fused_adam(step.optimizer, names, params, _warm.unpack_scales(g))
        assert step.optimizer._step == t0 and abs(step.optimizer.param_groups[0]["lr"] - float(g["lr"])) < 1e-12
        step.iter_num = t0
        e_t1 = step.criterion_kd_path.embed_t.linear.weight.detach().clone()
        R = Report(f"{variant} option branch {name} vs REFERENCE golden (B=8, 64x64)")
        pre = name + "."
        grads = {}
        orig = step.optimizer.step

        def spy(*a, **k):
            grads["fc2"] = step.model.fc_new2.weight.grad.detach().clone()
            grads["l4"] = step.model.layer4[1].conv2.weight.grad.detach().abs().sum()
            return orig(*a, **k)
        step.optimizer.step = spy
        for it in range(2):
            bt = synthetic_batch(B, H, n_data=n_data, P=1, K=K, seed=(600 if variant == "mia2022" else 700) + it)
            if variant == "mia2023":
                bt["grade"] = labels[bt["index"]]
            out = step.step(_tuple(bt), epoch=int(g[pre + f"epoch{it}"]))
            idx = bt["index"].cuda()
            R.close(g[pre + f"logit_path{it}"], out["logit_path"], 1e-3, 0, f"logit_path step {it}")
            R.close(g[pre + f"loss_cls{it}"], out["loss_cls"], 1e-3, 1e-4, f"loss_cls step {it}")
            R.close(float(g[pre + f"loss_div{it}"]) * opt.alpha, out["loss_div1"] + out["loss_div2"], 1e-3, 1e-4, f"loss_div step {it}")
            R.close(float(g[pre + f"loss_kd{it}"]) * opt.beta, out["loss_kd1"] + out["loss_kd2"], 1e-3, 1e-4, f"loss_kd step {it}")
            R.close(g[pre + f"loss{it}"], out["loss"], 1e-3, 1e-4, f"loss step {it}")
            if pre + f"scale{it}" in g:
                R.close(g[pre + f"scale{it}"], out["scale"], 2e-3, 1e-3, f"GK-Refine scale step {it}")
            else:
                assert out["scale"] is None
            ptol = 5e-6 if it == 0 else 2e-5
            R.close(g[pre + f"g_fc2_{it}"], grads["fc2"], 1e-6, 2e-3, f"grad fc2 step {it}")
            R.close(g[pre + f"g_l4_1_conv2_abs{it}"], grads["l4"], 1e-4, 3e-3, f"grad l4.1.conv2 |.|_1 step {it}")
            R.close(g[pre + f"p_fc2_{it}"], step.model.fc_new2.weight, ptol, 0, f"param fc2 after step {it}")
            R.close(g[pre + f"ema_fc2_{it}"], step.ema_model.fc_new2.weight, ptol, 0, f"EMA fc2 after step {it}")
            R.close(g[pre + f"embed_s0_{it}"], step.criterion_kd.embed_s.linear.weight[:8], ptol, 0, f"embed_s (kd) after step {it}")
            R.close(g[pre + f"embed_t1_{it}"], step.criterion_kd_path.embed_t.linear.weight[:8], ptol, 0, f"embed_t (kd_path) after step {it}")
            R.close(g[pre + f"bank0_v1_rows{it}"], step.criterion_kd.contrast.memory_v1[idx], 1e-4, 0, f"bank0 rows step {it}")
            R.close(g[pre + f"bank1_v2_rows{it}"], step.criterion_kd_path.contrast.memory_v2[idx], 1e-4, 0, f"bank1 rows step {it}")
            R.close(g[pre + f"params0_{it}"], step.criterion_kd.contrast.params, 1e-2, 1e-5, f"CRD params / Z (kd) step {it}")
            skipped = set(str(s) for s in g[pre + f"no_grad{it}"])
            assert "crd1.embed_t.linear.weight" in skipped and torch.equal(step.criterion_kd_path.embed_t.linear.weight, e_t1)
        R.finish()
    finally:
        m.set_precision("bf16")
