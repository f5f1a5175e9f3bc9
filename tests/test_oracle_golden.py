"""Pins the CPU oracle (oracle/) against golden vectors produced by RUNNING the reference
(tests/golden/make_golden.py).  CPU-only; this is what makes the oracle trustworthy as the
checker for the HIP path."""
import os

import numpy as np
import pytest
import torch

import oracle
from oracle import weights as W
from oracle.step import DistillOracle, synthetic_batch, default_opt
from oracle.losses import CRDState, crd_loss, distill_kl


def _ld(golden_dir, name):
    return {k: v for k, v in np.load(os.path.join(golden_dir, name)).items()}


def _close(a, b, tol=1e-5, rel=1e-4):
    a = torch.as_tensor(np.asarray(a), dtype=torch.float64)
    b = torch.as_tensor(np.asarray(b.detach() if isinstance(b, torch.Tensor) else b), dtype=torch.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    err = (a - b).abs().max().item() if a.numel() else 0.0
    assert err <= tol + rel * a.abs().max().item(), f"max err {err} (ref max {a.abs().max().item()})"


def test_student_forward_backward(golden_dir):
    g = _ld(golden_dir, "modules_b4_h64.npz")
    bt = synthetic_batch(int(g["B"]), int(g["H"]), seed=int(g["batch_seed"]))
    sd = W.make_state_dict(W.student_shapes(), 1)
    params = {k: v.requires_grad_(True) for k, v in sd.items()
              if v.dtype.is_floating_point and "running" not in k and "output_" not in k}
    x = bt["x_path"].clone().requires_grad_(True)
    f3, feat, hazard, pred, none = oracle.resnet_forward(x, sd)
    assert none is None
    _close(g["f3"], f3); _close(g["feat"], feat); _close(g["hazard"], hazard); _close(g["pred"], pred)
    loss = (feat * torch.linspace(0.5, 1.5, 128)).sum() + (hazard * torch.tensor([1.0, -2.0, 0.5])).sum() \
        + 0.1 * f3.sum()
    loss.backward()
    _close(g["dx"], x.grad, 1e-6, 1e-3)
    _close(g["g_conv1"], sd["conv1.weight"].grad, 1e-5, 1e-3)
    _close(g["g_bn1_w"], sd["bn1.weight"].grad, 1e-5, 1e-3)
    _close(g["g_l1_0_conv1"], sd["layer1.0.conv1.weight"].grad, 1e-5, 1e-3)
    _close(g["g_l2_0_ds"], sd["layer2.0.downsample.0.weight"].grad, 1e-5, 1e-3)
    _close(g["g_l4_1_bn2_w"], sd["layer4.1.bn2.weight"].grad, 1e-5, 1e-3)
    _close(g["g_fc1_w"], sd["fc_new1.0.weight"].grad, 1e-5, 1e-3)
    _close(g["g_fc2_w"], sd["fc_new2.weight"].grad, 1e-5, 1e-3)
    _close(g["g_l3_1_conv2_abs"], sd["layer3.1.conv2.weight"].grad.abs().sum(), 1e-4, 1e-3)
    _close(g["rm_bn1"], sd["bn1.running_mean"]); _close(g["rv_bn1"], sd["bn1.running_var"])
    _close(g["rm_l4"], sd["layer4.1.bn2.running_mean"]); _close(g["rv_l4"], sd["layer4.1.bn2.running_var"])


def test_teacher_forward(golden_dir):
    g = _ld(golden_dir, "modules_b4_h64.npz")
    bt = synthetic_batch(int(g["B"]), int(g["H"]), seed=int(g["batch_seed"]))
    sd = W.make_state_dict(W.teacher_shapes(320), 3)
    with torch.no_grad():
        t = oracle.pathomic_forward(bt["x_path"], bt["x_omic"], sd)
    assert len(t) == 11 and t[8] is None and t[9] is None and t[10] is None
    _close(g["t_fuse"], t[0]); _close(g["t_path_vec"], t[1]); _close(g["t_omic_vec"], t[2])
    _close(g["t_f3"], t[3]); _close(g["t_h_path"], t[4][0]); _close(g["t_h_omic"], t[4][1])
    _close(g["t_h_fuse"], t[4][2]); _close(g["t_pred"], t[5]); _close(g["t_pred_path"], t[6])
    _close(g["t_pred_omic"], t[7])
    sd2 = W.make_state_dict(W.teacher_shapes(320), 3)
    with torch.no_grad():
        # the golden omic/fusion calls ran after the teacher forward (BN running stats differ, outputs do not)
        om = oracle.maxnet_forward(bt["x_omic"], sd2, "omic_net.")
        fo = oracle.bilinear_fusion_forward(torch.as_tensor(g["fus_in1"]), torch.as_tensor(g["fus_in2"]), sd2)
    _close(g["omic_feat"], om[0]); _close(g["omic_out"], om[1]); _close(g["omic_pred"], om[2])
    _close(g["fus_out"], fo)


def test_distill_kl(golden_dir):
    g = _ld(golden_dir, "modules_b4_h64.npz")
    for T in (1, 4):
        ys = torch.as_tensor(g["kl_ys"]).requires_grad_(True)
        kl = distill_kl(ys, torch.as_tensor(g["kl_yt"]), float(T))
        gr, = torch.autograd.grad(kl, ys)
        _close(g[f"kl_T{T}"], kl, 1e-6); _close(g[f"kl_T{T}_g"], gr, 1e-6)


@pytest.mark.parametrize("mode", ["mid", "hard"])
def test_crd_loss(golden_dir, mode):
    g = _ld(golden_dir, f"crd_{mode}.npz")
    st = CRDState(1024, seed=int(g["bank_seed"]),
                  embed_s=W.make_state_dict(W.embed_shapes(), 10),
                  embed_t=W.make_state_dict(W.embed_shapes(), 11))
    for d in (st.embed_s, st.embed_t):
        for v in d.values():
            v.requires_grad_(True)
    for it in range(2):
        f_s = torch.as_tensor(g[f"f_s{it}"]).requires_grad_(True)
        ranks = g["ranks"][it] if mode == "mid" else None
        loss = crd_loss(st, f_s, torch.as_tensor(g[f"f_t{it}"]), torch.as_tensor(g[f"index{it}"]),
                        torch.as_tensor(g[f"sidx{it}"]), 20, 512, mode, ranks)
        gs = torch.autograd.grad(loss, [f_s, st.embed_s["linear.weight"], st.embed_t["linear.weight"],
                                        st.embed_s["linear.bias"]])
        _close(g[f"loss{it}"], loss, 1e-5, 1e-5)
        _close(g[f"g_fs{it}"], gs[0], 1e-6, 1e-3); _close(g[f"g_ws{it}"], gs[1], 1e-6, 1e-3)
        _close(g[f"g_wt{it}"], gs[2], 1e-6, 1e-3); _close(g[f"g_bs{it}"], gs[3], 1e-6, 1e-3)
        _close(g[f"params{it}"], st.params, 1e-2, 1e-5)
        idx = torch.as_tensor(g[f"index{it}"])
        _close(g[f"bank_v1_rows{it}"], st.memory_v1[idx], 1e-6); _close(g[f"bank_v2_rows{it}"], st.memory_v2[idx], 1e-6)


@pytest.mark.parametrize("mode", ["mid", "hard"])
def test_crd_loss_sample_kd(golden_dir, mode):
    """`--sample_KD True` inside CRDLoss (CRD_loss.py:148-149 -> :246-250): the [B] per-sample losses, two calls."""
    g = _ld(golden_dir, "crd_samplekd.npz")
    st = CRDState(int(g["n_data"]), seed=int(g["bank_seed"]), embed_s=W.make_state_dict(W.embed_shapes(), 10),
                  embed_t=W.make_state_dict(W.embed_shapes(), 11))
    for d in (st.embed_s, st.embed_t):
        for v in d.values():
            v.requires_grad_(True)
    for it in range(2):
        t = f"{mode}{it}"
        f_s = torch.as_tensor(g[f"f_s_{t}"]).requires_grad_(True)
        ranks = g[f"ranks_{mode}"][it] if mode == "mid" else None
        rows = crd_loss(st, f_s, torch.as_tensor(g[f"f_t_{t}"]), torch.as_tensor(g[f"index_{t}"]), torch.as_tensor(g[f"sidx_{t}"]),
                        int(g["nce_p2"]), int(g["nce_k2"]), mode, ranks, sample_KD="True")
        assert rows.shape == (f_s.shape[0],)
        gs = torch.autograd.grad((rows * torch.as_tensor(g[f"w_{t}"])).sum(),
                                 [f_s, st.embed_s["linear.weight"], st.embed_t["linear.weight"], st.embed_s["linear.bias"]])
        _close(g[f"rows_{t}"], rows, 1e-4, 1e-5)
        _close(g[f"g_fs_{t}"], gs[0], 1e-5, 1e-3); _close(g[f"g_ws_{t}"], gs[1], 1e-5, 1e-3)
        _close(g[f"g_wt_{t}"], gs[2], 1e-5, 1e-3); _close(g[f"g_bs_{t}"], gs[3], 1e-5, 1e-3)
        _close(g[f"params_{t}"], st.params, 1e-2, 1e-5)
        idx = torch.as_tensor(g[f"index_{t}"])
        _close(g[f"bank_v1_rows_{t}"], st.memory_v1[idx], 1e-6); _close(g[f"bank_v2_rows_{t}"], st.memory_v2[idx], 1e-6)


@pytest.mark.parametrize("faithful", [False, True])
def test_full_step_b16_h224(golden_dir, faithful):
    """Three consecutive steps of train_test_path_multi_distill.py:249-330 vs the reference."""
    g = _ld(golden_dir, "step_b16_h224.npz")
    torch.set_num_threads(8)
    orc = DistillOracle(default_opt(), seed=int(g["seed"]), n_data=int(g["n_data"]))
    nsteps = 3 if not faithful else 1
    for it in range(nsteps):
        bt = synthetic_batch(int(g["B"]), int(g["H"]), seed=100 + it)
        out = orc.step(bt, mid_ranks=[g["ranks"][2 * it], g["ranks"][2 * it + 1]], faithful=faithful)
        tol = 2e-5 * (1 + 20 * it)      # trajectories drift apart slowly (Adam amplifies rounding)
        _close(g[f"logit_path{it}"], out["logit_path"], tol, 1e-4)
        _close(g[f"ema_logit{it}"], out["ema_logit"], tol, 1e-4)
        _close(g[f"fuse_logit{it}"], out["fuse_logit"], tol, 1e-4)
        for k in ("loss_cls", "loss"):
            _close(g[f"{k}{it}"], out[k], tol * 10, 1e-4)
        for k in ("loss_div1", "loss_div2", "loss_kd1", "loss_kd2"):
            _close(g[f"{k}_{it}"], out[k], tol * 10, 1e-4)
        _close(g[f"scale{it}"], out["scale"], 1e-3 * (1 + 5 * it), 1e-3)
        if it == 0:
            _close(g["g0_conv1"], out["grads"]["student.conv1.weight"], 1e-5, 2e-3)
            _close(g["g0_fc2_w"], out["grads"]["student.fc_new2.weight"], 1e-5, 2e-3)
            _close(g["g0_embed_s0"], out["grads"]["crd0.embed_s.linear.weight"], 1e-6, 2e-3)
            _close(g["g0_embed_t1"], out["grads"]["crd1.embed_t.linear.weight"], 1e-6, 2e-3)
        _close(g[f"p_fc2_{it}"], orc.student["fc_new2.weight"], 2e-4 if it else 2e-5, 0)
        _close(g[f"ema_fc2_{it}"], orc.ema["fc_new2.weight"], 2e-4, 0)
        _close(g[f"params0_{it}"], orc.crd[0].params, 1.0, 1e-4)
        _close(g[f"bank0_v1_rows{it}"], orc.crd[0].memory_v1[bt["index"]], 1e-4)
        _close(g[f"bank1_v2_rows{it}"], orc.crd[1].memory_v2[bt["index"]], 1e-4)


def test_eval_mode_forward(golden_dir):
    """module.eval(): running-statistics BatchNorm (the reference's test())."""
    g = _ld(golden_dir, "modules_eval_b4_h96.npz")
    bt = synthetic_batch(int(g["B"]), int(g["H"]), seed=int(g["batch_seed"]))
    from oracle.nets import Rounding
    Rounding.training = False
    try:
        with torch.no_grad():
            f3, feat, hazard, pred, _ = oracle.resnet_forward(bt["x_path"], W.make_state_dict(W.student_shapes(), 1))
            t = oracle.pathomic_forward(bt["x_path"], bt["x_omic"], W.make_state_dict(W.teacher_shapes(320), 3))
    finally:
        Rounding.training = True
    _close(g["f3"], f3); _close(g["feat"], feat); _close(g["hazard"], hazard); _close(g["pred"], pred)
    _close(g["t_fuse"], t[0]); _close(g["t_h_fuse"], t[4][2]); _close(g["t_pred"], t[5]); _close(g["t_pred_omic"], t[7])


def _teacher_bwd_check(named_grads, dx_omic, loss, g, tag, rtol):
    """Compare gradients {name: tensor} with the (partly sampled) fixture entries of one tag."""
    import numpy as np
    assert abs(float(loss) - float(g[f"{tag}_loss"])) <= 1e-4 * abs(float(g[f"{tag}_loss"]))
    ref = g[f"{tag}_dx_omic"]
    assert np.abs(dx_omic - ref).max() <= rtol * max(np.abs(ref).max(), 1e-6)
    n = 0
    for key in g.files:
        if key.startswith(f"{tag}_g_"):
            name = key[len(tag) + 3:]
            got, ref = named_grads[name], g[key]
            # (a Linear bias in front of a BatchNorm has a mathematically zero gradient: 1e-8-level rounding noise)
            assert np.abs(got - ref).max() <= rtol * np.abs(ref).max() + 2e-6, name
            n += 1
        elif key.startswith(f"{tag}_gs_"):
            name = key[len(tag) + 4:]
            flat = named_grads[name].reshape(-1)
            stride = flat.size // 4096
            ref = g[key]
            assert np.abs(flat[::stride][:4096] - ref).max() <= rtol * np.abs(ref).max() + 2e-6, name
            nrm = float(g[f"{tag}_gn_{name}"])
            assert abs(float(np.linalg.norm(flat.astype(np.float64))) - nrm) <= rtol * nrm, name
            n += 1
    assert n >= 30


def test_teacher_backward_vs_reference_golden(golden_dir):
    """Row f-1: gradients of the three-branch NLL through MaxNet / BilinearFusion / PathomicNet, with and without
    cut_fuse_grad (fixture: the reference's PathomicNet run by tests/golden/make_golden_teacher_bwd.py)."""
    import os
    import numpy as np
    import torch
    import torch.nn.functional as F
    from oracle import weights as W
    from oracle.nets import pathomic_forward
    from oracle.step import synthetic_batch
    g = np.load(os.path.join(golden_dir, "teacher_bwd_b4_h64.npz"))
    bt = synthetic_batch(4, 64, seed=7)
    for tag, cut in (("cut", True), ("nocut", False)):
        sd = {k: (v.clone().requires_grad_(True) if (v.is_floating_point() and "running_" not in k) else v.clone())
              for k, v in W.make_state_dict(W.teacher_shapes(320), 3).items()}
        x_omic = bt["x_omic"].clone().requires_grad_(True)
        out = pathomic_forward(bt["x_path"], x_omic, sd, cut_fuse_grad=cut)
        pred, pred_path, pred_omic = out[5], out[6], out[7]
        loss = F.nll_loss(pred_path, bt["grade"]) + F.nll_loss(pred_omic, bt["grade"]) + F.nll_loss(pred, bt["grade"])
        names = [k for k, v in sd.items() if v.is_floating_point() and v.requires_grad]
        grads = torch.autograd.grad(loss, [sd[k] for k in names] + [x_omic], allow_unused=True)
        named = {k: (gr.numpy() if gr is not None else np.zeros(tuple(sd[k].shape), np.float32)) for k, gr in zip(names, grads[:-1])}
        _teacher_bwd_check(named, grads[-1].numpy(), loss.item(), g, tag, 2e-4)


def load_midstate(g, orc):
    """Put a DistillOracle into the mid-training state of tests/golden/make_golden_midstate.py (also used by the GPU test
    for the names / recipe)."""
    scales = dict(zip([str(s) for s in g["scale_names"]], g["scale_values"]))
    trainable = [(n, tuple(p.shape)) for n, p in orc.trainable()]
    mom = W.adam_moments(trainable, scales, int(g["seed"]) + 30)
    orc.adam_t = int(g["t0"])
    orc.iter_num = int(g["t0"])
    orc.lr = float(g["lr"])            # the trainer's LambdaLR has applied its epoch-0 factor (define_scheduler, :212)
    for n, (m, v) in mom.items():
        orc._m[n], orc._v[n] = m.clone(), v.clone()
    for i, key in enumerate(("Z0", "Z1")):
        orc.crd[i].params[2:4] = torch.as_tensor(g[key])
    return mom


def test_two_steps_from_mid_training_state(golden_dir):
    """The update path at the north-star tolerance: from a state with non-zero Adam moments (step count 7), an EMA
    model of its own, Z already set, the reference's SECOND step (logits, losses, GK-Refine weights after one Adam +
    EMA + bank update) is reproduced within 1e-3 - which the zero-moment start of the three-step fixture cannot show,
    because Adam's first updates are sign(g)."""
    g = _ld(golden_dir, "midstate_b8_h96.npz")
    torch.set_num_threads(8)
    orc = DistillOracle(default_opt(), seed=int(g["seed"]), n_data=int(g["n_data"]))
    load_midstate(g, orc)
    cut = lambda t: t.reshape(-1)[:4096]      # noqa: E731
    for it in range(2):
        bt = synthetic_batch(int(g["B"]), int(g["H"]), n_data=int(g["n_data"]), seed=310 + it)
        out = orc.step(bt, mid_ranks=[g["ranks"][2 * it], g["ranks"][2 * it + 1]])
        for k in ("logit_path", "ema_logit", "fuse_logit"):
            _close(g[f"{k}{it}"], out[k], 1e-3, 0)
        for k in ("loss_cls", "loss_div1", "loss_div2", "loss_kd1", "loss_kd2", "loss"):
            _close(g[f"{k}{it}"], out[k], 1e-3, 1e-4)
        _close(g[f"scale{it}"], out["scale"], 2e-3, 1e-3)
        for k in ("conv1.weight", "layer2.0.conv1.weight", "layer4.1.bn2.weight", "fc_new1.0.weight", "fc_new2.weight",
                  "fc_new2.bias"):
            if it == 0:
                _close(g["g0_" + k], cut(out["grads"]["student." + k]), 1e-6, 2e-3)
            _close(g[f"p{it}_{k}"], cut(orc.student[k]), 2e-6, 0)          # lr = 5e-4: a wrong update is ~1e-4
            _close(g[f"e{it}_{k}"], cut(orc.ema[k]), 2e-6, 0)
            _close(g[f"m{it}_{k}"], cut(orc._m["student." + k]), 1e-7, 2e-3)
            _close(g[f"v{it}_{k}"], cut(orc._v["student." + k]), 1e-10, 2e-3)
        _close(g[f"params0_{it}"], orc.crd[0].params, 1e-3, 1e-6)             # Z stays what it was
        _close(g[f"bank0_v1_rows{it}"], orc.crd[0].memory_v1[bt["index"]], 1e-5)
        _close(g[f"bank1_v2_rows{it}"], orc.crd[1].memory_v2[bt["index"]], 1e-5)


@pytest.mark.parametrize("mode", ["mid", "hard"])
def test_contrast_memory_v3_forward_outputs(golden_dir, mode):
    """The oracle's ContrastMemory_v3.forward against the reference class run STANDALONE (tests/golden/
    make_golden_crd_forward.py): (out_v1, out_v2), their gradients, Z and the updated rows over two calls."""
    from oracle.losses import contrast_memory_v3
    g = np.load(os.path.join(golden_dir, "crd_forward.npz"))
    st = CRDState(int(g["n_data"]), seed=int(g["bank_seed"]), P=int(g["P"]), K=int(g["K"]))
    for it in range(2):
        t = f"{mode}{it}"
        v1 = torch.as_tensor(g[f"v1_{t}"]).requires_grad_(True); v2 = torch.as_tensor(g[f"v2_{t}"]).requires_grad_(True)
        ranks = g[f"ranks_{mode}"][it] if mode == "mid" else None
        o1, o2, _ = contrast_memory_v3(st, v1, v2, torch.as_tensor(g[f"y_{t}"]), torch.as_tensor(g[f"idx_{t}"]),
                                       int(g["P2"]), int(g["K2"]), mode, ranks)
        gv1, gv2 = torch.autograd.grad((o1 * torch.as_tensor(g[f"w1_{t}"])).sum() + (o2 * torch.as_tensor(g[f"w2_{t}"])).sum(),
                                       [v1, v2])
        np.testing.assert_allclose(o1.detach().numpy(), g[f"out1_{t}"], rtol=1e-5, atol=1e-9)
        np.testing.assert_allclose(o2.detach().numpy(), g[f"out2_{t}"], rtol=1e-5, atol=1e-9)
        np.testing.assert_allclose(gv1.numpy(), g[f"gv1_{t}"], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(gv2.numpy(), g[f"gv2_{t}"], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(st.params.numpy(), g[f"params_{t}"], rtol=1e-6)
        y = torch.as_tensor(g[f"y_{t}"])
        np.testing.assert_allclose(st.memory_v1[y].numpy(), g[f"rows1_{t}"], atol=1e-6)


def test_contrast_loss_v2_standalone(golden_dir):
    from oracle.losses import contrast_loss_v2
    g = np.load(os.path.join(golden_dir, "crd_forward.npz"))
    x = torch.as_tensor(g["cl_x"]).requires_grad_(True)
    loss = contrast_loss_v2(x, int(g["P2"]), int(g["n_data"]))
    (gx,) = torch.autograd.grad(loss, [x])
    np.testing.assert_allclose(loss.item(), float(g["cl_loss_False"]), rtol=1e-6)
    np.testing.assert_allclose(gx.numpy(), g["cl_gx_False"], rtol=1e-5, atol=1e-7)


def test_sampler_draws_equal_the_reference(golden_dir):
    """oracle/sampler.py against the reference's `Pathomic_InstanceSample.__getitem__` run in the build container
    (tests/golden/make_golden_sampler.py; MICCAI-2022/data_loaders_MT.py:229-249 and the MIA-2023 neg_mode variants): the same
    seed gives the same `sample_idx` rows, draw for draw, over a sequence of items (the stream position carries over)."""
    from oracle.sampler import class_lists, sample_item
    g = _ld(golden_dir, "sampler_draws.npz")
    labels, order, n = g["labels"], g["order"], int(g["n"])
    cls_pos, cls_neg = class_lists(labels, 3)
    assert len(g["cases"]) == 11
    for name in g["cases"]:
        rng = np.random.RandomState(int(g[f"{name}_seed"]))
        want = g[f"{name}_rows"]
        for r, index in enumerate(order):
            got = sample_item(rng, int(index), int(labels[index]), cls_pos, cls_neg, n, int(g[f"{name}_P"]), int(g[f"{name}_K"]),
                              str(g[f"{name}_pos_mode"]), str(g[f"{name}_neg_mode"]))
            assert got.shape == want[r].shape and (got == want[r]).all(), (str(name), r, str(g[f"{name}_ref"]))


BRANCH_OPTS = {"t1_fuse_crd": (1, "fuse", "crd", "False"), "t1_ema_crd": (1, "self_EMA", "crd", "False"),
               "t1_fuse_kd": (1, "fuse", "kd", "False"), "t1_ema_kd": (1, "self_EMA", "kd", "False"),
               "t2_kd_gk": (2, "fuse", "kd", "True"), "t2_kd_sum": (2, "fuse", "kd", "False"),
               "t2_crd_sum": (2, "fuse", "crd", "False")}


def branch_opt(name, **kw):
    nt, wt, distill, aw = BRANCH_OPTS[name]
    return default_opt(num_teachers=nt, which_teacher=wt, distill=distill, assign_weights=aw, **kw)


def load_branch_state(g, orc):
    """The mid-training Adam state of tests/golden/make_golden_branches.py (tests/golden/_warm.py recipe)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import _warm
    scales = _warm.unpack_scales(g)
    trainable = [(n, tuple(p.shape)) for n, p in orc.trainable() if n in scales]
    mom = W.adam_moments(trainable, scales, _warm.SEED)
    orc.adam_t = orc.iter_num = int(g["t0"])
    orc.lr = float(g["lr"])            # the trainer's LambdaLR has applied its epoch-0 factor (define_scheduler, :212)
    for n, (m, v) in mom.items():
        orc._m[n], orc._v[n] = m.clone(), v.clone()
    return mom


@pytest.mark.parametrize("name", list(BRANCH_OPTS))
def test_option_branches_of_the_batch_body(golden_dir, name):
    """The batch body's non-default option branches (train_test_path_multi_distill.py:263-309: one teacher - fused or the mean
    teacher -, `--distill kd`, fixed weights instead of GK-Refine) against the reference run under those options
    (tests/golden/make_golden_branches.py), two steps from a mid-training optimiser state: logits, every loss term, GK-Refine
    weights, gradients, updated parameters / EMA, the CRD state - and the parameters the reference's optimiser never touches."""
    g = _ld(golden_dir, "branches_b8_h64.npz")
    torch.set_num_threads(8)
    orc = DistillOracle(branch_opt(name), seed=0, n_data=int(g["n_data"]))
    load_branch_state(g, orc)
    e_s0 = orc.crd[0].embed_s["linear.weight"].clone(); e_t1 = orc.crd[1].embed_t["linear.weight"].clone()
    ranks = g[name + ".ranks"]
    per = 0 if len(ranks) == 0 else len(ranks) // 2
    for it in range(2):
        bt = synthetic_batch(int(g["B"]), int(g["H"]), n_data=int(g["n_data"]), seed=500 + it)
        out = orc.step(bt, mid_ranks=list(ranks[per * it: per * (it + 1)]) if per else None)
        pre = name + "."
        _close(g[pre + f"logit_path{it}"], out["logit_path"], 1e-3, 0)
        _close(g[pre + f"loss_cls{it}"], out["loss_cls"], 1e-3, 1e-4)
        _close(g[pre + f"loss_div{it}"], out["loss_div1"] + out["loss_div2"], 1e-3, 1e-4)
        # (the UNSCALED CRD loss, ~16 per criterion: what enters the objective is beta = 0.02 times it; the reference's own fp32
        # value sits 5e-4 relative from its float64 run at step 1, tests/golden/branches_b8_h64_fp64.npz)
        _close(g[pre + f"loss_kd{it}"], out["loss_kd1"] + out["loss_kd2"], 1e-3, 1e-3)
        _close(g[pre + f"loss_KD{it}"], out["loss_KD"], 1e-3, 1e-4)
        _close(g[pre + f"loss{it}"], out["loss"], 1e-3, 1e-4)
        if pre + f"scale{it}" in g:
            _close(g[pre + f"scale{it}"], out["scale"], 2e-3, 1e-3)
        else:
            assert out["scale"] is None
        _close(g[pre + f"g_fc2_{it}"], out["grads"]["student.fc_new2.weight"], 1e-6, 2e-3)
        _close(g[pre + f"p_fc2_{it}"], orc.student["fc_new2.weight"], 2e-6, 0)
        _close(g[pre + f"ema_fc2_{it}"], orc.ema["fc_new2.weight"], 2e-6, 0)
        _close(g[pre + f"embed_s0_{it}"], orc.crd[0].embed_s["linear.weight"][:8], 2e-6, 0)
        _close(g[pre + f"embed_t1_{it}"], orc.crd[1].embed_t["linear.weight"][:8], 2e-6, 0)
        _close(g[pre + f"bank0_v1_rows{it}"], orc.crd[0].memory_v1[bt["index"]], 1e-5)
        _close(g[pre + f"bank1_v2_rows{it}"], orc.crd[1].memory_v2[bt["index"]], 1e-5)
        _close(g[pre + f"params0_{it}"], orc.crd[0].params, 1e-3, 1e-6)
        _close(g[pre + f"params1_{it}"], orc.crd[1].params, 1e-3, 1e-6)
        # parameters without a gradient: the reference's optimiser skips them (no weight decay either)
        skipped = set(str(s) for s in g[pre + f"no_grad{it}"])
        assert ("crd0.embed_s.linear.weight" in skipped) == (BRANCH_OPTS[name][2] == "kd")
        assert ("crd1.embed_t.linear.weight" in skipped) == (BRANCH_OPTS[name][2] == "kd" or BRANCH_OPTS[name][0] == 1)
        if "crd0.embed_s.linear.weight" in skipped:
            assert torch.equal(orc.crd[0].embed_s["linear.weight"], e_s0)
        if "crd1.embed_t.linear.weight" in skipped:
            assert torch.equal(orc.crd[1].embed_t["linear.weight"], e_t1)


def test_one_teacher_with_gk_refine_fails_like_the_reference(golden_dir):
    """`--num_teachers 1 --assign_weights True`: the reference's batch body reads KD_loss_list unbound (:293-304)."""
    g = _ld(golden_dir, "branches_b8_h64.npz")
    assert str(g["t1_gk_error"]) == "UnboundLocalError"
    orc = DistillOracle(default_opt(num_teachers=1, which_teacher="fuse", assign_weights="True"), seed=0, n_data=1024)
    with pytest.raises(UnboundLocalError):
        orc.step(synthetic_batch(2, 32, seed=1))
