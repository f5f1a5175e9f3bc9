"""Pins the MIA-2022 variant restatements (oracle/variants.py) against golden vectors produced by running the
reference's own code (tests/golden/make_golden_variants.py)."""
import os

import numpy as np
import torch

from oracle import weights as W
from oracle.variants import CRDv3State, crd_v3_loss, momentum_aekd_loss
from tests.test_oracle_golden import _close


def test_crd_v3(golden_dir):
    g = np.load(os.path.join(golden_dir, "mia2022_crd_v3.npz"))
    st = CRDv3State(int(g["n_data"]), K=int(g["K"]), seed=int(g["bank_seed"]),
                    embed_s=W.make_state_dict(W.embed_shapes(), 30), embed_t=W.make_state_dict(W.embed_shapes(), 31))
    for d in (st.embed_s, st.embed_t):
        for v in d.values():
            v.requires_grad_(True)
    for it in range(2):
        f_s = torch.as_tensor(g[f"f_s{it}"]).requires_grad_(True)
        loss = crd_v3_loss(st, float(g[f"w{it}"]), f_s, torch.as_tensor(g[f"f_t{it}"]),
                           torch.as_tensor(g[f"index{it}"]), torch.as_tensor(g[f"sidx{it}"]))
        gs = torch.autograd.grad(loss.sum(), [f_s, st.embed_s["linear.weight"], st.embed_t["linear.weight"]])
        _close(g[f"loss{it}"], loss, 1e-5, 1e-5)
        _close(g[f"g_fs{it}"], gs[0], 1e-6, 1e-3); _close(g[f"g_ws{it}"], gs[1], 1e-6, 1e-3)
        _close(g[f"g_wt{it}"], gs[2], 1e-6, 1e-3)
        _close(g[f"params{it}"], st.params, 1e-2, 1e-5)
        _close(g[f"bank_v1_rows{it}"], st.memory_v1[torch.as_tensor(g[f"index{it}"])], 1e-6)


def test_momentum_gk(golden_dir):
    g = np.load(os.path.join(golden_dir, "mia2022_momentum_gk.npz"))
    ws = torch.as_tensor(g["ws"])
    for name, gth, th in (("plain", "False", 0.0), ("thresh", "True", 0.25)):
        mo = None
        for it in range(3):
            feat = (torch.as_tensor(g["feat"]) * (1 + 0.1 * it)).clone().requires_grad_(True)
            losses = [((feat * w).sum(1) ** 2).mean() * (0.1 + i) + (feat ** 2).mean() * (i % 2) for i, w in enumerate(ws)]
            mo, total = momentum_aekd_loss(losses[4], feat, losses[:4], mo, 0.9, gth, th)
            mo = mo.detach()
            _close(g[f"{name}_scale{it}"], mo, 1e-5, 1e-5)
            _close(g[f"{name}_total{it}"], total, 1e-5, 1e-5)


def test_mia2023_crd_v10(golden_dir):
    from oracle.variants import CRDv10State, crd_v10_loss
    g = np.load(os.path.join(golden_dir, "mia2023_crd_v10.npz"))
    st = CRDv10State(int(g["n_data"]), g["labels"], K=int(g["K"]), seed=int(g["bank_seed"]),
                     embed_s=W.make_state_dict(W.embed_shapes(), 50), embed_t=W.make_state_dict(W.embed_shapes(), 51))
    for d in (st.embed_s, st.embed_t):
        for v in d.values():
            v.requires_grad_(True)
    for it in range(2):
        f_s = torch.as_tensor(g[f"f_s{it}"]).requires_grad_(True)
        loss, sl, _ = crd_v10_loss(st, torch.as_tensor(g[f"w{it}"]), f_s, torch.as_tensor(g[f"f_t{it}"]),
                                   torch.as_tensor(g[f"grade{it}"]), torch.as_tensor(g[f"index{it}"]),
                                   torch.as_tensor(g[f"sidx{it}"]), int(g["num_pos"]))
        gs = torch.autograd.grad(loss, [f_s, st.embed_s["linear.weight"], st.embed_t["linear.weight"]])
        _close(g[f"loss{it}"], loss, 1e-5, 1e-5); _close(g[f"sample_loss{it}"], sl, 1e-4, 1e-5)
        _close(g[f"g_fs{it}"], gs[0], 1e-6, 1e-3); _close(g[f"g_ws{it}"], gs[1], 1e-6, 1e-3)
        _close(g[f"g_wt{it}"], gs[2], 1e-6, 1e-3); _close(g[f"params{it}"], st.params, 1e-2, 1e-5)
        _close(g[f"bank_v1_rows{it}"], st.memory_v1[torch.as_tensor(g[f"index{it}"])], 1e-6)


def test_mia2023_crd_v10_centers(golden_dir):
    """pos_extra == "centers", nce_p == 2 (class-mean centres) against the reference's CRDLoss, two calls."""
    from oracle.variants import CRDv10State, crd_v10_centers_loss
    g = np.load(os.path.join(golden_dir, "mia2023_crd_v10_centers.npz"))
    labels = g["labels"]
    class_idx = [np.nonzero(labels == c)[0] for c in range(3)]
    st = CRDv10State(int(g["n_data"]), labels, K=int(g["K"]), seed=int(g["bank_seed"]),
                     embed_s=W.make_state_dict(W.embed_shapes(), 52), embed_t=W.make_state_dict(W.embed_shapes(), 53))
    for d in (st.embed_s, st.embed_t):
        for v in d.values():
            v.requires_grad_(True)
    for it in range(2):
        f_s = torch.as_tensor(g[f"f_s{it}"]).requires_grad_(True)
        loss, sl = crd_v10_centers_loss(st, torch.as_tensor(g[f"w{it}"]), f_s, torch.as_tensor(g[f"f_t{it}"]),
                                        torch.as_tensor(g[f"grade{it}"]), torch.as_tensor(g[f"index{it}"]),
                                        torch.as_tensor(g[f"sidx{it}"]), class_idx, int(g["num_pos"]))
        gs = torch.autograd.grad(loss, [f_s, st.embed_s["linear.weight"], st.embed_t["linear.weight"]])
        _close(g[f"loss{it}"], loss, 1e-5, 1e-5); _close(g[f"sample_loss{it}"], sl, 1e-4, 1e-5)
        _close(g[f"g_fs{it}"], gs[0], 1e-6, 1e-3); _close(g[f"g_ws{it}"], gs[1], 1e-6, 1e-3)
        _close(g[f"g_wt{it}"], gs[2], 1e-6, 1e-3); _close(g[f"params{it}"], st.params, 1e-2, 1e-5)
        _close(g[f"bank_v1_rows{it}"], st.memory_v1[torch.as_tensor(g[f"index{it}"])], 1e-6)
        _close(g[f"bank_v2_rows{it}"], st.memory_v2[torch.as_tensor(g[f"index{it}"])], 1e-6)


def test_mia2023_rows(golden_dir):
    from oracle.variants import distill_kl_per_sample, assign_sample_weights, gk_refine_thresh
    g = np.load(os.path.join(golden_dir, "mia2023_rows.npz"))
    B = g["ys"].shape[0]
    for T in (1, 2):
        ys = torch.as_tensor(g["ys"]).requires_grad_(True)
        loss, sl = distill_kl_per_sample(ys, torch.as_tensor(g["yt"]), float(T))
        gg, = torch.autograd.grad((sl * torch.arange(1, B + 1).float()).sum(), ys)
        _close(g[f"kl_loss_T{T}"], loss, 1e-6); _close(g[f"kl_rows_T{T}"], sl, 1e-6); _close(g[f"kl_g_T{T}"], gg, 1e-5)
    d = assign_sample_weights(torch.softmax(torch.as_tensor(g["ys"]), 1), torch.softmax(torch.as_tensor(g["yt"]), 1),
                              torch.as_tensor(g["grade"]), 1.0)
    _close(g["discrep"], d, 1e-6)
    ws = torch.as_tensor(g["ws"])
    for name, use, th in (("thr", "True", 0.25), ("relu", "False", 0.2)):
        feat = torch.as_tensor(g["feat"]).clone().requires_grad_(True)
        rows = [((feat * w).sum(1) ** 2) * (0.1 + i) + (feat ** 2).mean(1) * (i % 2) for i, w in enumerate(ws)]
        scale, total, _ = gk_refine_thresh(rows[4].mean(), feat, rows[:4], use, th)
        _close(g[f"gk_{name}_scale"], scale, 1e-5, 1e-5); _close(g[f"gk_{name}_total"], total, 1e-5, 1e-5)


def test_tsvd_adjacency_and_penalty_vs_reference_golden(golden_dir):
    """Row a16: update_adj_tensor + Frobenius penalty of "MIA 2022/train_test_tSVD.py" (fixture produced by running them)."""
    import numpy as np
    import torch
    from oracle import variants as V
    g = np.load(os.path.join(golden_dir, "mia2022_tsvd.npz"))
    feats = [torch.tensor(g[f"feat{v}"], requires_grad=True) for v in range(4)]
    aux = [torch.tensor(g[f"aux{v}"]) for v in range(4)]
    adj = V.update_adj_tensor(feats)
    loss = V.tsvd_penalty(adj, aux, float(g["mu"]))
    grads = torch.autograd.grad(loss, feats)
    assert abs(loss.item() - float(g["loss"])) <= 1e-6 * abs(float(g["loss"]))
    for v in range(4):
        assert np.allclose(adj[v].detach().numpy(), g[f"adj{v}"], atol=1e-6)
        assert np.allclose(grads[v].numpy(), g[f"g_feat{v}"], atol=1e-6)


def test_tsvd_update_aux_prox_properties():
    """update_aux is parity-unpinned (absent from the reference); the oracle must at least be the TNN proximal operator:
    tau = 0 is the identity, a huge tau gives zero, and the result minimises tau*TNN(Y) + 1/2 ||Y - X||^2 against
    random perturbations."""
    import numpy as np
    from oracle import variants as V
    rng = np.random.default_rng(5)
    x = rng.standard_normal((12, 12, 4))
    y0, t0 = V.update_aux(x, 0.0)
    assert np.allclose(y0, x, atol=1e-10)
    y1, t1 = V.update_aux(x, 1e6)
    assert np.allclose(y1, 0.0, atol=1e-10) and t1 == 0.0

    def tnn(t):
        tf = np.fft.fft(t, axis=2)
        return sum(np.linalg.svd(tf[:, :, k], compute_uv=False).sum() for k in range(t.shape[2])) / t.shape[2]

    tau = 0.7
    y, _ = V.update_aux(x, tau)
    # scaling: the FFT-domain problem carries a factor V on the quadratic term, the oracle thresholds by tau directly
    obj = lambda t: tau * tnn(t) * x.shape[2] / x.shape[2] + 0.5 * np.sum((t - x) ** 2) / 1.0
    base = tau * tnn(y) + 0.5 * np.sum((y - x) ** 2) / 1.0
    # the prox of the slice-wise soft-thresholding is optimal for  tau * sum_k ||Yhat_k||_* + 1/2 sum_k ||Yhat_k - Xhat_k||^2
    xf, yf = np.fft.fft(x, axis=2), np.fft.fft(y, axis=2)
    fobj = lambda tf: sum(tau * np.linalg.svd(tf[:, :, k], compute_uv=False).sum()
                          + 0.5 * np.sum(np.abs(tf[:, :, k] - xf[:, :, k]) ** 2) for k in range(4))
    best = fobj(yf)
    for _ in range(20):
        pert = yf + 1e-3 * (rng.standard_normal(yf.shape) + 1j * rng.standard_normal(yf.shape))
        assert fobj(pert) >= best - 1e-9
    assert np.isfinite(base) and np.isfinite(obj(y))


def _embed2_state(seed):
    """Seed recipe of tests/golden/make_golden_stage1_terms.py for the two-layer projection heads."""
    g = torch.Generator().manual_seed(seed)
    return {"linear.0.weight": torch.randn(128, 128, generator=g) * 0.08, "linear.0.bias": torch.randn(128, generator=g) * 0.02,
            "linear.2.weight": torch.randn(128, 128, generator=g) * 0.08, "linear.2.bias": torch.randn(128, generator=g) * 0.02}


def test_stage1_orth_and_vanilla_crd_vs_reference_golden(golden_dir):
    """Stage-1 optional terms (row f-1): OrthLoss and the vanilla CRD criterion with two-layer heads."""
    from oracle.variants import orth_loss, crd_v0_loss
    g = np.load(os.path.join(golden_dir, "stage1_terms.npz"))
    x1 = torch.as_tensor(g["orth_x1"]).requires_grad_(True); x2 = torch.as_tensor(g["orth_x2"]).requires_grad_(True)
    lo = orth_loss(x1, x2)
    g1, g2 = torch.autograd.grad(lo, [x1, x2])
    _close(g["orth_loss"], lo, 1e-9, 1e-5); _close(g["orth_g1"], g1, 1e-9, 1e-4); _close(g["orth_g2"], g2, 1e-9, 1e-4)
    st = CRDv3State(int(g["n_data"]), K=int(g["K"]), seed=80, embed_s=_embed2_state(70), embed_t=_embed2_state(71))
    for d in (st.embed_s, st.embed_t):
        for v in d.values():
            v.requires_grad_(True)
    for it in range(2):
        f_s = torch.as_tensor(g[f"f_s{it}"]).requires_grad_(True)
        idx = torch.as_tensor(g[f"index{it}"])
        loss = crd_v0_loss(st, f_s, torch.as_tensor(g[f"f_t{it}"]), idx, torch.as_tensor(g[f"sidx{it}"]))
        gs = torch.autograd.grad(loss.sum(), [f_s, st.embed_s["linear.0.weight"], st.embed_s["linear.2.weight"],
                                              st.embed_t["linear.2.bias"]])
        _close(g[f"loss{it}"], loss, 1e-5, 1e-5)
        _close(g[f"g_fs{it}"], gs[0], 1e-6, 1e-3); _close(g[f"g_w0{it}"], gs[1], 1e-6, 1e-3)
        _close(g[f"g_w2{it}"], gs[2], 1e-6, 1e-3); _close(g[f"g_tb2{it}"], gs[3], 1e-6, 1e-3)
        _close(g[f"params{it}"], st.params, 1e-2, 1e-5)
        _close(g[f"bank_v1_rows{it}"], st.memory_v1[idx], 1e-6); _close(g[f"bank_v2_rows{it}"], st.memory_v2[idx], 1e-6)


def test_sampler_rule_restatement_properties():
    """oracle/sampler.py (numpy restatement of data_loaders_MT.py:187-249): structural rules of one draw."""
    from oracle.sampler import class_lists, sample_item
    labels = np.random.RandomState(1).randint(0, 3, 500)
    cp, cn = class_lists(labels, 3)
    assert sum(len(c) for c in cp) == 500 and all(len(cn[i]) == 500 - len(cp[i]) for i in range(3))
    rng = np.random.RandomState(2)
    for index in (0, 17, 499):
        g = int(labels[index])
        s = sample_item(rng, index, g, cp, cn, 500, 20, 64)
        assert s.shape == (84,) and s[0] == index and (labels[s[:20]] == g).all() and (labels[s[20:]] != g).all()
        assert len(set(s[1:20].tolist())) == 19 and len(set(s[20:].tolist())) == 64
        s2 = sample_item(rng, index, g, cp, cn, 500, 20, 700, neg_mode="all_others")       # K > list: with replacement
        assert (s2[20:] != index).all() and len(set(s2[20:].tolist())) < 700


def test_augment_oracle_structure():
    """oracle/augment.py (numpy restatement of the loader transform; the colour arithmetic is pinned against Pillow in
    tests/test_oracle_augment.py): neutral factors with the hue step dropped reduce it to flip + crop + normalise
    exactly, an enabled hue step with factor 0 costs Pillow's uint8 HSV round trip; a brightness factor scales with
    truncation; the contrast mean is the rounded luma mean of the image as it enters that step."""
    from oracle import augment as OA
    rng = np.random.default_rng(3)
    src = rng.integers(0, 256, (40, 56, 3), dtype=np.uint8)
    base = dict(flipH=1, flipV=1, top=4, left=9, S=32, b=1.0, c=1.0, s=1.0, h=0.0, order=(0, 1, 2, 3))
    out, mean = OA.one_view(src, dict(base, skip=(3,)))
    ref = src[::-1, ::-1][4:36, 9:41].astype(np.float32).transpose(2, 0, 1) / 255
    assert np.array_equal(out, (ref - np.float32(0.5)) / np.float32(0.5))
    out_h, _ = OA.one_view(src, base)     # hue enabled with factor 0: the uint8 HSV round trip moves a few grey levels
    assert 0 < np.abs(out_h - out).max() <= 2.0 / 255 * 8
    assert mean == int(OA.luma(src[::-1, ::-1][4:36, 9:41]).mean() + 0.5)
    out_b, _ = OA.one_view(src, dict(base, b=0.5, order=(0, 2, 1, 3), h=0.0, flipH=0, flipV=0, skip=(3,)))
    crop = src[4:36, 9:41].astype(np.int64)
    half = (crop * 0.5).astype(np.int64)
    got = np.rint((out_b * 0.5 + 0.5) * 255).astype(np.int64).transpose(1, 2, 0)
    assert np.abs(got - half).max() <= 1
