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
