#!/usr/bin/env python3
"""Golden vectors for the MIA-2023 stage-2 variant, produced by running the reference's
"MIA 2023/stage2_unimodal_student/CL_utils/CRD_criterion_v10.py", KD_loss.py and the assign_sample_weights /
GK_refine_thresh functions of its trainer.  Build container only.  Writes tests/golden/mia2023_*.npz."""
import contextlib
import importlib
import io
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
REF = "/root/reference/MIA 2023/stage2_unimodal_student"


def main():
    from make_golden import install_shims, npz
    install_shims()
    sys.path.insert(0, REF)
    os.chdir(REF)
    from oracle import weights as W
    from oracle.variants import CRDv10State
    with contextlib.redirect_stdout(io.StringIO()):
        v10 = importlib.import_module("CL_utils.CRD_criterion_v10")
        kd = importlib.import_module("KD_loss")
    n_data, K, NP, B = 4096, 512, 6, 8
    g = torch.Generator().manual_seed(11)
    labels = torch.randint(0, 3, (n_data,), generator=g)
    class_idx = [np.nonzero((labels == c).numpy())[0] for c in range(3)]
    opt = types.SimpleNamespace(s_dim=128, t_dim=128, feat_dim=128, nce_k=K, nce_t=0.07, nce_m=0.5, nce_p=NP,
                                pos_extra="neighbors")
    torch.manual_seed(4)
    with contextlib.redirect_stdout(io.StringIO()):
        crd = v10.CRDLoss(opt, n_data, class_idx)
    crd.embed_s.load_state_dict(W.make_state_dict(W.embed_shapes(), 50))
    crd.embed_t.load_state_dict(W.make_state_dict(W.embed_shapes(), 51))
    st = CRDv10State(n_data, labels, K=K, seed=60)
    crd.contrast.memory_v1.copy_(st.memory_v1); crd.contrast.memory_v2.copy_(st.memory_v2)
    rec = dict(n_data=n_data, K=K, num_pos=NP, bank_seed=60, labels=labels)
    for it in range(2):
        f_s = torch.randn(B, 128, generator=g).relu_().requires_grad_(True)
        f_t = torch.randn(B, 128, generator=g).relu_()
        index = torch.randperm(n_data, generator=g)[:B]
        sidx = torch.randint(0, n_data, (B, K + 1), generator=g); sidx[:, 0] = index
        grade = labels[index]
        w = (1 + torch.rand(B, generator=g)).view(-1, 1)
        with contextlib.redirect_stdout(io.StringIO()):
            loss, sample_loss = crd(w, f_s, f_t, grade, index, sidx)
        gs = torch.autograd.grad(loss, [f_s, crd.embed_s.linear.weight, crd.embed_t.linear.weight], retain_graph=True)
        rec.update({f"f_s{it}": f_s, f"f_t{it}": f_t, f"index{it}": index, f"sidx{it}": sidx, f"grade{it}": grade,
                    f"w{it}": w, f"loss{it}": loss, f"sample_loss{it}": sample_loss, f"g_fs{it}": gs[0],
                    f"g_ws{it}": gs[1], f"g_wt{it}": gs[2], f"params{it}": crd.contrast.params.clone(),
                    f"bank_v1_rows{it}": crd.contrast.memory_v1[index].clone()})
    np.savez_compressed(os.path.join(HERE, "mia2023_crd_v10.npz"), **npz(rec))

    # ---- per-sample KL, assign_sample_weights, GK_refine_thresh
    src = open(os.path.join(REF, "train_test_path_multi_distill.py")).read()
    ns = {"torch": torch, "np": np, "F": torch.nn.functional, "Variable": torch.autograd.Variable}
    from sklearn.metrics.pairwise import cosine_similarity
    ns["cosine_similarity"] = cosine_similarity
    s0 = src.index("def GK_refine_thresh"); s1 = src.index("def assign_sample_weights"); s2 = src.index("def train(")
    exec(compile(src[s0:s2], "mia2023<reference>", "exec"), ns)
    g = torch.Generator().manual_seed(21)
    B = 16
    ys = torch.randn(B, 3, generator=g).requires_grad_(True)
    yt = torch.randn(B, 3, generator=g)
    grade = torch.randint(0, 3, (B,), generator=g)
    rec = dict(ys=ys, yt=yt, grade=grade)
    for T in (1.0, 2.0):
        loss, sl = kd.DistillKL(T)(ys, yt)
        gg, = torch.autograd.grad((sl * torch.arange(1, B + 1).float()).sum(), ys)
        rec.update({f"kl_loss_T{int(T)}": loss, f"kl_rows_T{int(T)}": sl, f"kl_g_T{int(T)}": gg})
    rec["discrep"] = ns["assign_sample_weights"](torch.softmax(ys, 1), torch.softmax(yt, 1), grade, 1, 1)

    class Opt:
        def zero_grad(self, *a, **k): pass
    feat_c = torch.randn(B, 128, generator=g)
    ws = [torch.randn(128, generator=g) for _ in range(5)]
    rec.update(feat=feat_c, ws=torch.stack(ws))
    for name, use, th in (("thr", "True", 0.25), ("relu", "False", 0.2)):
        o = types.SimpleNamespace(CE_grads=True, batch_size=B, use_grads_thresh=use, grads_thresh=th)
        feat = feat_c.clone().requires_grad_(True)
        f2 = feat * 1.0
        rows = [((f2 * w).sum(1) ** 2) * (0.1 + i) + (f2 ** 2).mean(1) * (i % 2) for i, w in enumerate(ws)]   # per-sample [B]
        scale, total = ns["GK_refine_thresh"](o, Opt(), rows[4].mean(), f2, rows[:4])
        rec.update({f"gk_{name}_scale": scale, f"gk_{name}_total": total.detach()})
    np.savez_compressed(os.path.join(HERE, "mia2023_rows.npz"), **npz(rec))
    print("written mia2023_*.npz")


if __name__ == "__main__":
    main()
