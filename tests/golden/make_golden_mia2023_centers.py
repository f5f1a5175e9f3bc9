#!/usr/bin/env python3
"""Golden vectors for the class-centre positives of the MIA-2023 CRD bank (`--pos_extra centers --nce_p 2`,
"MIA 2023/stage2_unimodal_student/CL_utils/CRD_criterion_v10.py":81-101,118-139 with ContrastLoss :241-277), produced
by running the reference's CRDLoss for two calls.  Build container only.  Writes tests/golden/mia2023_crd_v10_centers.npz.
(nce_p > 2 clusters every class with sklearn KMeans from a random initialisation: not reproducible, not pinned.)"""
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
    n_data, K, NP, B = 2048, 256, 2, 8
    g = torch.Generator().manual_seed(13)
    labels = torch.randint(0, 3, (n_data,), generator=g)
    class_idx = [np.nonzero((labels == c).numpy())[0] for c in range(3)]
    opt = types.SimpleNamespace(s_dim=128, t_dim=128, feat_dim=128, nce_k=K, nce_t=0.07, nce_m=0.5, nce_p=NP,
                                pos_extra="centers")
    torch.manual_seed(4)
    with contextlib.redirect_stdout(io.StringIO()):
        crd = v10.CRDLoss(opt, n_data, class_idx)
    crd.embed_s.load_state_dict(W.make_state_dict(W.embed_shapes(), 52))
    crd.embed_t.load_state_dict(W.make_state_dict(W.embed_shapes(), 53))
    st = CRDv10State(n_data, labels, K=K, seed=61)
    crd.contrast.memory_v1.copy_(st.memory_v1); crd.contrast.memory_v2.copy_(st.memory_v2)
    rec = dict(n_data=n_data, K=K, num_pos=NP, bank_seed=61, labels=labels)
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
                    f"bank_v1_rows{it}": crd.contrast.memory_v1[index].clone(),
                    f"bank_v2_rows{it}": crd.contrast.memory_v2[index].clone()})
    np.savez_compressed(os.path.join(HERE, "mia2023_crd_v10_centers.npz"), **npz(rec))
    print("written mia2023_crd_v10_centers.npz", float(rec["loss0"]), float(rec["loss1"]))


if __name__ == "__main__":
    main()
