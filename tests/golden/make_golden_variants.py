#!/usr/bin/env python3
"""Golden vectors for the MIA-2022 stage-2 variant, produced by importing and running the
reference's own code ("MIA 2022/CL_utils/CRD_criterion_v3.py", momentum_AEKD_loss of
"MIA 2022/train_test_path_multi_distill_v2.py").  Build container only.  Writes tests/golden/mia2022_*.npz."""
import os
import sys
import types
import contextlib
import io

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
REF = "/root/reference/MIA 2022"


def main():
    from make_golden import install_shims, npz
    install_shims()
    # the v2 trainer imports the whole distiller zoo (needs dgl) and data loaders (torchvision): stub them
    for name in ("distiller_zoo", "data_loaders_MT", "torchvision", "torchvision.transforms"):
        m = types.ModuleType(name)
        sys.modules[name] = m
    for n in ("DistillKL", "feats_KL", "HintLoss", "Attention", "Similarity", "Correlation", "VIDLoss", "RKDLoss",
              "PKT", "ABLoss", "FactorTransfer", "KDSVD", "FSP", "NSTLoss", "HKDLoss"):
        setattr(sys.modules["distiller_zoo"], n, object)
    sys.path.insert(0, REF)
    os.chdir(REF)
    from oracle import weights as W
    from oracle.variants import CRDv3State
    with contextlib.redirect_stdout(io.StringIO()):
        import importlib
        crdv3 = importlib.import_module("CL_utils.CRD_criterion_v3")
    opt = types.SimpleNamespace(s_dim=128, t_dim=128, feat_dim=128, nce_k=1024, nce_t=0.07, nce_m=0.5)
    n_data = 2048
    torch.manual_seed(3)
    with contextlib.redirect_stdout(io.StringIO()):
        crd = crdv3.CRDLoss(opt, n_data)
    crd.embed_s.load_state_dict(W.make_state_dict(W.embed_shapes(), 30))
    crd.embed_t.load_state_dict(W.make_state_dict(W.embed_shapes(), 31))
    st = CRDv3State(n_data, K=1024, seed=40)
    crd.contrast.memory_v1.copy_(st.memory_v1); crd.contrast.memory_v2.copy_(st.memory_v2)
    g = torch.Generator().manual_seed(77)
    rec = dict(n_data=n_data, bank_seed=40, K=1024)
    B = 8
    for it in range(2):
        f_s = torch.randn(B, 128, generator=g).relu_().requires_grad_(True)
        f_t = torch.randn(B, 128, generator=g).relu_()
        index = torch.randperm(n_data, generator=g)[:B]
        sidx = torch.randint(0, n_data, (B, 1025), generator=g); sidx[:, 0] = index
        w = 0.3 + 0.1 * it            # the shipped call passes epoch/niter_decay (a scalar)
        with contextlib.redirect_stdout(io.StringIO()):
            loss = crd(w, f_s, f_t, index, sidx)
        gs = torch.autograd.grad(loss.sum(), [f_s, crd.embed_s.linear.weight, crd.embed_t.linear.weight])
        rec.update({f"f_s{it}": f_s, f"f_t{it}": f_t, f"index{it}": index, f"sidx{it}": sidx, f"w{it}": w,
                    f"loss{it}": loss, f"g_fs{it}": gs[0], f"g_ws{it}": gs[1], f"g_wt{it}": gs[2],
                    f"params{it}": crd.contrast.params.clone(),
                    f"bank_v1_rows{it}": crd.contrast.memory_v1[index].clone()})
    np.savez_compressed(os.path.join(HERE, "mia2022_crd_v3.npz"), **npz(rec))

    # ---- momentum_AEKD_loss: run the reference function on a small differentiable graph
    src = open(os.path.join(REF, "train_test_path_multi_distill_v2.py")).read()
    start = src.index("def momentum_AEKD_loss"); end = src.index("def AEKD_loss")
    ns = {"torch": torch, "Variable": torch.autograd.Variable}
    exec(compile(src[start:end], "momentum_AEKD_loss<reference>", "exec"), ns)   # executes the reference's function body
    ref_fn = ns["momentum_AEKD_loss"]

    class Opt:   # zero_grad() stand-in: the function only calls optimizer.zero_grad()
        def zero_grad(self, *a, **k): pass
    g = torch.Generator().manual_seed(5)
    feat_c = torch.randn(16, 128, generator=g)
    ws = [torch.randn(128, generator=g) for _ in range(5)]
    rec = dict(feat=feat_c, ws=torch.stack(ws))
    for name, gth, th in (("plain", "False", 0.0), ("thresh", "True", 0.25)):
        o = types.SimpleNamespace(CE_grads=True, grads_thresh=gth, thresh=th, grads_m=0.9)
        mo = None
        for it in range(3):
            feat = (feat_c * (1 + 0.1 * it)).clone().requires_grad_(True)
            f2 = feat * 1.0      # non-leaf so that register_hook sees every backward
            losses = [((f2 * w).sum(1) ** 2).mean() * (0.1 + i) + (f2 ** 2).mean() * (i % 2) for i, w in enumerate(ws)]
            mo, total = ref_fn(o, Opt(), losses[4], f2, losses[:4], mo)
            mo = mo.detach()
            rec.update({f"{name}_scale{it}": mo.clone(), f"{name}_total{it}": total.detach()})
    np.savez_compressed(os.path.join(HERE, "mia2022_momentum_gk.npz"), **npz(rec))
    print("written mia2022_*.npz")


if __name__ == "__main__":
    main()
