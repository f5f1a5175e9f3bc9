#!/usr/bin/env python3
"""Golden vectors for the t-SVD stage-1 constraint, produced by importing "MIA 2022/train_test_tSVD.py"
and running ITS update_adj_tensor and penalty expression.  update_aux itself is absent from the reference (my_utils is
not in the repository) - it is stubbed for the import and not part of the fixture.  Build container only.
Writes tests/golden/mia2022_tsvd.npz."""
import contextlib
import io
import os
import sys
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
REF = "/root/reference/MIA 2022"


def main():
    from make_golden import install_shims, npz
    install_shims()
    for name in ("data_loaders_MT", "torchvision", "torchvision.transforms", "my_utils", "my_utils.TSVD_update_aux",
                 "my_utils.compute_gradients", "distiller_zoo"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["data_loaders_MT"].omic_transform = lambda *a, **k: None
    sys.modules["my_utils.TSVD_update_aux"].update_aux = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("absent in the reference"))
    sys.modules["my_utils.compute_gradients"].get_grad_embedding = lambda *a, **k: None
    sys.path.insert(0, REF)
    os.chdir(REF)
    with contextlib.redirect_stdout(io.StringIO()):
        import importlib
        ref = importlib.import_module("train_test_tSVD")
    g = torch.Generator().manual_seed(11)
    B, D, V = 16, 32, 4
    feats = [torch.randn(B, D, generator=g).relu_().requires_grad_(True) for _ in range(V)]
    aux = [torch.rand(B, B, generator=g) * 0.3 for _ in range(V)]
    mu = 0.037
    adj = ref.update_adj_tensor([None] * V, feats)
    loss = 0
    for v in range(V):   # the penalty expression of train_test_tSVD.py:421 (tSVD_mode == "path")
        loss = loss + mu / 2.0 * (torch.norm(adj[v] - aux[v])) ** 2
    grads = torch.autograd.grad(loss, feats)
    rec = dict(mu=mu, loss=loss)
    for v in range(V):
        rec[f"feat{v}"] = feats[v]; rec[f"aux{v}"] = aux[v]; rec[f"adj{v}"] = adj[v]; rec[f"g_feat{v}"] = grads[v]
    import numpy as np
    np.savez_compressed(os.path.join(HERE, "mia2022_tsvd.npz"), **npz(rec))
    print("wrote mia2022_tsvd.npz")


if __name__ == "__main__":
    main()
