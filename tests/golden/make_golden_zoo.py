#!/usr/bin/env python3
"""Golden vectors for the distiller-zoo losses that are built (SP, feats_KL, RKD, PKT): the reference's own classes, loaded
from their files ("MIA 2022/distiller_zoo/SP.py", "feats_KL.py", "RKD.py", "PKT.py"; the package __init__ imports dgl, which is absent).
Build container only.  Writes tests/golden/zoo_sp_featskl.npz."""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
ZOO = "/root/reference/MIA 2022/distiller_zoo"


def load(name):
    spec = importlib.util.spec_from_file_location("ref_" + name, os.path.join(ZOO, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    from make_golden import npz
    sp, fk = load("SP").Similarity(), load("feats_KL").feats_KL()
    g = torch.Generator().manual_seed(17)
    rec = {}
    for B in (8, 64):
        f_s = torch.randn(B, 128, generator=g).relu_().requires_grad_(True)
        f_t = torch.randn(B, 128, generator=g).relu_()
        l1 = sp(f_s, f_t)
        g1, = torch.autograd.grad(l1.sum(), f_s)
        l2 = fk(f_s, f_t)
        g2, = torch.autograd.grad(l2, f_s)
        rec.update({f"f_s{B}": f_s, f"f_t{B}": f_t, f"sp{B}": l1, f"sp_g{B}": g1, f"fkl{B}": l2, f"fkl_g{B}": g2})
    rkd, pkt = load("RKD").RKDLoss(), load("PKT").PKT()
    for B in (8, 64, 128):
        f_s = torch.randn(B, 128, generator=g).relu_().requires_grad_(True)
        f_t = torch.randn(B, 128, generator=g).relu_()
        l3 = rkd(f_s, f_t)
        g3, = torch.autograd.grad(l3, f_s)
        l4 = pkt(f_s, f_t)
        g4, = torch.autograd.grad(l4, f_s)
        rec.update({f"r_f_s{B}": f_s, f"r_f_t{B}": f_t, f"rkd{B}": l3, f"rkd_g{B}": g3, f"pkt{B}": l4, f"pkt_g{B}": g4})
    np.savez_compressed(os.path.join(HERE, "zoo_sp_featskl.npz"), **npz(rec))
    print("wrote zoo_sp_featskl.npz", float(rec["sp8"]), float(rec["fkl8"]))


if __name__ == "__main__":
    main()
