#!/usr/bin/env python3
"""Golden vectors for CoxLoss: the reference's own function (MICCAI-2022/utils.py:361-376), compiled from the file where it
lies (the module imports lifelines / imblearn, absent here), with its gradient.  Writes tests/golden/cox_loss.npz."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)


def main():
    from make_golden import npz
    src = open("/root/reference/MICCAI-2022/utils.py").read()
    s0 = src.index("def CoxLoss("); s1 = src.index("def accuracy(", s0)
    ns = {"np": np, "torch": torch}
    exec(compile(src[s0:s1], "CoxLoss<reference>", "exec"), ns)
    g = torch.Generator().manual_seed(4)
    rec = {}
    for B in (8, 64, 300):
        theta = torch.randn(B, 1, generator=g).requires_grad_(True)
        t = torch.randint(1, 60, (B,), generator=g).float()            # ties included
        c = (torch.rand(B, generator=g) > 0.3).float()
        loss = ns["CoxLoss"](t, c, theta, torch.device("cpu"))
        gr, = torch.autograd.grad(loss, theta)
        rec.update({f"theta{B}": theta, f"t{B}": t, f"c{B}": c, f"loss{B}": loss, f"g{B}": gr})
    np.savez_compressed(os.path.join(HERE, "cox_loss.npz"), **npz(rec))
    print("wrote cox_loss.npz", [float(rec[f"loss{B}"]) for B in (8, 64, 300)])


if __name__ == "__main__":
    main()
