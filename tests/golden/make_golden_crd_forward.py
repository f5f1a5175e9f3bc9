#!/usr/bin/env python3
"""Golden vectors for the STANDALONE calls of the reference's ContrastMemory_v3.forward (CL_utils/memory_new.py:249-397,
returns (out_v1, out_v2)) and ContrastLoss_v2.forward (CL_utils/CRD_loss.py:221-252), produced by importing and running
the reference classes on CPU (build container only; shims of make_golden.py).  Two consecutive calls per mode (Z is set
on the first, frozen on the second; the second scores against the momentum-updated bank).  Saved: inputs, the host-RNG
rank lists the reference drew, outputs, gradients of a fixed linear functional of the outputs, params, updated rows.

Usage:  python tests/golden/make_golden_crd_forward.py        # writes tests/golden/crd_forward.npz
"""
import contextlib
import io
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG   # noqa: E402


def main():
    MG.install_shims()
    sys.path.insert(0, MG.REF)
    os.chdir(MG.REF)
    with contextlib.redirect_stdout(io.StringIO()):
        from CL_utils.memory_new import ContrastMemory_v3
        from CL_utils.CRD_loss import ContrastLoss_v2
    from oracle.losses import CRDState
    n_data, B, D, P, K, P2, K2 = 1024, 8, 128, 300, 700, 20, 512
    rec = dict(n_data=n_data, P=P, K=K, P2=P2, K2=K2, T=0.07, momentum=0.5, bank_seed=21)
    for mode in ("mid", "hard"):
        mem = ContrastMemory_v3(D, n_data, P, K, 0.07, 0.5, True, P2, "True", K2)
        st0 = CRDState(n_data, seed=21)
        mem.memory_v1.copy_(st0.memory_v1); mem.memory_v2.copy_(st0.memory_v2)
        g = torch.Generator().manual_seed(77)
        ranks_all = []
        _choice = np.random.choice

        def rec_choice(*a, **k):
            r = _choice(*a, **k); ranks_all.append(np.asarray(r)); return r
        np.random.choice = rec_choice
        np.random.seed(11)
        for it in range(2):
            v1 = torch.nn.functional.normalize(torch.randn(B, D, generator=g), dim=1).requires_grad_(True)
            v2 = torch.nn.functional.normalize(torch.randn(B, D, generator=g), dim=1).requires_grad_(True)
            y = torch.randperm(n_data, generator=g)[:B]
            idx = torch.randint(0, n_data, (B, P + K), generator=g); idx[:, 0] = y
            w1 = torch.randn(B, P2 + K2, 1, generator=g); w2 = torch.randn(B, P2 + K2, 1, generator=g)
            with contextlib.redirect_stdout(io.StringIO()):
                o1, o2 = mem(0.1, v1, v2, y, idx, select_pos_mode=mode)
            gv1, gv2 = torch.autograd.grad((o1 * w1).sum() + (o2 * w2).sum(), [v1, v2])
            t = f"{mode}{it}"
            rec.update({f"v1_{t}": v1, f"v2_{t}": v2, f"y_{t}": y, f"idx_{t}": idx, f"w1_{t}": w1, f"w2_{t}": w2,
                        f"out1_{t}": o1, f"out2_{t}": o2, f"gv1_{t}": gv1, f"gv2_{t}": gv2,
                        f"params_{t}": mem.params.clone(), f"rows1_{t}": mem.memory_v1[y].clone(),
                        f"rows2_{t}": mem.memory_v2[y].clone()})
        np.random.choice = _choice
        rec[f"ranks_{mode}"] = np.stack(ranks_all) if ranks_all else np.zeros((0, P2), dtype=np.int64)
    # ContrastLoss_v2 on its own: x = scores / Z in the range the step produces
    g = torch.Generator().manual_seed(5)
    x = (torch.rand(B, P2 + K2, 1, generator=g) * 3e-3 + 1e-5).requires_grad_(True)
    for kd in ("False", "True"):
        crit = ContrastLoss_v2(n_data, kd)
        loss = crit(x, P2)
        wv = torch.randn(loss.shape, generator=g) if loss.dim() else torch.tensor(1.0)
        (gx,) = torch.autograd.grad((loss * wv).sum(), [x])
        rec.update({f"cl_loss_{kd}": loss, f"cl_w_{kd}": wv, f"cl_gx_{kd}": gx})
    rec["cl_x"] = x
    np.savez_compressed(os.path.join(HERE, "crd_forward.npz"), **MG.npz(rec))
    print("wrote crd_forward.npz")


if __name__ == "__main__":
    main()
