#!/usr/bin/env python3
"""Golden vectors for CRDLoss with `--sample_KD True` (MICCAI-2022/CL_utils/CRD_loss.py:127-175 calling
ContrastLoss_v2's per-sample branch, :246-250): the criterion then returns the [B] per-sample losses s_loss + t_loss
instead of a 0-d mean.  Produced by importing and running the reference classes on CPU (build container only; shims and
option parsing of make_golden.py).  Two consecutive calls (Z is set on the first, frozen on the second; the second scores
against the momentum-updated bank).  Saved: inputs, the host-RNG rank lists the reference drew, the per-sample losses,
gradients of a fixed linear functional of them, params, updated bank rows.

Usage:  python tests/golden/make_golden_crd_samplekd.py        # writes tests/golden/crd_samplekd.npz
"""
import contextlib
import io
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG   # noqa: E402


def main():
    MG.install_shims()
    sys.path.insert(0, MG.REF)
    os.chdir(MG.REF)
    opt = MG.ref_opt(tempfile.mkdtemp(), extra=("--sample_KD", "True"))
    assert opt.sample_KD == "True"
    with contextlib.redirect_stdout(io.StringIO()):
        from CL_utils.CRD_loss import CRDLoss
    from oracle import weights as W
    from oracle.losses import CRDState
    n_data, Bc = 1024, 8
    rec = dict(n_data=n_data, bank_seed=23, nce_p=opt.nce_p, nce_k=opt.nce_k, nce_p2=opt.nce_p2, nce_k2=opt.nce_k2)
    for mode in ("mid", "hard"):
        opt.select_pos_mode = mode
        torch.manual_seed(20)
        with contextlib.redirect_stdout(io.StringIO()):
            crd = CRDLoss(opt, n_data)
        crd.embed_s.load_state_dict(W.make_state_dict(W.embed_shapes(), 10))
        crd.embed_t.load_state_dict(W.make_state_dict(W.embed_shapes(), 11))
        st0 = CRDState(n_data, seed=23)
        crd.contrast.memory_v1.copy_(st0.memory_v1); crd.contrast.memory_v2.copy_(st0.memory_v2)
        g = torch.Generator().manual_seed(41)
        ranks_all = []
        _choice = np.random.choice

        def rec_choice(*a, **k):
            r = _choice(*a, **k); ranks_all.append(np.asarray(r)); return r
        np.random.choice = rec_choice
        np.random.seed(2019)
        for it in range(2):
            f_s = torch.randn(Bc, 128, generator=g).relu_().requires_grad_(True)
            f_t = torch.randn(Bc, 128, generator=g).relu_()
            index = torch.randperm(n_data, generator=g)[:Bc]
            sidx = torch.randint(0, n_data, (Bc, opt.nce_p + opt.nce_k), generator=g); sidx[:, 0] = index
            wv = torch.randn(Bc, generator=g)
            with contextlib.redirect_stdout(io.StringIO()):
                rows = crd(0.1, f_s, f_t, index, sidx)
            assert rows.shape == (Bc,), rows.shape
            gs = torch.autograd.grad((rows * wv).sum(), [f_s, crd.embed_s.linear.weight, crd.embed_t.linear.weight,
                                                         crd.embed_s.linear.bias])
            t = f"{mode}{it}"
            rec.update({f"f_s_{t}": f_s, f"f_t_{t}": f_t, f"index_{t}": index, f"sidx_{t}": sidx, f"w_{t}": wv,
                        f"rows_{t}": rows, f"g_fs_{t}": gs[0], f"g_ws_{t}": gs[1], f"g_wt_{t}": gs[2], f"g_bs_{t}": gs[3],
                        f"params_{t}": crd.contrast.params.clone(),
                        f"bank_v1_rows_{t}": crd.contrast.memory_v1[index].clone(),
                        f"bank_v2_rows_{t}": crd.contrast.memory_v2[index].clone()})
        np.random.choice = _choice
        rec[f"ranks_{mode}"] = np.stack(ranks_all) if ranks_all else np.zeros((0, opt.nce_p2), dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, "crd_samplekd.npz"), **MG.npz(rec))
    print("wrote crd_samplekd.npz")


if __name__ == "__main__":
    main()
