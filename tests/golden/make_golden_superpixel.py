#!/usr/bin/env python3
"""Golden vectors for the superpixel attention masks: the reference's own statements
("MIA 2023/stage1_multi_modal_teacher/train_test_MT_SP_Masking.py":77-98, the part of `superpixel_attention_mask` that
follows the backward pass), compiled from the file where it lies and run on synthetic input gradients and a synthetic
superpixel label map.  Build container only.  Writes tests/golden/superpixel_masks.npz."""
import os
import sys
import textwrap

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
REF = "/root/reference/MIA 2023/stage1_multi_modal_teacher/train_test_MT_SP_Masking.py"


def label_map(B, H, W, cells, gen):
    """A superpixel-like label map: a jittered grid of `cells` x `cells` regions (some labels may stay unused)."""
    ys = torch.arange(H).view(1, H, 1).expand(B, H, W)
    xs = torch.arange(W).view(1, 1, W).expand(B, H, W)
    jy = torch.randint(-3, 4, (B, H, W), generator=gen)
    jx = torch.randint(-3, 4, (B, H, W), generator=gen)
    cy = ((ys + jy).clamp(0, H - 1) * cells // H)
    cx = ((xs + jx).clamp(0, W - 1) * cells // W)
    return (cy * cells + cx).long()


def main():
    from make_golden import install_shims, npz
    install_shims()
    src = open(REF).read()
    s0 = src.index("    # sp_mask  B, W, H"); s1 = src.index("    model.train()", s0)
    body = textwrap.dedent(src[s0:s1])
    code = compile(body, "superpixel_attention_mask<reference tail>", "exec")
    gen = torch.Generator().manual_seed(23)
    rec = {}
    for tag, (B, H, W, cells, PK, D, OK) in {"a": (4, 64, 64, 6, 5, 80, 8), "b": (3, 96, 128, 9, 12, 320, 30)}.items():
        sp_mask = label_map(B, H, W, cells, gen)
        x_path_grad = torch.randn(B, 3, H, W, generator=gen) * 1e-3
        x_omic_grad = torch.randn(B, D, generator=gen) * 1e-2
        ns = dict(torch=torch, F=F, sp_mask=sp_mask, x_path_grad=x_path_grad, x_omic_grad=x_omic_grad, Path_K=PK,
                  Omic_K=OK, device=torch.device("cpu"))
        exec(code, ns)
        rec.update({f"{tag}_sp_mask": sp_mask, f"{tag}_x_path_grad": x_path_grad, f"{tag}_x_omic_grad": x_omic_grad,
                    f"{tag}_Path_K": PK, f"{tag}_Omic_K": OK, f"{tag}_mean": ns["inputs_grad_aggre_mean"],
                    f"{tag}_path_mask": ns["x_path_super_mask"].float(), f"{tag}_omic_mask": ns["x_omic_super_mask"].float()})
        print(tag, "N =", int(sp_mask.max()) + 1, "mask pixels", int(ns["x_path_super_mask"].sum()), "omic", int(ns["x_omic_super_mask"].sum()))
    np.savez_compressed(os.path.join(HERE, "superpixel_masks.npz"), **npz(rec))
    print("wrote superpixel_masks.npz")


if __name__ == "__main__":
    main()
