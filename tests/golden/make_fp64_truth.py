#!/usr/bin/env python3
"""fp64 'truth' trajectory of the B=16 / 224x224 three-step fixture, computed by the CPU oracle in double
precision (the oracle is pinned to the reference by tests/test_oracle_golden.py).

Why: the reference's fp32 trajectory is chaotic under Adam - its own fp32 run sits ~1e-2 from this fp64 truth
on the step-1 logits - so parity after the first parameter update can only be asserted relative to that noise
floor (tests/test_gpu_step.py).  Writes tests/golden/step_b16_h224_fp64.npz.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle.step import DistillOracle, default_opt, synthetic_batch  # noqa: E402


def main():
    g = np.load(os.path.join(HERE, "step_b16_h224.npz"))
    dt = torch.float64
    orc = DistillOracle(default_opt(), seed=int(g["seed"]), n_data=int(g["n_data"]))
    for d in (orc.student, orc.ema, orc.teacher):
        for k, v in d.items():
            if v.dtype.is_floating_point:
                d[k] = v.to(dt)
    for c in orc.crd:
        c.memory_v1 = c.memory_v1.to(dt); c.memory_v2 = c.memory_v2.to(dt); c.params = c.params.to(dt)
        for d in (c.embed_s, c.embed_t):
            for k, v in d.items():
                d[k] = v.to(dt)
    rec = {}
    for it in range(3):
        bt = synthetic_batch(int(g["B"]), int(g["H"]), seed=100 + it)
        bt = {k: (v.to(dt) if v.dtype.is_floating_point else v) for k, v in bt.items()}
        o = orc.step(bt, mid_ranks=[g["ranks"][2 * it], g["ranks"][2 * it + 1]])
        rec.update({f"logit_path{it}": o["logit_path"].numpy(), f"path_feat{it}": o["path_feat"].numpy(),
                    f"ema_logit{it}": o["ema_logit"].numpy(), f"loss{it}": o["loss"].numpy(),
                    f"loss_cls{it}": o["loss_cls"].numpy(), f"scale{it}": o["scale"].numpy(),
                    f"loss_div1_{it}": o["loss_div1"].numpy(), f"loss_kd1_{it}": o["loss_kd1"].numpy(),
                    f"loss_kd2_{it}": o["loss_kd2"].numpy(),
                    f"bank0_v1_rows{it}": orc.crd[0].memory_v1[bt["index"]].numpy(),
                    f"bank1_v2_rows{it}": orc.crd[1].memory_v2[bt["index"]].numpy()})
        if it == 0:
            rec["g0_conv1"] = o["grads"]["student.conv1.weight"].numpy()
        print("fp64 step", it, "loss", float(o["loss"]))
    np.savez_compressed(os.path.join(HERE, "step_b16_h224_fp64.npz"), **rec)


if __name__ == "__main__":
    main()
