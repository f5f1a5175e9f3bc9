#!/usr/bin/env python3
"""fp64 'truth' of step 0 of the mid-training-state fixture (make_golden_midstate.py), computed by the CPU oracle in double
precision (the oracle is pinned to the reference by tests/test_oracle_golden.py::test_two_steps_from_mid_training_state).

Why: a ReLU network's gradient is discontinuous in its pre-activations.  At B = 8 / 96 x 96 a single activation whose
pre-activation lies within ~1e-6 of zero - there are a few in every batch of this size - decides 0.5 % of the first layers'
weight gradient, and any two fp32-grade evaluations (the reference's own included) can put it on different sides.  The GPU
test therefore also states every gradient against this fp64 truth, bounded by the reference's OWN distance to it
(tests/test_gpu_step.py).  Writes tests/golden/midstate_b8_h96_fp64.npz."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle.step import DistillOracle, default_opt, synthetic_batch  # noqa: E402
from tests.test_oracle_golden import load_midstate  # noqa: E402

WATCH = ("conv1.weight", "layer2.0.conv1.weight", "layer4.1.bn2.weight", "fc_new1.0.weight", "fc_new2.weight", "fc_new2.bias")


def main():
    g = np.load(os.path.join(HERE, "midstate_b8_h96.npz"))
    dt = torch.float64
    torch.set_num_threads(8)
    orc = DistillOracle(default_opt(), seed=int(g["seed"]), n_data=int(g["n_data"]))
    load_midstate(g, orc)
    for d in (orc.student, orc.ema, orc.teacher):
        for k, v in d.items():
            if v.dtype.is_floating_point:
                d[k] = v.to(dt)
    for c in orc.crd:
        c.memory_v1 = c.memory_v1.to(dt); c.memory_v2 = c.memory_v2.to(dt); c.params = c.params.to(dt)
        for d in (c.embed_s, c.embed_t):
            for k, v in d.items():
                d[k] = v.to(dt)
    for n in list(orc._m):
        orc._m[n] = orc._m[n].to(dt); orc._v[n] = orc._v[n].to(dt)
    bt = synthetic_batch(int(g["B"]), int(g["H"]), n_data=int(g["n_data"]), seed=310)
    bt = {k: (v.to(dt) if v.dtype.is_floating_point else v) for k, v in bt.items()}
    o = orc.step(bt, mid_ranks=[g["ranks"][0], g["ranks"][1]])
    cut = lambda t: t.reshape(-1)[:4096]      # noqa: E731
    rec = {"logit_path0": o["logit_path"].numpy(), "loss0": o["loss"].numpy()}
    for k in WATCH:
        rec["g0_" + k] = cut(o["grads"]["student." + k]).numpy()
        ref = np.asarray(g["g0_" + k], dtype=np.float64)
        print("%-28s |reference fp32 - fp64 truth| max %.3e  (max|ref| %.3e, rel %.2e)" %
              (k, np.abs(ref - rec["g0_" + k]).max(), np.abs(ref).max(), np.abs(ref - rec["g0_" + k]).max() / np.abs(ref).max()))
    rec["g0_embed_s0"] = cut(o["grads"]["crd0.embed_s.linear.weight"]).numpy()
    print("logits: |reference - truth| %.3e" % np.abs(np.asarray(g["logit_path0"], dtype=np.float64) - rec["logit_path0"]).max())
    np.savez_compressed(os.path.join(HERE, "midstate_b8_h96_fp64.npz"), **rec)


if __name__ == "__main__":
    main()
