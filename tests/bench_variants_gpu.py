"""Step time of the MIA-2022 / MIA-2023 stage-2 variants at BASELINE sizes (not a test): python tests/bench_variants_gpu.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multimodal_learning_amd as m

B, H = 64, 512
m.set_precision("bf16")


def run(name, step, batch, R=10):
    step.enable_graph()
    for _ in range(4):
        step.step(batch, epoch=5)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(R):
        step.step(batch, epoch=5)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / R * 1e3
    print(f"{name}: {ms:.2f} ms/step = {B / ms * 1e3:.0f} tiles/s (B={B}, {H}x{H}, graph replay)")


def batch(n_data, K, labels=None, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(B, 3, H, H, generator=g) * 2 - 1
    index = torch.randperm(n_data, generator=g)[:B]
    sidx = torch.randint(0, n_data, (B, K + 1), generator=g); sidx[:, 0] = index
    grade = labels[index] if labels is not None else torch.randint(0, 3, (B,), generator=g)
    z = torch.zeros(B)
    d = lambda t: t.cuda()
    return ((d(x), d(x + 0.01 * torch.randn(B, 3, H, H, generator=g))), d(z), d(torch.randn(B, 320, generator=g)), d(z), d(z),
            d(grade), d(index), d(sidx))


# MIA-2022 stage 2: vanilla K+1 bank, momentum GK-Refine
opt = m.stage2_opt(dropout_rate=0.1, batch_size=B)
opt.nce_k, opt.grads_m, opt.grads_thresh, opt.thresh = 4096, 0.9, "False", 0.1
n_data = 16384
s22 = m.DistillStep(opt, n_data, device="cuda", variant="mia2022")
for c in (s22.criterion_kd, s22.criterion_kd_path):
    c.contrast.verbose = False
run("mia2022 (nce_k 4096, bank 16384)", s22, batch(n_data, 4096))
del s22
torch.cuda.empty_cache()
# MIA-2023 stage 2 at BASELINE config 5: bank of 65536 rows, class-aware KNN positives, per-sample GK-Refine
n_data = 65536
labels = torch.arange(n_data) % 3
opt = m.stage2_opt(dropout_rate=0.1, batch_size=B)
for k, v in dict(nce_k=4096, nce_p=4, pos_extra="neighbors", neg_mode="all_others", start_reweight=0, discrep_scale=1,
                 max_discrep=2.0, use_grads_thresh="True", grads_thresh=0.0, loss_weighting="GK_refine").items():
    setattr(opt, k, v)
cls = [np.nonzero((labels == c).numpy())[0] for c in range(3)]
s23 = m.DistillStep(opt, n_data, device="cuda", variant="mia2023", train_class_idx=cls)
for c in (s23.criterion_kd, s23.criterion_kd_path):
    c.contrast.verbose = False
run("mia2023 (bank 65536, nce_k 4096, 4 KNN positives)", s23, batch(n_data, 4096, labels))
