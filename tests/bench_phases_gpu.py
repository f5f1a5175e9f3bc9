#!/usr/bin/env python3
"""Phase timeline of the replayed distillation step (not a test): ph_prof_stamp markers inside the captured graph.
    python tests/bench_phases_gpu.py [B] [H]
Prints, over 20 replays, the median time between the markers DistillStep._device_body places."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import multimodal_learning_amd as m  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
H = int(sys.argv[2]) if len(sys.argv) > 2 else 512
dev = torch.device("cuda:0")
opt = m.stage2_opt(dropout_rate=0.1, batch_size=B)
step = m.DistillStep(opt, 1024, device=dev)
step._stamps = torch.zeros(16, dtype=torch.int64, device=dev)
bts = [bench.make_batch(B, H, 1024, opt, dev, seed=31 + i) for i in range(2)]
for c in (step.criterion_kd, step.criterion_kd_path):
    c.contrast.verbose = False
for i in range(3):
    step.step(bts[i % 2], epoch=5)
step.enable_graph()
rows = []
for i in range(24):
    step.step(bts[i % 2], epoch=5)
    torch.cuda.synchronize()
    if i >= 4:
        rows.append(step._stamps.cpu().numpy().astype(np.int64).copy())
# back-to-back replays (no host sync in between): the idle time between the end of one step's graph and the start of the next
# is (time per step) - (marker 6 - marker 0); a second stamp buffer is not needed for that
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for i in range(30):
    step.step(bts[i % 2], epoch=5)
torch.cuda.synchronize()
per_step_us = (time.perf_counter() - t0) / 30 * 1e6
t = np.stack(rows)
rel = (t - t[:, :1]) * 0.01      # us since marker 0
names = {1: "student forward done (main stream)", 8: "mean-teacher forward done (side stream)",
         9: "teacher forward done (side stream)", 2: "streams joined", 3: "loss head forward done", 4: "loss head backward done "
         "(gradient of the student feature)", 5: "backward done", 6: "optimizer + EMA done"}
for k in (1, 8, 9, 2, 3, 4, 5, 6):
    print("  %-62s %9.1f us   (p10 %.1f  p90 %.1f)" % (names[k], np.median(rel[:, k]), np.percentile(rel[:, k], 10),
                                                        np.percentile(rel[:, k], 90)))
body = np.median(rel[:, 6])
print("  back-to-back: %.1f us per step, graph body (marker 0 -> 6) %.1f us -> %.1f us per step outside the body (input copies, "
      "graph launch, idle)" % (per_step_us, body, per_step_us - body))
print("  head phase (join -> feature gradient): %.1f us;  trunk backward: %.1f us" %
      (np.median(rel[:, 4] - rel[:, 2]), np.median(rel[:, 5] - rel[:, 4])))
