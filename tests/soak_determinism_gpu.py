#!/usr/bin/env python3
"""Soak check of the multi-stream step at the benchmark size (not a test; ~1 minute on the GPU box):
    python tests/soak_determinism_gpu.py [steps] [miccai2022 | mia2022 | mia2023 | tsvd] [precision, e.g. fp16x3/x1]
Two fresh runs of `steps` graph-replayed distillation steps (B = 64, 512 x 512, same seeds) must produce bitwise the same loss
trajectory, parameters and bank rows: a missing dependency between the streams of the step (three forward streams, the
trunk backward's side stream, the loss head's) would show up as run-to-run noise sooner or later."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import multimodal_learning_amd as m  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
variant = sys.argv[2] if len(sys.argv) > 2 else "miccai2022"
if len(sys.argv) > 3:
    m.set_precision(sys.argv[3])      # (round 6: the half-pair kernels conv_tap5 / conv_tap6.hip count their own vmcnt - soak them too)
dev = torch.device("cuda:0")


def run_variant():
    torch.manual_seed(0)
    np.random.seed(2019)
    step, bts, _ = bench.variant_setup(variant, 128 if variant == "tsvd" else 64, 512, dev)
    if hasattr(step, "enable_graph"):
        step.enable_graph()
    losses = []
    for i in range(steps):
        out = step.step(bts[i % 2], epoch=5)
        losses.append(out["loss"].clone())
    torch.cuda.synchronize()
    sd = torch.cat([p.detach().flatten() for p in step.model.parameters()]).clone()
    res = (torch.stack(losses).cpu(), sd.cpu(), sd.cpu()[:1])
    del step
    torch.cuda.empty_cache()
    return res


def run():
    torch.manual_seed(0)
    np.random.seed(2019)
    opt = m.stage2_opt(dropout_rate=0.1, batch_size=64)
    step = m.DistillStep(opt, 1024, device=dev)
    for c in (step.criterion_kd, step.criterion_kd_path):
        c.contrast.verbose = False
    bts = [bench.make_batch(64, 512, 1024, opt, dev, seed=i) for i in range(2)]
    step.enable_graph()
    losses = []
    for i in range(steps):
        out = step.step(bts[i % 2], epoch=1 + i // 100)
        if i % 10 == 9 or i < 5:
            losses.append(out["loss"].clone())
    torch.cuda.synchronize()
    sd = torch.cat([p.detach().flatten() for p in step.model.parameters()]).clone()
    bank = step.criterion_kd.contrast.memory_v1.clone()
    res = (torch.stack(losses).cpu(), sd.cpu(), bank.cpu())
    for mod in (step.model, step.ema_model, step.fix_model.path_net):
        mod.release_workspaces()
    del step
    torch.cuda.empty_cache()
    return res


a = run() if variant == "miccai2022" else run_variant()
b = run() if variant == "miccai2022" else run_variant()
ok = all(torch.equal(x, y) for x, y in zip(a, b))
print("steps %d  last losses %s  finite %s  bitwise identical runs: %s" %
      (steps, [round(float(v), 5) for v in a[0][-3:]], bool(torch.isfinite(a[0]).all() and torch.isfinite(a[1]).all()), ok))
sys.exit(0 if ok else 1)
