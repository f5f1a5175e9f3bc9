#!/usr/bin/env python3
"""Where do the device-to-device copies of one distillation step come from (not a test):  python tests/prof_copies_gpu.py
One eager step at B = 8 / 128 x 128 under torch.profiler with Python stacks; prints the source lines that issue aten::copy_ /
aten::clone / aten::contiguous (each such call is one ~4 us launch on a latency-bound head chain)."""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multimodal_learning_amd as m      # noqa: E402
import bench                              # noqa: E402

dev = torch.device("cuda", 0)
opt = m.stage2_opt(dropout_rate=0.1, batch_size=8)
step = m.DistillStep(opt, 1024, device=dev)
for c in (step.criterion_kd, step.criterion_kd_path):
    c.contrast.verbose = False
bts = [bench.make_batch(8, 128, 1024, opt, dev, seed=i) for i in range(2)]
for i in range(3):
    step.step(bts[i % 2], epoch=1)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    step.step(bts[1], epoch=1)
torch.cuda.synchronize()
cnt = collections.Counter()
here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::clone", "aten::_to_copy", "aten::contiguous", "aten::cat", "aten::fill_", "aten::zero_"):
        src = "?"
        for fr in (ev.stack or []):
            if here in fr and "tests/" not in fr:
                src = fr.replace(here + "/", "")
                break
        cnt[(ev.name, src)] += 1
for (name, src), n in sorted(cnt.items(), key=lambda kv: -kv[1]):
    print("%3d  %-18s %s" % (n, name, src))
