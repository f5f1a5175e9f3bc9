#!/usr/bin/env python3
"""Where do the device-to-device copies of one distillation step come from (not a test):  python tests/prof_copies_gpu.py
One eager step of the device body (what a captured graph replays) at B = 8 / 128 x 128 under a TorchDispatchMode that records,
for every aten op that launches a copy / fill kernel, the innermost source line inside this repository.  Each such op is
one ~3-4 us launch (`__amd_rocclr_copyBuffer` / `fillBuffer` in the kernel trace) on a latency-bound head chain."""
import collections
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multimodal_learning_amd as m      # noqa: E402
import bench                              # noqa: E402

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WATCH = ("copy_", "clone", "_to_copy", "contiguous", "cat", "stack", "fill_", "zero_", "zeros", "zeros_like", "ones", "full",
         "arange", "repeat", "index", "index_select", "add", "mul", "sum", "dot", "div", "sub", "neg", "where", "expand")


class Spy(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.cnt = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__.split(".")[0]
        out = func(*args, **(kwargs or {}))
        on_gpu = any(torch.is_tensor(a) and a.is_cuda for a in list(args) + [out])
        if on_gpu and not name.startswith(("view", "reshape", "_unsafe_view", "detach", "alias", "as_strided", "t", "transpose",
                                            "select", "slice", "unsqueeze", "squeeze", "permute", "empty", "_local_scalar",
                                            "record_stream", "is_", "size", "stride", "unbind", "split", "expand")):
            src = "?"
            for fr in reversed(traceback.extract_stack()):
                if fr.filename.startswith(HERE) and "/tests/" not in fr.filename:
                    src = "%s:%d" % (fr.filename.replace(HERE + "/", ""), fr.lineno)
                    break
            self.cnt[(name, src)] += 1
        return out


dev = torch.device("cuda", 0)
variant = sys.argv[1] if len(sys.argv) > 1 else "miccai2022"
opt = m.stage2_opt(dropout_rate=0.1, batch_size=8)
step = m.DistillStep(opt, 1024, device=dev)
for c in (step.criterion_kd, step.criterion_kd_path):
    c.contrast.verbose = False
bts = [bench.make_batch(8, 128, 1024, opt, dev, seed=i) for i in range(2)]
for i in range(3):
    step.step(bts[i % 2], epoch=1)
torch.cuda.synchronize()
spy = Spy()
with spy:
    step.step(bts[1], epoch=1)
torch.cuda.synchronize()
tot = 0
for (name, src), n in sorted(spy.cnt.items(), key=lambda kv: -kv[1]):
    print("%3d  %-22s %s" % (n, name, src))
    tot += n
print("total aten GPU ops in one eager step (each a kernel launch of its own):", tot)
