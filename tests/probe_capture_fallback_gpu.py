#!/usr/bin/env python3
"""Probe (not a test): what happens when a graph capture of the distillation step FAILS.  Injects a capture-illegal call (a
device synchronize) at three points of the captured body (after every stream has been joined / while the teacher streams are
forked / inside the backward).  Finding on ROCm 7.2 / PyTorch 2.10: in all three cases - and in a ten-line capture with no fork
at all - hipStreamEndCapture leaves the capturing stream `Invalidated` and forked streams `Active`; the next host-to-device copy
fails with "operation not permitted when stream is capturing" and nothing ends the capture (hipStreamEndCapture: WrongThread /
Unmatched).  DistillStep therefore raises a clear error instead of pretending to continue, `bench.py --gpus N` restarts its
replicas eagerly, and replicas of an external launcher do not capture at all (bench.py: PH_BENCH_DDP_GRAPH).
    python tests/probe_capture_fallback_gpu.py"""
import os
import sys
import warnings

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import multimodal_learning_amd as m  # noqa: E402

dev = torch.device("cuda:0")
B, H = 8, 96


def run(where):
    opt = m.stage2_opt(dropout_rate=0.1, batch_size=B)
    step = m.DistillStep(opt, 1024, device=dev)
    for c in (step.criterion_kd, step.criterion_kd_path):
        c.contrast.verbose = False
    bts = [bench.make_batch(B, H, 1024, opt, dev, seed=i) for i in range(2)]
    armed = {"on": False}

    def bomb(*a, **k):
        if armed["on"] and torch.cuda.is_current_stream_capturing():
            armed["on"] = False
            torch.cuda.synchronize()          # illegal while capturing: raises and invalidates the capture

    if where == "forward":        # while the two teacher streams are forked (student forward starts)
        orig = step.model.forward
        step.model.forward = lambda *a, **k: (bomb(), orig(*a, **k))[1]
    elif where == "backward":     # inside the trunk backward's stream fork: a hook on the student feature's gradient
        orig_body = step._device_body

        def body(*a, **k):
            return orig_body(*a, **k)
        from multimodal_learning_amd import loss_head
        orig_bwd = loss_head.FusedDistillLossFn.backward

        def bwd(ctx, g):
            bomb()
            return orig_bwd(ctx, g)
        loss_head.FusedDistillLossFn.backward = staticmethod(bwd)
    else:                         # after everything has been joined
        orig_opt = step.optimizer.step
        step.optimizer.step = lambda *a, **k: (orig_opt(*a, **k), bomb())[0]
    step.enable_graph()
    losses = []
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        for i in range(6):
            if i == 2:
                armed["on"] = True
            out = step.step(bts[i % 2], epoch=1)
            losses.append(float(out["loss"]))
    torch.cuda.synchronize()
    fell_back = any("capture" in str(x.message) for x in w)
    print("%-9s fell back to eager: %s   losses %s" % (where, fell_back, [round(v, 4) for v in losses]), flush=True)
    if where == "backward":
        loss_head.FusedDistillLossFn.backward = staticmethod(orig_bwd)


for where in (sys.argv[1:] or ["end", "forward", "backward"]):
    run(where)
