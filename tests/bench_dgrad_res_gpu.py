#!/usr/bin/env python3
"""Timing of the dense 3x3 dgrad with and without the masked residual in its epilogue at the layer 2-4 shapes (not a test):
    python tests/bench_dgrad_res_gpu.py
Each number = 20 back-to-back C-ABI calls (weight pack included, the same for both arms), HIP events, alternating arms."""
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import multimodal_learning_amd as m                      # noqa: E402,F401
from multimodal_learning_amd._lib import lib, ptr, stream, check      # noqa: E402

L = lib()
st = stream()
for C, H in ((128, 64), (256, 32), (512, 16)):
    B = 64
    g = torch.Generator(device="cuda").manual_seed(0)
    dy = torch.randn(B, H, H, C, device="cuda", generator=g).bfloat16()
    rg = torch.randn(B, H, H, C, device="cuda", generator=g).bfloat16()
    ra = torch.randn(B, H, H, C, device="cuda", generator=g).relu_().bfloat16()
    w = torch.randn(C, C, 3, 3, device="cuda", generator=g) * 0.03
    dx = torch.empty(B, H, H, C, device="cuda", dtype=torch.bfloat16)
    ws = torch.empty(L.ph_conv2d_workspace_bytes(B, C, H, H, C, 3, 1, 1), device="cuda", dtype=torch.uint8)

    def plain():
        check(L.ph_conv2d_dgrad(ptr(dy), ptr(w), ptr(dx), B, C, H, H, C, 3, 1, 1, 0, ptr(ws), st), "dgrad")

    def res():
        check(L.ph_conv2d_dgrad_res(ptr(dy), ptr(w), ptr(dx), ptr(rg), ptr(ra), B, C, H, H, C, 3, 1, 1, 0, ptr(ws), st), "dgrad_res")

    out = {"plain": [], "residual": []}
    for rnd in range(5):
        for name, fn in (("plain", plain), ("residual", res)):
            fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record()
            torch.cuda.synchronize()
            if rnd:
                out[name].append(e0.elapsed_time(e1) * 50)
    print("C=%d H=%d: " % (C, H) + " | ".join("%s: %s us" % (k, " ".join("%.1f" % v for v in vs)) for k, vs in out.items()))
