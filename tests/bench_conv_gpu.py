#!/usr/bin/env python3
"""Per-layer micro-benchmark of the MFMA convolution kernels (not a test; run on the GPU box):
    python tests/bench_conv_gpu.py [B] [H]
times every distinct ResNet-18 conv shape (fwd / dgrad / wgrad, perf mode) with the in-library HIP-event timer."""
import ctypes
import sys

import torch

sys.path.insert(0, ".")
import multimodal_learning_amd as m
from multimodal_learning_amd._lib import lib, ptr, stream, check

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
H0 = int(sys.argv[2]) if len(sys.argv) > 2 else 512
L = lib()
shapes = [("layer1 3x3", 64, 64, H0 // 4, 3, 1, 1), ("layer2.0.c1 s2", 64, 128, H0 // 4, 3, 2, 1),
          ("layer2 3x3", 128, 128, H0 // 8, 3, 1, 1), ("layer2 ds 1x1 s2", 64, 128, H0 // 4, 1, 2, 0),
          ("layer3.0.c1 s2", 128, 256, H0 // 8, 3, 2, 1), ("layer3 3x3", 256, 256, H0 // 16, 3, 1, 1),
          ("layer4.0.c1 s2", 256, 512, H0 // 16, 3, 2, 1), ("layer4 3x3", 512, 512, H0 // 32, 3, 1, 1)]
buf = (ctypes.c_double * 24)()
print(f"B={B} input {H0}x{H0}   TFLOP/s (algorithmic)   [us per launch]")
for name, Cin, Cout, H, KS, S, pad in shapes:
    OH = (H + 2 * pad - KS) // S + 1
    x = torch.randn(B, H, H, Cin, device="cuda").bfloat16()
    w = torch.randn(Cout, Cin, KS, KS, device="cuda") * 0.05
    dy = torch.randn(B, OH, OH, Cout, device="cuda").bfloat16()
    y = torch.empty(B, OH, OH, Cout, device="cuda", dtype=torch.bfloat16)
    dx = torch.empty_like(x)
    dw = torch.empty_like(w)
    s1 = torch.empty(Cout, device="cuda"); s2 = torch.empty(Cout, device="cuda")
    ws = torch.empty(L.ph_conv2d_workspace_bytes(B, Cin, H, H, Cout, KS, S, pad), device="cuda", dtype=torch.uint8)
    res = []
    for what in ("fwd", "dgrad", "wgrad"):
        def run():
            if what == "fwd":
                check(L.ph_conv2d_fwd(ptr(x), ptr(w), ptr(y), ptr(s1), ptr(s2), B, Cin, H, H, Cout, KS, S, pad, 0, ptr(ws), stream()), what)
            elif what == "dgrad":
                check(L.ph_conv2d_dgrad(ptr(dy), ptr(w), ptr(dx), B, Cin, H, H, Cout, KS, S, pad, 0, ptr(ws), stream()), what)
            else:
                check(L.ph_conv2d_wgrad(ptr(x), ptr(dy), ptr(dw), B, Cin, H, H, Cout, KS, S, pad, 0, ptr(ws), stream()), what)
        for _ in range(3):
            run()
        L.ph_prof_reset(); L.ph_prof_enable(1)
        for _ in range(10):
            run()
        L.ph_prof_enable(0)
        L.ph_prof_summary(buf, 8)
        n = sum(buf[3 * c] for c in range(8)); ms = sum(buf[3 * c + 1] for c in range(8)); fl = sum(buf[3 * c + 2] for c in range(8))
        res.append(f"{what} {fl / (ms * 1e-3) / 1e12:7.1f} [{1000 * ms / 10:7.1f}]")
    print(f"{name:<18s} Cin {Cin:4d} Cout {Cout:4d} HW {H:4d}  " + "   ".join(res))
