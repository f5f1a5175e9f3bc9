#!/usr/bin/env python3
"""A/B timing of the weight-gradient kernel between builds of the library (not a test):
    python tests/bench_wgrad_ab_gpu.py "" _wgNOLDS _wgNODMA _wgNOSLAB     (libpathomic_hip_trace<tag>.so)
rocprof-free: HIP events around 20 x ph_conv2d_wgrad (wgrad + slab reduction) per layer shape, alternating libraries."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tags = sys.argv[1:] or ["", "_wgNOLDS", "_wgNODMA", "_wgNOSLAB"]
libs = [C.CDLL(os.path.join(ROOT, "multimodal-learning_amd", "libpathomic_hip_trace%s.so" % t)) for t in tags]
vp, i32 = C.c_void_p, C.c_int
for L in libs:
    L.ph_conv2d_wgrad.restype = i32
    L.ph_conv2d_wgrad.argtypes = [vp] * 3 + [i32] * 9 + [vp, vp]
    L.ph_conv2d_workspace_bytes.restype = C.c_size_t
    L.ph_conv2d_workspace_bytes.argtypes = [i32] * 8
B, H0 = 64, 512
ptr = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for name, Cc, H in (("layer2", 128, H0 // 8), ("layer3", 256, H0 // 16), ("layer4", 512, H0 // 32)):
    x = torch.randn(B, H, H, Cc, device="cuda").bfloat16()
    dy = torch.randn(B, H, H, Cc, device="cuda").bfloat16()
    dw = torch.empty(Cc, Cc, 3, 3, device="cuda")
    ws = torch.empty(libs[0].ph_conv2d_workspace_bytes(B, Cc, H, H, Cc, 3, 1, 1), device="cuda", dtype=torch.uint8)
    res = [[] for _ in libs]
    for rnd in range(4):
        for k, L in enumerate(libs):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                assert L.ph_conv2d_wgrad(ptr(x), ptr(dy), ptr(dw), B, Cc, H, H, Cc, 3, 1, 1, 0, ptr(ws), st) == 0
            e1.record()
            torch.cuda.synchronize()
            if rnd:
                res[k].append(e0.elapsed_time(e1) * 50)
    print(name, "  ".join("%s %.1f us" % (tags[k] or "base", min(res[k])) for k in range(len(libs))))
