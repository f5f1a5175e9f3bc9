#!/usr/bin/env python3
"""A/B timing of the dense 3x3 convolutions between two builds of the library (not a test):
    python tests/bench_conv_ab_gpu.py _tagA _tagB      (libpathomic_hip_trace<tag>.so; "" = the plain trace build)
Alternates the two libraries on the same box, 5 rounds x 20 launches per layer shape, HIP events."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tags = sys.argv[1:3] if len(sys.argv) > 2 else ["", "_m16"]
libs = [C.CDLL(os.path.join(ROOT, "multimodal-learning_amd", "libpathomic_hip_trace%s.so" % t)) for t in tags]
vp, i32 = C.c_void_p, C.c_int
for L in libs:
    L.ph_conv2d_fwd.restype = i32
    L.ph_conv2d_fwd.argtypes = [vp] * 5 + [i32] * 9 + [vp, vp]
    L.ph_conv2d_workspace_bytes.restype = C.c_size_t
    L.ph_conv2d_workspace_bytes.argtypes = [i32] * 8
B, H0 = 64, 512
ptr = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for name, Cc, H in (("layer2", 128, H0 // 8), ("layer3", 256, H0 // 16), ("layer4", 512, H0 // 32)):
    x = torch.randn(B, H, H, Cc, device="cuda").bfloat16()
    w = torch.randn(Cc, Cc, 3, 3, device="cuda") * 0.05
    y = torch.empty(B, H, H, Cc, device="cuda", dtype=torch.bfloat16)
    s1 = torch.empty(Cc, device="cuda"); s2 = torch.empty(Cc, device="cuda")
    ws = torch.empty(libs[0].ph_conv2d_workspace_bytes(B, Cc, H, H, Cc, 3, 1, 1), device="cuda", dtype=torch.uint8)
    res = [[], []]
    for rnd in range(6):
        for k, L in enumerate(libs):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                assert L.ph_conv2d_fwd(ptr(x), ptr(w), ptr(y), ptr(s1), ptr(s2), B, Cc, H, H, Cc, 3, 1, 1, 0, ptr(ws), st) == 0
            e1.record()
            torch.cuda.synchronize()
            if rnd:
                res[k].append(e0.elapsed_time(e1) * 50)
    fl = 2.0 * B * H * H * Cc * Cc * 9
    print(name, " ".join("%s: %s us (%.0f TF/s)" % (tags[k] or "base", " ".join("%.1f" % v for v in res[k]), fl / min(res[k]) / 1e6)
                         for k in range(2)))
