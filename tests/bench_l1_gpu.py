#!/usr/bin/env python3
"""Timing of the layer-1 convolution kernels at the benchmark's shape (not a test):  python tests/bench_l1_gpu.py
B = 64, 128 x 128 x 64 -> 64, 3x3: forward (+ BatchNorm partial sums), dgrad, dgrad + masked residual, dgrad with the fused
BatchNorm-backward sums; tapconv2_l1_kernel against conv_tap4.hip with the epilogue after / inside the tile stream.  Each
number = 20 back-to-back calls of the C-ABI entry point (which also packs the weights: ~6 us of small kernels per call, the
same for every arm), HIP events, alternating arms, 5 rounds."""
import ctypes as C
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import multimodal_learning_amd as m                      # noqa: E402
from multimodal_learning_amd._lib import lib, ptr, stream, check      # noqa: E402

L = lib()
B, H, Cc = 64, 128, 64
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(B, H, H, Cc, device="cuda", generator=g).bfloat16()
dy = torch.randn(B, H, H, Cc, device="cuda", generator=g).bfloat16()
rg = torch.randn(B, H, H, Cc, device="cuda", generator=g).bfloat16()
ra = torch.randn(B, H, H, Cc, device="cuda", generator=g).relu_().bfloat16()
yy = torch.randn(B, H, H, Cc, device="cuda", generator=g).bfloat16()
w = torch.randn(Cc, Cc, 3, 3, device="cuda", generator=g) * 0.05
y = torch.empty(B, H, H, Cc, device="cuda", dtype=torch.bfloat16)
s1 = torch.empty(Cc, device="cuda"); s2 = torch.empty(Cc, device="cuda"); s3 = torch.empty(3, Cc, device="cuda")
sc = torch.rand(Cc, device="cuda") + 0.5; sh = torch.randn(Cc, device="cuda") * 0.3; mu = torch.randn(Cc, device="cuda") * 0.3
ws = torch.empty(L.ph_conv2d_workspace_bytes(B, Cc, H, H, Cc, 3, 1, 1) + 3 * 4 * Cc * 1024, device="cuda", dtype=torch.uint8)
st = stream()


def fwd():
    check(L.ph_conv2d_fwd(ptr(x), ptr(w), ptr(y), ptr(s1), ptr(s2), B, Cc, H, H, Cc, 3, 1, 1, 0, ptr(ws), st), "fwd")


def dgrad():
    check(L.ph_conv2d_dgrad(ptr(dy), ptr(w), ptr(y), B, Cc, H, H, Cc, 3, 1, 1, 0, ptr(ws), st), "dgrad")


def dgrad_res():
    check(L.ph_conv2d_dgrad_res(ptr(dy), ptr(w), ptr(y), ptr(rg), ptr(ra), B, Cc, H, H, Cc, 3, 1, 1, 0, ptr(ws), st), "dgrad_res")


def bst_a():
    check(L.ph_conv2d_dgrad_bnstat(ptr(dy), ptr(w), ptr(y), None, None, ptr(yy), None, None, ptr(sc), ptr(sh), ptr(mu), None, ptr(s3),
                                   B, Cc, H, H, Cc, ptr(ws), st), "bst a")


def bst_b():
    check(L.ph_conv2d_dgrad_bnstat(ptr(dy), ptr(w), ptr(y), ptr(rg), ptr(ra), ptr(yy), ptr(ra), None, None, None, ptr(mu), None, ptr(s3),
                                   B, Cc, H, H, Cc, ptr(ws), st), "bst b")


arms = [("l1 two-group", 0, 0), ("tap4 epilogue after the tile", 1, 0), ("tap4 epilogue inside the next tile", 1, 1)]
for name, fn in (("forward", fwd), ("dgrad", dgrad), ("dgrad + masked residual", dgrad_res), ("dgrad + fused sums (own ReLU)", bst_a),
                 ("dgrad + residual + fused sums (mask tensor)", bst_b)):
    res = {a[0]: [] for a in arms}
    for rnd in range(6):
        for an, t4, ovl in arms:
            if t4 == 0 and fn in (bst_a, bst_b):
                continue
            L.ph_debug_set_tap4(t4); L.ph_debug_set_tap4_ovl(ovl)
            fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record()
            torch.cuda.synchronize()
            if rnd:
                res[an].append(e0.elapsed_time(e1) * 50)
    print("%-46s" % name, " | ".join("%s: %s us" % (an, " ".join("%.1f" % v for v in res[an])) for an, _, _ in arms if res[an]))
L.ph_debug_set_tap4(1); L.ph_debug_set_tap4_ovl(1)
