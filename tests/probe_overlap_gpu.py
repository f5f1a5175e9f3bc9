"""Probe (not a test): do an MFMA-bound kernel (conv wgrad) and an HBM-bound elementwise kernel overlap when they are
launched on two streams?  python tests/probe_overlap_gpu.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multimodal_learning_amd as m
from multimodal_learning_amd._lib import lib, check, ptr

m.set_precision("bf16")
dev = "cuda"
B, C, H = 64, 128, 64
x = torch.randn(B, H, H, C, device=dev).bfloat16()
dy = torch.randn(B, H, H, C, device=dev).bfloat16()
dw = torch.empty(C, C, 3, 3, device=dev)
ws = torch.empty(lib().ph_conv2d_workspace_bytes(B, C, H, H, C, 3, 1, 1), device=dev, dtype=torch.uint8)
n = 64 * 1024 * 1024
a = torch.randn(n, device=dev); b = torch.randn(n, device=dev); o = torch.empty(n, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def wgrad(st, reps):
    for _ in range(reps):
        check(lib().ph_conv2d_wgrad(ptr(x), ptr(dy), ptr(dw), B, C, H, H, C, 3, 1, 1, 0, ptr(ws), st.cuda_stream), "wgrad")


def elt(st, reps):
    for _ in range(reps):
        check(lib().ph_eltwise(ptr(a), ptr(b), ptr(o), n, 0, st.cuda_stream), "eltwise")


def timed(fn):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    s1.synchronize(); s2.synchronize()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)


R = 20
for _ in range(2):
    wgrad(s1, 3); elt(s2, 3)
t1 = timed(lambda: wgrad(s1, R))
t2 = timed(lambda: elt(s2, R))
t12 = timed(lambda: (wgrad(s1, R), elt(s2, R)))
ts = timed(lambda: (wgrad(s1, R), elt(s1, R)))
print(f"wgrad alone {t1 / R * 1e3:.1f} us/iter, eltwise alone {t2 / R * 1e3:.1f} us/iter ({3 * n * 4 / (t2 / R * 1e-3) / 1e12:.2f} TB/s), "
      f"two streams {t12 / R * 1e3:.1f}, one stream {ts / R * 1e3:.1f}")
