"""Timing of the layer-1 half-pair convolution alone (conv_tap5.hip) at the benchmark shape (64 x 128 x 128 x 64 -> 64): forward
(3 products) and hi-only dgrad; PH_LIB_VARIANT / PH_TAP5 choose the build / kernel.  usage: python tests/bench_tap5_gpu.py [B H]"""
import sys
import torch
sys.path.insert(0, ".")
from tests.test_gpu_conv import _setup
from tests.gpu_util import nhwc, hp_pack
m, L, ptr, stream, check = _setup()
B, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (64, 128)
g = torch.Generator().manual_seed(1)
x = torch.randn(B, 64, H, H, generator=g); w = torch.randn(64, 64, 3, 3, generator=g) * 0.06
ws = torch.empty(L.ph_conv2d_workspace_bytes(B, 64, H, H, 64, 3, 1, 1), device="cuda", dtype=torch.uint8)
xd = hp_pack(nhwc(x, torch.float32)); wd = w.cuda()
y = torch.empty((B, H, H, 64), device="cuda")
rg = torch.randn((B, H, H, 64), device="cuda"); ra = torch.randn((B, H, H, 64), device="cuda")
def run(kind):
    if kind == "fwd":
        return L.ph_conv2d_fwd(ptr(xd), ptr(wd), ptr(y), None, None, B, 64, H, H, 64, 3, 1, 1, 3, ptr(ws), stream())
    if kind == "dgrad_x1":
        return L.ph_conv2d_dgrad(ptr(xd), ptr(wd), ptr(y), B, 64, H, H, 64, 3, 1, 1, 4, ptr(ws), stream())
    return L.ph_conv2d_dgrad_res(ptr(xd), ptr(wd), ptr(y), ptr(rg), ptr(ra), B, 64, H, H, 64, 3, 1, 1, 4, ptr(ws), stream())
for kind in ("fwd", "dgrad_x1", "dgrad_x1_res"):
    for _ in range(3):
        check(run(kind), kind)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        run(kind)
    e1.record(); torch.cuda.synchronize()
    print("%-14s %.1f us per call (incl. the weight pack launch)" % (kind, e0.elapsed_time(e1) * 1000 / n))
