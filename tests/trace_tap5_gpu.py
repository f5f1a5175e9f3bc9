"""Slice timeline of conv_tap5.hip's kernel (debug build with -DPH_TAP_TRACE, PH_LIB_VARIANT selects it): per-workgroup 100 MHz
timestamps at kernel start / prologue end / the three slices + epilogue of the first two tiles / kernel end."""
import ctypes as C
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from tests.test_gpu_conv import _setup
from tests.gpu_util import nhwc, hp_pack
m, L, ptr, stream, check = _setup()
B, H = 64, 128
g = torch.Generator().manual_seed(1)
x = torch.randn(B, 64, H, H, generator=g); w = torch.randn(64, 64, 3, 3, generator=g) * 0.06
ws = torch.empty(L.ph_conv2d_workspace_bytes(B, 64, H, H, 64, 3, 1, 1), device="cuda", dtype=torch.uint8)
xd = hp_pack(nhwc(x, torch.float32)); wd = w.cuda()
y = torch.empty((B, H, H, 64), device="cuda")
for _ in range(3):
    check(L.ph_conv2d_fwd(ptr(xd), ptr(wd), ptr(y), None, None, B, 64, H, H, 64, 3, 1, 1, 3, ptr(ws), stream()), "fwd")
torch.cuda.synchronize()
nwg = 256
buf = np.zeros((nwg, 12), dtype=np.uint64)
L.ph_debug_tap5_trace.argtypes = [C.c_void_p, C.c_int]
assert L.ph_debug_tap5_trace(buf.ctypes.data_as(C.c_void_p), nwg) == 0
if len(sys.argv) > 1 and sys.argv[1] == "taps":      # build with -DPH5_TRACE_KIND=k: slot 0 = slice start, 1..9 = end of tap 0..8
    t = buf[:, :10].astype(np.int64)
    d = np.diff(t, axis=1) * 0.01
    print("per-tap medians (us):", " ".join("%.2f" % np.median(d[:, i]) for i in range(9)), " slice %.2f" % np.median((t[:, 9] - t[:, 0]) * 0.01))
    sys.exit(0)
t = buf[:, :11].astype(np.int64)
d = np.diff(t, axis=1) * 0.01
names = ["prologue", "tile0 S0 (hi.hi')", "tile0 S1 (hi.lo)", "tile0 S2 (lo.hi)", "tile0 epilogue", "tile1 S0", "tile1 S1", "tile1 S2",
         "tile1 epilogue", "tiles 2..7"]
for i, n in enumerate(names):
    print("%-20s median %7.2f us   p10 %7.2f   p90 %7.2f" % (n, np.median(d[:, i]), np.percentile(d[:, i], 10), np.percentile(d[:, i], 90)))
print("kernel span %.1f us; per-workgroup total median %.1f us" % ((t[:, 10].max() - t[:, 0].min()) * 0.01, np.median(t[:, 10] - t[:, 0]) * 0.01))
