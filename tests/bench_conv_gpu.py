"""Timing of one convolution through the C-ABI test hooks: python tests/bench_conv_gpu.py Cin Cout H stride prec [B]
(prec 3 = half-pair forward; kernel time = the rocprofv3 table of this command, the bracket here includes the weight pack launch)."""
import sys
import torch
sys.path.insert(0, ".")
from tests.test_gpu_conv import _setup
from tests.gpu_util import nhwc, hp_pack
m, L, ptr, stream, check = _setup()
Cin, Cout, H, S, prec = [int(a) for a in sys.argv[1:6]]
B = int(sys.argv[6]) if len(sys.argv) > 6 else 64
g = torch.Generator().manual_seed(1)
x = torch.randn(B, Cin, H, H, generator=g); w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.06
ws = torch.empty(L.ph_conv2d_workspace_bytes(B, Cin, H, H, Cout, 3, S, 1), device="cuda", dtype=torch.uint8)
xd = hp_pack(nhwc(x, torch.float32)) if prec >= 3 else nhwc(x, torch.bfloat16 if prec == 0 else torch.float32)
wd = w.cuda()
OH = (H + 2 - 3) // S + 1
y = torch.empty((B, OH, OH, Cout), device="cuda", dtype=torch.bfloat16 if prec == 0 else torch.float32)
run = lambda: L.ph_conv2d_fwd(ptr(xd), ptr(wd), ptr(y), None, None, B, Cin, H, H, Cout, 3, S, 1, prec, ptr(ws), stream())
for _ in range(3):
    check(run(), "fwd")
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
n = 20
e0.record()
for _ in range(n):
    run()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1000 / n
fl = 2.0 * B * OH * OH * Cout * 9 * Cin * (3 if prec == 3 else 1)
print("conv %d->%d H=%d S=%d prec=%d: %.1f us per call (incl. weight pack), %.0f TFLOP/s of MFMA work" % (Cin, Cout, H, S, prec, us, fl / us * 1e-6))
