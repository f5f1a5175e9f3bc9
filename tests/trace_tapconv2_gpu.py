#!/usr/bin/env python3
"""Phase accounting of the second-generation tap-conv kernel (conv_tap2.hip; not a test; needs `make trace`):
    python tests/trace_tapconv2_gpu.py [B] [H]
Per ResNet-18 3x3 stride-1 conv shape: launch span, and per persistent workgroup the shader cycles spent in the
MFMA stream (+ interleaved prefetch), waiting at the stage barrier, in the epilogues and in the halo refills."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = C.CDLL(os.path.join(ROOT, "multimodal-learning_amd", "libpathomic_hip_trace%s.so" % os.environ.get("PH_TRACE_TAG", "")))
vp, i32 = C.c_void_p, C.c_int
L.ph_conv2d_workspace_bytes.restype = C.c_size_t
L.ph_conv2d_workspace_bytes.argtypes = [i32] * 8
L.ph_conv2d_fwd.restype = i32
L.ph_conv2d_fwd.argtypes = [vp, vp, vp, vp, vp] + [i32] * 9 + [vp, vp]
L.ph_debug_tap2_trace.restype = i32
L.ph_debug_tap2_trace.argtypes = [vp, i32]

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
H0 = int(sys.argv[2]) if len(sys.argv) > 2 else 512
shapes = [("layer1 3x3", 64, 64, H0 // 4, 1), ("layer2 3x3", 128, 128, H0 // 8, 1), ("layer3 3x3", 256, 256, H0 // 16, 1),
          ("layer4 3x3", 512, 512, H0 // 32, 1),
          # stride-2 forward convs (masked grid: `stages` = live taps)
          ("layer2.0.conv1 3x3/2", 64, 128, H0 // 4, 2), ("layer3.0.conv1 3x3/2", 128, 256, H0 // 8, 2),
          ("layer4.0.conv1 3x3/2", 256, 512, H0 // 16, 2)]
ptr = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for name, Cin, Cout, H, S in shapes:
    x = torch.randn(B, H, H, Cin, device="cuda").bfloat16()
    w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.05
    y = torch.empty(B, H // S, H // S, Cout, device="cuda", dtype=torch.bfloat16)
    s1 = torch.empty(Cout, device="cuda"); s2 = torch.empty(Cout, device="cuda")
    ws = torch.empty(L.ph_conv2d_workspace_bytes(B, Cin, H, H, Cout, 3, S, 1), device="cuda", dtype=torch.uint8)
    for _ in range(3):
        assert L.ph_conv2d_fwd(ptr(x), ptr(w), ptr(y), ptr(s1), ptr(s2), B, Cin, H, H, Cout, 3, S, 1, 0, ptr(ws), st) == 0
    torch.cuda.synchronize()
    nwg = 256
    buf = np.zeros((nwg, 12), dtype=np.uint64)
    assert L.ph_debug_tap2_trace(buf.ctypes.data_as(vp), nwg) == 0
    t0 = buf[:, 0].astype(np.int64).min()
    span = (buf[:, 5].astype(np.int64).max() - t0) * 0.01
    life = (buf[:, 5].astype(np.int64) - buf[:, 0].astype(np.int64)) * 0.01
    prol = (buf[:, 1].astype(np.int64) - buf[:, 0].astype(np.int64)) * 0.01
    fl = 2.0 * B * (H // S) * (H // S) * Cout * 9 * Cin
    ns = np.median(buf[:, 11].astype(np.float64))
    loop = np.median(buf[:, 10].astype(np.float64))
    med = lambda k: np.median(buf[:, k].astype(np.float64))
    if Cin == 64 and Cout == 64 and not os.environ.get("PH_L1_ONE_GROUP"):
        # two-group layer-1 kernel: slot 11 = tiles of the workgroup; per group: MFMA phases, store phases, barrier waits
        K = ns
        print(f"{name}: span {span:.1f} us ({fl / span / 1e6:.0f} TFLOP/s), workgroup life {np.median(life):.1f} us, prologue "
              f"{np.median(prol):.2f} us, {K:.0f} tiles/workgroup, loop {loop:.0f} cyc ({loop / np.median(life - prol) / 1e3:.2f} GHz) "
              f"= {loop / (K + 1):.0f} per half-phase")
        print(f"    group 0 per tile: MFMA phase {med(6) / (K / 2):.0f} (floor 4608), store phase {med(3) / (K / 2):.0f}, barrier wait {med(8) / (K / 2):.0f};"
              f"  group 1: MFMA {med(4) / (K / 2):.0f}, store {med(9) / (K / 2):.0f}, barrier wait {med(7) / (K / 2):.0f}"
              "   (build with EXTRA=-DPH_TRACE_FINE: slots 4/9/7 = group 0's store phase split into DMA issue / epilogue / vmcnt wait)")
        continue
    ntile = max(ns / (9 * Cin / 64), 1)
    print(f"{name}: Cin {Cin} Cout {Cout} HW {H}: span {span:.1f} us ({fl / span / 1e6:.0f} TFLOP/s), workgroup life {np.median(life):.1f} us, "
          f"prologue {np.median(prol):.2f} us, {ns:.0f} stages/workgroup")
    print(f"    loop {loop:.0f} cyc = {loop / ns:.0f}/stage ({loop / np.median(life - prol) / 1e3:.2f} GHz): taps (MFMA stream + DMA issue + tap barriers) {med(6) / ns:.0f}, "
          f"pre-epilogue barrier {med(8) / ns:.0f}, epilogues {med(3) / ns:.0f} (= {med(3) / ntile:.0f} per tile), epilogue store part per tile {med(9) / ntile:.0f}, slice setup {med(4) / ntile:.0f}, tile switch {med(7) / ntile:.0f} per tile  (MFMA floor 1024/tap)")
