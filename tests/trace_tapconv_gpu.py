#!/usr/bin/env python3
"""Phase timeline of the tap-conv kernel (not a test; run on the GPU box after `make -C .../csrc trace`):
    python tests/trace_tapconv_gpu.py [B] [H]
Loads the debug build libpathomic_hip_trace.so (per-workgroup 100 MHz timestamps at the phase boundaries of
tapconv_kernel) and prints, per ResNet-18 conv shape, the median duration of each phase and the occupancy picture
(workgroups in flight, span of the launch)."""
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
for n in ("ph_conv2d_fwd",):
    getattr(L, n).restype = i32
    getattr(L, n).argtypes = [vp, vp, vp, vp, vp] + [i32] * 9 + [vp, vp]
L.ph_conv2d_dgrad.restype = i32
L.ph_conv2d_dgrad.argtypes = [vp, vp, vp] + [i32] * 9 + [vp, vp]
L.ph_debug_tap_trace.restype = i32
L.ph_debug_tap_trace.argtypes = [vp, i32]

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
H0 = int(sys.argv[2]) if len(sys.argv) > 2 else 512
shapes = [("layer1 3x3", 64, 64, H0 // 4, 3, 1, 1, 256, 64), ("layer2 3x3", 128, 128, H0 // 8, 3, 1, 1, 256, 128),
          ("layer3 3x3", 256, 256, H0 // 16, 3, 1, 1, 256, 128), ("layer4 3x3", 512, 512, H0 // 32, 3, 1, 1, 256, 128),
          ("layer2.0.c1 s2", 64, 128, H0 // 4, 3, 2, 1, 128, 128)]
ptr = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
names = ["prologue (first halo+weights -> LDS)", "main loop (MFMA stages)", "epilogue: acc -> LDS C tile",
         "epilogue: coalesced stores", "epilogue: BN partial stats"]
for name, Cin, Cout, H, KS, S, pad, BM, BN in shapes:
    OH = (H + 2 * pad - KS) // S + 1
    x = torch.randn(B, H, H, Cin, device="cuda").bfloat16()
    w = torch.randn(Cout, Cin, KS, KS, device="cuda") * 0.05
    y = torch.empty(B, OH, OH, Cout, device="cuda", dtype=torch.bfloat16)
    s1 = torch.empty(Cout, device="cuda"); s2 = torch.empty(Cout, device="cuda")
    ws = torch.empty(L.ph_conv2d_workspace_bytes(B, Cin, H, H, Cout, KS, S, pad), device="cuda", dtype=torch.uint8)
    for _ in range(3):
        rc = L.ph_conv2d_fwd(ptr(x), ptr(w), ptr(y), ptr(s1), ptr(s2), B, Cin, H, H, Cout, KS, S, pad, 0, ptr(ws), st)
        assert rc == 0, rc
    torch.cuda.synchronize()
    th = 16 if S == 1 else 8
    nwg = B * ((OH + th - 1) // th) * ((OH + 15) // 16) * (Cout // BN)
    nwg = min(nwg, 65536)
    buf = np.zeros((nwg, 12), dtype=np.uint64)
    assert L.ph_debug_tap_trace(buf.ctypes.data_as(vp), nwg) == 0
    t = buf[:, :6].astype(np.int64)
    t0 = t[:, 0].min()
    d = np.diff(t, axis=1) * 0.01          # us
    span = (t[:, 5].max() - t0) * 0.01
    total = (t[:, 5] - t[:, 0]) * 0.01
    # workgroups resident at the middle of the launch
    mid = t0 + (t[:, 5].max() - t0) // 2
    resident = int(((t[:, 0] <= mid) & (t[:, 5] > mid)).sum())
    hw = buf[:, 7]
    xcc = (hw >> np.uint64(32)) & np.uint64(0xF)
    cu = (hw >> np.uint64(8)) & np.uint64(0xF)
    se = (hw >> np.uint64(13)) & np.uint64(0x7)
    ncu = len(set(zip(xcc.tolist(), se.tolist(), cu.tolist())))
    fl = 2.0 * B * OH * OH * Cout * KS * KS * Cin
    print(f"{name}: Cin {Cin} Cout {Cout} HW {H}  workgroups {nwg}  launch span {span:.1f} us "
          f"({fl / span / 1e6:.0f} TFLOP/s)  resident at mid-launch {resident}  distinct CUs seen {ncu}")
    print(f"    per-workgroup lifetime: median {np.median(total):.2f} us  p10 {np.percentile(total, 10):.2f}  p90 {np.percentile(total, 90):.2f}")
    for k, nm in enumerate(names):
        print(f"    {nm:<40s} median {np.median(d[:, k]):6.2f} us   p90 {np.percentile(d[:, k], 90):6.2f}")
    ns = max(int(buf[0, 11]) & 0xff, 1)
    ca = np.median((buf[:, 11] >> np.uint64(8)) & np.uint64((1 << 28) - 1)) / ns
    cb = np.median(buf[:, 11] >> np.uint64(36)) / ns
    loop_us = np.median(d[:, 1]); loop_cyc = np.median(buf[:, 10].astype(np.float64))
    print(f"    main loop: {ns} stages, {loop_cyc:.0f} shader cycles = {loop_us:.2f} us -> {loop_cyc / loop_us / 1e3:.2f} GHz;  per stage: "
          f"compute {np.median(buf[:, 6]) / ns:.0f} cyc, wait at barrier {np.median(buf[:, 8]) / ns:.0f} cyc, "
          f"LDS refill (+2nd barrier) {np.median(buf[:, 9]) / ns:.0f} cyc, [double-buffered path: weight ds_write {ca:.0f}, load issue {cb:.0f}]   (MFMA floor: 2 waves/SIMD x taps x 16 x 32 cyc)")
    # start-time histogram: how the hardware feeds workgroups over the launch
    starts = np.sort((t[:, 0] - t0) * 0.01)
    q = [starts[int(len(starts) * f)] for f in (0.1, 0.25, 0.5, 0.75, 0.9)]
    print("    workgroup start times (us) at 10/25/50/75/90 % of the grid: " + " ".join(f"{v:.1f}" for v in q))
