"""Micro-benchmark of the stem conv with the pooled epilogue (conv_stem.hip stem_fwd_pool_kernel) and of its ablation
builds (not a test):   make -C multimodal-learning_amd/csrc trace TRACE_TAG=_s1 EXTRA=-DPH_STEM_ABL=1   (2, 4, 8, ...)
                       python tests/bench_stem_pool_gpu.py                 # times the product library and every trace build
Calls the library-internal launcher ph_stem_fwd_pool_launch(PhStemPool*, stream) through ctypes."""
import ctypes as C
import glob
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "multimodal-learning_amd")


class PhStemPool(C.Structure):
    _fields_ = [("x4", C.c_void_p), ("w", C.c_void_p), ("wplane", C.c_size_t), ("pooled", C.c_void_p), ("stats", C.c_void_p),
                ("gamma", C.c_void_p), ("B", C.c_int), ("IH", C.c_int), ("IW", C.c_int), ("OH", C.c_int), ("OW", C.c_int),
                ("PH", C.c_int), ("PW", C.c_int), ("nsplit", C.c_int)]


def main():
    B, H = int(os.environ.get("B", 64)), 512
    OH = H // 2; PH = OH // 2
    g = torch.Generator(device="cuda").manual_seed(0)
    x4 = (torch.rand(B, H, H, 4, device="cuda", generator=g) * 2 - 1).bfloat16()
    w = (torch.randn(7 * 64 * 32, device="cuda", generator=g) * 0.05).bfloat16()
    pooled = torch.empty(B, PH, PH, 64, device="cuda", dtype=torch.bfloat16)
    stats = torch.empty(16384 * 128, device="cuda")
    gamma = torch.ones(64, device="cuda")
    libs = [os.path.join(PKG, "libpathomic_hip.so")] + sorted(glob.glob(os.path.join(PKG, "libpathomic_hip_trace_s*.so")))
    if os.environ.get("ONLY_MAIN"):
        libs = libs[:1]
    st = torch.cuda.current_stream().cuda_stream
    for path in libs:
        L = C.CDLL(path)
        fn = getattr(L, "_Z23ph_stem_fwd_pool_launchPK10PhStemPoolP12ihipStream_t")   # library-internal C++ symbol
        fn.restype = C.c_int; fn.argtypes = [C.c_void_p, C.c_void_p]
        p = PhStemPool(x4.data_ptr(), w.data_ptr(), 7 * 64 * 32, pooled.data_ptr(), stats.data_ptr(), gamma.data_ptr(),
                       B, H, H, OH, OH, PH, PH, 0)
        for _ in range(3):
            assert fn(C.byref(p), st) == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        R = 20
        for _ in range(R):
            fn(C.byref(p), st)
        e1.record(); torch.cuda.synchronize()
        print(f"{os.path.basename(path):40s} {e0.elapsed_time(e1) / R * 1e3:8.1f} us")
        if "trace_st" in path:      # PH_STEM_TRACE build: s_memtime stamps (interval start, before barrier, after barrier)
            tr = stats.view(torch.int32)[(1 << 20):(1 << 20) + 8 * 256].cpu().view(8, 256).long() & 0xffffffff
            for row in (0, 1, 2, 3):
                t = tr[row]
                t0 = int(t[0])
                print("wg", row // 2, "group", row % 2, "work / barrier-wait cycles per interval:")
                print("   ", " ".join(f"{int(t[3 * k + 1] - t[3 * k]) & 0xffffffff}/{int(t[3 * k + 2] - t[3 * k + 1]) & 0xffffffff}" for k in range(37)))
                print("    total", (int(t[3 * 36 + 2]) - t0) & 0xffffffff)
                ks = [k for k in range(6, 30) if (k - row % 2) % 2 == 1]
                print("    S+P intervals: stage / halo store / halo load issue / pool:",
                      " ".join(f"{int(t[128 + 3 * k] - t[3 * k]) & 0xffffffff}/{int(t[128 + 3 * k + 1] - t[128 + 3 * k]) & 0xffffffff}/"
                               f"{int(t[128 + 3 * k + 2] - t[128 + 3 * k + 1]) & 0xffffffff}/{int(t[3 * k + 1] - t[128 + 3 * k + 2]) & 0xffffffff}" for k in ks[:6]))


if __name__ == "__main__":
    main()
