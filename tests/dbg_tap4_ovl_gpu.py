#!/usr/bin/env python3
"""debug helper (not a test): OVL-with-operands vs sequential epilogue, repeated launches"""
import sys, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import multimodal_learning_amd as m
from multimodal_learning_amd._lib import lib, ptr, stream, check
from tests.gpu_util import nhwc
L = lib()
H, B, C = (int(sys.argv[1]), int(sys.argv[2]), 64) if len(sys.argv) > 2 else (64, 24, 64)
g = torch.Generator().manual_seed(1)
dy = nhwc(torch.randn(B, C, H, H, generator=g), torch.bfloat16)
wd = (torch.randn(C, C, 3, 3, generator=g) * (2.0 / (C * 9)) ** 0.5).cuda()
y = nhwc(torch.randn(B, C, H, H, generator=g) * 1.5 + 0.3, torch.bfloat16)
scale = (torch.rand(C, generator=g) + 0.5).cuda(); shift = (torch.randn(C, generator=g) * 0.5).cuda()
mean = (torch.randn(C, generator=g) * 0.3 + 0.3).cuda()
ws = torch.empty(L.ph_conv2d_workspace_bytes(B, C, H, H, C, 3, 1, 1) + 3 * 4 * C * 1024, device="cuda", dtype=torch.uint8)
def run():
    base = torch.full((B, H, H, C), float("nan"), device="cuda", dtype=torch.bfloat16)
    if len(sys.argv) > 3:
        check(L.ph_conv2d_dgrad(ptr(dy), ptr(wd), ptr(base), B, C, H, H, C, 3, 1, 1, 0, ptr(ws), stream()), "dgrad")
    d = torch.full((B, H, H, C), float("nan"), device="cuda", dtype=torch.bfloat16)
    sums = torch.full((3, C), float("nan"), device="cuda")
    check(L.ph_conv2d_dgrad_bnstat(ptr(dy), ptr(wd), ptr(d), None, None, ptr(y), None, None, ptr(scale), ptr(shift), ptr(mean),
                                   None, ptr(sums), B, C, H, H, C, ptr(ws), stream()), "bnstat")
    torch.cuda.synchronize()
    return d, sums
L.ph_debug_set_tap4_ovl(0)
sd, ss = run()
sd2, ss2 = run()
print("seq repeatable:", torch.equal(sd.view(torch.int16), sd2.view(torch.int16)), torch.equal(ss, ss2))
L.ph_debug_set_tap4_ovl(1)
for rep in range(4):
    od, os_ = run()
    nd = int((od.view(torch.int16) != sd.view(torch.int16)).sum())
    diff = (os_ - ss)
    print("rep", rep, "dx mismatches", nd, "sum rows differing ch:", [int((diff[r] != 0).sum()) for r in range(3)],
          "max |d|", [float(diff[r].abs().max()) for r in range(3)], "rel", float(diff[0].abs().max() / ss[0].abs().max()))
    for r in range(2):
        ch = (diff[r] != 0).nonzero().flatten().tolist()
        if ch:
            print("  row", r, "ch", ch[:64], "d", [float(diff[r][c]) for c in ch[:8]], "ref", [float(ss[r][c]) for c in ch[:8]])
