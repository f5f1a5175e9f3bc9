"""Parity of the MFMA convolution kernels (forward / dgrad / wgrad) against the oracle's F.conv2d, through
the fine-grained C-ABI entry points ph_conv2d_*.  Bit-level expectations:
  * parity mode (bf16x6): |err| <= 2e-5 * max|ref|   (3-plane split reproduces fp32 products to ~2^-24)
  * perf mode  (bf16)   : inputs pre-rounded to bf16 on both sides; fp32 accumulate; the only differences are
    summation order and the final bf16 store (<= 2^-8 relative per element)."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CASES = [  # Cin, Cout, H, KS, stride, pad, B
    (64, 64, 32, 3, 1, 1, 2),
    (64, 128, 32, 3, 2, 1, 2),
    (128, 128, 20, 3, 1, 1, 3),      # ragged: 20 is not a multiple of the 8x16 tile
    (64, 128, 32, 1, 2, 0, 2),
    (256, 512, 14, 3, 2, 1, 2),      # 14 -> 7 (the 224^2 layer4 case)
    (512, 512, 7, 3, 1, 1, 2),
    (128, 256, 9, 3, 2, 1, 1),       # odd input size
    (64, 64, 24, 3, 1, 1, 8),        # layer-1 shape with ragged tiles (24 = 16 + 8)
    (64, 128, 24, 3, 2, 1, 4),       # stride-2 dgrad classes with 64 output channels, ragged
]


def _setup():
    import multimodal_learning_amd as m
    from multimodal_learning_amd._lib import lib, ptr, stream, check
    return m, lib(), ptr, stream, check


@pytest.mark.parametrize("prec", [1, 0])
@pytest.mark.parametrize("case", CASES)
def test_conv_fwd_dgrad_wgrad(case, prec):
    from tests.gpu_util import nhwc, nchw_cpu, assert_close
    m, L, ptr, stream, check = _setup()
    Cin, Cout, H, KS, S, pad, B = case
    g = torch.Generator().manual_seed(Cin * 7 + Cout + H)
    x = torch.randn(B, Cin, H, H, generator=g)
    w = torch.randn(Cout, Cin, KS, KS, generator=g) * (2.0 / (Cin * KS * KS)) ** 0.5
    OH = (H + 2 * pad - KS) // S + 1
    dy = torch.randn(B, Cout, OH, OH, generator=g)
    dt = torch.float32 if prec == 1 else torch.bfloat16
    if prec == 0:   # like-for-like operand rounding
        x = x.bfloat16().float(); w_r = w.bfloat16().float(); dy = dy.bfloat16().float()
    else:
        w_r = w
    xr = x.clone().requires_grad_(True); wr = w_r.clone().requires_grad_(True)
    y_ref = F.conv2d(xr, wr, None, S, pad)
    y_ref.backward(dy)
    ws = torch.empty(L.ph_conv2d_workspace_bytes(B, Cin, H, H, Cout, KS, S, pad), device="cuda", dtype=torch.uint8)
    xd = nhwc(x, dt); wd = w.cuda(); dyd = nhwc(dy, dt)
    y = torch.empty(B, OH, OH, Cout, device="cuda", dtype=dt)
    s1 = torch.empty(Cout, device="cuda"); s2 = torch.empty(Cout, device="cuda")
    check(L.ph_conv2d_fwd(ptr(xd), ptr(wd), ptr(y), ptr(s1), ptr(s2), B, Cin, H, H, Cout, KS, S, pad, prec, ptr(ws),
                          stream()), "fwd")
    tol = 2e-5 if prec == 1 else 1.0 / 128
    assert_close(y_ref.detach(), nchw_cpu(y), 1e-6, tol, "conv fwd")
    assert_close(y_ref.detach().sum(dim=(0, 2, 3)), s1.cpu(), 1e-3, 2e-4, "channel sum")
    assert_close((y_ref.detach() ** 2).sum(dim=(0, 2, 3)), s2.cpu(), 1e-3, 2e-4, "channel sumsq")
    dx = torch.full((B, H, H, Cin), float("nan"), device="cuda", dtype=dt)
    check(L.ph_conv2d_dgrad(ptr(dyd), ptr(wd), ptr(dx), B, Cin, H, H, Cout, KS, S, pad, prec, ptr(ws), stream()), "dgrad")
    assert_close(xr.grad, nchw_cpu(dx), 1e-6, tol, "conv dgrad")
    dw = torch.empty_like(wd)
    check(L.ph_conv2d_wgrad(ptr(xd), ptr(dyd), ptr(dw), B, Cin, H, H, Cout, KS, S, pad, prec, ptr(ws), stream()), "wgrad")
    assert_close(wr.grad, dw.cpu(), 1e-6, 2e-5 if prec == 1 else 2e-3, "conv wgrad")


def test_half_pair_storage_round_trip():
    """ph_hp_pack / ph_hp_unpack (PH_PREC_FP16X3 storage): x ~= hi + lo * 2^-11 keeps 22 significant bits over the fp16
    exponent range, and a power-of-two scale applied before the split is exact."""
    from tests.gpu_util import hp_pack, hp_unpack
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(3, 7, 5, 128, generator=g) * torch.exp(torch.randn(3, 7, 5, 128, generator=g) * 3)).clamp(-6e4, 6e4).cuda()
    x[0, 0, 0, :8] = torch.tensor([0.0, -0.0, 1.0, -1.0, 65000.0, 6e-5, 1e-7, -3.3e-6]).cuda()
    y = hp_unpack(hp_pack(x))
    rel = ((y - x).abs() / x.abs().clamp_min(1e-4)).max().item()
    assert rel < 2.0 ** -21, rel                       # (|x| >= 1e-4: both halves normal or the residual below 2^-25 absolute)
    assert (y - x).abs().max().item() <= 2.0 ** -21 * x.abs().max().item()
    small = x.abs() < 1e-4
    assert ((y - x).abs()[small] <= 2.0 ** -24).all()  # fp16 subnormal spacing of the hi half, absolute
    z = hp_unpack(hp_pack(x * 2.0 ** -20, scale=2.0 ** 20))
    assert torch.equal(z, y)


@pytest.mark.parametrize("case", CASES)
def test_conv_fwd_dgrad_wgrad_half_pair(case):
    """The three convolution kernels in PH_PREC_FP16X3 (fp16 pairs in, three fp16 MFMA products, fp32 out) against fp32
    F.conv2d on the same fp32 operands: 22-bit operands and fp32 accumulation leave ~1e-6 of the largest output."""
    from tests.gpu_util import nhwc, nchw_cpu, assert_close, hp_pack
    m, L, ptr, stream, check = _setup()
    Cin, Cout, H, KS, S, pad, B = case
    g = torch.Generator().manual_seed(Cin * 7 + Cout + H)
    x = torch.randn(B, Cin, H, H, generator=g)
    w = torch.randn(Cout, Cin, KS, KS, generator=g) * (2.0 / (Cin * KS * KS)) ** 0.5
    OH = (H + 2 * pad - KS) // S + 1
    dy = torch.randn(B, Cout, OH, OH, generator=g)
    xr = x.clone().requires_grad_(True); wr = w.clone().requires_grad_(True)
    y_ref = F.conv2d(xr.double(), wr.double(), None, S, pad)
    y_ref.backward(dy.double())
    ws = torch.empty(L.ph_conv2d_workspace_bytes(B, Cin, H, H, Cout, KS, S, pad), device="cuda", dtype=torch.uint8)
    xd = hp_pack(nhwc(x, torch.float32)); wd = w.cuda(); dyd = hp_pack(nhwc(dy, torch.float32))
    y = torch.full((B, OH, OH, Cout), float("nan"), device="cuda")
    s1 = torch.empty(Cout, device="cuda"); s2 = torch.empty(Cout, device="cuda")
    check(L.ph_conv2d_fwd(ptr(xd), ptr(wd), ptr(y), ptr(s1), ptr(s2), B, Cin, H, H, Cout, KS, S, pad, 3, ptr(ws),
                          stream()), "fwd")
    assert_close(y_ref.detach(), nchw_cpu(y), 1e-6, 3e-6, "conv fwd")
    assert_close(y_ref.detach().sum(dim=(0, 2, 3)), s1.cpu(), 1e-3, 2e-4, "channel sum")
    assert_close((y_ref.detach() ** 2).sum(dim=(0, 2, 3)), s2.cpu(), 1e-3, 2e-4, "channel sumsq")
    dx = torch.full((B, H, H, Cin), float("nan"), device="cuda")
    check(L.ph_conv2d_dgrad(ptr(dyd), ptr(wd), ptr(dx), B, Cin, H, H, Cout, KS, S, pad, 3, ptr(ws), stream()), "dgrad")
    assert_close(xr.grad, nchw_cpu(dx), 1e-6, 3e-6, "conv dgrad")
    dw = torch.empty_like(wd)
    check(L.ph_conv2d_wgrad(ptr(xd), ptr(dyd), ptr(dw), B, Cin, H, H, Cout, KS, S, pad, 3, ptr(ws), stream()), "wgrad")
    assert_close(wr.grad, dw.cpu(), 1e-6, 5e-6, "conv wgrad")


# ---- second-generation kernel (conv_tap2.hip): persistent workgroups walking SEVERAL tiles each.  The cases above
# give every workgroup one tile; these are sized past 256 tiles so that the cross-tile operand streams (next tile's
# halo in the other A buffer, the weight ring running across the tile edge, the epilogue between two tiles) are
# checked against F.conv2d, and the launch is repeated to catch a schedule-dependent (racy) result.
MULTITILE = [  # Cin, Cout, H, B
    (128, 128, 64, 20),     # <2,2,4,false>: 2 slices/tile, 320 tiles
    (64, 64, 64, 24),       # <4,1,2,true> (resident weights): 384 tiles
    (256, 256, 24, 40),     # ragged 24 = 16 + 8: partial tiles in the stream, 4 slices/tile, 320 tiles
    (64, 64, 40, 40),       # resident-weights configuration with partial tiles (40 = 16 + 16 + 8), 360 tiles
    (512, 512, 16, 80),     # 8 slices/tile, 4 Cout blocks: a workgroup's tile list crosses Cout blocks, 320 tiles
]


@pytest.mark.parametrize("case", MULTITILE)
def test_tapconv2_multitile_stream(case):
    from tests.gpu_util import nhwc, nchw_cpu, assert_close
    m, L, ptr, stream, check = _setup()
    Cin, Cout, H, B = case
    g = torch.Generator().manual_seed(Cin + Cout + H + B)
    x = (torch.randn(B, Cin, H, H, generator=g)).bfloat16().float()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (Cin * 9)) ** 0.5)
    w_r = w.bfloat16().float()
    dy = torch.randn(B, Cout, H, H, generator=g).bfloat16().float()
    torch.set_num_threads(8)
    y_ref = F.conv2d(x, w_r, None, 1, 1)
    dx_ref = F.conv_transpose2d(dy, w_r, None, 1, 1)
    ws = torch.empty(L.ph_conv2d_workspace_bytes(B, Cin, H, H, Cout, 3, 1, 1), device="cuda", dtype=torch.uint8)
    xd = nhwc(x, torch.bfloat16); wd = w.cuda(); dyd = nhwc(dy, torch.bfloat16)
    s1 = torch.empty(Cout, device="cuda"); s2 = torch.empty(Cout, device="cuda")
    outs = []
    for rep in range(4):
        y = torch.full((B, H, H, Cout), float("nan"), device="cuda", dtype=torch.bfloat16)
        check(L.ph_conv2d_fwd(ptr(xd), ptr(wd), ptr(y), ptr(s1), ptr(s2), B, Cin, H, H, Cout, 3, 1, 1, 0, ptr(ws),
                              stream()), "fwd")
        outs.append((y.clone(), s1.clone(), s2.clone()))
    assert_close(y_ref, nchw_cpu(outs[0][0]), 1e-6, 1.0 / 128, "multi-tile conv fwd")
    assert_close(y_ref.sum(dim=(0, 2, 3)), outs[0][1].cpu(), 1e-2, 1e-3, "channel sum")
    assert_close((y_ref ** 2).sum(dim=(0, 2, 3)), outs[0][2].cpu(), 1e-2, 1e-3, "channel sumsq")
    for rep in range(1, 4):   # bitwise repeatable: no schedule-dependent reads of a buffer still being filled
        assert torch.equal(outs[0][0].view(torch.int16), outs[rep][0].view(torch.int16)), "fwd differs between launches"
        assert torch.equal(outs[0][1], outs[rep][1]) and torch.equal(outs[0][2], outs[rep][2])
    dx = torch.full((B, H, H, Cin), float("nan"), device="cuda", dtype=torch.bfloat16)
    check(L.ph_conv2d_dgrad(ptr(dyd), ptr(wd), ptr(dx), B, Cin, H, H, Cout, 3, 1, 1, 0, ptr(ws), stream()), "dgrad")
    assert_close(dx_ref, nchw_cpu(dx), 1e-6, 1.0 / 128, "multi-tile conv dgrad")


@pytest.mark.parametrize("case", [(128, 128, 64, 20), (256, 256, 24, 40), (512, 512, 16, 80)])
def test_tapconv3_half_pair_multitile_stream(case):
    """The third-generation dense kernel in the half-pair arithmetic (conv_tap3.hip, tapconv3_kernel<false, true>: 3 x Cin / 64
    slices of fp16 operand blocks, fp32 16-byte stores): persistent workgroups walking several tiles / Cout blocks, ragged maps,
    forward (+ BatchNorm partial sums) and dgrad against fp64 F.conv2d on the same fp32 operands; repeated launches bitwise."""
    from tests.gpu_util import nhwc, nchw_cpu, assert_close, hp_pack
    m, L, ptr, stream, check = _setup()
    Cin, Cout, H, B = case
    g = torch.Generator().manual_seed(Cin + Cout + H + B + 1)
    x = torch.randn(B, Cin, H, H, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (Cin * 9)) ** 0.5
    dy = torch.randn(B, Cout, H, H, generator=g)
    torch.set_num_threads(8)
    y_ref = F.conv2d(x.double(), w.double(), None, 1, 1)
    dx_ref = F.conv_transpose2d(dy.double(), w.double(), None, 1, 1)
    ws = torch.empty(L.ph_conv2d_workspace_bytes(B, Cin, H, H, Cout, 3, 1, 1), device="cuda", dtype=torch.uint8)
    xd = hp_pack(nhwc(x, torch.float32)); wd = w.cuda(); dyd = hp_pack(nhwc(dy, torch.float32))
    s1 = torch.empty(Cout, device="cuda"); s2 = torch.empty(Cout, device="cuda")
    outs = []
    for rep in range(4):
        y = torch.full((B, H, H, Cout), float("nan"), device="cuda")
        check(L.ph_conv2d_fwd(ptr(xd), ptr(wd), ptr(y), ptr(s1), ptr(s2), B, Cin, H, H, Cout, 3, 1, 1, 3, ptr(ws), stream()), "fwd")
        outs.append((y.clone(), s1.clone(), s2.clone()))
    assert_close(y_ref, nchw_cpu(outs[0][0]), 1e-6, 3e-6, "multi-tile half-pair conv fwd")
    assert_close(y_ref.sum(dim=(0, 2, 3)), outs[0][1].cpu(), 1e-2, 1e-4, "channel sum")
    assert_close((y_ref ** 2).sum(dim=(0, 2, 3)), outs[0][2].cpu(), 1e-2, 1e-4, "channel sumsq")
    for rep in range(1, 4):
        assert torch.equal(outs[0][0], outs[rep][0]), "fwd differs between launches"
        assert torch.equal(outs[0][1], outs[rep][1]) and torch.equal(outs[0][2], outs[rep][2])
    dx = torch.full((B, H, H, Cin), float("nan"), device="cuda")
    check(L.ph_conv2d_dgrad(ptr(dyd), ptr(wd), ptr(dx), B, Cin, H, H, Cout, 3, 1, 1, 3, ptr(ws), stream()), "dgrad")
    assert_close(dx_ref, nchw_cpu(dx), 1e-6, 3e-6, "multi-tile half-pair conv dgrad")


@pytest.mark.parametrize("case", [(64, 24), (40, 40), (128, 5), (16, 3), (33, 7), (72, 9)])
def test_tapconv5_half_pair_layer1_stream(case):
    """The half-pair kernel of the 64 -> 64 convolutions (conv_tap5.hip: 32 x 16 tiles with both halo planes in LDS, weight
    fragments from global memory through a register window, hand-counted vmcnt, two barriers per tile): persistent workgroups
    walking several tiles, ragged maps (40 = 32 + 8 rows, 2.5 column tiles; 33; 72), fewer tiles than compute units, forward
    (+ BatchNorm partial sums) and dgrad (+ fused residual / mask) against fp64 F.conv2d on the same fp32 operands, the hi-only
    backward form (PH_PREC_FP16X1) at its 11-bit operand tolerance, repeated launches bitwise, and the first-generation kernel
    (PH_TAP5 off) as the second opinion."""
    from tests.gpu_util import nhwc, nchw_cpu, assert_close, hp_pack
    m, L, ptr, stream, check = _setup()
    H, B = case
    Cin = Cout = 64
    g = torch.Generator().manual_seed(H * 100 + B)
    x = torch.randn(B, Cin, H, H, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (Cin * 9)) ** 0.5
    dy = torch.randn(B, Cout, H, H, generator=g)
    res_g = torch.randn(B, H, H, Cin, generator=g)
    res_a = torch.randn(B, H, H, Cin, generator=g)
    torch.set_num_threads(8)
    y_ref = F.conv2d(x.double(), w.double(), None, 1, 1)
    dx_ref = F.conv_transpose2d(dy.double(), w.double(), None, 1, 1)
    ws = torch.empty(L.ph_conv2d_workspace_bytes(B, Cin, H, H, Cout, 3, 1, 1), device="cuda", dtype=torch.uint8)
    xd = hp_pack(nhwc(x, torch.float32)); wd = w.cuda(); dyd = hp_pack(nhwc(dy, torch.float32))
    s1 = torch.empty(Cout, device="cuda"); s2 = torch.empty(Cout, device="cuda")
    outs = []
    for rep in range(4):
        y = torch.full((B, H, H, Cout), float("nan"), device="cuda")
        check(L.ph_conv2d_fwd(ptr(xd), ptr(wd), ptr(y), ptr(s1), ptr(s2), B, Cin, H, H, Cout, 3, 1, 1, 3, ptr(ws), stream()), "fwd")
        outs.append((y.clone(), s1.clone(), s2.clone()))
    assert_close(y_ref, nchw_cpu(outs[0][0]), 1e-6, 3e-6, "layer-1 half-pair conv fwd")
    assert_close(y_ref.sum(dim=(0, 2, 3)), outs[0][1].cpu(), 1e-2, 1e-4, "channel sum")
    assert_close((y_ref ** 2).sum(dim=(0, 2, 3)), outs[0][2].cpu(), 1e-2, 1e-4, "channel sumsq")
    for rep in range(1, 4):
        assert torch.equal(outs[0][0], outs[rep][0]), "fwd differs between launches"
        assert torch.equal(outs[0][1], outs[rep][1]) and torch.equal(outs[0][2], outs[rep][2])
    L.ph_debug_set_tap5(0)
    try:
        y1 = torch.full((B, H, H, Cout), float("nan"), device="cuda")
        check(L.ph_conv2d_fwd(ptr(xd), ptr(wd), ptr(y1), ptr(s1), ptr(s2), B, Cin, H, H, Cout, 3, 1, 1, 3, ptr(ws), stream()), "fwd gen1")
    finally:
        L.ph_debug_set_tap5(1)
    assert (y1 - outs[0][0]).abs().max().item() <= 4e-6 * y_ref.abs().max().item()
    # dgrad: plain, + residual, + masked residual; 3 products and the hi-only form
    rg = res_g.cuda(); ra = res_a.cuda()
    for prec, rtol in ((3, 3e-6), (4, 3e-3)):
        dx = torch.full((B, H, H, Cin), float("nan"), device="cuda")
        check(L.ph_conv2d_dgrad(ptr(dyd), ptr(wd), ptr(dx), B, Cin, H, H, Cout, 3, 1, 1, prec, ptr(ws), stream()), "dgrad")
        assert_close(dx_ref, nchw_cpu(dx), 1e-6, rtol, "layer-1 half-pair conv dgrad prec %d" % prec)
        dx1 = torch.full((B, H, H, Cin), float("nan"), device="cuda")
        check(L.ph_conv2d_dgrad_res(ptr(dyd), ptr(wd), ptr(dx1), ptr(rg), None, B, Cin, H, H, Cout, 3, 1, 1, prec, ptr(ws), stream()), "dgrad+res")
        assert (dx1 - (dx + rg)).abs().max().item() <= 1e-6 * dx_ref.abs().max().item() + 1e-6
        dx2 = torch.full((B, H, H, Cin), float("nan"), device="cuda")
        check(L.ph_conv2d_dgrad_res(ptr(dyd), ptr(wd), ptr(dx2), ptr(rg), ptr(ra), B, Cin, H, H, Cout, 3, 1, 1, prec, ptr(ws), stream()), "dgrad+masked res")
        assert (dx2 - (dx + rg * (ra > 0))).abs().max().item() <= 1e-6 * dx_ref.abs().max().item() + 1e-6
        dx3 = torch.full((B, H, H, Cin), float("nan"), device="cuda")
        check(L.ph_conv2d_dgrad(ptr(dyd), ptr(wd), ptr(dx3), B, Cin, H, H, Cout, 3, 1, 1, prec, ptr(ws), stream()), "dgrad again")
        assert torch.equal(dx, dx3), "dgrad differs between launches"


@pytest.mark.parametrize("case", [(128, 64, 20), (256, 24, 40), (512, 16, 80), (128, 40, 9), (256, 32, 64), (128, 9, 3)])
def test_tapconv7_bitwise_equals_the_ring_kernel(case):
    """conv_tap7.hip (the plain perf-mode form of the dense 3x3 convolutions with Cin = Cout >= 128 on the register-window machinery:
    fragment-major weights four taps ahead, one barrier per slice) against conv_tap3.hip (PH_TAP7 off) - the same GEMM in the same
    summation order: forward outputs, BatchNorm sums, dgrad outputs with no / plain / masked residual BITWISE equal; persistent workgroups walking several tiles and Cout blocks, ragged maps (24, 40, 9), repeated
    launches bitwise."""
    from tests.gpu_util import nhwc
    m, L, ptr, stream, check = _setup()
    Cn, H, B = case
    g = torch.Generator().manual_seed(Cn + H + B)
    x = nhwc(torch.randn(B, Cn, H, H, generator=g), torch.bfloat16)
    dy = nhwc(torch.randn(B, Cn, H, H, generator=g), torch.bfloat16)
    rg = nhwc(torch.randn(B, Cn, H, H, generator=g), torch.bfloat16)
    ra = nhwc(torch.randn(B, Cn, H, H, generator=g).relu_(), torch.bfloat16)
    wd = (torch.randn(Cn, Cn, 3, 3, generator=g) * (2.0 / (Cn * 9)) ** 0.5).cuda()
    ws = torch.empty(L.ph_conv2d_workspace_bytes(B, Cn, H, H, Cn, 3, 1, 1), device="cuda", dtype=torch.uint8)

    def run_all():
        out = []
        y = torch.full((B, H, H, Cn), float("nan"), device="cuda", dtype=torch.bfloat16)
        s1 = torch.empty(Cn, device="cuda"); s2 = torch.empty(Cn, device="cuda")
        check(L.ph_conv2d_fwd(ptr(x), ptr(wd), ptr(y), ptr(s1), ptr(s2), B, Cn, H, H, Cn, 3, 1, 1, 0, ptr(ws), stream()), "fwd")
        out += [y, s1, s2]
        for g_, a_ in ((None, None), (rg, None), (rg, ra)):
            dx = torch.full((B, H, H, Cn), float("nan"), device="cuda", dtype=torch.bfloat16)
            check(L.ph_conv2d_dgrad_res(ptr(dy), ptr(wd), ptr(dx), ptr(g_) if g_ is not None else None, ptr(a_) if a_ is not None else None,
                                        B, Cn, H, H, Cn, 3, 1, 1, 0, ptr(ws), stream()), "dgrad")
            out.append(dx)
        torch.cuda.synchronize()
        return out

    new = run_all()
    again = run_all()
    L.ph_debug_set_tap7(0)
    try:
        old = run_all()
    finally:
        L.ph_debug_set_tap7(1)
    assert not torch.isnan(new[0].float()).any()
    for i in (0, 3, 4, 5):
        assert torch.equal(new[i].view(torch.int16), old[i].view(torch.int16)), "output %d differs from conv_tap3.hip" % i
        assert torch.equal(new[i].view(torch.int16), again[i].view(torch.int16)), "output %d differs between launches" % i
    for i in (1, 2):
        assert torch.equal(new[i], again[i]) and torch.equal(new[i], old[i]), "BatchNorm sums differ"


@pytest.mark.parametrize("case", [(64, 128, 64, 20), (64, 128, 128, 20), (128, 256, 48, 8), (256, 512, 30, 5), (128, 256, 64, 40), (256, 512, 32, 64), (64, 128, 9, 3)])
def test_tapconv6_half_pair_stride2_stream(case):
    """The half-pair kernel of the 3x3 / stride-2 forward convolutions (conv_tap6.hip: eight parity-plane images per 64-channel group
    through four LDS buffers on a compile-time DMA schedule, fragment-major weights through a register window): persistent workgroups
    walking several tiles, ragged / odd maps (24 = 16 + 8, 15, 5), 1 / 2 / 4 channel groups and Cout blocks, fewer tiles than compute
    units, against fp64 F.conv2d on the same fp32 operands (+ BatchNorm partial sums), repeated launches bitwise, and the
    first-generation kernel (PH_TAP6 off) as the second opinion."""
    from tests.gpu_util import nhwc, nchw_cpu, assert_close, hp_pack
    m, L, ptr, stream, check = _setup()
    Cin, Cout, H, B = case
    g = torch.Generator().manual_seed(Cin + Cout + H * 7 + B)
    x = torch.randn(B, Cin, H, H, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (Cin * 9)) ** 0.5
    torch.set_num_threads(8)
    y_ref = F.conv2d(x.double(), w.double(), None, 2, 1)
    OH = y_ref.shape[-1]
    ws = torch.empty(L.ph_conv2d_workspace_bytes(B, Cin, H, H, Cout, 3, 2, 1), device="cuda", dtype=torch.uint8)
    xd = hp_pack(nhwc(x, torch.float32)); wd = w.cuda()
    s1 = torch.empty(Cout, device="cuda"); s2 = torch.empty(Cout, device="cuda")
    outs = []
    for rep in range(4):
        y = torch.full((B, OH, OH, Cout), float("nan"), device="cuda")
        check(L.ph_conv2d_fwd(ptr(xd), ptr(wd), ptr(y), ptr(s1), ptr(s2), B, Cin, H, H, Cout, 3, 2, 1, 3, ptr(ws), stream()), "fwd")
        outs.append((y.clone(), s1.clone(), s2.clone()))
    assert_close(y_ref, nchw_cpu(outs[0][0]), 1e-6, 3e-6, "stride-2 half-pair conv fwd")
    assert_close(y_ref.sum(dim=(0, 2, 3)), outs[0][1].cpu(), 1e-2, 1e-4, "channel sum")
    assert_close((y_ref ** 2).sum(dim=(0, 2, 3)), outs[0][2].cpu(), 1e-2, 1e-4, "channel sumsq")
    for rep in range(1, 4):
        assert torch.equal(outs[0][0], outs[rep][0]), "fwd differs between launches"
        assert torch.equal(outs[0][1], outs[rep][1]) and torch.equal(outs[0][2], outs[rep][2])
    L.ph_debug_set_tap6(0)
    try:
        y1 = torch.full((B, OH, OH, Cout), float("nan"), device="cuda")
        check(L.ph_conv2d_fwd(ptr(xd), ptr(wd), ptr(y1), ptr(s1), ptr(s2), B, Cin, H, H, Cout, 3, 2, 1, 3, ptr(ws), stream()), "fwd gen1")
    finally:
        L.ph_debug_set_tap6(1)
    assert (y1 - outs[0][0]).abs().max().item() <= 4e-6 * y_ref.abs().max().item()


MASKED_S2 = [  # Cin, Cout, H (input), B
    (64, 128, 64, 40),      # 1 slice per plane, 32 x 32 maps: 160 tiles ... x 1 Cout block
    (64, 128, 128, 12),     # 64 x 64 maps: 192 tiles (the 512^2 layer-2 shape)
    (128, 256, 48, 48),     # 2 slices per plane, ragged 24 x 24 maps, 2 Cout blocks: 384 tiles
    (256, 512, 30, 40),     # 4 slices per plane, 15 x 15 maps (odd: the last input row / column is never read), 160 x 4 tiles
    (128, 256, 9, 3),       # one partial tile per image
]


@pytest.mark.parametrize("case", MASKED_S2)
def test_tapconv2_masked_stride2_stream(case):
    """3x3 / stride 2 / pad 1 forward in perf mode = the persistent tap-conv kernel over a MASKED 3x3 grid on the four
    pixel-parity planes of the input (conv_tap2.hip: tapconv2_kernel<2,2,4,false,true>, ph_tapconv2_setup_s2_fwd): the
    weight ring runs over the 9 live (plane, tap) pairs per 64 channels only, dead taps carry the halo DMAs.  Sized
    so that workgroups walk several tiles and Cout blocks (cross-tile weight prefetch, plane and slice boundaries for
    1 / 2 / 4 slices per plane), ragged and odd maps; against F.conv2d with the same operand rounding; statistics;
    repeated launches must be bitwise equal (a read of a buffer still being filled would not be)."""
    from tests.gpu_util import nhwc, nchw_cpu, assert_close
    m, L, ptr, stream, check = _setup()
    Cin, Cout, H, B = case
    g = torch.Generator().manual_seed(3 * Cin + Cout + H + B)
    x = (torch.randn(B, Cin, H, H, generator=g)).bfloat16().float()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (Cin * 9)) ** 0.5)
    torch.set_num_threads(8)
    y_ref = F.conv2d(x, w.bfloat16().float(), None, 2, 1)
    OH = y_ref.shape[-1]
    ws = torch.empty(L.ph_conv2d_workspace_bytes(B, Cin, H, H, Cout, 3, 2, 1), device="cuda", dtype=torch.uint8)
    xd = nhwc(x, torch.bfloat16); wd = w.cuda()
    s1 = torch.empty(Cout, device="cuda"); s2 = torch.empty(Cout, device="cuda")
    outs = []
    for rep in range(4):
        y = torch.full((B, OH, OH, Cout), float("nan"), device="cuda", dtype=torch.bfloat16)
        check(L.ph_conv2d_fwd(ptr(xd), ptr(wd), ptr(y), ptr(s1), ptr(s2), B, Cin, H, H, Cout, 3, 2, 1, 0, ptr(ws),
                              stream()), "fwd")
        outs.append((y.clone(), s1.clone(), s2.clone()))
    assert_close(y_ref, nchw_cpu(outs[0][0]), 1e-6, 1.0 / 128, "masked stride-2 conv fwd")
    assert_close(y_ref.sum(dim=(0, 2, 3)), outs[0][1].cpu(), 1e-2, 1e-3, "channel sum")
    assert_close((y_ref ** 2).sum(dim=(0, 2, 3)), outs[0][2].cpu(), 1e-2, 1e-3, "channel sumsq")
    for rep in range(1, 4):
        assert torch.equal(outs[0][0].view(torch.int16), outs[rep][0].view(torch.int16)), "fwd differs between launches"
        assert torch.equal(outs[0][1], outs[rep][1]) and torch.equal(outs[0][2], outs[rep][2])
    # round 6: the launch above went to conv_tap6b.hip (parity-plane images, register-window weights); conv_tap2.hip's masked grid
    # (PH_TAP6B off) as the second opinion: the same products summed in another order and rounded once to bf16
    L.ph_debug_set_tap6b(0)
    try:
        y1 = torch.full((B, OH, OH, Cout), float("nan"), device="cuda", dtype=torch.bfloat16)
        check(L.ph_conv2d_fwd(ptr(xd), ptr(wd), ptr(y1), ptr(s1), ptr(s2), B, Cin, H, H, Cout, 3, 2, 1, 0, ptr(ws), stream()), "fwd masked grid")
    finally:
        L.ph_debug_set_tap6b(1)
    d = (y1.float() - outs[0][0].float()).abs()
    assert d.max().item() <= 2.0 ** -7 * y_ref.abs().max().item() and d.mean().item() <= 2e-3 * y_ref.abs().mean().item(), (d.max().item(), d.mean().item())
    assert (s1 - outs[0][1]).abs().max().item() <= 1e-5 * outs[0][1].abs().max().item() + 1e-3


RES_CASES = [  # Cin (= dgrad output channels), Cout, H, B
    (64, 64, 32, 8),        # layer-1 kernel (two wave groups), full tiles
    (64, 64, 40, 12),       # layer-1 kernel, partial tiles, several tiles per group
    (128, 128, 24, 6),      # <2,2,4,false>, partial tiles
    (256, 256, 16, 4),
]


@pytest.mark.parametrize("mode", ["masked", "plain"])
@pytest.mark.parametrize("case", RES_CASES)
def test_tapconv2_dgrad_with_fused_residual(case, mode):
    """The BasicBlock backward fuses d_x = dgrad(conv1) + d_out * (out > 0) into the dgrad epilogue (res_g / res_a of
    PhTapConv; resnet_plan.hip: conv_dgrad).  Perf mode: like-for-like against F.conv_transpose2d with the same rounding
    points (the dgrad is rounded to bf16, the residual is added in fp32, the sum is rounded again)."""
    from tests.gpu_util import nhwc, nchw_cpu, assert_close
    m, L, ptr, stream, check = _setup()
    Cin, Cout, H, B = case
    g = torch.Generator().manual_seed(Cin + 3 * Cout + H + B)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (Cin * 9)) ** 0.5)
    dy = torch.randn(B, Cout, H, H, generator=g).bfloat16().float()
    res_g = torch.randn(B, Cin, H, H, generator=g).bfloat16().float()
    res_a = torch.randn(B, Cin, H, H, generator=g).relu_().bfloat16().float()      # about half the entries are zero
    torch.set_num_threads(8)
    core = F.conv_transpose2d(dy, w.bfloat16().float(), None, 1, 1).bfloat16().float()
    ref = (core + (res_g * (res_a > 0) if mode == "masked" else res_g)).bfloat16().float()
    ws = torch.empty(L.ph_conv2d_workspace_bytes(B, Cin, H, H, Cout, 3, 1, 1), device="cuda", dtype=torch.uint8)
    dx = torch.full((B, H, H, Cin), float("nan"), device="cuda", dtype=torch.bfloat16)
    gd, ad = nhwc(res_g, torch.bfloat16), nhwc(res_a, torch.bfloat16)
    dyd, wd = nhwc(dy, torch.bfloat16), w.cuda()      # (named: a temporary would be freed - and reused - before the launch)
    check(L.ph_conv2d_dgrad_res(ptr(dyd), ptr(wd), ptr(dx), ptr(gd),
                                ptr(ad) if mode == "masked" else None, B, Cin, H, H, Cout, 3, 1, 1, 0, ptr(ws), stream()),
          "dgrad_res")
    got = nchw_cpu(dx)
    # the residual term is exact; the dgrad term carries the usual bf16 noise (summation order + one rounding)
    assert_close(ref, got, 1e-6, 1.0 / 64, "dgrad + residual")
    if mode == "masked":
        mask = res_a == 0
        assert_close(core[mask], got[mask], 1e-6, 1.0 / 128, "masked-out positions carry the plain dgrad")


@pytest.mark.parametrize("prec", [1, 0])
@pytest.mark.parametrize("case", [(64, 128, 32, 1, 0, 3), (128, 256, 18, 1, 0, 2), (64, 128, 32, 3, 1, 2)])
def test_stride2_dgrad_accumulates_in_place(case, prec):
    """The downsample path of a BasicBlock backward adds its 1x1 / stride-2 dgrad INTO the gradient buffer that already
    holds the main path's dgrad (resnet_plan.hip: conv_dgrad(..., gnext, gnext, nullptr)): res_g == dx, one launch per
    output parity class, classes no tap reaches untouched.  Both precisions, first-generation kernels."""
    from tests.gpu_util import nhwc, nchw_cpu, assert_close
    m, L, ptr, stream, check = _setup()
    Cin, Cout, H, KS, pad, B = case
    g = torch.Generator().manual_seed(Cin + Cout + H + KS)
    w = torch.randn(Cout, Cin, KS, KS, generator=g) * (2.0 / (Cin * KS * KS)) ** 0.5
    OH = (H + 2 * pad - KS) // 2 + 1
    dy = torch.randn(B, Cout, OH, OH, generator=g)
    base = torch.randn(B, Cin, H, H, generator=g)
    dt = torch.float32 if prec == 1 else torch.bfloat16
    if prec == 0:
        dy = dy.bfloat16().float(); base = base.bfloat16().float(); w_r = w.bfloat16().float()
    else:
        w_r = w
    core = F.conv_transpose2d(dy, w_r, None, 2, pad, output_padding=H - ((OH - 1) * 2 - 2 * pad + KS))
    if prec == 0:
        core = core.bfloat16().float()
    ref = core + base
    dx = nhwc(base, dt).clone()
    dyd, wd = nhwc(dy, dt), w.cuda()
    ws = torch.empty(L.ph_conv2d_workspace_bytes(B, Cin, H, H, Cout, KS, 2, pad), device="cuda", dtype=torch.uint8)
    check(L.ph_conv2d_dgrad_res(ptr(dyd), ptr(wd), ptr(dx), ptr(dx), None, B, Cin, H, H, Cout, KS, 2, pad, prec, ptr(ws),
                                stream()), "dgrad_res s2")
    assert_close(ref, nchw_cpu(dx), 1e-5, 2e-5 if prec == 1 else 1.0 / 64, "stride-2 dgrad accumulated in place")


@pytest.mark.parametrize("B,H,W", [(1, 33, 47), (2, 16, 16), (1, 130, 18)])
def test_stem_dgrad_vs_conv_transpose(B, H, W):
    """ph_stem_dgrad (input gradient of the 7x7 / stride 2 / pad 3 stem conv) against torch autograd, odd and tiny sizes."""
    import multimodal_learning_amd as m
    from multimodal_learning_amd._lib import lib, check, ptr, stream
    g = torch.Generator().manual_seed(H)
    w = torch.randn(64, 3, 7, 7, generator=g) * 0.1
    x = torch.randn(B, 3, H, W, generator=g, requires_grad=True)
    y = torch.nn.functional.conv2d(x, w, None, 2, 3)
    dy = torch.randn(y.shape, generator=g)
    ref, = torch.autograd.grad(y, x, dy, retain_graph=True)
    dy_nhwc = dy.permute(0, 2, 3, 1).contiguous().cuda()
    dx = torch.empty(B, 3, H, W, device="cuda")
    wc = w.cuda().contiguous()
    check(lib().ph_stem_dgrad(ptr(dy_nhwc), ptr(wc), ptr(dx), B, H, W, 1, stream()), "ph_stem_dgrad")     # fp32 dy
    assert float((dx.cpu() - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
    dyb = dy_nhwc.bfloat16()
    check(lib().ph_stem_dgrad(ptr(dyb), ptr(wc), ptr(dx), B, H, W, 0, stream()), "ph_stem_dgrad")         # bf16 dy
    ref_b, = torch.autograd.grad(y, x, dyb.float().cpu().permute(0, 3, 1, 2))
    assert float((dx.cpu() - ref_b).abs().max()) <= 2e-5 * float(ref_b.abs().max())


TAP4_CASES = [  # H, B  (Cin = Cout = 64)
    (128, 64),      # the benchmark's layer-1 shape: 4096 tiles, 16 per workgroup
    (128, 32),      # 2048 tiles: 8 per workgroup - the steady state of the tile stream (both halo buffers and both accumulator
                    # sets several times over, the previous tile's epilogue riding in the next tile's taps)
    (32, 8),        # full tiles, one tile per workgroup
    (40, 12),       # partial tiles (40 = 16 + 16 + 8)
    (64, 24),       # 384 tiles: several tiles per workgroup, both halo buffers, XCD-contiguous lists
    (24, 40),       # ragged, 160 tiles
]


@pytest.mark.parametrize("case", TAP4_CASES)
def test_tapconv4_bitwise_equals_the_two_group_layer1_kernel(case):
    """the fourth-generation layer-1 kernel (conv_tap4.hip: one wave per SIMD, 16x16x32 fragments, resident
    weights, one barrier per tile) against tapconv2_l1_kernel on the same inputs - forward (+ BatchNorm partial sums), dgrad,
    dgrad with the masked residual, forward with the input's BatchNorm + ReLU applied in LDS are out of this entry point's
    reach and covered by the trunk tests.  Same products, same fp32 accumulation order: outputs must be BITWISE equal; the
    channel sums (another reduction tree) to fp32 rounding.  Repeated launches must be bitwise repeatable."""
    from tests.gpu_util import nhwc
    m, L, ptr, stream, check = _setup()
    H, B = case
    C = 64
    g = torch.Generator().manual_seed(H * 131 + B)
    x = nhwc(torch.randn(B, C, H, H, generator=g), torch.bfloat16)
    dy = nhwc(torch.randn(B, C, H, H, generator=g), torch.bfloat16)
    res_g = nhwc(torch.randn(B, C, H, H, generator=g), torch.bfloat16)
    res_a = nhwc(torch.randn(B, C, H, H, generator=g).relu_(), torch.bfloat16)
    wd = (torch.randn(C, C, 3, 3, generator=g) * (2.0 / (C * 9)) ** 0.5).cuda()
    ws = torch.empty(L.ph_conv2d_workspace_bytes(B, C, H, H, C, 3, 1, 1), device="cuda", dtype=torch.uint8)

    def run(which):
        out = {}
        y = torch.full((B, H, H, C), float("nan"), device="cuda", dtype=torch.bfloat16)
        s1 = torch.empty(C, device="cuda"); s2 = torch.empty(C, device="cuda")
        check(L.ph_conv2d_fwd(ptr(x), ptr(wd), ptr(y), ptr(s1), ptr(s2), B, C, H, H, C, 3, 1, 1, 0, ptr(ws), stream()), "fwd")
        out["fwd"], out["s1"], out["s2"] = y, s1, s2
        dx = torch.full((B, H, H, C), float("nan"), device="cuda", dtype=torch.bfloat16)
        check(L.ph_conv2d_dgrad(ptr(dy), ptr(wd), ptr(dx), B, C, H, H, C, 3, 1, 1, 0, ptr(ws), stream()), "dgrad")
        out["dgrad"] = dx
        for mode in ("masked", "plain"):
            d2 = torch.full((B, H, H, C), float("nan"), device="cuda", dtype=torch.bfloat16)
            check(L.ph_conv2d_dgrad_res(ptr(dy), ptr(wd), ptr(d2), ptr(res_g), ptr(res_a) if mode == "masked" else None, B, C, H, H, C,
                                        3, 1, 1, 0, ptr(ws), stream()), "dgrad_res")
            out["res_" + mode] = d2
        torch.cuda.synchronize()
        return out
    try:
        L.ph_debug_set_tap4(0)
        old = run("l1")
        L.ph_debug_set_tap4(1)
        new = run("tap4")
        again = run("tap4")
    finally:
        L.ph_debug_set_tap4(1)
    for k in ("fwd", "dgrad", "res_masked", "res_plain"):
        assert torch.isfinite(new[k].float()).all(), k
        assert torch.equal(old[k].view(torch.int16), new[k].view(torch.int16)), k + ": not bitwise the two-group kernel's output"
        assert torch.equal(new[k].view(torch.int16), again[k].view(torch.int16)), k + ": differs between launches"
    for k in ("s1", "s2"):
        assert torch.equal(new[k], again[k])
        ref = old[k].double()
        assert ((new[k].double() - ref).abs() <= 2e-5 * ref.abs().max()).all(), k


@pytest.mark.parametrize("mode", ["self_mask", "act_mask", "act_mask_two"])
@pytest.mark.parametrize("case", [(32, 8, 64), (40, 12, 64), (64, 24, 64), (24, 6, 128), (64, 20, 128), (16, 20, 256), (16, 40, 512)])
def test_fused_batchnorm_backward_sums_of_the_dgrad_kernels(case, mode):
    """the BatchNorm-backward sums taken in the dgrad epilogue (PhTapConv::bst_y) against the definition,
    evaluated in float64 from the kernel's OWN stored gradient (so that the comparison sees the reduction alone): dbeta = sum dz,
    the centred second sum = sum dz (y - mean), with dz = dx * mask; mask = the BatchNorm's own ReLU re-derived from y
    (self_mask: bn1 behind conv2's dgrad) or a block output > 0 (act_mask: bn2 behind conv1's dgrad + masked residual), and
    a second y for the downsample BatchNorm that shares the dz (act_mask_two).  The gradient itself must stay bitwise the
    un-fused launch's."""
    from tests.gpu_util import nhwc
    m, L, ptr, stream, check = _setup()
    H, B, C = case      # C = 64: conv_tap4.hip (layer 1); C >= 128: conv_tap3.hip (layers 2-4, several Cout blocks per workgroup)
    g = torch.Generator().manual_seed(H * 7 + B + len(mode) + C)
    dy = nhwc(torch.randn(B, C, H, H, generator=g), torch.bfloat16)
    wd = (torch.randn(C, C, 3, 3, generator=g) * (2.0 / (C * 9)) ** 0.5).cuda()
    y = nhwc(torch.randn(B, C, H, H, generator=g) * 1.5 + 0.3, torch.bfloat16)
    y2 = nhwc(torch.randn(B, C, H, H, generator=g) * 0.7 - 0.2, torch.bfloat16)
    act = nhwc(torch.randn(B, C, H, H, generator=g).relu_(), torch.bfloat16)
    res_g = nhwc(torch.randn(B, C, H, H, generator=g), torch.bfloat16)
    res_a = nhwc(torch.randn(B, C, H, H, generator=g).relu_(), torch.bfloat16)
    scale = (torch.rand(C, generator=g) + 0.5).cuda(); shift = (torch.randn(C, generator=g) * 0.5).cuda()
    mean = (torch.randn(C, generator=g) * 0.3 + 0.3).cuda(); mean2 = (torch.randn(C, generator=g) * 0.3 - 0.2).cuda()
    ws = torch.empty(L.ph_conv2d_workspace_bytes(B, C, H, H, C, 3, 1, 1) + 3 * 4 * C * 1024, device="cuda", dtype=torch.uint8)
    masked_res = mode != "self_mask"
    base = torch.full((B, H, H, C), float("nan"), device="cuda", dtype=torch.bfloat16)
    if masked_res:
        check(L.ph_conv2d_dgrad_res(ptr(dy), ptr(wd), ptr(base), ptr(res_g), ptr(res_a), B, C, H, H, C, 3, 1, 1, 0, ptr(ws), stream()), "dgrad_res")
    else:
        check(L.ph_conv2d_dgrad(ptr(dy), ptr(wd), ptr(base), B, C, H, H, C, 3, 1, 1, 0, ptr(ws), stream()), "dgrad")
    sums = torch.full((3, C), float("nan"), device="cuda")
    dx = torch.full((B, H, H, C), float("nan"), device="cuda", dtype=torch.bfloat16)
    runs = []
    for rep in range(2):
        check(L.ph_conv2d_dgrad_bnstat(ptr(dy), ptr(wd), ptr(dx), ptr(res_g) if masked_res else None, ptr(res_a) if masked_res else None,
                                       ptr(y), ptr(act) if masked_res else None, ptr(y2) if mode == "act_mask_two" else None,
                                       None if masked_res else ptr(scale), None if masked_res else ptr(shift), ptr(mean),
                                       ptr(mean2) if mode == "act_mask_two" else None, ptr(sums), B, C, H, H, C, ptr(ws), stream()),
              "dgrad_bnstat")
        runs.append((dx.clone(), sums.clone()))
    assert torch.equal(base.view(torch.int16), dx.view(torch.int16)), "the fused launch changed the gradient"
    assert torch.equal(runs[0][0].view(torch.int16), runs[1][0].view(torch.int16)) and torch.equal(runs[0][1], runs[1][1])
    dxf, yf = dx.double(), y.double()
    if masked_res:
        mask = act.float() > 0
    else:
        mask = (yf * scale.double() + shift.double()) > 0      # the sign of the exact value = the sign of the kernel's fused multiply-add
    dz = torch.where(mask, dxf, torch.zeros_like(dxf))
    want = torch.stack([dz.sum(dim=(0, 1, 2)), (dz * (yf - mean.double())).sum(dim=(0, 1, 2)),
                        (dz * (y2.double() - mean2.double())).sum(dim=(0, 1, 2)) if mode == "act_mask_two" else torch.zeros(C, device="cuda", dtype=torch.float64)])
    mag = torch.stack([dz.abs().sum(dim=(0, 1, 2)), (dz * (yf - mean.double())).abs().sum(dim=(0, 1, 2)),
                       (dz * (y2.double() - mean2.double())).abs().sum(dim=(0, 1, 2)) + 1e-30])
    err = ((sums.double() - want).abs() / mag.max(dim=1, keepdim=True).values).max().item()
    print(f"\nfused BatchNorm-backward sums {mode} H={H} B={B}: max error {err:.2e} of sum |terms|")
    assert err <= 3e-6, err


@pytest.mark.parametrize("case", [(128, 64), (128, 20), (40, 12), (24, 40), (16, 3)])
def test_tapconv4_epilogue_with_operands_inside_the_tile_stream(case):
    """Round 5: dgrad launches whose epilogue READS (masked residual; fused BatchNorm-backward sums with the own-ReLU mask) also
    run it inside the next tile's taps, their operands prefetched a few k-steps ahead through a rotating register window (head
    pieces from the tile's own stream).  Against the same kernel with the epilogue after the tile (ph_debug_set_tap4_ovl(0)):
    gradients AND sums bitwise (same per-lane accumulation order), on full / partial / ragged tile lists, 1 to 16 tiles per
    workgroup (the last tile of a workgroup takes the sequential path in both), repeated launches."""
    from tests.gpu_util import nhwc
    m, L, ptr, stream, check = _setup()
    H, B = case
    C = 64
    g = torch.Generator().manual_seed(H * 17 + B)
    dy = nhwc(torch.randn(B, C, H, H, generator=g), torch.bfloat16)
    wd = (torch.randn(C, C, 3, 3, generator=g) * (2.0 / (C * 9)) ** 0.5).cuda()
    y = nhwc(torch.randn(B, C, H, H, generator=g) * 1.5 + 0.3, torch.bfloat16)
    res_g = nhwc(torch.randn(B, C, H, H, generator=g), torch.bfloat16)
    res_a = nhwc(torch.randn(B, C, H, H, generator=g).relu_(), torch.bfloat16)
    scale = (torch.rand(C, generator=g) + 0.5).cuda(); shift = (torch.randn(C, generator=g) * 0.5).cuda()
    mean = (torch.randn(C, generator=g) * 0.3 + 0.3).cuda()
    ws = torch.empty(L.ph_conv2d_workspace_bytes(B, C, H, H, C, 3, 1, 1) + 3 * 4 * C * 1024, device="cuda", dtype=torch.uint8)

    def run():
        d_res = torch.full((B, H, H, C), float("nan"), device="cuda", dtype=torch.bfloat16)
        check(L.ph_conv2d_dgrad_res(ptr(dy), ptr(wd), ptr(d_res), ptr(res_g), ptr(res_a), B, C, H, H, C, 3, 1, 1, 0, ptr(ws), stream()), "dgrad_res")
        d_bst = torch.full((B, H, H, C), float("nan"), device="cuda", dtype=torch.bfloat16)
        sums = torch.full((3, C), float("nan"), device="cuda")
        check(L.ph_conv2d_dgrad_bnstat(ptr(dy), ptr(wd), ptr(d_bst), None, None, ptr(y), None, None, ptr(scale), ptr(shift), ptr(mean),
                                       None, ptr(sums), B, C, H, H, C, ptr(ws), stream()), "dgrad_bnstat")
        torch.cuda.synchronize()
        return d_res, d_bst, sums
    try:
        L.ph_debug_set_tap4_ovl(0)
        seq = run()
        L.ph_debug_set_tap4_ovl(1)
        ovl = run()
        again = run()
    finally:
        L.ph_debug_set_tap4_ovl(1)
    for k, name in enumerate(("dgrad + masked residual", "dgrad under fused sums")):
        assert torch.isfinite(ovl[k].float()).all(), name
        assert torch.equal(seq[k].view(torch.int16), ovl[k].view(torch.int16)), name + ": differs from the epilogue after the tile"
        assert torch.equal(ovl[k].view(torch.int16), again[k].view(torch.int16)), name + ": differs between launches"
    assert torch.isfinite(ovl[2]).all()
    assert torch.equal(seq[2], ovl[2]), "fused sums differ from the epilogue after the tile: max |d| = %g" % float((seq[2] - ovl[2]).abs().max())
    assert torch.equal(ovl[2], again[2])


@pytest.mark.parametrize("case", [(128, 64, 64), (128, 32, 64), (40, 12, 64), (64, 24, 64), (64, 20, 128), (24, 40, 256)])
def test_input_batchnorm_applied_in_lds_equals_the_separate_pass(case):
    """PhTapConv::in_scale at the kernel level (the trunk-level test_fused_input_batchnorm_equals_separate_pass sees it through
    a whole network): conv(relu(x * s + h) -> bf16) with the map applied to each halo tile in LDS must be BITWISE the
    convolution of the tensor the stand-alone pass writes, deep in the tile stream (8 tiles per workgroup), on partial
    tiles, in repeated launches - outputs and BatchNorm partial sums."""
    from tests.gpu_util import nhwc
    m, L, ptr, stream, check = _setup()
    H, B, C = case
    g = torch.Generator().manual_seed(H + 3 * B + C)
    xr = torch.randn(B, C, H, H, generator=g) * 1.3
    x = nhwc(xr, torch.bfloat16)
    sc = (torch.rand(C, generator=g) + 0.5).cuda(); sh = (torch.randn(C, generator=g) * 0.4).cuda()
    wd = (torch.randn(C, C, 3, 3, generator=g) * (2.0 / (C * 9)) ** 0.5).cuda()
    # what the stand-alone pass stores: one fused multiply-add in fp32 (evaluated here in float64 and rounded once), ReLU, bf16
    a = torch.relu((x.double() * sc.double() + sh.double()).float()).bfloat16()
    ws = torch.empty(L.ph_conv2d_workspace_bytes(B, C, H, H, C, 3, 1, 1), device="cuda", dtype=torch.uint8)
    y0 = torch.full((B, H, H, C), float("nan"), device="cuda", dtype=torch.bfloat16)
    s1 = torch.empty(C, device="cuda"); s2 = torch.empty(C, device="cuda")
    check(L.ph_conv2d_fwd(ptr(a), ptr(wd), ptr(y0), ptr(s1), ptr(s2), B, C, H, H, C, 3, 1, 1, 0, ptr(ws), stream()), "fwd")
    ref = (y0.clone(), s1.clone(), s2.clone())
    for rep in range(3):
        y1 = torch.full((B, H, H, C), float("nan"), device="cuda", dtype=torch.bfloat16)
        t1 = torch.empty(C, device="cuda"); t2 = torch.empty(C, device="cuda")
        check(L.ph_conv2d_fwd_fused_in(ptr(x), ptr(sc), ptr(sh), ptr(wd), ptr(y1), ptr(t1), ptr(t2), B, C, H, H, C, ptr(ws), stream()), "fused in")
        bad = (ref[0].view(torch.int16) != y1.view(torch.int16))
        nbad = int(bad.sum().item())
        if nbad:
            idx = bad.nonzero()
            print("mismatch rep", rep, "count", nbad, "of", y1.numel(), "first", idx[:6].tolist(), "last", idx[-3:].tolist(),
                  "images", sorted(set(idx[:, 0].tolist()))[:10], "rows", sorted(set(idx[:, 1].tolist()))[:20], "cols", sorted(set(idx[:, 2].tolist()))[:20])
        assert nbad == 0, (rep, nbad, y1.numel())
        assert torch.equal(ref[1], t1) and torch.equal(ref[2], t2), rep
