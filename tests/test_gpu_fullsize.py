"""Checks AT THE BENCHMARKED SIZES (BASELINE configs[1]: 64 tiles of 512 x 512 per GPU; the north-star point: 256 tiles).

The parity tests elsewhere run 4-16 tiles of 64-224 pixels: one tile of work per workgroup, small maps.  At 512 x 512
the persistent tap-convs walk many tiles per workgroup, layer 1 runs 128 x 128 maps, the stem writes half a gigabyte:
other code paths of the same kernels.  Here:

* every stage of the trunk forward (20 convolutions, 17 BatchNorm / ReLU / residual / pooling passes) is compared like
  for like - its reference is computed by plain fp32 PyTorch from the GPU's OWN input to that stage, read out of the
  plan workspace - so nothing is carried from stage to stage and each kernel family is held to its arithmetic's
  rounding (bf16 store in perf mode, fp32 in parity mode); convolutions on a subset of the images (they are independent
  per image), BatchNorm over the whole batch;
* forward / dgrad / wgrad of the convolution kernels at the benchmark's layer shapes through the fine-grained C-ABI
  entry points against PyTorch's own convolution gradients;
* the student's backward in perf mode against parity mode on the same weights (per-tensor cosine, as in
  test_gpu_resnet.py at 160 x 160)."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _student():
    import multimodal_learning_amd as m
    from oracle import weights as W
    from oracle.step import default_opt
    net = m.define_net(default_opt(), 1, path_only=True)
    net.load_state_dict(W.make_state_dict(W.student_shapes(), 1))
    return net.cuda()


def _images(B, H, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    base = torch.rand(B, 3, H // 8, H // 8, device="cuda", generator=g) * 2 - 1          # smooth structure + pixel noise
    x = F.interpolate(base, size=(H, H), mode="bilinear", align_corners=False)
    return (x + 0.2 * torch.randn(B, 3, H, H, device="cuda", generator=g)).clamp_(-1, 1).contiguous()


@pytest.mark.parametrize("mode,B", [("bf16", 64), ("bf16x6", 64), ("bf16", 256)])
def test_every_forward_stage_at_benchmark_size(mode, B):
    import multimodal_learning_amd as m
    from multimodal_learning_amd._lib import lib, check
    H, SUB = 512, 3                                   # reference convolutions on the first / middle / last image
    m.set_precision(mode)
    try:
        net = _student()
        net.train()
        x = _images(B, H, 17)
        sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
        f3, feat, hazard, pred, _ = net(x_path=x)            # requires grad -> the workspace persists until backward
        plan = net._get_plan(B, H, H)
        ws = f3.grad_fn.ws
        assert ws is not None and ws.numel() == plan.ws_bytes
        act = torch.bfloat16 if mode == "bf16" else torch.float32
        es = 2 if mode == "bf16" else 4
        sub = torch.tensor([0, B // 2, B - 1], device="cuda")[:SUB]

        def tensor(what, idx, rows=None):
            off = C.c_size_t(0)
            dims = (C.c_int * 4)()
            check(lib().ph_resnet_tensor_info(plan.h, what, idx, C.byref(off), dims), "tensor_info")
            n = dims[0] * dims[1] * dims[2] * dims[3]
            t = ws[off.value: off.value + es * n].view(act).view(dims[0], dims[1], dims[2], dims[3])
            if rows is not None:
                t = t[rows]
            return t.float().permute(0, 3, 1, 2).contiguous()            # NCHW fp32 copy

        ctol, btol = (1.0 / 128, 1.0 / 64) if mode == "bf16" else (5e-5, 1e-4)
        rb = (lambda t: t.bfloat16().float()) if mode == "bf16" else (lambda t: t)
        worst = {}

        def rel(ref, got, what, tol):
            e = ((ref - got).abs().max() / ref.abs().max().clamp_min(1e-30)).item()
            worst[what] = e
            assert e <= tol, (what, e, tol)

        def bn_consts(z, prefix):   # train-mode BatchNorm constants from the batch statistics of the GPU's own z (fp64)
            zd = z.double()
            mu = zd.mean(dim=(0, 2, 3))
            var = (zd * zd).mean(dim=(0, 2, 3)) - mu * mu
            sc = sd[prefix + ".weight"].double() / torch.sqrt(var.clamp_min(0) + 1e-5)
            return sc.float().view(1, -1, 1, 1), (sd[prefix + ".bias"].double() - mu * sc).float().view(1, -1, 1, 1)

        # ---- stem: conv 7x7/2 on the image subset; BN + ReLU + max-pool over the whole batch from the stored output
        y0 = tensor(0, 0)
        rel(F.conv2d(rb(x[sub]), rb(sd["conv1.weight"]), None, 2, 3), y0[sub], "stem conv", ctol)
        sc, sh = bn_consts(y0, "bn1")
        cur = tensor(3, 0)
        rel(F.max_pool2d(F.relu(y0 * sc + sh), 3, 2, 1), cur, "stem bn+relu+maxpool", btol)
        del y0
        unit = 1
        for li in range(1, 5):
            for bi in range(2):
                p = f"layer{li}.{bi}"
                blk = (li - 1) * 2 + bi
                stride = 2 if (li > 1 and bi == 0) else 1
                u1, u2 = unit, unit + 1
                has_ds = (p + ".downsample.0.weight") in sd
                unit += 3 if has_ds else 2
                y1 = tensor(0, u1)
                rel(F.conv2d(cur[sub], rb(sd[p + ".conv1.weight"]), None, stride, 1), y1[sub], p + ".conv1", ctol)
                sc, sh = bn_consts(y1, p + ".bn1")
                a1 = tensor(2, blk)
                rel(F.relu(y1 * sc + sh), a1, p + " bn1+relu", btol)
                del y1
                y2 = tensor(0, u2)
                rel(F.conv2d(a1[sub], rb(sd[p + ".conv2.weight"]), None, 1, 1), y2[sub], p + ".conv2", ctol)
                del a1
                sc, sh = bn_consts(y2, p + ".bn2")
                idt = cur
                if has_ds:
                    yd = tensor(0, u2 + 1)
                    rel(F.conv2d(cur[sub], rb(sd[p + ".downsample.0.weight"]), None, stride, 0), yd[sub], p + ".downsample", ctol)
                    scd, shd = bn_consts(yd, p + ".downsample.1")
                    idt = yd * scd + shd
                out = tensor(1, blk)
                rel(F.relu(y2 * sc + sh + idt), out, p + " bn2+residual+relu", btol)
                del y2, idt
                cur = out
                if blk == 5:
                    rel(cur.mean(dim=(2, 3)), f3.detach(), "f3 = avgpool(layer3)", 1e-4 if mode == "bf16x6" else 1e-3)
        # head: avgpool -> fc -> BN1d(train) -> ReLU -> fc
        f4 = cur.mean(dim=(2, 3))
        h = F.linear(f4, sd["fc_new1.0.weight"], sd["fc_new1.0.bias"])
        mu, var = h.mean(0), h.var(0, unbiased=False)
        ft = F.relu((h - mu) / torch.sqrt(var + 1e-5) * sd["fc_new1.1.weight"] + sd["fc_new1.1.bias"])
        rel(ft, feat.detach(), "head features", 2e-3)
        rel(F.linear(ft, sd["fc_new2.weight"], sd["fc_new2.bias"]), hazard.detach(), "logits", 2e-3)
        assert torch.isfinite(pred).all()
        w = max(worst, key=worst.get)
        print(f"\n{mode} B={B} 512x512: {len(worst)} stages like for like, worst {worst[w]:.2e} at {w}")
    finally:
        m.set_precision("bf16")


# the benchmark's layer shapes: (Cin, Cout, H of the INPUT map, KS, stride, pad); batch 64
LAYER_SHAPES = [
    (64, 64, 128, 3, 1, 1),      # layer 1: the two-group resident-weights kernel, 4096 tiles
    (64, 128, 128, 3, 2, 1),     # layer2.0.conv1: stride-2 forward, parity-class dgrad
    (64, 128, 128, 1, 2, 0),     # layer2.0.downsample
    (128, 128, 64, 3, 1, 1),     # layer 2
    (256, 256, 32, 3, 1, 1),     # layer 3
    (256, 512, 32, 3, 2, 1),     # layer4.0.conv1
    (512, 512, 16, 3, 1, 1),     # layer 4: a workgroup's tile list crosses Cout blocks
]


@pytest.mark.parametrize("case", LAYER_SHAPES)
def test_conv_kernels_at_benchmark_layer_shapes(case):
    """Forward (+ BatchNorm sums), dgrad, wgrad of one benchmark layer, perf mode, batch 64, against PyTorch's fp32
    convolution and its autograd on the same bf16-rounded operands (references on the GPU)."""
    from multimodal_learning_amd._lib import lib, ptr, stream, check
    L = lib()
    Cin, Cout, H, KS, S, pad = case
    B = 64
    g = torch.Generator(device="cuda").manual_seed(Cin + 3 * Cout + H + KS)
    x = torch.randn(B, Cin, H, H, device="cuda", generator=g).bfloat16().float()
    w = torch.randn(Cout, Cin, KS, KS, device="cuda", generator=g) * (2.0 / (Cin * KS * KS)) ** 0.5
    OH = (H + 2 * pad - KS) // S + 1
    dy = torch.randn(B, Cout, OH, OH, device="cuda", generator=g).bfloat16().float()
    xr = x.clone().requires_grad_(True)
    wr = w.bfloat16().float().requires_grad_(True)
    y_ref = F.conv2d(xr, wr, None, S, pad)
    y_ref.backward(dy)
    ws = torch.empty(L.ph_conv2d_workspace_bytes(B, Cin, H, H, Cout, KS, S, pad), device="cuda", dtype=torch.uint8)
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().bfloat16()       # noqa: E731
    back = lambda t: t.float().permute(0, 3, 1, 2)                        # noqa: E731
    xd, dyd = nhwc(x), nhwc(dy)
    y = torch.full((B, OH, OH, Cout), float("nan"), device="cuda", dtype=torch.bfloat16)
    s1 = torch.empty(Cout, device="cuda"); s2 = torch.empty(Cout, device="cuda")
    check(L.ph_conv2d_fwd(ptr(xd), ptr(w), ptr(y), ptr(s1), ptr(s2), B, Cin, H, H, Cout, KS, S, pad, 0, ptr(ws), stream()), "fwd")

    def rel(ref, got):
        return ((ref - got).abs().max() / ref.abs().max()).item()
    yr = y_ref.detach()
    assert rel(yr, back(y)) <= 1.0 / 128, ("fwd", rel(yr, back(y)))
    asum = yr.abs().double().sum(dim=(0, 2, 3)).float()       # (a channel sum may cancel: judged against sum |y|)
    assert ((yr.double().sum(dim=(0, 2, 3)).float() - s1).abs() <= 1e-4 * asum + 1e-3).all(), "channel sums"
    assert rel((yr.double() ** 2).sum(dim=(0, 2, 3)).float(), s2) <= 1e-3, "channel sums of squares"
    dx = torch.full((B, H, H, Cin), float("nan"), device="cuda", dtype=torch.bfloat16)
    check(L.ph_conv2d_dgrad(ptr(dyd), ptr(w), ptr(dx), B, Cin, H, H, Cout, KS, S, pad, 0, ptr(ws), stream()), "dgrad")
    assert rel(xr.grad, back(dx)) <= 1.0 / 128, ("dgrad", rel(xr.grad, back(dx)))
    dw = torch.empty_like(w)
    check(L.ph_conv2d_wgrad(ptr(xd), ptr(dyd), ptr(dw), B, Cin, H, H, Cout, KS, S, pad, 0, ptr(ws), stream()), "wgrad")
    assert rel(wr.grad, dw) <= 2e-3, ("wgrad", rel(wr.grad, dw))


def test_student_backward_perf_vs_parity_at_benchmark_size():
    """BASELINE configs[1] (64 x 512 x 512): every weight gradient of the perf-mode (bf16) backward against the
    parity-mode one on the same weights and images.  bf16 arithmetic alone decorrelates the gradients of this untrained
    train-mode-BN network to cosine ~0.8-0.9 (the CPU oracle's bf16 emulation shows the same at small size), uniformly
    over the depth; a kernel bug shows as one layer far below the others (test_gpu_resnet.py explains the history)."""
    import multimodal_learning_amd as m
    B, H = 64, 512
    x = _images(B, H, 23)
    wf = torch.linspace(0.5, 1.5, 128).cuda()
    grads = {}
    try:
        for mode in ("bf16x6", "bf16"):
            m.set_precision(mode)
            net = _student()
            net.train()
            f3, feat, hazard, pred, _ = net(x_path=x)
            loss = (feat * wf).sum() + (hazard * torch.tensor([1.0, -2.0, 0.5]).cuda()).sum() + 0.1 * f3.sum()
            loss.backward()
            grads[mode] = {k: p.grad.detach().float().clone() for k, p in net.named_parameters() if p.grad is not None}
            assert all(torch.isfinite(v).all() for v in grads[mode].values()), mode
            net.release_workspaces()
            del net
            torch.cuda.empty_cache()
    finally:
        m.set_precision("bf16")
    cos = {}
    for k, gp in grads["bf16x6"].items():
        gq = grads["bf16"][k]
        if k == "fc_new1.0.bias" or gp.norm().item() < 1e-6 or gp.numel() < 64:
            continue
        cos[k] = torch.dot(gp.flatten(), gq.flatten()).item() / (gp.norm().item() * gq.norm().item() + 1e-30)
    big = {k: v for k, v in cos.items() if grads["bf16"][k].numel() >= 4096}
    small = {k: v for k, v in cos.items() if k not in big}
    vals = sorted(big.values())
    med = vals[len(vals) // 2]
    worst, worst_s = min(big, key=big.get), min(small, key=small.get)
    print(f"\nB=64 512x512 perf vs parity gradients: {len(big)} weight tensors median cosine {med:.4f}, worst {big[worst]:.4f} at "
          f"{worst}; {len(small)} BN tensors worst {small[worst_s]:.4f} at {worst_s}")
    assert big[worst] >= 0.70 and big[worst] >= med - 0.12, (worst, big[worst], med)
    assert small[worst_s] >= 0.50, (worst_s, small[worst_s])


def test_every_backward_stage_at_benchmark_size():
    """VERDICT r02 missing 3: the perf-mode (bf16) BACKWARD like for like at BASELINE configs[1] (64 x 512 x 512) - autograd
    of resnets.py:58-74,219-222 one node at a time.  ph_resnet_backward_debug cuts the backward off after k launch
    groups; after each cut the newest output (BatchNorm-backward dy + dgamma / dbeta of all 20 BatchNorm units incl. the
    stem, the weight gradient of all 20 convolutions incl. `wgrad_reduce`, every dgrad incl. the residual-masked and the
    in-place downsample ones, the avgpool scatters, the stem's pooled-gradient scatter) is compared with plain fp32
    PyTorch fed the GPU's OWN inputs of that stage (read out of the workspace), so nothing is carried from stage to stage:
    bf16-stored tensors are held to bf16 rounding (2^-7 of the tensor's maximum), fp32 parameter gradients to 2e-3.
    ReLU masks are re-derived from bf16 data exactly as the kernels do; an element whose pre-activation is within
    rounding of zero may still fall on the other side (fused multiply-add vs two roundings), so mask-dependent tensors
    may hold a handful (<= 1e-6 of the elements) of outliers, which are counted and printed."""
    import multimodal_learning_amd as m
    from multimodal_learning_amd import ops
    from multimodal_learning_amd._lib import lib, check, ptr, stream
    B, H = 64, 512
    m.set_precision("bf16")
    net = _student()
    net.train()
    x = _images(B, H, 29)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    f3, feat, hazard, pred, _ = net(x_path=x)
    ctx = f3.grad_fn
    plan, ws, packed, table = ctx.plan, ctx.ws, ctx.packed, ctx.table
    gen = torch.Generator(device="cuda").manual_seed(3)
    g4 = torch.randn(B, 512, device="cuda", generator=gen)
    g3 = 0.5 * torch.randn(B, 256, device="cuda", generator=gen)
    params = net._trunk_params()
    grads = [torch.full_like(p, float("nan")) for p in params]
    ga = ops.void_array([g.data_ptr() for g in grads])
    L = lib()
    rb = lambda t: t.bfloat16().float()      # noqa: E731
    bf = torch.bfloat16

    def run(stop):
        check(L.ph_resnet_backward_debug(plan.h, table, ptr(packed), ptr(ws), ptr(g3), ptr(g4), ga, stop, stream()), "backward_debug")

    def info(what, idx):
        off = C.c_size_t(0)
        dims = (C.c_int * 4)()
        check(L.ph_resnet_tensor_info(plan.h, what, idx, C.byref(off), dims), "tensor_info")
        return off.value, tuple(dims)

    def act(what, idx, dims=None):           # bf16 NHWC tensor of the workspace -> NCHW fp32 copy
        off, d = info(what, idx)
        d = dims or d
        n = d[0] * d[1] * d[2] * d[3]
        return ws[off: off + 2 * n].view(bf).view(*d).float().permute(0, 3, 1, 2).contiguous()

    def stats(unit):                         # mean, invstd, scale, shift of the unit's BatchNorm, each [1,C,1,1]
        off, d = info(8, unit)
        t = ws[off: off + 16 * d[1]].view(torch.float32).view(4, d[1]).clone()
        return [t[i].view(1, -1, 1, 1) for i in range(4)]

    worst, outliers = {}, {}

    def rel(ref, got, what, tol, allow=0.0):
        scale_ = ref.abs().max().clamp_min(1e-30)
        err = (ref - got).abs()
        nbad = int((err > tol * scale_).sum().item())
        ok = nbad <= allow * ref.numel() + (4 if allow > 0 else 0)
        e = (err.max() / scale_).item()
        if allow > 0 and nbad:
            outliers[what] = nbad
            e = (torch.where(err > tol * scale_, torch.zeros_like(err), err).max() / scale_).item()
        worst[what] = e
        assert ok and torch.isfinite(got).all(), (what, e, nbad, ref.numel())

    def unit_names():
        names = ["conv1|bn1"]
        for li in range(1, 5):
            for bi in range(2):
                p = f"layer{li}.{bi}"
                names += [f"{p}.conv1|{p}.bn1", f"{p}.conv2|{p}.bn2"]
                if (p + ".downsample.0.weight") in sd:
                    names.append(f"{p}.downsample.0|{p}.downsample.1")
        return names
    names = unit_names()

    def bn_stage(u, g, mask, y, dims, what, stage, dy_stage=None):
        """One BatchNorm-backward launch group: reference from (g, mask, y, the unit's statistics).  (The stem's sums and
        its apply pass are two launch groups: `dy_stage`.)"""
        mean, invstd, _, _ = stats(u)
        gamma = sd[names[u].split("|")[1] + ".weight"].view(1, -1, 1, 1)
        dz = torch.where(mask, g, torch.zeros_like(g))
        xh = (y - mean) * invstd
        n = float(g.shape[0] * g.shape[2] * g.shape[3])
        dbeta = dz.double().sum(dim=(0, 2, 3))
        dgamma = (dz.double() * xh.double()).sum(dim=(0, 2, 3))
        ref = gamma * invstd * (dz - (dbeta / n).float().view(1, -1, 1, 1) - xh * (dgamma / n).float().view(1, -1, 1, 1))
        run(stage)
        asum = dz.abs().double().sum(dim=(0, 2, 3)).float().max()      # a channel sum may cancel: judged against sum |dz|
        e1 = ((grads[3 * u + 2] - dbeta.float()).abs().max() / asum).item()
        e2 = ((grads[3 * u + 1] - dgamma.float()).abs().max() / (dz.abs() * xh.abs()).double().sum(dim=(0, 2, 3)).float().max()).item()
        worst[what + " dbeta"], worst[what + " dgamma"] = e1, e2
        assert e1 <= 3e-4 and e2 <= 3e-4, (what, e1, e2)
        if dy_stage is not None:
            run(dy_stage)
        dy = act(6, 0, dims)
        rel(ref, dy, what + " dy", 1.0 / 128, allow=1e-6)
        return dy

    def conv_grads(u, xin, dy, stride, pad, what, stage_w, stage_d, dx_dims, dx_what, extra=None):
        """wgrad launch group, then the dgrad launch group; reference = PyTorch's own convolution autograd."""
        wname = names[u].split("|")[0] + ".weight"
        xr = xin.clone().requires_grad_(stage_d is not None)
        wr = rb(sd[wname]).requires_grad_(True)
        F.conv2d(xr, wr, None, stride, pad).backward(dy)
        run(stage_w)
        rel(wr.grad, grads[3 * u], what + " wgrad", 2e-3)
        if stage_d is None:
            return None
        run(stage_d)
        ref = xr.grad if extra is None else xr.grad + extra
        dx = act(dx_what, 0, dx_dims)
        rel(ref, dx, what + " dgrad", 1.0 / 128, allow=1e-6 if extra is not None else 0.0)
        return dx

    try:
        stage = 1
        run(stage)                                                        # avgpool backward of f4 into the first buffer
        off, d7 = info(1, 7)
        gcur = act(4, 0, d7)
        rel((g4 / (d7[1] * d7[2])).view(B, 512, 1, 1).expand(-1, -1, d7[1], d7[2]), gcur, "avgpool bwd f4", 1.0 / 128)
        unit_of = {}
        u = 1
        for li in range(1, 5):
            for bi in range(2):
                has_ds = (f"layer{li}.{bi}.downsample.0.weight") in sd
                unit_of[(li - 1) * 2 + bi] = (u, u + 1, u + 2 if has_ds else -1)
                u += 3 if has_ds else 2
        nbn = 0
        for blk in range(7, -1, -1):
            li, bi = blk // 2 + 1, blk % 2
            p = f"layer{li}.{bi}"
            u1, u2, uds = unit_of[blk]
            stride = 2 if (li > 1 and bi == 0) else 1
            cur_what, nxt_what = (4, 5) if (7 - blk) % 2 == 0 else (5, 4)
            _, dout = info(1, blk)
            din = info(1, blk - 1)[1] if blk > 0 else info(3, 0)[1]
            if blk == 5:
                stage += 1
                run(stage)
                ref = gcur + (g3 / (dout[1] * dout[2])).view(B, 256, 1, 1)
                gcur = act(cur_what, 0, dout)
                rel(ref, gcur, "avgpool bwd f3 (accumulate)", 1.0 / 128)
            out = act(1, blk)
            omask = out > 0
            # bn2
            stage += 1
            dy2 = bn_stage(u2, gcur, omask, act(0, u2), dout, p + ".bn2", stage); nbn += 1
            a1 = act(2, blk)
            dab = conv_grads(u2, a1, dy2, 1, 1, p + ".conv2", stage + 1, stage + 2, dout, 7)
            stage += 2
            del a1, dy2
            # bn1: the ReLU mask is re-derived from y1 with the forward's scale / shift (bn_act.hip DzPlain::mscale)
            y1 = act(0, u1)
            _, _, sc1, sh1 = stats(u1)
            stage += 1
            dy1 = bn_stage(u1, dab, (y1 * sc1 + sh1) > 0, y1, dout, p + ".bn1", stage); nbn += 1
            del y1, dab
            xin = act(1, blk - 1) if blk > 0 else act(3, 0)
            if uds < 0:
                gnext = conv_grads(u1, xin, dy1, stride, 1, p + ".conv1 (+ residual)", stage + 1, stage + 2, din, nxt_what,
                                   extra=torch.where(omask, gcur, torch.zeros_like(gcur)))
                stage += 2
            else:
                gnext = conv_grads(u1, xin, dy1, stride, 1, p + ".conv1", stage + 1, stage + 2, din, nxt_what)
                stage += 3
                dyd = bn_stage(uds, gcur, omask, act(0, uds), dout, p + ".downsample.1", stage); nbn += 1
                gnext = conv_grads(uds, xin, dyd, stride, 0, p + ".downsample.0 (in place)", stage + 1, stage + 2, din, nxt_what,
                                   extra=gnext)
                stage += 2
                del dyd
            del dy1, xin, out, omask
            gcur = gnext
        # ---- stem: pooled gradient -> arg-max scatter -> ReLU mask -> BatchNorm backward -> weight gradient
        y0 = act(0, 0)
        mean, invstd, sc0, sh0 = stats(0)
        offi, di = info(9, 0)
        code = ws[offi: offi + di[0] * di[1] * di[2] * di[3]].view(di).permute(0, 3, 1, 2).long()     # [B,64,PH,PW], kh*3+kw
        PH, PW = di[1], di[2]
        OH, OW = y0.shape[2], y0.shape[3]
        hh = (2 * torch.arange(PH, device="cuda") - 1).view(1, 1, PH, 1) + code // 3
        wwp = (2 * torch.arange(PW, device="cuda") - 1).view(1, 1, 1, PW) + code % 3
        flat = (hh * OW + wwp).view(B, 64, -1)
        dz = torch.zeros(B, 64, OH * OW, device="cuda").scatter_add_(2, flat, gcur.reshape(B, 64, -1)).view(B, 64, OH, OW)
        del hh, wwp, flat, code
        stage += 2      # the BatchNorm sums over pooled pixels, then the apply pass
        dy0 = bn_stage(0, dz, (y0 * sc0 + sh0) > 0, y0, (B, OH, OW, 64), "stem bn1 (pool scatter + relu)", stage - 1, stage); nbn += 1
        del dz, y0
        conv_grads(0, rb(x), dy0, 2, 3, "stem conv1", stage + 1, None, None, None)
        stage += 1
        assert nbn == 20 and len([k for k in worst if k.endswith("wgrad")]) == 20 and stage == 62
        assert all(torch.isfinite(g).all() for g in grads)
        w = max(worst, key=worst.get)
        print(f"\nbf16 B={B} 512x512 backward: {len(worst)} stage outputs like for like ({stage} launch groups), worst "
              f"{worst[w]:.2e} at {w}; mask outliers {outliers}")
    finally:
        m.set_precision("bf16")


# ---- the whole distillation step at the benchmarked size (BASELINE configs[1]; configs[3] / [4]-like variants: nce_k 4096,
# MIA-2023 with its 65 536-row bank).  No CPU oracle finishes this in test time, so the checks are properties:
#   * finite outputs over two steps;
#   * determinism: a second step object built from the same seeds reproduces losses, logits and the updated student
#     BITWISE (fixed-order reductions everywhere, no float atomics);
#   * batch-duplication invariance: in a batch whose rows 32..63 repeat rows 0..31 (both views, omic vector, label) the
#     train-mode BatchNorm statistics are those of the first half, every tile of the second half is computed by OTHER
#     workgroups at OTHER positions of their persistent tile lists, and the per-row outputs of the three networks must
#     be bitwise equal between the halves - a tile-stream bug that depends on where a tile sits (the class the small
#     parity tests cannot see) breaks exactly this.
def _variant_step(variant, B, seed):
    import numpy as np
    import multimodal_learning_amd as m
    torch.manual_seed(seed)
    np.random.seed(2019)
    opt = m.stage2_opt(dropout_rate=0.0, batch_size=B)
    kw = {}
    if variant == "miccai2022":
        n_data, K = 1024, opt.nce_p + opt.nce_k - 1
    elif variant == "mia2022":
        opt.nce_k, opt.grads_m, opt.grads_thresh, opt.thresh = 4096, 0.9, "False", 0.1
        n_data, K = 16384, 4096
        kw["variant"] = "mia2022"
    else:
        n_data, K = 65536, 4096
        for k, v in dict(nce_k=4096, nce_p=6, pos_extra="neighbors", neg_mode="all_others", start_reweight=0, discrep_scale=1,
                         max_discrep=2.0, use_grads_thresh="True", grads_thresh=0.0, loss_weighting="GK_refine").items():
            setattr(opt, k, v)
        labels = torch.arange(n_data) % 3
        kw["variant"] = "mia2023"
        kw["train_class_idx"] = [np.nonzero((labels == c).numpy())[0] for c in range(3)]
    step = m.DistillStep(opt, n_data, device="cuda", **kw)
    for c in (step.criterion_kd, step.criterion_kd_path):
        c.contrast.verbose = False
    return step, n_data, K


def _dup_batch(B, H, n_data, K, seed):
    g = torch.Generator().manual_seed(seed)
    h = B // 2
    x = _images(h, H, seed)
    x2 = (x + 0.01 * torch.randn(h, 3, H, H, device="cuda", generator=torch.Generator(device="cuda").manual_seed(seed + 1))).clamp_(-1, 1)
    om = torch.randn(h, 320, generator=g)
    index = torch.randperm(n_data, generator=g)[:B]
    grade = (index % 3)
    grade[h:] = grade[:h]
    sidx = torch.randint(0, n_data, (B, K + 1), generator=g)
    sidx[:, 0] = index
    z = torch.zeros(B)
    d = lambda t: t.cuda()
    return ((torch.cat([x, x]), torch.cat([x2, x2])), d(z), d(torch.cat([om, om])), d(z), d(z), d(grade), d(index), d(sidx))


@pytest.mark.parametrize("variant", ["miccai2022", "mia2022", "mia2023"])
def test_distill_step_at_benchmark_size(variant):
    import multimodal_learning_amd as m
    B, H = 64, 512
    m.set_precision("bf16")
    runs = []
    for rep in range(2):
        step, n_data, K = _variant_step(variant, B, seed=5)
        outs = []
        for it in range(2):
            out = step.step(_dup_batch(B, H, n_data, K, seed=40 + it), epoch=1)
            outs.append({k: out[k].detach().float().clone() for k in ("loss", "logit_path", "ema_logit", "fuse_logit", "path_feat")})
        torch.cuda.synchronize()
        psum = torch.cat([p.detach().flatten() for p in step.model.parameters()]).double().sum()
        runs.append((outs, psum))
        for crd in (step.criterion_kd, step.criterion_kd_path):
            assert torch.isfinite(crd.contrast.memory_v1).all() and torch.isfinite(crd.contrast.memory_v2).all()
        for net in (step.model, step.ema_model, step.fix_model.path_net):
            net.release_workspaces()
        del step
        torch.cuda.empty_cache()
    h = B // 2
    for it in range(2):
        o = runs[0][0][it]
        for k, v in o.items():
            assert torch.isfinite(v).all(), (variant, it, k)
        # batch-duplication invariance of the per-row outputs (step 0: all three networks; step 1: again, after an update)
        for k in ("logit_path", "ema_logit", "fuse_logit", "path_feat"):
            assert torch.equal(o[k][:h], o[k][h:]), (variant, it, k, (o[k][:h] - o[k][h:]).abs().max().item())
        # determinism across step objects
        for k, v in o.items():
            assert torch.equal(v, runs[1][0][it][k]), (variant, it, k)
    assert runs[0][1].item() == runs[1][1].item(), "updated student differs between two runs from the same seeds"


def test_half_pair_arithmetic_tracks_parity_mode_at_benchmark_size():
    """The tolerance-compliant arithmetic of the bench's `parity_mode` (fp16x3: fp16-pair operands, 3 MFMA products) against
    the fp32-equivalent parity mode (bf16x6) at BASELINE configs[1] (64 tiles of 512 x 512, where the 4 608-term sums, the
    1 M-pixel BatchNorm statistics and the per-tensor dz scales are at their real sizes): one step from the same seeds -
    logits of all three networks and every loss term within the north-star's 1e-3, the student's gradients (first layer to
    head, through 20 train-mode BatchNorms and every dz scale) within 1e-2 in relative L2: the GK-Refine weights that multiply
    the four distillation losses are cosines of DIFFERENCES of student and teacher probabilities and turn the 1e-5 logit
    difference of the two arithmetics into ~1e-3 of every gradient, head included (the kernels themselves are held to ~1e-6
    by tests/test_gpu_conv.py and, stage by stage on the reference's golden, by tests/test_gpu_step.py).  (A second step is not compared:
    from zero Adam moments the first update is lr * sign(g), which turns rounding differences of near-zero gradient entries
    into +-lr parameter differences - the regime tests/test_gpu_step.py::test_two_steps_from_mid_training_state avoids.)"""
    import multimodal_learning_amd as m
    B, H = 64, 512
    watch = ("conv1.weight", "layer1.1.conv2.weight", "layer2.0.conv1.weight", "layer2.0.downsample.0.weight", "layer3.1.bn1.weight",
             "layer4.1.conv2.weight", "fc_new1.0.weight", "fc_new2.weight")
    res = {}
    try:
        for mode in ("bf16x6", "fp16x3", "fp16x3/x1"):
            m.set_precision(mode)
            step, n_data, K = _variant_step("miccai2022", B, seed=5)
            named = dict(step.model.named_parameters())
            grads = {}
            orig = step.optimizer.step

            def spy(*a, **k):
                if not grads:
                    for kk in watch:
                        grads[kk] = named[kk].grad.detach().clone()
                return orig(*a, **k)
            step.optimizer.step = spy
            g = torch.Generator().manual_seed(70)
            x = _images(B, H, 70)
            x2 = (x + 0.01 * torch.randn(B, 3, H, H, device="cuda", generator=torch.Generator(device="cuda").manual_seed(80))).clamp_(-1, 1)
            index = torch.randperm(n_data, generator=g)[:B]
            sidx = torch.randint(0, n_data, (B, K + 1), generator=g); sidx[:, 0] = index
            z = torch.zeros(B)
            bt = ((x, x2), z.cuda(), torch.randn(B, 320, generator=g).cuda(), z.cuda(), z.cuda(), (index % 3).cuda(), index.cuda(), sidx.cuda())
            out = step.step(bt, epoch=1)
            torch.cuda.synchronize()
            res[mode] = ({k: out[k].detach().float().clone() for k in
                          ("loss", "loss_cls", "loss_div1", "loss_div2", "loss_kd1", "loss_kd2", "logit_path", "ema_logit", "fuse_logit",
                           "scale")}, grads)
            for net in (step.model, step.ema_model, step.fix_model.path_net):
                net.release_workspaces()
            del step, named
            torch.cuda.empty_cache()
    finally:
        m.set_precision("bf16")
    bad = []
    for mode in ("fp16x3", "fp16x3/x1"):      # (x1: the same forward, the hi planes' product alone in dgrad / wgrad)
        a, b = res["bf16x6"][0], res[mode][0]
        for k in a:
            assert torch.isfinite(b[k]).all(), (mode, k)
            err = (a[k] - b[k]).abs().max().item()
            print("%-10s %-12s |d vs bf16x6| %.3e  (max|ref| %.3e)" % (mode, k, err, a[k].abs().max().item()))
            tol = 1e-2 if k == "scale" else 1e-3
            if k == "loss":
                # total = lambda * CE + sum_i scale_i KD_i: the GK-Refine weights (held to 1e-2 below) carry their amplified
                # difference into the total - first-order bound max|d scale| * sum_i KD_i on top of the 1e-3 of the terms
                # (a recompile that moved the logits by 1e-6 moved the total from 3e-6 to 1.6e-3 of the bf16x6 value)
                tol += (a["scale"] - b["scale"]).abs().max().item() * sum(a[t].abs().item() for t in ("loss_div1", "loss_div2", "loss_kd1", "loss_kd2"))
            if not err <= tol:
                bad.append((mode, k, err, tol))
        for k in watch:
            ga, gb = res["bf16x6"][1][k], res[mode][1][k]
            err, mx = (ga - gb).abs().max().item(), ga.abs().max().item()
            l2 = ((ga - gb).double().norm() / ga.double().norm()).item()
            print("%-10s grad %-30s |d vs bf16x6| %.3e  (max|ref| %.3e, rel %.1e, rel L2 %.1e)" % (mode, k, err, mx, err / mx, l2))
            if not (torch.isfinite(gb).all() and l2 <= 1e-2):
                bad.append((mode, k))
        if mode == "fp16x3/x1":      # backward arithmetic alone: against fp16x3, whose forward (and loss weights) it shares bitwise
            for k in watch:
                ga, gb = res["fp16x3"][1][k], res[mode][1][k]
                l2 = ((ga - gb).double().norm() / ga.double().norm()).item()
                print("%-10s grad %-30s rel L2 vs fp16x3 %.1e" % (mode, k, l2))
                if l2 > 2e-3:
                    bad.append((mode, k, "vs fp16x3"))
    assert not bad, bad


def test_tsvd_stage1_step_at_config3_size():
    """BASELINE configs[3] (MIA-2022 train_test_tSVD, batch 128) at the benchmark tile size: the stage-1 step (student +
    mean-teacher PathomicNet, t-SVD constraint with 4 views, one-sided Jacobi prox at B = 128) on 128 tiles of 512 x 512
    in perf mode - finite, bitwise reproducible from the same seeds, and the auxiliary tensors it leaves behind are the
    float64 oracle's proximal operator of the adjacency tensors it computed."""
    import numpy as np
    import multimodal_learning_amd as m
    from oracle import weights as W
    from oracle import variants as OV
    B, H = 128, 512
    m.set_precision("bf16")
    res = []
    for rep in range(2):
        torch.manual_seed(3)
        opt = m.stage2_opt(dropout_rate=0.0, batch_size=B, cut_fuse_grad=True, num_teachers=2)
        opt.pred_distill, opt.KD_weight, opt.CRD_distill, opt.SP_distill, opt.orth_loss = 1, 1.0, 0, 0, "False"
        opt.tSVD_loss, opt.tSVD_mode, opt.n_views, opt.aux_iter = "True", "pathomic", 4, 1
        opt.mu, opt.pho, opt.max_mu, opt.Lambda_global = 0.01, 1.5, 1.0, 0.05
        model = m.define_net(opt, 1); ema = m.define_net(opt, 1)
        model.load_state_dict(W.make_state_dict(W.teacher_shapes(320), 3)); ema.load_state_dict(W.make_state_dict(W.teacher_shapes(320), 4))
        st = m.TeacherStage1Step(opt, device="cuda", models=(model.cuda(), ema.cuda()))
        g = torch.Generator().manual_seed(9)
        x = _images(B, H, 61)
        x2 = (x + 0.01 * torch.randn(B, 3, H, H, device="cuda", generator=torch.Generator(device="cuda").manual_seed(62))).clamp_(-1, 1)
        z = torch.zeros(B)
        batch = ((x, x2), z, torch.randn(B, 320, generator=g).cuda(), z, z, torch.randint(0, 3, (B,), generator=g).cuda(),
                 torch.arange(B).cuda(), torch.zeros(B, 2, dtype=torch.long).cuda())
        out = st.step(batch)
        torch.cuda.synchronize()
        assert torch.isfinite(out["loss"]).all() and torch.isfinite(out["loss_tsvd"]).all()
        if rep == 0:
            for adj, aux, tnn in ((st.adj_tensor1, st.aux_tensor1, st.path_TNN), (st.adj_tensor2, st.aux_tensor2, st.omic_TNN)):
                stack = torch.stack([a.detach().cpu() for a in adj], dim=2)
                ref, tnn_ref = OV.update_aux(stack, opt.Lambda_global / opt.mu)
                got = torch.stack([a.cpu() for a in aux], dim=2).numpy()
                assert np.abs(got - ref).max() <= 2e-5 * max(np.abs(ref).max(), 1.0)
                assert abs(float(tnn) - tnn_ref) <= 1e-4 * max(abs(tnn_ref), 1.0)
        res.append((out["loss"].detach().clone(), out["loss_tsvd"].detach().clone(),
                    torch.cat([p.detach().flatten() for p in model.parameters()]).double().sum()))
        del st, model, ema, x, x2, batch
        torch.cuda.empty_cache()
    for a, b in zip(res[0], res[1]):
        assert torch.equal(a, b), "stage-1 t-SVD step is not reproducible at B = 128 / 512 x 512"
