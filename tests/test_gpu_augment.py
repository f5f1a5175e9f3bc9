"""On-device input pipeline: bit for bit against vectors produced by the real Pillow calls torchvision
makes (tests/golden/colorjitter_pil.npz), against the numpy restatement of the loader's transform (oracle/augment.py,
itself held to Pillow), against Pillow directly where it imports, and for the structural rules."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _src(B, SH, SW, seed):
    g = torch.Generator().manual_seed(seed)
    base = torch.randint(0, 256, (B, SH // 8, SW // 8, 3), generator=g, dtype=torch.uint8)
    img = base.repeat_interleave(8, 1).repeat_interleave(8, 2).int() + torch.randint(-20, 21, (B, SH, SW, 3), generator=g)
    return img.clamp(0, 255).to(torch.uint8)


def _opt(S):
    import types
    return types.SimpleNamespace(input_size_path=S)


def test_apply_equals_oracle_for_given_draws():
    """All 24 step orders, both flips, crops at the corners, factors on both sides of 1: every output pixel equals the
    oracle's exactly (all of it is integer / correctly rounded arithmetic)."""
    import itertools
    import multimodal_learning_amd as m
    from oracle import augment as OA
    B, SH, SW, S = 12, 96, 80, 64
    src = _src(B, SH, SW, 1)
    orders = list(itertools.permutations(range(4)))
    rng = np.random.default_rng(0)
    prm = torch.zeros(B, 2, 16)
    dicts = {}
    for b in range(B):
        for v in range(2):
            k = b * 2 + v
            d = dict(flipH=int(k & 1), flipV=int((k >> 1) & 1), top=int([0, SH - S, rng.integers(0, SH - S + 1)][k % 3]),
                     left=int([SW - S, 0, rng.integers(0, SW - S + 1)][k % 3]), S=S, b=float(np.float32(rng.uniform(0.9, 1.1))),
                     c=float(np.float32(rng.uniform(0.9, 1.1))), s=float(np.float32(rng.uniform(0.95, 1.05))),
                     h=float(np.float32(rng.uniform(-0.01, 0.01))), order=orders[k])
            dicts[(b, v)] = d
            prm[b, v, :12] = torch.tensor([d["flipH"], d["flipV"], d["top"], d["left"], d["b"], d["c"], d["s"], d["h"], *d["order"]])
    aug = m.augment.DeviceAugment(_opt(S))
    o0, o1 = aug(src.cuda(), params=prm)
    outs = (o0.cpu().numpy(), o1.cpu().numpy())
    worst = 0.0
    for (b, v), d in dicts.items():
        ref, mean = OA.one_view(src[b].numpy(), d)
        assert int(aug.last_params[b, v, 12]) == mean, (b, v)
        worst = max(worst, float(np.abs(outs[v][b] - ref).max()))
    assert worst == 0.0, worst


def _golden_params(g):
    prm = torch.zeros(g["params"].shape[0], 2, 16)
    prm[:, :, :12] = torch.from_numpy(g["params"])
    return prm


def test_apply_equals_pillow_golden_bit_for_bit(golden_dir):
    """The kernel against the fixture written by the real Pillow calls (ImageEnhance.Brightness / Contrast / Color, the
    uint8 HSV hue shift) for all 24 orders, in-range and far out-of-range factors: identical uint8 images, i.e.
    identical normalised floats."""
    import os
    import multimodal_learning_amd as m
    g = np.load(os.path.join(golden_dir, "colorjitter_pil.npz"))
    S = int(g["S"])
    o0, o1 = m.augment.DeviceAugment(_opt(S))(torch.from_numpy(g["src"]).cuda(), params=_golden_params(g))
    want = ((g["out_u8"].astype(np.float32) / np.float32(255) - np.float32(0.5)) / np.float32(0.5)).transpose(0, 1, 4, 2, 3)
    for v, o in enumerate((o0, o1)):
        got = o.cpu().numpy()
        bad = np.argwhere(got != want[:, v])
        assert bad.size == 0, (v, bad[:4], float(np.abs(got - want[:, v]).max()) * 127.5)


def test_apply_equals_live_pillow_on_random_draws():
    """Where Pillow imports (it is in this image): device draws -> the same parameters through the Pillow calls."""
    Image = pytest.importorskip("PIL.Image")
    from PIL import ImageEnhance
    import multimodal_learning_amd as m
    B, SH, SW, S = 16, 72, 96, 56
    src = _src(B, SH, SW, 11)
    aug = m.augment.DeviceAugment(_opt(S), seed=5, brightness=0.4, contrast=0.4, saturation=0.4, hue=0.3)
    o0, o1 = aug(src.cuda())
    prm = aug.last_params.cpu().numpy()

    def hue(im, f):
        h, s, v = im.convert("HSV").split()
        nh = np.array(h, dtype=np.uint8)
        nh += np.array(f * 255).astype(np.int32).astype(np.uint8)
        return Image.merge("HSV", (Image.fromarray(nh, "L"), s, v)).convert("RGB")
    ops = (lambda im, p: ImageEnhance.Brightness(im).enhance(float(p[4])), lambda im, p: ImageEnhance.Contrast(im).enhance(float(p[5])),
           lambda im, p: ImageEnhance.Color(im).enhance(float(p[6])), lambda im, p: hue(im, float(p[7])))
    for b in range(B):
        for v, o in enumerate((o0, o1)):
            p = prm[b, v]
            im = Image.fromarray(src[b].numpy(), "RGB")
            if p[0]:
                im = im.transpose(Image.FLIP_LEFT_RIGHT)
            if p[1]:
                im = im.transpose(Image.FLIP_TOP_BOTTOM)
            im = im.crop((int(p[3]), int(p[2]), int(p[3]) + S, int(p[2]) + S))
            for k in p[8:12].astype(int):
                im = ops[k](im, p)
            want = ((np.array(im).astype(np.float32) / np.float32(255) - np.float32(0.5)) / np.float32(0.5)).transpose(2, 0, 1)
            assert np.array_equal(o[b].cpu().numpy(), want), (b, v, p[8:12])


def test_neutral_jitter_is_crop_flip_normalise_exactly():
    import multimodal_learning_amd as m
    B, SH, SW, S = 3, 72, 88, 48
    src = _src(B, SH, SW, 2)
    prm = torch.zeros(B, 2, 16)
    prm[:, :, 4:7] = 1.0
    prm[:, :, 13] = 8.0          # hue step dropped (ColorJitter(hue=0)): enabled, it would run Pillow's lossy HSV round trip
    prm[:, :, 8:12] = torch.tensor([2.0, 0.0, 3.0, 1.0])
    prm[:, 0, 0] = 1; prm[:, 1, 1] = 1
    prm[:, :, 2] = 5; prm[:, :, 3] = 11
    o0, o1 = m.augment.DeviceAugment(_opt(S))(src.cuda(), params=prm)
    f = src.float() / 255
    v0 = f.flip(2)[:, 5:5 + S, 11:11 + S].permute(0, 3, 1, 2)
    v1 = f.flip(1)[:, 5:5 + S, 11:11 + S].permute(0, 3, 1, 2)
    assert torch.equal(o0.cpu(), (v0 - 0.5) / 0.5) and torch.equal(o1.cpu(), (v1 - 0.5) / 0.5)


def test_device_draws_are_in_range_reproducible_and_spread():
    import multimodal_learning_amd as m
    B, SH, SW, S = 256, 40, 48, 32
    src = _src(4, SH, SW, 3).repeat(64, 1, 1, 1).cuda()
    a1 = m.augment.DeviceAugment(_opt(S), seed=7); a2 = m.augment.DeviceAugment(_opt(S), seed=7)
    x1 = a1(src); p1 = a1.last_params.cpu(); x2 = a2(src); p2 = a2.last_params.cpu()
    assert torch.equal(p1, p2) and torch.equal(x1[0], x2[0]) and torch.equal(x1[1], x2[1])
    a1(src); p3 = a1.last_params.cpu()
    assert not torch.equal(p1, p3)                                   # the step counter moves the stream
    p = p1.reshape(-1, 16).numpy()
    assert set(np.unique(p[:, 0])) <= {0.0, 1.0} and 0.35 < p[:, 0].mean() < 0.65 and 0.35 < p[:, 1].mean() < 0.65
    assert p[:, 2].min() >= 0 and p[:, 2].max() <= SH - S and p[:, 3].max() <= SW - S and len(np.unique(p[:, 3])) > 8
    for col, lo, hi in ((4, 0.9, 1.1), (5, 0.9, 1.1), (6, 0.95, 1.05), (7, -0.01, 0.01)):
        assert p[:, col].min() >= lo - 1e-6 and p[:, col].max() <= hi + 1e-6 and p[:, col].std() > 0.2 * (hi - lo)
    orders = {tuple(r) for r in p[:, 8:12].astype(int).tolist()}
    assert all(sorted(o) == [0, 1, 2, 3] for o in orders) and len(orders) == 24
    assert not torch.equal(x1[0], x1[1])                             # the two views are drawn independently
    assert float(x1[0].min()) >= -1.0 and float(x1[0].max()) <= 1.0


def test_resident_loader_batches_straight_from_the_store():
    """ResidentTileLoader: a batch drawn by row indices out of the resident uint8 store equals augmenting the gathered
    tiles with the same draws; the tuple has the loader's layout and feeds DistillStep.step."""
    import types
    import multimodal_learning_amd as m
    n, SH, S = 40, 96, 64
    tiles = _src(n, SH, SH, 5).cuda()
    labels = torch.arange(n) % 3
    opt = types.SimpleNamespace(input_size_path=S, nce_p=4, nce_k=16, pos_mode="multi_pos", label_dim=3)
    ld = m.augment.ResidentTileLoader(opt, tiles, torch.randn(n, 80), labels, seed=3)
    idx = torch.tensor([7, 0, 33, 12, 12, 39])
    (xa, xb), _, xo, _, _, gr, index, sidx = ld.batch(idx)
    prm = ld.aug.last_params.clone()
    ya, yb = m.augment.DeviceAugment(opt)(tiles[idx.cuda()], params=prm)
    assert torch.equal(xa, ya) and torch.equal(xb, yb)
    assert xa.shape == (6, 3, S, S) and xo.shape == (6, 80) and torch.equal(gr.cpu(), labels[idx]) and torch.equal(index.cpu(), idx)
    assert sidx.shape == (6, 4 + 16) and torch.equal(sidx[:, 0].cpu(), idx)


def test_resident_loader_refills_a_batch_in_place():
    """batch(index, into=previous) rewrites the previous tuple's tensors at their addresses (what a captured step graph
    that adopted them needs) with a new draw."""
    import types
    import multimodal_learning_amd as m
    n, SH, S = 24, 64, 48
    tiles = _src(n, SH, SH, 6).cuda()
    labels = torch.arange(n) % 3
    opt = types.SimpleNamespace(input_size_path=S, nce_p=3, nce_k=8, pos_mode="multi_pos", label_dim=3)
    ld = m.augment.ResidentTileLoader(opt, tiles, torch.randn(n, 16), labels, seed=1)
    bt = ld.batch(torch.tensor([1, 5, 9, 20]))
    ptrs = [t.data_ptr() for t in (bt[0][0], bt[0][1], bt[2], bt[5], bt[6], bt[7])]
    old = bt[0][0].clone()
    idx2 = torch.tensor([2, 6, 10, 23])
    bt2 = ld.batch(idx2, into=bt)
    assert [t.data_ptr() for t in (bt2[0][0], bt2[0][1], bt2[2], bt2[5], bt2[6], bt2[7])] == ptrs
    ya, yb = m.augment.DeviceAugment(opt)(tiles[idx2.cuda()], params=ld.aug.last_params.clone())
    assert torch.equal(bt2[0][0], ya) and torch.equal(bt2[0][1], yb) and not torch.equal(bt2[0][0], old)
    assert torch.equal(bt2[6].cpu(), idx2) and torch.equal(bt2[5].cpu(), labels[idx2]) and torch.equal(bt2[7][:, 0].cpu(), idx2)
    assert torch.equal(bt2[2].cpu(), ld.x_omic[idx2.cuda()].cpu())


def test_distill_step_fed_by_the_resident_loader_inside_its_graph():
    """step.loader = ResidentTileLoader; step.step(None): eager steps draw their batch eagerly, graph steps replay ONE
    graph that contains the shuffle-index draw, both augmented views, the contrast indices and the step.  Every replay
    consumes a new batch (the index buffer changes, an epoch of n/B batches visits every row once), losses stay finite."""
    import types
    import multimodal_learning_amd as m
    from oracle.step import default_opt
    B, S, n_data = 8, 64, 96
    opt = default_opt(batch_size=B, nce_p=30, nce_p2=8, nce_k=28, nce_k2=16, select_pos_mode="hard", n_data=n_data)
    opt.input_size_path, opt.pos_mode, opt.label_dim = S, "multi_pos", 3
    tiles = _src(16, 2 * S, 2 * S, 8).cuda()
    labels = torch.arange(n_data) % 3          # three classes of 32 rows: nce_p = 30 positives exist without replacement
    ld = m.augment.ResidentTileLoader(opt, tiles, torch.randn(n_data, 320), labels, seed=2)
    ld.row_to_tile = torch.arange(n_data, device="cuda") % 16
    step = m.DistillStep(opt, n_data, device="cuda")
    for crd in (step.criterion_kd, step.criterion_kd_path):
        crd.contrast.verbose = False
    step.loader = ld
    step.enable_graph()
    seen, losses = [], []
    for it in range(14):
        out = step.step(None, epoch=1)
        seen.append(step._loader_bt[6].clone())
        losses.append(out["loss"].clone())
    torch.cuda.synchronize()
    assert step._slots and step._slots[0]["graph"] is not None, "the graph path was not taken"
    assert all(torch.isfinite(l) for l in losses)
    rows = torch.cat(seen[:12]).cpu()                    # 12 batches of 8 = one epoch of 96 rows
    assert sorted(rows.tolist()) == list(range(n_data))
    assert not torch.equal(seen[12].cpu(), seen[0].cpu())  # the next epoch uses another permutation
    assert int(ld.batch_no.item()) == 14


def test_caller_supplied_params_validate_the_disabled_step_mask():
    """column 13 of a caller-supplied parameter block is the bit mask of disabled colour steps (0..15); anything
    else - a block from before the layout change, garbage - must raise instead of silently skipping jitter steps."""
    import types
    import multimodal_learning_amd as m
    from multimodal_learning_amd.augment import DeviceAugment, NPARAM
    opt = types.SimpleNamespace(input_size_path=32)
    aug = DeviceAugment(opt, "cuda", seed=1)
    src = torch.randint(0, 256, (2, 64, 64, 3), dtype=torch.uint8, device="cuda")
    aug(src)
    good = aug.last_params.clone()
    aug(src, params=good)                       # a block the kernel produced itself is accepted
    for bad in (99.0, -1.0, 2.5):
        p = good.clone()
        p[0, 1, 13] = bad
        with pytest.raises(ValueError):
            aug(src, params=p)
