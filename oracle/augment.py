"""TEST INFRASTRUCTURE (oracle): numpy restatement of the training loader's image transform
(MICCAI-2022/data_loaders_MT.py:168-175: RandomHorizontalFlip, RandomVerticalFlip, RandomCrop, ColorJitter(0.1, 0.1,
0.05, 0.01), ToTensor, Normalize(0.5, 0.5), applied twice by TransformTwice :51-53) for GIVEN random draws.

PARITY UNPINNED: torchvision and PIL are absent from this image, so the colour arithmetic cannot be checked against the
reference's dependency.  It follows their published algorithms: PIL `ImageEnhance` = `Image.blend(degenerate, image,
factor)` on uint8 with truncation (ImagingBlend: `(UINT8)(in1 + alpha * (in2 - in1))`, clipped when alpha is outside
[0, 1]); Brightness blends with black, Contrast with the solid grey `int(mean(L) + 0.5)`, Color (saturation) with the
per-pixel L; L = (19595 R + 38470 G + 7471 B + 0x8000) >> 16; ColorJitter applies the four steps in a random order;
hue is a float HSV round trip with H shifted by the factor (the PIL version works on uint8 HSV - the largest
deviation one should expect from the real dependency, at most a few grey levels at hue 0.01).  Geometry and
normalisation are unambiguous.  Only tests/ may import this module."""
import numpy as np


def luma(rgb):
    r, g, b = rgb[..., 0].astype(np.int64), rgb[..., 1].astype(np.int64), rgb[..., 2].astype(np.int64)
    return (r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16


def blend8(deg, img, f):
    f = np.float32(f)
    t = np.float32(deg) + f * (img.astype(np.float32) - np.float32(deg)).astype(np.float32)
    t = t.astype(np.float32)
    if 0.0 <= f <= 1.0:
        return t.astype(np.int64)
    return np.where(t <= 0, 0, np.where(t >= 255, 255, t.astype(np.int64)))


def hue_shift(rgb, hf):
    x = rgb.astype(np.float32) / np.float32(255)
    R, G, B = x[..., 0], x[..., 1], x[..., 2]
    mx, mn = x.max(-1), x.min(-1)
    d = mx - mn
    safe = np.where(d > 0, d, np.float32(1))
    h = np.where(mx == R, (G - B) / safe, np.where(mx == G, np.float32(2) + (B - R) / safe, np.float32(4) + (R - G) / safe))
    h = (h / np.float32(6)).astype(np.float32)
    h = np.where(d > 0, h - np.floor(h), np.float32(0)).astype(np.float32)
    s = np.where(mx > 0, d / np.where(mx > 0, mx, np.float32(1)), np.float32(0)).astype(np.float32)
    v = mx
    h = (h + np.float32(hf)).astype(np.float32)
    h = (h - np.floor(h)).astype(np.float32)
    h6 = (h * np.float32(6)).astype(np.float32)
    i = np.floor(h6).astype(np.int64) % 6
    fr = (h6 - np.floor(h6)).astype(np.float32)
    one = np.float32(1)
    p = (v * (one - s)).astype(np.float32); q = (v * (one - s * fr)).astype(np.float32); t = (v * (one - s * (one - fr))).astype(np.float32)
    rr = np.choose(i, [v, q, p, p, t, v]); gg = np.choose(i, [t, v, v, q, p, p]); bb = np.choose(i, [p, p, t, v, v, q])
    out = np.stack([rr, gg, bb], -1).astype(np.float32) * np.float32(255)
    return np.clip(np.rint(out), 0, 255).astype(np.int64)


def one_view(src, prm):
    """src uint8 [SH, SW, 3]; prm = dict(flipH, flipV, top, left, S, b, c, s, h, order) -> (f32 [3, S, S], grey mean)."""
    img = src
    if prm["flipH"]:
        img = img[:, ::-1]
    if prm["flipV"]:
        img = img[::-1]
    S = prm["S"]
    img = img[prm["top"]:prm["top"] + S, prm["left"]:prm["left"] + S].astype(np.int64)
    mean = None
    for op in prm["order"]:
        if op == 0:
            img = blend8(0, img, prm["b"])
        elif op == 1:
            mean = int(luma(img).mean() + 0.5)
            img = blend8(mean, img, prm["c"])
        elif op == 2:
            img = blend8(luma(img)[..., None], img, prm["s"])
        else:
            img = hue_shift(img, prm["h"])
    out = (img.astype(np.float32) / np.float32(255) - np.float32(0.5)) / np.float32(0.5)
    return np.ascontiguousarray(out.transpose(2, 0, 1)), mean
