"""TEST INFRASTRUCTURE (oracle): numpy restatement of the training loader's image transform
(MICCAI-2022/data_loaders_MT.py:168-175: RandomHorizontalFlip, RandomVerticalFlip, RandomCrop, ColorJitter(0.1, 0.1,
0.05, 0.01), ToTensor, Normalize(0.5, 0.5), applied twice by TransformTwice :51-53) for GIVEN random draws.

PINNED against the reference's own dependency: the algorithm lives in torchvision's PIL backend (absent from this
image) over Pillow (present, 12.2.0).  torchvision's `adjust_brightness / contrast / saturation` are
`PIL.ImageEnhance.Brightness / Contrast / Color(img).enhance(factor)` and `adjust_hue` is `img.convert("HSV")`, a
wrapping uint8 add of `uint8(hue_factor * 255)` on H, `convert("RGB")`; ColorJitter applies the four in a drawn order.
tests/golden/make_golden_colorjitter.py runs exactly those Pillow calls and writes tests/golden/colorjitter_pil.npz;
tests/test_oracle_augment.py holds this file to it bit for bit (and, where Pillow imports, to Pillow itself over all
2^24 colours for both colour-space conversions).

Pillow's arithmetic as restated here: `ImageEnhance` = `Image.blend(degenerate, image, factor)` on uint8 with
truncation (`(UINT8)(in1 + alpha * (in2 - in1))` in float, clipped when alpha is outside [0, 1]); Brightness blends
with black, Contrast with the solid grey `int(mean(L) + 0.5)`, Color with the per-pixel L; L = (19595 R + 38470 G +
7471 B + 0x8000) >> 16; RGB -> HSV and back follow Pillow's Convert.c (float / double mix reproduced below).
Only tests/ may import this module."""
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


def rgb_to_hsv_u8(rgb):
    """Pillow `convert("HSV")` (Convert.c rgb2hsv_row): uint8 H, S, V.  The C code keeps rc / gc / bc / s / h in
    float variables and evaluates the expressions that contain a double literal in double."""
    r, g, b = rgb[..., 0].astype(np.int64), rgb[..., 1].astype(np.int64), rgb[..., 2].astype(np.int64)
    maxc = np.maximum(r, np.maximum(g, b)); minc = np.minimum(r, np.minimum(g, b))
    cr = (maxc - minc).astype(np.float32)
    crs = np.where(cr > 0, cr, np.float32(1))
    s = (cr / np.where(maxc > 0, maxc, 1).astype(np.float32)).astype(np.float32)
    rc = ((maxc - r).astype(np.float32) / crs).astype(np.float32)
    gc = ((maxc - g).astype(np.float32) / crs).astype(np.float32)
    bc = ((maxc - b).astype(np.float32) / crs).astype(np.float32)
    h = np.where(r == maxc, (bc - gc).astype(np.float32).astype(np.float64),
                 np.where(g == maxc, 2.0 + rc.astype(np.float64) - bc.astype(np.float64),
                          4.0 + gc.astype(np.float64) - rc.astype(np.float64))).astype(np.float32)
    h = np.fmod(h.astype(np.float64) / 6.0 + 1.0, 1.0).astype(np.float32)
    uh = np.clip((h.astype(np.float64) * 255.0).astype(np.int64), 0, 255)
    us = np.clip((s.astype(np.float64) * 255.0).astype(np.int64), 0, 255)
    grey = maxc == minc
    return np.where(grey, 0, uh), np.where(grey, 0, us), maxc


def hsv_to_rgb_u8(h, s, v):
    """Pillow `convert("RGB")` from HSV (Convert.c hsv2rgb): double arithmetic, C round() (half away from zero)."""
    h6 = h.astype(np.float32).astype(np.float64) * 6.0 / 255.0
    i = np.floor(h6)
    f = h6 - i
    fs = s.astype(np.float32).astype(np.float64) / 255.0
    vf = v.astype(np.float32).astype(np.float64)
    rnd = lambda x: np.clip(np.floor(x + 0.5), 0, 255).astype(np.int64)       # noqa: E731  (arguments are >= 0)
    p, q, t = rnd(vf * (1.0 - fs)), rnd(vf * (1.0 - fs * f)), rnd(vf * (1.0 - fs * (1.0 - f)))
    ii = i.astype(np.int64) % 6
    V = v.astype(np.int64)
    r = np.choose(ii, [V, q, p, p, t, V]); g = np.choose(ii, [t, V, V, q, p, p]); b = np.choose(ii, [p, p, t, V, V, q])
    grey = s == 0
    return np.stack([np.where(grey, V, r), np.where(grey, V, g), np.where(grey, V, b)], -1)


def hue_delta(hf):
    """torchvision `np.uint8(hue_factor * 255)`: truncation toward zero, then the uint8 wrap."""
    return int(np.float64(hf) * 255.0) & 255


def hue_shift(rgb, hf):
    """torchvision functional_pil.adjust_hue: wrap-add on the uint8 H channel of the HSV image."""
    h, s, v = rgb_to_hsv_u8(rgb)
    return hsv_to_rgb_u8((h + hue_delta(hf)) & 255, s, v)


def one_view(src, prm):
    """src uint8 [SH, SW, 3]; prm = dict(flipH, flipV, top, left, S, b, c, s, h, order[, skip]) -> (f32 [3, S, S], grey
    mean).  `skip`: steps torchvision's ColorJitter drops because their range is zero (an enabled hue step with factor 0
    still runs Pillow's lossy uint8 HSV round trip)."""
    img = src
    if prm["flipH"]:
        img = img[:, ::-1]
    if prm["flipV"]:
        img = img[::-1]
    S = prm["S"]
    img = img[prm["top"]:prm["top"] + S, prm["left"]:prm["left"] + S].astype(np.int64)
    mean = None
    for op in prm["order"]:
        if op in prm.get("skip", ()):
            continue
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
