#!/usr/bin/env python3
"""Golden vectors for the loader's colour jitter (MICCAI-2022/data_loaders_MT.py:164-170:
`transforms.ColorJitter(brightness=0.1, contrast=0.1, saturation=0.05, hue=0.01)` on PIL images), produced with the
REAL Pillow calls torchvision's PIL backend makes (torchvision itself is absent from this image; its
functional_pil.adjust_* are thin wrappers):

    adjust_brightness(img, f) = ImageEnhance.Brightness(img).enhance(f)
    adjust_contrast(img, f)   = ImageEnhance.Contrast(img).enhance(f)
    adjust_saturation(img, f) = ImageEnhance.Color(img).enhance(f)
    adjust_hue(img, f)        = h, s, v = img.convert("HSV").split(); h += uint8(f * 255) (wrapping);
                                Image.merge("HSV", (h, s, v)).convert("RGB")

applied in a given order after the geometric steps (flips + crop, which are index arithmetic), then ToTensor +
Normalize(0.5, 0.5).  Writes tests/golden/colorjitter_pil.npz: source tiles, the draws, the uint8 result of every
(tile, view) plus a 16 384-colour sample of Pillow's RGB -> HSV -> RGB conversions."""
import itertools
import os

import numpy as np
from PIL import Image, ImageEnhance

HERE = os.path.dirname(os.path.abspath(__file__))


def adjust_hue(img, hue_factor):
    h, s, v = img.convert("HSV").split()
    np_h = np.array(h, dtype=np.uint8)
    with np.errstate(over="ignore"):
        np_h += np.array(hue_factor * 255).astype(np.int32).astype(np.uint8)   # numpy >= 2 rejects np.uint8(negative)
    h = Image.fromarray(np_h, "L")
    return Image.merge("HSV", (h, s, v)).convert("RGB")


OPS = (lambda im, p: ImageEnhance.Brightness(im).enhance(p["b"]),
       lambda im, p: ImageEnhance.Contrast(im).enhance(p["c"]),
       lambda im, p: ImageEnhance.Color(im).enhance(p["s"]),
       lambda im, p: adjust_hue(im, p["h"]))


def pil_view(src, p):
    im = Image.fromarray(src, "RGB")
    if p["flipH"]:
        im = im.transpose(Image.FLIP_LEFT_RIGHT)
    if p["flipV"]:
        im = im.transpose(Image.FLIP_TOP_BOTTOM)
    im = im.crop((p["left"], p["top"], p["left"] + p["S"], p["top"] + p["S"]))
    for op in p["order"]:
        im = OPS[op](im, p)
    return np.array(im)


def main():
    rng = np.random.default_rng(7)
    n, SH, SW, S = 12, 80, 72, 64
    base = rng.integers(0, 256, (n, SH // 8, SW // 8, 3))
    src = np.clip(np.repeat(np.repeat(base, 8, 1), 8, 2) + rng.integers(-24, 25, (n, SH, SW, 3)), 0, 255).astype(np.uint8)
    src[0, :8] = 255; src[0, 8:16] = 0; src[1, :, :8] = [255, 0, 0]; src[1, :, 8:16] = [3, 3, 3]      # extremes and greys
    orders = list(itertools.permutations(range(4)))
    prm = np.zeros((n, 2, 12), dtype=np.float32)
    out_u8 = np.zeros((n, 2, S, S, 3), dtype=np.uint8)
    for b in range(n):
        for v in range(2):
            k = b * 2 + v
            # factors of the reference's ranges ([0.9, 1.1], [0.9, 1.1], [0.95, 1.05], [-0.01, 0.01]) and, on the last
            # tiles, far outside them (blend factors beyond [0, 1], large hue shifts of both signs)
            wide = b >= n - 3
            p = dict(flipH=k & 1, flipV=(k >> 1) & 1, top=int(rng.integers(0, SH - S + 1)), left=int(rng.integers(0, SW - S + 1)),
                     S=S, b=float(np.float32(rng.uniform(0.3, 1.9) if wide else rng.uniform(0.9, 1.1))),
                     c=float(np.float32(rng.uniform(0.3, 1.9) if wide else rng.uniform(0.9, 1.1))),
                     s=float(np.float32(rng.uniform(0.0, 2.5) if wide else rng.uniform(0.95, 1.05))),
                     h=float(np.float32(rng.uniform(-0.5, 0.5) if wide else rng.uniform(-0.01, 0.01))), order=orders[k])
            prm[b, v] = [p["flipH"], p["flipV"], p["top"], p["left"], p["b"], p["c"], p["s"], p["h"], *p["order"]]
            out_u8[b, v] = pil_view(src[b], p)
    # (ToTensor = uint8 -> float32 / 255 and Normalize = sub 0.5, div 0.5 in float32 are stated in the tests)
    cols = rng.integers(0, 256, (16384, 3)).astype(np.uint8)
    cols[:256] = np.arange(256)[:, None]                                   # the grey axis
    hsv = np.array(Image.fromarray(cols.reshape(128, 128, 3), "RGB").convert("HSV")).reshape(-1, 3)
    back = np.array(Image.fromarray(cols.reshape(128, 128, 3), "HSV").convert("RGB")).reshape(-1, 3)   # cols read as HSV
    np.savez_compressed(os.path.join(HERE, "colorjitter_pil.npz"), src=src, params=prm, out_u8=out_u8,
                        S=S, colours=cols, rgb2hsv=hsv, hsv2rgb=back)
    print("wrote colorjitter_pil.npz", out_u8.shape, "Pillow", Image.__version__)


if __name__ == "__main__":
    main()
