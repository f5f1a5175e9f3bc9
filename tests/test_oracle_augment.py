"""The oracle's restatement of the loader's colour jitter (oracle/augment.py) against vectors produced by the real
Pillow calls torchvision makes (tests/golden/make_golden_colorjitter.py), bit for bit; and, where Pillow imports,
against Pillow itself over ALL 2^24 colours for both colour-space conversions."""
import os

import numpy as np
import pytest


def _dicts(prm, S):
    out = {}
    for b in range(prm.shape[0]):
        for v in range(2):
            p = prm[b, v]
            out[(b, v)] = dict(flipH=int(p[0]), flipV=int(p[1]), top=int(p[2]), left=int(p[3]), S=S, b=float(p[4]), c=float(p[5]),
                               s=float(p[6]), h=float(p[7]), order=tuple(int(x) for x in p[8:12]))
    return out


def test_oracle_jitter_equals_pillow_golden(golden_dir):
    from oracle import augment as OA
    g = np.load(os.path.join(golden_dir, "colorjitter_pil.npz"))
    S = int(g["S"])
    for (b, v), d in _dicts(g["params"], S).items():
        ref, _ = OA.one_view(g["src"][b], d)
        want = ((g["out_u8"][b, v].astype(np.float32) / np.float32(255) - np.float32(0.5)) / np.float32(0.5)).transpose(2, 0, 1)
        assert np.array_equal(ref, want), (b, v, d["order"], float(np.abs(ref - want).max()) * 127.5)
    h, s, v = OA.rgb_to_hsv_u8(g["colours"])
    assert np.array_equal(np.stack([h, s, v], -1), g["rgb2hsv"])
    c = g["colours"].astype(np.int64)
    assert np.array_equal(OA.hsv_to_rgb_u8(c[:, 0], c[:, 1], c[:, 2]), g["hsv2rgb"])
    assert OA.hue_delta(-0.01) == 254 and OA.hue_delta(0.01) == 2 and OA.hue_delta(-0.003) == 0


def test_oracle_colour_conversions_equal_pillow_exhaustively():
    Image = pytest.importorskip("PIL.Image")
    from oracle import augment as OA
    v = np.arange(1 << 24, dtype=np.uint32)
    for lo in range(0, 1 << 24, 1 << 22):                      # four slabs keep the float64 temporaries small
        w = v[lo:lo + (1 << 22)]
        cols = np.stack([(w >> 16) & 255, (w >> 8) & 255, w & 255], -1).astype(np.uint8)
        img = cols.reshape(2048, 2048, 3)
        hsv = np.array(Image.fromarray(img, "RGB").convert("HSV")).reshape(-1, 3)
        h, s, vv = OA.rgb_to_hsv_u8(cols)
        assert np.array_equal(np.stack([h, s, vv], -1), hsv), lo
        rgb = np.array(Image.fromarray(img, "HSV").convert("RGB")).reshape(-1, 3)
        c = cols.astype(np.int64)
        assert np.array_equal(OA.hsv_to_rgb_u8(c[:, 0], c[:, 1], c[:, 2]), rgb), lo
