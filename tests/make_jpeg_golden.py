#!/usr/bin/env python3
"""Writes tests/golden/jpeg_cases.npz: small JPEG files (bytes) and the RGB pixels libjpeg-turbo (through Pillow)
decodes them to.  The host decoder (simple-path-tracer_amd/csrc/host/jpeg.cpp) restates the IJG arithmetic - islow
IDCT, fancy upsampling, 16-bit YCbCr -> RGB - and must reproduce these pixels exactly (tests/test_jpeg.py).
Needs Pillow; the fixture is committed so the tests do not.  Usage: python tests/make_jpeg_golden.py"""
import io
import os

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))


def synthetic(h, w, seed):
    """smooth colour gradients + an edge + noise: every coefficient class and both chroma signs show up"""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    img = np.stack([127 + 120 * np.sin(x / 7.0 + y / 13.0), 127 + 120 * np.cos(x / 11.0 - y / 5.0), 255.0 * ((x + 2 * y) % 37 > 18)], axis=-1)
    img += rng.normal(0, 12, img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)


CASES = [  # name, (h, w), mode, kwargs
    ("base_444", (40, 52), "RGB", dict(quality=85, subsampling=0)),
    ("base_422_odd", (37, 45), "RGB", dict(quality=60, subsampling=1)),
    ("base_420_odd", (35, 51), "RGB", dict(quality=75, subsampling=2)),
    ("base_420_restart", (48, 64), "RGB", dict(quality=50, subsampling=2, restart_marker_blocks=2)),
    ("prog_420", (41, 57), "RGB", dict(quality=80, subsampling=2, progressive=True)),
    ("prog_444_restart", (24, 40), "RGB", dict(quality=92, subsampling=0, progressive=True, restart_marker_blocks=1)),
    ("gray", (33, 29), "L", dict(quality=70)),
    ("gray_prog", (16, 16), "L", dict(quality=95, progressive=True)),
    ("tiny_1x1", (1, 1), "RGB", dict(quality=75, subsampling=2)),
    ("optimized_tables", (30, 30), "RGB", dict(quality=65, subsampling=2, optimize=True)),
]


def main():
    out = {}
    for k, (name, (h, w), mode, kw) in enumerate(CASES):
        img = Image.fromarray(synthetic(h, w, k)).convert(mode)
        buf = io.BytesIO()
        img.save(buf, format="JPEG", **kw)
        data = buf.getvalue()
        rgb = np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))
        out[name + "_jpg"] = np.frombuffer(data, dtype=np.uint8)
        out[name + "_rgb"] = rgb
        print(name, len(data), "bytes", rgb.shape)
    np.savez_compressed(os.path.join(HERE, "golden", "jpeg_cases.npz"), **out)


if __name__ == "__main__":
    main()
