#!/usr/bin/env python3
"""Generates tests/golden/jpeg_ref.npz: 8-bit pictures (inputs) and the bytes of the file the REFERENCE's own Image::saveJPG
(src/image.cpp:60-74 -> stbi_write_jpg, quality 90; compiled in place into oracle/_ref/libref_loaders.so) writes for each (expected
outputs).  Run in the build container only (needs /root/reference):   python tests/golden/make_jpeg_golden.py"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import binding as ob  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def pictures():
    """Sizes that are and are not multiples of 8, one pixel, flat, gradients, noise (long runs of zeros and none), saturated."""
    rng = np.random.default_rng(90)
    out = []
    for (w, h) in [(1, 1), (8, 8), (7, 5), (16, 9), (33, 17), (64, 48), (3, 40), (40, 3)]:
        yy, xx = np.mgrid[0:h, 0:w]
        grad = np.stack([(xx * 255 // max(w - 1, 1)), (yy * 255 // max(h - 1, 1)), ((xx + yy) * 255 // max(w + h - 2, 1))], -1).astype(np.uint8)
        noise = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        soft = (128 + 40 * np.sin(xx / 3.0)[..., None] + rng.normal(0, 3, (h, w, 3))).clip(0, 255).astype(np.uint8)
        flat = np.full((h, w, 3), rng.integers(0, 256, 3, dtype=np.uint8), np.uint8)
        sat = np.where(rng.uniform(size=(h, w, 3)) < 0.5, 0, 255).astype(np.uint8)
        out += [grad, noise, soft, flat, sat]
    return out


def reference_bytes(R, img):
    """The reference's file for an 8-bit picture: Image holds floats; saveJPG clamps to [0, 1], scales by 255 and truncates, so
    (b + 0.5) / 255 comes back as b."""
    h, w, _ = img.shape
    px = np.ascontiguousarray((img.astype(np.float32) + np.float32(0.5)) / np.float32(255.0))
    assert np.array_equal((np.clip(px, 0, 1) * np.float32(255.0)).astype(np.uint8), img)
    with tempfile.TemporaryDirectory() as d:
        base = os.path.join(d, "shot")
        saved = os.dup(1); os.dup2(2, 1)                 # saveJPG prints "Saved ..." on stdout
        try:
            R.ref_save_jpg(base.encode(), w, h, px.reshape(-1))
        finally:
            os.dup2(saved, 1); os.close(saved)
        return np.frombuffer(open(base + ".jpg", "rb").read(), np.uint8).copy()


def main():
    R = ob.ref_loaders()
    assert R is not None, "build oracle/_ref first (make -C oracle ref)"
    g = {}
    for i, img in enumerate(pictures()):
        g[f"in_{i}"] = img
        g[f"out_{i}"] = reference_bytes(R, img)
    np.savez_compressed(os.path.join(OUT, "jpeg_ref.npz"), **g)
    print("wrote jpeg_ref.npz:", len(g) // 2, "pictures,", sum(v.size for k, v in g.items() if k.startswith("out_")), "bytes of JPEG")


if __name__ == "__main__":
    main()
