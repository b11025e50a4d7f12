#!/usr/bin/env python3
"""Generate tests/golden/jpeg_golden.npz with libjpeg-turbo (through Pillow).

The reference's JPEG stage is libjpeg-turbo driven by jpeg_io.hpp:211-330
(writeJPEG: JCS_RGB, jpeg_set_defaults, jpeg_set_quality(q, TRUE)) and
jpeg_io.hpp:90-192 (readJPEG: library defaults).  Pillow bundles libjpeg-turbo
and drives it with the same settings when called as below, so its output is
the golden vector for that third-party stage.  Pillow exists only in the build
container; this script is run there once and the vectors are committed.

    python tests/golden/make_jpeg_golden.py
"""
import io
import os

import numpy as np
from PIL import Image, features

assert features.check_feature("libjpeg_turbo"), "Pillow must bundle libjpeg-turbo"

CASES = [  # (h, w, quality, kind)
    (1, 256, 85, "smooth"), (5, 256, 85, "smooth"), (8, 256, 75, "smooth"), (9, 256, 85, "rand"),
    (17, 256, 85, "smooth"), (40, 256, 1, "smooth"), (16, 256, 100, "rand"),
    (1, 2048, 85, "smooth"), (1, 2049, 75, "smooth"), (1, 37, 85, "rand"),
    (1, 1, 85, "rand"), (1, 2, 75, "rand"), (1, 4, 75, "rand"), (3, 5, 50, "rand"), (5, 2, 85, "rand"),
    (31, 33, 30, "smooth"),
]


def image(h, w, kind, rng):
    if kind == "rand":
        return rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    y, x = np.mgrid[0:h, 0:w]
    img = np.stack([(x * 3 + y) % 256, (x + 2 * y) % 256, (x * y // 7) % 256], -1)
    return np.clip(img + rng.integers(-8, 9, img.shape), 0, 255).astype(np.uint8)


def main():
    rng = np.random.default_rng(0x4A50)
    out = {}
    for i, (h, w, q, kind) in enumerate(CASES):
        img = image(h, w, kind, rng)
        b = io.BytesIO()
        Image.fromarray(img, "RGB").save(b, "JPEG", quality=q, subsampling=2, optimize=False)
        jpg = b.getvalue()
        dec = np.array(Image.open(io.BytesIO(jpg)).convert("RGB"))
        out["in_%02d" % i] = img
        out["q_%02d" % i] = np.int32(q)
        out["jpg_%02d" % i] = np.frombuffer(jpg, dtype=np.uint8)
        out["dec_%02d" % i] = dec
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "jpeg_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(CASES), "cases")


if __name__ == "__main__":
    main()
