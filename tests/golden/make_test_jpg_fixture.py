"""Generates tests/golden/test_jpg_rgb.npy: the DECODED pixels of the reference's one real input image,
/root/reference/assets/image/test.jpg (540 x 360; `example/inference.rs` and `example/correctness.rs` read it through the
`image` crate's JPEG decoder and `to_rgb8`). The fixture is DATA -- a uint8 [360, 540, 3] array -- not the file itself.

Decoder: Pillow (libjpeg-turbo, ISLOW IDCT). The `image` crate's pure-Rust decoder may differ from it by +-1 in single
pixels; BASELINE config 3-(iii) needs "the test.jpg frame", and both sides of every test here read THIS array.
Run in the build container (it reads /root/reference): python tests/golden/make_test_jpg_fixture.py
"""
import hashlib
import os

import numpy as np
from PIL import Image

SRC = "/root/reference/assets/image/test.jpg"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "test_jpg_rgb.npy")

if __name__ == "__main__":
    rgb = np.asarray(Image.open(SRC).convert("RGB"), dtype=np.uint8)
    assert rgb.shape == (360, 540, 3), rgb.shape
    np.save(DST, rgb)
    print(DST, rgb.shape, rgb.dtype, "sha256", hashlib.sha256(rgb.tobytes()).hexdigest()[:16], "mean", rgb.mean(axis=(0, 1)))
