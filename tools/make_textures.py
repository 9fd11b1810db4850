"""Decode the reference's two JPEG textures (presets/textures/*.jpg) to raw RGB8 fixtures.

The reference decodes them with image 0.25.5 -> zune-jpeg 0.4.14 (src/scene/image_loader.rs:5-8), which is
not available here; PIL/libjpeg is used instead.  Both files are baseline 8-bit 4:4:4 1024x512, so the
only possible difference is IDCT rounding (<= +-1 LSB per channel): texel parity is UNPINNED.
Run in the dev container only (the GPU box has no /root/reference); the .npz is committed.
"""
import os
import sys

import numpy as np
from PIL import Image

src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/presets/textures"
dst = os.path.join(os.path.dirname(__file__), "..", "raytracer_amd", "assets", "textures.npz")
out = {}
for name, key in (("earthmap.jpg", "earthmap"), ("mars_1k_color.jpg", "mars_1k_color")):
    im = Image.open(os.path.join(src, name)).convert("RGB")
    out[key] = np.asarray(im, dtype=np.uint8)
    print(key, out[key].shape, out[key].dtype, int(out[key].astype(np.uint64).sum()))
np.savez_compressed(dst, **out)
print("wrote", os.path.abspath(dst), os.path.getsize(dst), "bytes")
