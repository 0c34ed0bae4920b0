"""Generates ``tests/golden/g9_resize.npz`` with the REAL Pillow in the build container (torchvision's Resize on PIL images
is ``Image.resize(size[::-1], BILINEAR)``): inputs are regenerated from seeds, stored are the expected outputs (the small
case in full, the large cases as a 24 x 24 corner + a checksum).  TEST INFRASTRUCTURE ONLY.
    python oracle/make_resize_golden.py"""
import hashlib
import os

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = {"small": (37, 53, 24, 24), "down": (375, 500, 224, 224), "down_odd": (641, 479, 224, 224), "up": (100, 80, 224, 224),
         "mixed": (300, 150, 224, 224), "same_w": (448, 224, 224, 224),      # (H_in, W_in, H_out, W_out)
         # more than 100 x taller than wide and shrinking in height: Image.resize goes vertical-first (PIL/Image.py); 801 / 800 x 8
         # are the two sides of that boundary, tall_grow grows in height and stays horizontal-first
         "tall_thin": (1275, 9, 224, 224), "tall_801": (801, 8, 50, 40), "tall_800": (800, 8, 50, 40),
         "tall_wide_out": (1168, 11, 41, 205), "tall_grow": (350, 3, 351, 10), "tall_shrink": (350, 3, 349, 50)}


def image(name, h, w):
    g = np.random.Generator(np.random.Philox(key=[sum(name.encode()), h * 1000 + w]))
    base = g.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    smooth = ((np.sin(yy / 7.0)[..., None] + np.cos(xx / 5.0)[..., None]) * 60 + 128).clip(0, 255).astype(np.uint8)
    return np.where(g.random((h, w, 1)) < 0.5, base, smooth).astype(np.uint8)       # noise + structure (saturating edges)


def main():
    out = {}
    for name, (h, w, oh, ow) in CASES.items():
        img = image(name, h, w)
        res = np.asarray(Image.fromarray(img, "RGB").resize((ow, oh), Image.BILINEAR))
        out[f"{name}_shape"] = np.array([h, w, oh, ow])
        out[f"{name}_corner"] = res[:24, :24].copy()
        out[f"{name}_tail"] = res[-8:, -8:].copy()
        out[f"{name}_sha"] = np.frombuffer(hashlib.sha256(res.tobytes()).digest(), dtype=np.uint8)
        if name == "small":
            out["small_full"] = res.copy()
    import PIL
    out["pillow_version"] = np.array(PIL.__version__)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "g9_resize.npz"), **out)
    print("wrote g9_resize.npz with Pillow", PIL.__version__)


if __name__ == "__main__":
    main()
