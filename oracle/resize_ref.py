"""CPU restatement (numpy, integer arithmetic) of Pillow's ``Image.resize(size, BILINEAR)`` for 8-bit images -- the resize
``torchvision.transforms.Resize((224, 224))`` performs on the PIL images of the reference's pipeline
(deephumor_demo.ipynb:565: Resize -> ToTensor -> Normalize; torchvision's default interpolation is BILINEAR and PIL
always antialiases).  TEST INFRASTRUCTURE ONLY: the oracle for ``dh_resize_u8_hwc``.

Pillow is a third-party dependency absent from /root/reference (requirements.txt pins nothing); its published algorithm
(src/libImaging/Resample.c, ImagingResampleHorizontal_8bpc / Vertical_8bpc) is restated here and pinned by
``tests/golden/g9_resize.npz`` -- outputs of the real Pillow 12.2.0 in the build container (``oracle/make_resize_golden.py``):
  * per output coordinate: center = (i + 0.5) * scale, support = max(scale, 1) (triangle filter of width 1, stretched when
    down-scaling = antialiasing), window [int(center - support + 0.5), int(center + support + 0.5)) clipped to the image,
    weights triangle((x - center + 0.5) / max(scale, 1)) normalised to sum 1 (double precision);
  * fixed point: round-half-away weights at 22 fractional bits (PRECISION_BITS = 32 - 8 - 2), accumulator starts at 2^21,
    result = clip(acc >> 22, 0, 255);
  * two passes, the intermediate image is 8-bit (rounded): HORIZONTAL first, then vertical -- except that ``Image.resize`` itself
    (PIL/Image.py, Pillow >= 9.x incl. the 12.2.0 here: ``if self.size[1] > self.size[0] * 100 and size[1] < self.size[1]``) resizes an
    image more than 100 times taller than wide whose height shrinks VERTICALLY first (two C-level resizes).  Found by
    tools/fuzz_encoder.py; pinned by the ``tall_*`` cases of the golden (incl. both sides of the 100 x boundary).
"""
import numpy as np

PRECISION_BITS = 32 - 8 - 2


def coefficients(in_size, out_size):
    """-> (bounds int32 [out, 2] = (first source index, count), weights int32 [out, ksize]) exactly as
    precompute_coeffs + normalize_coeffs_8bpc (Resample.c) produce them for the bilinear filter."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        x = np.arange(xmax, dtype=np.float64)
        arg = (x + xmin - center + 0.5) * ss
        w = np.where(np.abs(arg) < 1.0, 1.0 - np.abs(arg), 0.0)
        ww = w.sum()
        if ww != 0.0:
            w = w / ww
        fixed = np.where(w < 0, -0.5 + w * (1 << PRECISION_BITS), 0.5 + w * (1 << PRECISION_BITS)).astype(np.int64)   # C cast: truncation
        kk[xx, :xmax] = fixed
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _pass(img, bounds, kk, axis):
    """img uint8 [H, W, C]; resamples ``axis`` (0 = vertical, 1 = horizontal)."""
    out_n = bounds.shape[0]
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((out_n,) + src.shape[1:], dtype=np.uint8)
    for i in range(out_n):
        lo, n = bounds[i]
        acc = np.full(src.shape[1:], 1 << (PRECISION_BITS - 1), dtype=np.int64)
        acc += np.tensordot(kk[i, :n].astype(np.int64), src[lo:lo + n], axes=(0, 0))
        out[i] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def resize_bilinear_u8(img, out_h, out_w):
    """img uint8 [H, W, C] -> uint8 [out_h, out_w, C], bit-identical to PIL.Image.resize((out_w, out_h), BILINEAR)."""
    h, w = img.shape[:2]
    if h > 100 * w and out_h < h:                   # PIL/Image.py, Image.resize: very tall images shrink vertically first
        img = _pass(img, *coefficients(h, out_h), axis=0)
        h = out_h
    if w != out_w:
        img = _pass(img, *coefficients(w, out_w), axis=1)
    if h != out_h:
        img = _pass(img, *coefficients(h, out_h), axis=0)
    return img
