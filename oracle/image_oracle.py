"""CPU restatement of the ViLT image pre-processing the reference runs on its training thread (SURVEY.md §8(f) row F1;
REF/modeling/vilt.py:83-96 -> `ViltProcessor` -> transformers `image_processing_pil_vilt.py:70-98, 127-242`), including the
arithmetic of the third-party resize it delegates to: Pillow's `Image.resize(..., BICUBIC)` (libImaging/Resample.c, the 8-bit
two-pass convolution with 22-bit fixed-point coefficients).  TEST INFRASTRUCTURE ONLY: tests/ and smoke() use it as the checker.

Pinned (tests/test_image_pipeline.py, CPU) against Pillow and against transformers' own ViltImageProcessor, bit for bit.
"""
from __future__ import annotations

import math
from typing import List, Tuple

import numpy as np

PRECISION_BITS = 32 - 8 - 2          # Resample.c: coefficients are rounded to 22 fractional bits
BICUBIC_SUPPORT = 2.0


def _bicubic(x: float) -> float:
    """Resample.c bicubic_filter, a = -0.5 (Keys)."""
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def resample_coeffs(in_size: int, out_size: int) -> Tuple[np.ndarray, np.ndarray, int]:
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for the box (0, in_size).
    Returns (bounds [out,2] = (first input index, count), kk [out, ksize] int32 fixed point, ksize)."""
    scale = filterscale = float(in_size) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = BICUBIC_SUPPORT * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)           # C (int): truncation toward zero
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        if ww != 0.0:
            w = [v / ww for v in w]
        for x, v in enumerate(w):
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk, ksize


def _pass(src: np.ndarray, bounds: np.ndarray, kk: np.ndarray) -> np.ndarray:
    """One 8-bit convolution pass along axis 1 of src [rows, n, C] (ImagingResampleHorizontal_8bpc; the vertical pass is the same
    arithmetic along the other axis): int32 accumulate from 2^21, arithmetic shift, clip to [0, 255]."""
    rows, _, C = src.shape
    out = np.empty((rows, bounds.shape[0], C), dtype=np.uint8)
    s32 = src.astype(np.int64)
    for xx in range(bounds.shape[0]):
        x0, n = int(bounds[xx, 0]), int(bounds[xx, 1])
        acc = (s32[:, x0:x0 + n, :] * kk[xx, :n].astype(np.int64)[None, :, None]).sum(axis=1) + (1 << (PRECISION_BITS - 1))
        acc = ((acc + 2 ** 31) % 2 ** 32 - 2 ** 31)          # the C accumulator is a 32-bit int
        out[:, xx, :] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return out


def pil_bicubic_resize(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """uint8 [H, W, C] -> uint8 [out_h, out_w, C]: horizontal pass, then vertical pass (ImagingResample; it skips a pass whose
    size does not change, which is the same as running it: the identity coefficients reproduce the input exactly)."""
    h, w, _ = img.shape
    bh, kh, _ = resample_coeffs(w, out_w)
    bv, kv, _ = resample_coeffs(h, out_h)
    tmp = _pass(img, bh, kh)                                               # [h, out_w, C]
    return _pass(tmp.transpose(1, 0, 2), bv, kv).transpose(1, 0, 2)        # [out_h, out_w, C]


def vilt_output_size(h: int, w: int, shorter: int = 384, size_divisor: int = 32) -> Tuple[int, int]:
    """image_processing_pil_vilt.py:70-98 with longer = int(1333 / 800 * shorter) (:147)."""
    longer = int(1333 / 800 * shorter)
    scale = shorter / min(h, w)
    if h < w:
        nh, nw = shorter, scale * w
    else:
        nh, nw = scale * h, shorter
    if max(nh, nw) > longer:
        scale = longer / max(nh, nw)
        nh, nw = scale * nh, scale * nw
    nh, nw = int(nh + 0.5), int(nw + 0.5)
    return nh // size_divisor * size_divisor, nw // size_divisor * size_divisor


def normalize_lut() -> np.ndarray:
    """rescale (image_transforms.rescale: float64 product, cast to float32) then normalize with mean = std = 0.5 in float32
    (image_transforms.normalize): 256 possible inputs, so the whole float path is a table."""
    v = (np.arange(256, dtype=np.uint8).astype(np.float64) * (1 / 255)).astype(np.float32)
    return ((v - np.float32(0.5)) / np.float32(0.5)).astype(np.float32)


def vilt_image_batch(images: List[np.ndarray]):
    """uint8 RGB [H, W, 3] images -> (pixel_values [B, 3, Hc, Wc] float32, pixel_mask [B, Hc, Wc] int64) exactly as
    ViltImageProcessor returns them (resize, rescale, normalize, pad bottom/right with zeros to the batch maximum)."""
    lut = normalize_lut()
    outs = []
    for img in images:
        oh, ow = vilt_output_size(img.shape[0], img.shape[1])
        outs.append(lut[pil_bicubic_resize(img, oh, ow)].transpose(2, 0, 1))
    Hc, Wc = max(o.shape[1] for o in outs), max(o.shape[2] for o in outs)
    px = np.zeros((len(outs), 3, Hc, Wc), dtype=np.float32)
    pm = np.zeros((len(outs), Hc, Wc), dtype=np.int64)
    for b, o in enumerate(outs):
        px[b, :, :o.shape[1], :o.shape[2]] = o
        pm[b, :o.shape[1], :o.shape[2]] = 1
    return px, pm
