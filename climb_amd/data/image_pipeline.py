"""Image half of the input pipeline on the device.

The reference calls `ViltProcessor` on its training thread every step (REF/modeling/vilt.py:83-96): Pillow bicubic resize to
shortest edge 384 (longest capped at 640, both floored to multiples of 32), x 1/255, normalise with mean = std = 0.5, pad to the
batch maximum, build `pixel_mask`, then copy 113 MB of fp32 pixels + 75 MB of int64 mask to the GPU (64 images).  Here the host only
 * decodes to uint8 RGB (what PIL hands over anyway),
 * computes output sizes and Pillow's fixed-point resampling coefficients (a few KB per distinct size, cached),
 * copies the RAW bytes (<= 1/4 of the fp32 volume, no mask) through one pinned staging buffer,
and `climb_image_resample` / `climb_image_normalize_pad` (climb_amd/csrc/image.hip) produce `pixel_values` and `pixel_mask`
directly in HBM -- bit-identical to the reference's tensors (tests/test_image_pipeline.py).
"""
from __future__ import annotations

import functools
import math
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch

from .. import _lib

PRECISION_BITS = 22            # Pillow Resample.c: 32 - 8 - 2
_DESC = 16


def vilt_output_size(h: int, w: int, shorter: int = 384, size_divisor: int = 32) -> Tuple[int, int]:
    """transformers image_processing_pil_vilt.py:70-98 (`longer = int(1333 / 800 * shorter)`, :147)."""
    longer = int(1333 / 800 * shorter)
    scale = shorter / min(h, w)
    nh, nw = (shorter, scale * w) if h < w else (scale * h, shorter)
    if max(nh, nw) > longer:
        scale = longer / max(nh, nw)
        nh, nw = scale * nh, scale * nw
    nh, nw = int(nh + 0.5), int(nw + 0.5)
    return nh // size_divisor * size_divisor, nw // size_divisor * size_divisor


@functools.lru_cache(maxsize=4096)
def resample_coefficients(in_size: int, out_size: int) -> Tuple[np.ndarray, np.ndarray, int]:
    """Pillow's bicubic (a = -0.5, support 2) coefficient table for resampling `in_size` -> `out_size` samples, in its 22-bit
    fixed point (Resample.c precompute_coeffs + normalize_coeffs_8bpc), vectorised over the output index.  Every floating-point
    operation is done in the order the C code does it (float64), including the running sum of the weights, so the integers
    match Pillow's bit for bit.  Returns (bounds [out, 2] int32 = first input sample and count, kk [out, ksize] int32, ksize)."""
    scale = filterscale = float(in_size) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    ss = 1.0 / filterscale
    center = (np.arange(out_size, dtype=np.float64) + 0.5) * scale
    xmin = np.maximum(np.trunc(center - support + 0.5).astype(np.int64), 0)
    xmax = np.minimum(np.trunc(center + support + 0.5).astype(np.int64), in_size) - xmin
    x = np.arange(ksize, dtype=np.int64)[None, :]
    arg = (((x + xmin[:, None]).astype(np.float64) - center[:, None]) + 0.5) * ss
    a = np.abs(arg)
    w = np.where(a < 1.0, ((1.5 * a - 2.5) * a) * a + 1, np.where(a < 2.0, (((a - 5) * a + 8) * a - 4) * -0.5, 0.0))
    w = np.where(x < xmax[:, None], w, 0.0)
    ww = np.cumsum(w, axis=1)[:, -1:]                      # sequential sum, as the C loop (np.sum would add pairwise)
    w = np.where(ww != 0.0, w / np.where(ww != 0.0, ww, 1.0), w)
    fx = w * float(1 << PRECISION_BITS)
    kk = np.where(w < 0, np.trunc(-0.5 + fx), np.trunc(0.5 + fx)).astype(np.int32)
    bounds = np.stack([xmin, xmax], axis=1).astype(np.int32)
    return bounds, kk, ksize


def normalize_table() -> np.ndarray:
    """rescale (float64 product cast to float32, image_transforms.rescale) then (v - 0.5) / 0.5 in float32
    (image_transforms.normalize) for each of the 256 possible bytes."""
    v = (np.arange(256, dtype=np.uint8).astype(np.float64) * (1 / 255)).astype(np.float32)
    return ((v - np.float32(0.5)) / np.float32(0.5)).astype(np.float32)


def _as_rgb_u8(img) -> np.ndarray:
    if isinstance(img, np.ndarray):
        a = img
    elif isinstance(img, torch.Tensor):
        a = img.cpu().numpy()
    else:                                                    # PIL.Image
        a = np.asarray(img.convert("RGB"))
    if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3:
        raise TypeError(f"expected uint8 RGB [H, W, 3] images, got {a.dtype} {a.shape}")
    return np.ascontiguousarray(a)


class DeviceImagePipeline:
    """`pipe(list of PIL / uint8 [H,W,3] images) -> {'pixel_values': [B,3,Hc,Wc] f32, 'pixel_mask': [B,Hc,Wc] i64}` on `device`."""

    def __init__(self, device, shorter: int = 384, size_divisor: int = 32):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("DeviceImagePipeline runs on the HIP device; there is no CPU fallback (use ViltProcessor on the host)")
        _lib.load()
        self.shorter, self.size_divisor = shorter, size_divisor
        self.lut = torch.from_numpy(normalize_table()).to(self.device)
        # pinned staging ring: a buffer is rewritten only after the event recorded behind its last host->device copy has
        # completed, so a prefetching caller (PrefetchLoader: worker thread, side stream, no host syncs) can never overwrite
        # raw bytes a DMA is still reading
        self._ring = [None] * 3          # [pinned uint8 tensor, torch.cuda.Event]
        self._ring_at = 0

    def plan(self, shapes: Sequence[Tuple[int, int]]):
        """Host-side layout of one batch: descriptor table, coefficient arena, arena sizes, canvas size."""
        table = np.zeros((len(shapes), _DESC), dtype=np.int64)
        coef_parts: List[np.ndarray] = []
        coef_at: Dict[Tuple[int, int], Tuple[int, int, int]] = {}
        coef_len = 0

        def coef_for(n_in, n_out):
            nonlocal coef_len
            key = (n_in, n_out)
            if key not in coef_at:
                bounds, kk, ksize = resample_coefficients(n_in, n_out)
                coef_at[key] = (coef_len, coef_len + kk.size, ksize)
                coef_parts.extend([kk.reshape(-1), bounds.reshape(-1)])
                coef_len += kk.size + bounds.size
            return coef_at[key]

        src_off = tmp_off = dst_off = 0
        max_elems = 0
        Hc = Wc = 0
        for b, (sh, sw) in enumerate(shapes):
            dh, dw = vilt_output_size(sh, sw, self.shorter, self.size_divisor)
            kh, bh, ksh = coef_for(sw, dw)
            kv, bv, ksv = coef_for(sh, dh)
            table[b, :13] = (src_off, tmp_off, dst_off, sh, sw, dh, dw, kh, bh, ksh, kv, bv, ksv)
            src_off += sh * sw * 3
            tmp_off += sh * dw * 3
            dst_off += dh * dw * 3
            max_elems = max(max_elems, sh * dw * 3, dh * dw * 3)
            Hc, Wc = max(Hc, dh), max(Wc, dw)
        coef = np.concatenate(coef_parts).astype(np.int32)
        return table, coef, (src_off, tmp_off, dst_off), max_elems, (Hc, Wc)

    def __call__(self, images) -> Dict[str, torch.Tensor]:
        arrs = [_as_rgb_u8(im) for im in images]
        table, coef, (nsrc, ntmp, ndst), max_elems, (Hc, Wc) = self.plan([a.shape[:2] for a in arrs])
        slot = self._ring_at
        self._ring_at = (slot + 1) % len(self._ring)
        if self._ring[slot] is not None:
            self._ring[slot][1].synchronize()                 # the copy that last read this buffer has finished
        if self._ring[slot] is None or self._ring[slot][0].numel() < nsrc:
            self._ring[slot] = [torch.empty(max(nsrc, 1 << 20), dtype=torch.uint8).pin_memory(), torch.cuda.Event()]
        pinned, copied = self._ring[slot]
        stage = pinned.numpy()
        o = 0
        for a in arrs:
            stage[o:o + a.size] = a.reshape(-1)
            o += a.size
        dev = self.device
        src = torch.empty(nsrc, dtype=torch.uint8, device=dev)
        src.copy_(pinned[:nsrc], non_blocking=True)
        copied.record(torch.cuda.current_stream())
        tmp = torch.empty(ntmp, dtype=torch.uint8, device=dev)
        dst = torch.empty(ndst, dtype=torch.uint8, device=dev)
        table_d = torch.from_numpy(table.reshape(-1)).to(dev, non_blocking=True)
        coef_d = torch.from_numpy(coef).to(dev, non_blocking=True)
        B = len(arrs)
        pixel_values = torch.empty(B, 3, Hc, Wc, dtype=torch.float32, device=dev)
        pixel_mask = torch.empty(B, Hc, Wc, dtype=torch.int64, device=dev)
        st = torch.cuda.current_stream().cuda_stream
        _lib.call("climb_image_resample", src, tmp, dst, coef_d, table_d, B, max_elems, st)
        _lib.call("climb_image_normalize_pad", dst, table_d, self.lut, pixel_values, pixel_mask, B, Hc, Wc, st)
        # largest number of valid patches of any image (known on the host): lets the engine size packed sequences without a sync
        pixel_mask._climb_max_patches = int(((table[:, 5] // 32) * (table[:, 6] // 32)).max())
        return {"pixel_values": pixel_values, "pixel_mask": pixel_mask}
