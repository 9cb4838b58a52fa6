"""TEST INFRASTRUCTURE -- CPU restatement (numpy) of the image transforms of the reference's input pipeline.

Only ``tests/`` may import this module.  The reference applies, per sample on the host
(``configs/experiment/coop/clipseg.yaml:78-120``): ``albumentations.Resize(interpolation=cv2.INTER_CUBIC)``,
``Affine(scale, translate_percent, rotate, INTER_CUBIC, BORDER_REPLICATE, p=0.2)``, ``PadIfNeeded``, ``CropNonEmptyMaskIfExists``,
``RandomBrightnessContrast(p=0.2)``, ``Normalize``, ``ToTensorV2``.  Both libraries are third-party, un-vendored
(``requirements.txt``: albumentations, opencv-python) and ABSENT from this image, so what follows restates their published
algorithms from their sources as recalled -- **parity unpinned**: no output of cv2 / albumentations is available here to pin them.

* ``resize_cubic_u8``: OpenCV ``cv::resize`` for 8-bit images, ``INTER_CUBIC`` (imgproc/src/resize.cpp): half-pixel centres
  ``fx = (dx + 0.5) * scale - 0.5``, Keys cubic with A = -0.75 evaluated in float (``interpolateCubic``), the four weights quantised to
  ``short`` at 11 fractional bits (``cvRound(w * 2048)``), horizontal pass in int32, vertical pass in int32, result
  ``(v + 2^21) >> 22`` saturated to 8 bits; taps beyond the border replicate the edge pixel.  (OpenCV's SIMD vertical pass goes through
  float and may differ from this scalar path by one grey level on exact ties.)
* ``resize_nearest_u8``: ``INTER_NEAREST``: ``sx = min(floor(dx * scale), w - 1)`` (no half-pixel shift).
* ``warp_affine_cubic_u8`` / ``_nearest``: ``cv::warpAffine`` semantics (dst -> src through the inverse matrix, ``BORDER_REPLICATE``) with
  the same Keys kernel evaluated in float at the exact coordinate (OpenCV rounds the coordinate to 1/32 pixel and uses a 15-bit table:
  a documented approximation here).
* ``brightness_contrast_u8``: albumentations' uint8 path: a 256-entry table ``clip(v * alpha + beta * 255, 0, 255)`` truncated to uint8.
* ``affine_matrix``: albumentations ``Affine`` parameters -> forward matrix about the image centre ``((w - 1) / 2, (h - 1) / 2)``:
  translate . rotate . scale.
"""
from __future__ import annotations

import math

import numpy as np

A_CUBIC = np.float32(-0.75)


def cubic_weights(x: np.ndarray) -> np.ndarray:
    """``interpolateCubic`` in float32: weights of the taps at -1, 0, +1, +2 for fractional position x in [0, 1)."""
    x = x.astype(np.float32)
    A = A_CUBIC
    one = np.float32(1)
    w0 = ((A * (x + one) - np.float32(5) * A) * (x + one) + np.float32(8) * A) * (x + one) - np.float32(4) * A
    w1 = ((A + np.float32(2)) * x - (A + np.float32(3))) * x * x + one
    y = one - x
    w2 = ((A + np.float32(2)) * y - (A + np.float32(3))) * y * y + one
    w3 = one - w0 - w1 - w2
    return np.stack([w0, w1, w2, w3], -1).astype(np.float32)


def _axis_tables(dst: int, src: int):
    scale = src / dst   # double
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    frac = (f - s.astype(np.float32)).astype(np.float32)
    w = np.rint(cubic_weights(frac).astype(np.float64) * 2048.0).astype(np.int64)   # cvRound: half to even
    w = np.clip(w, -32768, 32767)
    idx = np.clip(s[:, None] + np.arange(-1, 3)[None, :], 0, src - 1)
    return idx, w


def resize_cubic_u8(img: np.ndarray, H: int, W: int) -> np.ndarray:
    """img [h, w, C] (or [h, w]) uint8 -> [H, W, C] uint8."""
    squeeze = img.ndim == 2
    src = img[..., None] if squeeze else img
    h, w, _ = src.shape
    xi, xw = _axis_tables(W, w)
    yi, yw = _axis_tables(H, h)
    s = src.astype(np.int64)
    rows = (s[:, xi, :] * xw[None, :, :, None]).sum(2)            # [h, W, C] horizontal pass, scaled by 2^11
    out = (rows[yi, :, :] * yw[:, :, None, None]).sum(1)          # [H, W, C] scaled by 2^22
    out = np.clip((out + (1 << 21)) >> 22, 0, 255).astype(np.uint8)
    return out[..., 0] if squeeze else out


def resize_nearest_u8(img: np.ndarray, H: int, W: int) -> np.ndarray:
    h, w = img.shape[:2]
    ys = np.minimum(np.floor(np.arange(H) * (h / H)).astype(np.int64), h - 1)
    xs = np.minimum(np.floor(np.arange(W) * (w / W)).astype(np.int64), w - 1)
    return img[ys][:, xs]


def affine_matrix(h: int, w: int, scale_x: float, scale_y: float, angle_deg: float, tx_frac: float, ty_frac: float) -> np.ndarray:
    """Forward 2x3 matrix (src -> dst): scale, then rotate, about the image centre, then translate by fractions of the size."""
    cx, cy = (w - 1) / 2.0, (h - 1) / 2.0
    a = math.radians(angle_deg)
    c, s = math.cos(a), math.sin(a)
    L = np.array([[c * scale_x, -s * scale_y], [s * scale_x, c * scale_y]])
    t = np.array([cx + tx_frac * w, cy + ty_frac * h]) - L @ np.array([cx, cy])
    return np.concatenate([L, t[:, None]], 1)


def invert_affine(M: np.ndarray) -> np.ndarray:
    L, t = M[:, :2], M[:, 2]
    Li = np.linalg.inv(L)
    return np.concatenate([Li, (-Li @ t)[:, None]], 1)


def warp_affine_cubic_u8(img: np.ndarray, Minv: np.ndarray) -> np.ndarray:
    """dst[y, x] = cubic sample of img at Minv . (x, y, 1), replicate border; float32 arithmetic, round-half-even, saturate."""
    h, w = img.shape[:2]
    src = (img if img.ndim == 3 else img[..., None]).astype(np.float32)
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float32)
    m = Minv.astype(np.float32)
    sx = m[0, 0] * xs + m[0, 1] * ys + m[0, 2]
    sy = m[1, 0] * xs + m[1, 1] * ys + m[1, 2]
    x0, y0 = np.floor(sx), np.floor(sy)
    wx, wy = cubic_weights(sx - x0), cubic_weights(sy - y0)
    out = np.zeros(src.shape, np.float32)
    for j in range(4):
        yy = np.clip(y0.astype(np.int64) + j - 1, 0, h - 1)
        row = np.zeros(src.shape, np.float32)
        for i in range(4):
            xx = np.clip(x0.astype(np.int64) + i - 1, 0, w - 1)
            row += src[yy, xx] * wx[..., i, None]
        out += row * wy[..., j, None]
    out = np.clip(np.rint(out), 0, 255).astype(np.uint8)
    return out if img.ndim == 3 else out[..., 0]


def warp_affine_nearest_u8(img: np.ndarray, Minv: np.ndarray) -> np.ndarray:
    h, w = img.shape[:2]
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float32)
    m = Minv.astype(np.float32)
    sx = np.rint(m[0, 0] * xs + m[0, 1] * ys + m[0, 2]).astype(np.int64)
    sy = np.rint(m[1, 0] * xs + m[1, 1] * ys + m[1, 2]).astype(np.int64)
    return img[np.clip(sy, 0, h - 1), np.clip(sx, 0, w - 1)]


def brightness_contrast_u8(img: np.ndarray, alpha: float, beta: float) -> np.ndarray:
    lut = np.arange(256, dtype=np.float32)
    if alpha != 1:
        lut = lut * np.float32(alpha)
    if beta != 0:
        lut = lut + np.float32(beta) * np.float32(255)
    return np.clip(lut, 0, 255).astype(np.uint8)[img]


def normalize_chw(img_u8: np.ndarray, mean, std) -> np.ndarray:
    """albumentations Normalize (max_pixel_value 255) + ToTensorV2: (v / 255 - mean) / std, HWC -> CHW float32."""
    x = img_u8.astype(np.float32) / np.float32(255)
    x = (x - np.asarray(mean, np.float32)) / np.asarray(std, np.float32)
    return np.ascontiguousarray(x.transpose(2, 0, 1))
