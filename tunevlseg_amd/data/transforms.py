"""The albumentations pipeline of the reference's experiment files, as a DEVICE pipeline over the whole batch.

The reference (``configs/experiment/coop/clipseg.yaml:78-120``) composes, per sample on a DataLoader worker:
``Resize(INTER_CUBIC) -> Affine(p=0.2) -> PadIfNeeded -> CropNonEmptyMaskIfExists -> RandomBrightnessContrast(p=0.2) -> Normalize ->
ToTensorV2``.  Here the classes below carry the SAME constructor keywords (the YAML's ``_target_: albumentations.X`` entries are mapped
onto them by ``config_loader.TARGET_MAP``) but hold parameters only: samples leave the dataset decoded and untouched (uint8, their own
size), the collator packs the ragged batch, and :class:`DeviceTransform` runs the pipeline on the GPU in two launches --
``tvl_resize_u8`` (OpenCV's 8-bit ``INTER_CUBIC`` for the images, ``INTER_NEAREST`` for the masks) and ``tvl_augment_u8`` (per-sample
affine warp + brightness / contrast + normalisation, one read of the uint8 batch, one write of the fp32 network input).  The random
draws (who is augmented, with what parameters) are made on the host per batch from a seeded ``numpy`` generator.

cv2 / albumentations are not in the image: their arithmetic is restated from the published algorithms (the tests hold the kernels to a
numpy restatement of the same definitions) and stays "unpinned" against the libraries themselves.  ``PadIfNeeded`` and
``CropNonEmptyMaskIfExists`` at the resized size are identities (every experiment file of the reference uses ``img_size`` for all
three); other sizes raise."""
from __future__ import annotations

import math
from collections.abc import Mapping, Sequence
from typing import Any

import numpy as np
import torch

from .. import hip

# cv2's constants, for ${import_eval:cv2.X} when cv2 itself is absent (config_loader.import_resolver)
CV2_CONSTANTS = {"INTER_NEAREST": 0, "INTER_LINEAR": 1, "INTER_CUBIC": 2, "INTER_AREA": 3, "BORDER_CONSTANT": 0, "BORDER_REPLICATE": 1,
                 "BORDER_REFLECT": 2, "BORDER_REFLECT_101": 4}


class _Op:
    p = 1.0

    def __repr__(self) -> str:
        return f"{type(self).__name__}({', '.join(f'{k}={v!r}' for k, v in vars(self).items())})"


class Resize(_Op):
    def __init__(self, height: int, width: int, interpolation: int = 1, p: float = 1.0, **_ignored: Any) -> None:
        self.height, self.width, self.interpolation, self.p = int(height), int(width), int(interpolation), float(p)
        if self.interpolation != CV2_CONSTANTS["INTER_CUBIC"]:
            raise NotImplementedError(f"Resize: only cv2.INTER_CUBIC (2) is built (the reference's setting), got {interpolation}")


class Affine(_Op):
    def __init__(self, scale=1.0, translate_percent=None, rotate=0.0, shear=0.0, interpolation: int = 1, mode: int = 0, p: float = 0.5,
                 keep_ratio: bool = False, **_ignored: Any) -> None:
        rng = lambda v: (float(v), float(v)) if not isinstance(v, (Sequence, tuple, list)) else (float(v[0]), float(v[1]))  # noqa: E731
        self.scale, self.rotate = rng(scale), rng(rotate)
        self.translate_percent = rng(translate_percent) if translate_percent is not None else (0.0, 0.0)
        self.keep_ratio, self.p = bool(keep_ratio), float(p)
        if shear not in (0, 0.0, None):
            raise NotImplementedError("Affine: shear is not built (the reference uses none)")
        if int(interpolation) != CV2_CONSTANTS["INTER_CUBIC"] or int(mode) != CV2_CONSTANTS["BORDER_REPLICATE"]:
            raise NotImplementedError("Affine: only INTER_CUBIC + BORDER_REPLICATE are built (the reference's setting)")

    def sample(self, g: np.random.Generator, h: int, w: int) -> np.ndarray:
        """Inverse matrix (dst -> src) of one draw: scale, rotate about the image centre, translate by fractions of the size."""
        sx = g.uniform(*self.scale)
        sy = sx if self.keep_ratio else g.uniform(*self.scale)
        ang = math.radians(g.uniform(*self.rotate))
        tx, ty = g.uniform(*self.translate_percent) * w, g.uniform(*self.translate_percent) * h
        cx, cy = (w - 1) / 2.0, (h - 1) / 2.0
        c, s = math.cos(ang), math.sin(ang)
        L = np.array([[c * sx, -s * sy], [s * sx, c * sy]])
        t = np.array([cx + tx, cy + ty]) - L @ np.array([cx, cy])
        Li = np.linalg.inv(L)
        return np.concatenate([Li, (-Li @ t)[:, None]], 1).astype(np.float32)


class PadIfNeeded(_Op):
    def __init__(self, min_height: int, min_width: int, border_mode: int = 4, p: float = 1.0, **_ignored: Any) -> None:
        self.min_height, self.min_width = int(min_height), int(min_width)


class CropNonEmptyMaskIfExists(_Op):
    def __init__(self, height: int, width: int, p: float = 1.0, **_ignored: Any) -> None:
        self.height, self.width = int(height), int(width)


class RandomBrightnessContrast(_Op):
    def __init__(self, brightness_limit=0.2, contrast_limit=0.2, brightness_by_max: bool = True, p: float = 0.5, **_ignored: Any) -> None:
        lim = lambda v: (-float(v), float(v)) if not isinstance(v, (Sequence, tuple, list)) else (float(v[0]), float(v[1]))  # noqa: E731
        self.brightness_limit, self.contrast_limit, self.p = lim(brightness_limit), lim(contrast_limit), float(p)
        if not brightness_by_max:
            raise NotImplementedError("RandomBrightnessContrast: brightness_by_max=False (image mean) is not built")

    def sample(self, g: np.random.Generator) -> tuple[float, float]:
        return 1.0 + g.uniform(*self.contrast_limit), g.uniform(*self.brightness_limit)


class Normalize(_Op):
    def __init__(self, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225), max_pixel_value: float = 255.0, p: float = 1.0, **_ignored: Any) -> None:
        self.mean, self.std = tuple(float(v) for v in mean), tuple(float(v) for v in std)
        if float(max_pixel_value) != 255.0:
            raise NotImplementedError("Normalize: max_pixel_value must be 255")


class ToTensorV2(_Op):
    def __init__(self, transpose_mask: bool = False, p: float = 1.0, **_ignored: Any) -> None:
        self.transpose_mask = bool(transpose_mask)


class Compose:
    """``albumentations.Compose(transforms=[...])`` as a parameter holder; :meth:`plan` checks that the list is one this pipeline runs."""

    def __init__(self, transforms: Sequence[_Op], p: float = 1.0, **_ignored: Any) -> None:
        self.transforms = list(transforms)

    def plan(self) -> dict[str, Any]:
        ops = {type(t).__name__: t for t in self.transforms}
        unknown = [type(t).__name__ for t in self.transforms if not isinstance(t, _Op)]
        if unknown:
            raise NotImplementedError(f"transforms outside the reference's pipeline: {unknown}")
        if "Resize" not in ops or "Normalize" not in ops:
            raise ValueError("the device pipeline needs a Resize and a Normalize entry (every experiment file of the reference has both)")
        r = ops["Resize"]
        for name in ("PadIfNeeded", "CropNonEmptyMaskIfExists"):
            o = ops.get(name)
            if o is not None:
                hh, ww = (o.min_height, o.min_width) if name == "PadIfNeeded" else (o.height, o.width)
                if (hh, ww) != (r.height, r.width):
                    raise NotImplementedError(f"{name}({hh}, {ww}) differs from Resize({r.height}, {r.width}): only the identity case is built")
        return {"size": (r.height, r.width), "affine": ops.get("Affine"), "bc": ops.get("RandomBrightnessContrast"), "normalize": ops["Normalize"]}

    def __call__(self, **_kw):  # the albumentations calling convention is NOT how this pipeline runs
        raise RuntimeError("transforms run on the device for the whole batch (data.transforms.DeviceTransform), not per sample on the host")


class RaggedCollator:
    """Collate decoded samples of different sizes: ``image_bytes`` uint8 [sum h*w*3] + ``image_hw`` int32 [B, 2] (+ the same for masks),
    token rows padded to the longest (``PadToLongestCollator``), everything else by the default rules.  One pinned host buffer per
    batch = one H2D copy."""

    def __init__(self, token_collator) -> None:
        self.tokens = token_collator

    def __call__(self, features: list[Mapping[str, Any]]) -> dict[str, Any]:
        imgs = [np.ascontiguousarray(np.asarray(f["image"], dtype=np.uint8)) for f in features]
        msks = [np.ascontiguousarray(np.asarray(f["mask"], dtype=np.uint8)) for f in features]
        for a, m in zip(imgs, msks):
            if a.ndim != 3 or a.shape[2] != 3 or m.shape[:2] != a.shape[:2]:
                raise ValueError(f"expected an [h, w, 3] uint8 image and an [h, w] mask of the same size, got {a.shape} / {m.shape}")
        rest = self.tokens([{k: v for k, v in f.items() if k not in ("image", "mask")} for f in features])
        rest["image_bytes"] = torch.from_numpy(np.concatenate([a.reshape(-1) for a in imgs]))
        rest["mask_bytes"] = torch.from_numpy(np.concatenate([m.reshape(-1) for m in msks]))
        rest["image_hw"] = torch.tensor([a.shape[:2] for a in imgs], dtype=torch.int32)
        return rest


class DeviceTransform:
    """Ragged uint8 batch -> ``{"image": fp32 [B,3,H,W], "mask": fp32 [B,1,H,W], input_ids, attention_mask, ...}`` on the device."""

    def __init__(self, compose: Compose, device="cuda", seed: int = 0) -> None:
        self.plan = compose.plan()
        self.device = torch.device(device)
        self.rng = np.random.default_rng(seed)

    def set_epoch(self, epoch: int, seed: int = 0, rank: int = 0) -> None:
        self.rng = np.random.default_rng([seed, rank, epoch])

    @staticmethod
    def _h2d(t: torch.Tensor, dev) -> torch.Tensor:
        """Small host tensor -> device without stalling the host: from pageable memory the copy is synchronous, i.e. the host waits until
        the stream has drained (the whole previous train step) and host enqueue stops overlapping GPU execution; from pinned memory it is
        one more asynchronous operation in stream order."""
        if torch.device(dev).type == "cuda" and not t.is_cuda and not t.is_pinned():
            t = t.pin_memory()
        return t.to(dev, non_blocking=True)

    def __call__(self, batch: Mapping[str, Any]) -> dict[str, Any]:
        out = {k: v for k, v in batch.items() if k not in ("image_bytes", "mask_bytes", "image_hw")}
        hw = batch["image_hw"]
        B = hw.shape[0]
        H, W = self.plan["size"]
        px = hw[:, 0].long() * hw[:, 1].long()
        m_offs = torch.cat([torch.zeros(1, dtype=torch.int64), px.cumsum(0)[:-1]])
        dev = self.device
        img_b, msk_b = self._h2d(batch["image_bytes"], dev), self._h2d(batch["mask_bytes"], dev)
        hw_d = self._h2d(hw, dev)
        img = hip.resize_u8(img_b, self._h2d(3 * m_offs, dev), hw_d, 3, H, W, hip.INTER_CUBIC)
        msk = hip.resize_u8(msk_b, self._h2d(m_offs, dev), hw_d, 1, H, W, hip.INTER_NEAREST).view(B, H, W)
        params = np.zeros((B, 8), np.float32)
        params[:, [0, 4, 6]] = 1.0
        flags = np.zeros(B, np.int32)
        aff, bc = self.plan["affine"], self.plan["bc"]
        for b in range(B):
            if aff is not None and self.rng.random() < aff.p:
                params[b, :6] = aff.sample(self.rng, H, W).reshape(-1)
                flags[b] |= 1
            if bc is not None and self.rng.random() < bc.p:
                params[b, 6], params[b, 7] = bc.sample(self.rng)
                flags[b] |= 2
        nz = self.plan["normalize"]
        image, mask = hip.augment_u8(img, msk, self._h2d(torch.from_numpy(params), dev), self._h2d(torch.from_numpy(flags), dev), nz.mean, nz.std)
        out["image"], out["mask"] = image, mask
        for k in ("input_ids", "attention_mask"):
            out[k] = self._h2d(batch[k], dev)
        return out
