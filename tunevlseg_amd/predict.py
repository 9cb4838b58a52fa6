"""Predict / offline-eval tail.

* :func:`save_predictions` -- reference ``src/utils/save_utils.py:18-110``: run ``predict_step`` over a loader, resize every
  probability map to its original ``mask_shape`` (bicubic, antialias off) and write it as an 8-bit PNG under
  ``output_masks_dir / mask_name``.  Resize + quantisation is one HIP kernel per map (``tvl_bicubic_resize_u8``); the PNG has the
  three equal channels ``torchvision.utils.save_image`` writes for a single-channel tensor.
* :func:`eval_metrics` -- reference ``scripts/eval_metrics.py:49-137``: per-image IoU / Dice / "all-ones Dice" difference (x100)
  between PNG folders, thresholded at 127 (ground truth) and ``threshold`` (prediction), written as a CSV sorted by file name.
  Offline host tooling in the reference (cv2 + monai on the CPU) and here (PIL + numpy); the monai ``ignore_empty=False``
  conventions are restated from its published source (monai is not installed: *parity unpinned*).
"""
from __future__ import annotations

import csv
import logging
from pathlib import Path
from typing import Any, Iterable

import numpy as np
import torch

from . import hip

log = logging.getLogger(__name__)


def save_predictions(module, dataloader: Iterable[dict[str, Any]], output_masks_dir: str | Path | None = None,
                     overwrite_outputs: bool = False) -> int:
    """Returns the number of masks written (0 if the directory exists and ``overwrite_outputs`` is false, like the reference)."""
    from PIL import Image

    if output_masks_dir is None:
        output_masks_dir = "output_masks"
        log.warning("`output_masks_dir` was not passed in the config. Defaulting to %s", output_masks_dir)
    out_dir = Path(output_masks_dir)
    if out_dir.exists():
        log.warning("%s exists. The output masks may override the previous ones.", out_dir)
        if not overwrite_outputs:
            log.info("`overwrite_outputs` was not passed or if passed as False. So stopping the prediction instead of overwriting.")
            return 0
    n = 0
    for batch in dataloader:
        p = module.predict_step(batch)
        preds, names, shapes = p["preds"], p["mask_name"], p["mask_shape"]
        if names is None or shapes is None:
            raise ValueError("predict batches must carry `mask_name` and `mask_shape` (reference image_text_mask_dataset.py:74-96)")
        if len(names) != preds.shape[0] or len(shapes) != preds.shape[0]:
            raise ValueError("preds / mask_name / mask_shape lengths differ")  # zip(strict=True) in the reference
        for pred, name, shape in zip(preds, names, shapes):
            h, w = (int(v) for v in (shape.tolist() if isinstance(shape, torch.Tensor) else list(shape)))
            grey = hip.bicubic_resize_u8(pred.reshape(pred.shape[-2], pred.shape[-1]).float().contiguous(), h, w).cpu().numpy()
            path = out_dir / str(name)
            path.parent.mkdir(parents=True, exist_ok=True)
            Image.fromarray(np.repeat(grey[:, :, None], 3, axis=2)).save(path)  # save_image: 1 channel -> 3 equal channels
            n += 1
    log.info("Saved %d masks in directory %s", n, out_dir)
    return n


def _load_grey(path: Path) -> np.ndarray:
    from PIL import Image

    if not path.exists():
        raise ValueError(f"Image Not found: {path}")
    return np.asarray(Image.open(path).convert("L"))


def binary_scores(pred: np.ndarray, gt: np.ndarray) -> tuple[float, float]:
    """(IoU, Dice) of boolean maps with monai's ``ignore_empty=False`` rules: empty ground truth scores 1 if the
    prediction is empty too, else 0."""
    inter = float(np.logical_and(pred, gt).sum())
    p_o, y_o = float(pred.sum()), float(gt.sum())
    if y_o > 0:
        return inter / (y_o + p_o - inter), 2.0 * inter / (y_o + p_o)
    empty = 1.0 if p_o <= 0 else 0.0
    return empty, empty


def compute_metrics(gt_img_path: Path, pred_img_path: Path, threshold: int) -> dict[str, float]:
    gt, pred = _load_grey(gt_img_path), _load_grey(pred_img_path)
    if gt.shape != pred.shape:
        raise AssertionError(f"Images {gt_img_path} and {pred_img_path} are of different sizes")
    gt_b, pred_b = gt > 127, pred > threshold
    iou, dice = binary_scores(pred_b, gt_b)
    _, ones_dice = binary_scores(np.ones_like(pred_b), gt_b)
    return {"iou": 100.0 * iou, "dice": 100.0 * dice, "ones_dice_diff": 100.0 * (dice - ones_dice)}


def eval_metrics(seg_path: str | Path, gt_path: str | Path, csv_path: str | Path, threshold: int = 127) -> list[dict[str, Any]]:
    """Writes ``filename,iou,dice,ones_dice_diff`` rows (``%.4f``), sorted by file name; returns the rows."""
    seg_path, gt_path = Path(seg_path), Path(gt_path)
    rows = []
    for f in sorted(seg_path.glob("*.png")):
        try:
            rows.append({"filename": str(f), **compute_metrics(gt_path / f.name, seg_path / f.name, threshold)})
        except Exception as exc:  # the reference reports and skips (eval_metrics.py:117-120)
            print(f"{f} generated an exception: {exc}")
    with open(csv_path, "w", newline="") as fh:
        wr = csv.writer(fh)
        wr.writerow(["filename", "iou", "dice", "ones_dice_diff"])
        for r in rows:
            wr.writerow([r["filename"], *(f"{r[k]:.4f}" for k in ("iou", "dice", "ones_dice_diff"))])
    return rows
