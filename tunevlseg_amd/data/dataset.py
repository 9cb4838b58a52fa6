"""The reference's image / mask / prompt dataset on this repo's input side.

Mirrors ``src/data/core_datasets/image_text_mask_dataset.py:20-128`` and ``basedataset.py:25-130``: same wire format (``image_dir``,
``mask_dir``, a task file = JSON list of ``{"img_name", "mask_name", "prompts": {"p0": ..., "p1": ...}}``), same constructor keywords,
same prompt selection (``prompt_index`` >= 0 -> ``p<index>``; negative -> a random key other than ``p0``; a list-valued prompt -> a
random element; ``override_prompt``; ``insert_stop_at_last``), same item keys (``image``, ``mask``, ``mask_shape``, ``mask_name``,
``prompt`` + the tokenizer's ``input_ids`` / ``attention_mask``).

What differs, deliberately: decoding is PIL (cv2 and albumentations are not in the image), and samples stay **uint8** -- HWC RGB image,
grey-level mask -- so that ``/255``, ``A.Normalize`` and ``ToTensorV2`` happen on the device for the whole batch (``DeviceBatchPrep``:
``tvl_normalize_u8`` / ``tvl_mask_u8``) instead of per sample on the host.  The default geometric transform is a plain resize to
``img_size`` (PIL bicubic for the image, nearest for the mask); cv2's ``INTER_CUBIC`` is not restated, so resized pixels are not
bit-comparable with the reference's loader ("unpinned")."""
from __future__ import annotations

import json
import random
from collections.abc import Callable, Mapping, Sequence
from pathlib import Path
from typing import Any

import numpy as np
import torch
from torch.utils.data import Dataset

from .. import hip


def load_image(path, mode: str) -> np.ndarray:
    """``mode`` "RGB" -> [H, W, 3] uint8, "L" -> [H, W] uint8 (reference ``load_image`` + BGR2RGB, basedataset.py:66-97)."""
    from PIL import Image

    p = Path(path)
    if not p.exists():
        raise ValueError(f"Image not found in the path: {p}")
    with Image.open(p) as im:
        return np.asarray(im.convert(mode))


class ResizeTransform:
    """``transforms(image=..., mask=...) -> {"image", "mask"}`` with the albumentations calling convention: resize to a square."""

    def __init__(self, img_size: int) -> None:
        self.size = int(img_size)

    def __call__(self, *, image: np.ndarray, mask: np.ndarray) -> dict[str, np.ndarray]:
        from PIL import Image

        s = (self.size, self.size)
        img = np.asarray(Image.fromarray(image).resize(s, Image.BICUBIC))
        m = np.asarray(Image.fromarray(mask).resize(s, Image.NEAREST))
        return {"image": img, "mask": m}


def resolve_tokenizer(tokenizer=None, tokenizer_pretrained_path=None, model_max_length=None):
    """The reference's datasets build ``AutoTokenizer.from_pretrained(tokenizer_pretrained_path)`` (basedataset.py:52-60); offline that
    means a local CLIP tokenizer directory / ``merges.txt`` (or ``TVL_CLIP_BPE``).  A hub id such as ``CIDAS/clipseg-rd64`` cannot be
    fetched: it falls back to ``TVL_CLIP_BPE`` and fails loudly when that is unset too."""
    if tokenizer is not None:
        return tokenizer
    from .tokenizer import ClipBpeTokenizer

    path = tokenizer_pretrained_path if tokenizer_pretrained_path is not None and Path(str(tokenizer_pretrained_path)).exists() else None
    return ClipBpeTokenizer(path, **({"model_max_length": int(model_max_length)} if model_max_length else {}))


def load_tokenizer(pretrained_model_name_or_path=None, *args, **kwargs):
    """``transformers.AutoTokenizer.from_pretrained`` of the reference's collate_fn config, offline (see ``resolve_tokenizer``)."""
    return resolve_tokenizer(None, pretrained_model_name_or_path, kwargs.get("model_max_length"))


class _TransformMixin:
    """``transforms`` is either a host callable with the albumentations calling convention (equal-size uint8 outputs: the legacy
    ``ResizeTransform``) or a ``data.transforms.Compose`` -- then samples leave the dataset decoded and untouched and the whole pipeline
    runs on the device per batch (``DeviceTransform``)."""

    def apply_transforms(self, image, mask):
        from .transforms import Compose

        if self.transforms is None or isinstance(self.transforms, Compose):
            return image, mask
        out = self.transforms(image=image, mask=mask)
        return out["image"], out["mask"]


class ImageTextMaskDataset(_TransformMixin, Dataset):
    def __init__(self, *, image_dir, mask_dir, task_path, prompt_index: int, tokenizer=None, transforms: Callable | None = None,
                 override_prompt: str | None = None, insert_stop_at_last: bool = False, collate_fn=None, tokenizer_pretrained_path=None,
                 model_max_length=None, return_tensors=None, **_ignored: Any) -> None:
        self.tasks = self.get_tasks(task_path)
        self.image_dir, self.mask_dir = Path(image_dir), Path(mask_dir)
        self.prompt_map_index = f"p{prompt_index}" if prompt_index >= 0 else "random"
        self.override_prompt, self.insert_stop_at_last = override_prompt, insert_stop_at_last
        self.tokenizer = resolve_tokenizer(tokenizer, tokenizer_pretrained_path, model_max_length)
        self.transforms, self.collate_fn = transforms, collate_fn

    @staticmethod
    def get_tasks(task_path) -> list[dict[str, Any]]:
        with open(task_path, encoding="utf-8") as fp:
            tasks = json.load(fp)
        if not isinstance(tasks, list):
            raise TypeError(f"Expected the task file to hold a list, got {type(tasks)}")
        return tasks

    def __len__(self) -> int:
        return len(self.tasks)

    def get_curr_prompt(self, task: Mapping[str, Any]) -> str:
        prompts = task["prompts"]
        if not isinstance(prompts, Mapping):
            raise TypeError(f"Expected `prompts` to be a `Mapping` but got: {type(prompts)} instead.")
        if self.override_prompt is not None:
            return self.override_prompt
        if self.prompt_map_index == "random":
            keys = sorted(prompts, key=lambda k: int(k[1:]))
            key = random.choice(keys[1:])   # any prompt except p0
        else:
            key = self.prompt_map_index
        prompt = prompts[key]
        return prompt if isinstance(prompt, str) else random.choice(list(prompt))

    def __getitem__(self, index: int) -> dict[str, Any]:
        task = self.tasks[index]
        image = load_image(self.image_dir / str(task["img_name"]), "RGB")
        mask_name = str(task["mask_name"])
        mask = load_image(self.mask_dir / mask_name, "L")
        mask_shape = np.array(mask.shape)   # the ORIGINAL shape: the predict tail resizes back to it
        image, mask = self.apply_transforms(image, mask)
        prompt = self.get_curr_prompt(task)
        if self.insert_stop_at_last and prompt[-1] != ".":
            prompt += "."
        text = self.tokenizer(prompt, truncation=True)
        return {"image": torch.from_numpy(np.array(image, copy=True)), "mask": torch.from_numpy(np.array(mask, copy=True)),
                "mask_shape": mask_shape, "mask_name": mask_name, "prompt": prompt,
                "input_ids": text["input_ids"], "attention_mask": text["attention_mask"]}


class ImageDirTextMaskDataset(_TransformMixin, Dataset):
    """One directory per class under ``mask_dir`` (reference ``image_dir_mask_text_dataset.py:17-116`` -- how the reference runs the
    "20-class Pascal-VOC" data: 20 binary problems, the class name is the prompt): every ``mask_dir/<class>/<name><mask_suffix>`` is a
    sample, its image is ``image_dir/<name><image_suffix>``, its prompt ``<class>`` (+ "." with ``insert_stop_at_last``)."""

    def __init__(self, *, image_dir, mask_dir, image_suffix: str, mask_suffix: str, insert_stop_at_last: bool = False, tokenizer=None,
                 transforms: Callable | None = None, collate_fn=None, tokenizer_pretrained_path=None, model_max_length=None,
                 return_tensors=None, **_ignored: Any) -> None:
        if image_suffix[0] != ".":
            raise ValueError(f"image_suffix must start with a period: {image_suffix=}")
        if mask_suffix[0] != ".":
            raise ValueError(f"mask_suffix must start with a period: {mask_suffix=}")
        self.image_dir, self.mask_dir = Path(image_dir), Path(mask_dir)
        self.image_suffix, self.mask_suffix, self.insert_stop_at_last = image_suffix, mask_suffix, insert_stop_at_last
        if not any(p.is_dir() for p in self.mask_dir.iterdir()):
            raise ValueError(f"No directories found in {self.mask_dir}")
        self.tasks = [{"class_name": p.parent.name, "mask_name": p.name} for p in sorted(self.mask_dir.glob(f"*/*{mask_suffix}"))]
        self.tokenizer = resolve_tokenizer(tokenizer, tokenizer_pretrained_path, model_max_length)
        self.transforms, self.collate_fn = transforms, collate_fn

    def __len__(self) -> int:
        return len(self.tasks)

    def __getitem__(self, index: int) -> dict[str, Any]:
        task = self.tasks[index]
        cls = str(task["class_name"])
        prompt = f"{cls}." if self.insert_stop_at_last and cls[-1] != "." else cls
        name = Path(task["mask_name"])
        image = load_image(self.image_dir / name.with_suffix(self.image_suffix), "RGB")
        mask_name = Path(cls) / name
        mask = load_image(self.mask_dir / mask_name, "L")
        mask_shape = np.array(mask.shape)
        image, mask = self.apply_transforms(image, mask)
        text = self.tokenizer(prompt, truncation=True)
        return {"image": torch.from_numpy(np.array(image, copy=True)), "mask": torch.from_numpy(np.array(mask, copy=True)),
                "mask_shape": mask_shape, "mask_name": str(mask_name), "prompt": prompt,
                "input_ids": text["input_ids"], "attention_mask": text["attention_mask"]}


class DeviceBatchPrep:
    """Collated uint8 batch -> what ``ImageTextMaskModule`` consumes: ``image`` [B,3,H,W] fp32 normalised with the experiment's
    mean / std (``configs/experiment/coop/clipseg.yaml:65-66``), ``mask`` [B,1,H,W] fp32 = grey / 255 (targets are ``mask.long()``
    downstream, as in the reference), token tensors on the device.  One H2D copy of the uint8 bytes (4x fewer than fp32), then two
    HIP kernels."""

    def __init__(self, mean: Sequence[float], std: Sequence[float], device="cuda") -> None:
        self.mean, self.std, self.device = tuple(mean), tuple(std), torch.device(device)

    def _h2d(self, t: torch.Tensor) -> torch.Tensor:
        # pageable -> device copies are synchronous (the host waits for the stream to drain: the previous train step); pinned ones are not
        if self.device.type == "cuda" and not t.is_cuda and not t.is_pinned():
            t = t.pin_memory()
        return t.to(self.device, non_blocking=True)

    def __call__(self, batch: Mapping[str, Any]) -> dict[str, Any]:
        out = dict(batch)
        img = self._h2d(batch["image"])
        msk = self._h2d(batch["mask"])
        out["image"] = hip.normalize_u8(img.contiguous(), self.mean, self.std)
        out["mask"] = hip.mask_u8(msk.contiguous())
        for k in ("input_ids", "attention_mask"):
            out[k] = self._h2d(batch[k])
        return out
