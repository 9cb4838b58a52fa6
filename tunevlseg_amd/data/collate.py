"""Batch collation of the reference's datamodule (``src/data/components/data_collator.py:17-34``): the tokenizer outputs are
padded to the longest row of the batch (HF ``DataCollatorWithPadding(padding=True)``), everything else goes through the default
stacking collate.  No HF tokenizer object is needed: only its ``pad_token_id`` (attention mask pads are 0)."""
from __future__ import annotations

from collections.abc import Iterable
from typing import Any

import torch
from torch.utils.data import default_collate


class PadToLongestCollator:
    def __init__(self, padding_keys: Iterable[str] = ("input_ids", "attention_mask"), pad_token_id: int = 49407,
                 pad_to_multiple_of: int | None = None, tokenizer=None, **_hf_kwargs: Any) -> None:
        self.padding_keys = set(padding_keys)
        if not self.padding_keys:
            raise ValueError("`padding_keys` should not be empty.")
        self.pad_token_id = int(getattr(tokenizer, "pad_token_id", pad_token_id) if tokenizer is not None else pad_token_id)
        self.pad_to_multiple_of = pad_to_multiple_of

    def __call__(self, features: list[dict[str, Any]]) -> dict[str, Any]:
        width = max(len(f["input_ids"]) for f in features)
        if self.pad_to_multiple_of:
            width = -(-width // self.pad_to_multiple_of) * self.pad_to_multiple_of
        padded = {}
        for key in self.padding_keys:
            fill = self.pad_token_id if key == "input_ids" else 0
            rows = [list(f[key].tolist() if isinstance(f[key], torch.Tensor) else f[key]) for f in features]
            padded[key] = torch.tensor([r + [fill] * (width - len(r)) for r in rows], dtype=torch.long)
        rest = default_collate([{k: v for k, v in f.items() if k not in self.padding_keys} for f in features])
        return {**rest, **padded}
