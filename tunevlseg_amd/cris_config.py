"""Geometry of the CRIS path (BASELINE configs[2]): CLIP-RN50 towers + FPN neck + cross-attention decoder + projector.

The reference derives the CLIP geometry from the RN50 checkpoint (``build_model``, reference
``src/models/components/cris_model/clip.py:578-647``) and takes the rest from ``model_cfg``
(``configs/model/coop/cris.yaml:5-20``, ``cris_model/__init__.py:23-61``); both halves are explicit here because the
checkpoint is not reachable offline.
"""
from __future__ import annotations

from dataclasses import asdict, dataclass
from typing import Any


@dataclass
class CRISConfig:
    # CLIP (clip.py:406-458)
    embed_dim: int = 1024
    image_resolution: int = 224
    vision_layers: tuple[int, int, int, int] = (3, 4, 6, 3)
    vision_width: int = 64
    context_length: int = 77
    vocab_size: int = 49408
    transformer_width: int = 512
    transformer_heads: int = 8
    transformer_layers: int = 12
    # CRIS head (configs/model/coop/cris.yaml:7-17)
    fpn_in: tuple[int, int, int] = (512, 1024, 1024)
    fpn_out: tuple[int, int, int] = (256, 512, 1024)
    vis_dim: int = 512
    word_dim: int = 1024
    num_layers: int = 3
    num_head: int = 8
    dim_ffn: int = 2048
    dropout: float = 0.2
    img_size: int = 416
    max_length: int = 77  # CRIS.max_length, cris_model/__init__.py:21

    @property
    def vision_heads(self) -> int:  # clip.py:441
        return self.vision_width * 32 // 64

    @property
    def vision_embed(self) -> int:  # clip.py:238
        return self.vision_width * 32

    @classmethod
    def from_dict(cls, d: dict[str, Any]) -> "CRISConfig":
        d = {k: v for k, v in dict(d).items() if k in cls.__dataclass_fields__}
        for k in ("vision_layers", "fpn_in", "fpn_out"):
            if k in d:
                d[k] = tuple(d[k])
        return cls(**d)

    def to_dict(self) -> dict[str, Any]:
        out = asdict(self)
        for k in ("vision_layers", "fpn_in", "fpn_out"):
            out[k] = list(out[k])
        return out

    @classmethod
    def rn50(cls, img_size: int = 416) -> "CRISConfig":
        return cls(img_size=img_size)

    @classmethod
    def tiny(cls, img_size: int = 96) -> "CRISConfig":
        """Reduced geometry for the golden fixtures: every code path of the full model (attention-pool position resize,
        strided bottlenecks with the avg-pool shortcut, three-scale neck, key-padding mask, dynamic conv, bicubic head)."""
        return cls(
            embed_dim=32, image_resolution=64, vision_layers=(1, 2, 1, 1), vision_width=8,
            context_length=77, vocab_size=64, transformer_width=32, transformer_heads=2, transformer_layers=3,
            fpn_in=(64, 128, 32), fpn_out=(16, 32, 32), vis_dim=32, word_dim=32, num_layers=2, num_head=2, dim_ffn=64,
            dropout=0.2, img_size=img_size,
        )
