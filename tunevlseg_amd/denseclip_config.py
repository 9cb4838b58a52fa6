"""Geometry of the DenseCLIP path (BASELINE configs[4]): CLIP ViT-B/16 backbone with four FPN taps + CLIP text encoder
over learnable contexts + context decoder + pixel-text score map.

The reference spells it as an mmseg config (``denseclip_configs/denseclip_fpn_vit-b_640x640_80k.py:8-61``): ``backbone``
(``CLIPVisionTransformer``, ``models.py:530-714``), ``text_encoder`` (``CLIPTextContextEncoder``, ``models.py:805-903``),
``context_decoder`` (``ContextDecoder``, ``models.py:907-960``) and the segmentor's own ``context_length`` / ``tau`` /
``score_concat_index`` / ``text_dim`` (``denseclip.py:31-53``).  The mmseg FPN neck and FPNHead behind the score map are
outside the hot path (mmseg / mmengine are absent from the image; no runnable counterpart).
"""
from __future__ import annotations

from dataclasses import asdict, dataclass
from typing import Any


@dataclass
class DenseCLIPConfig:
    # backbone: CLIPVisionTransformer (models.py:531-545)
    input_resolution: int = 640
    patch_size: int = 16
    width: int = 768
    layers: int = 12
    heads: int = 12
    output_dim: int = 512
    out_indices: tuple[int, ...] = (3, 5, 7, 11)
    # text encoder: CLIPTextContextEncoder (models.py:806-817)
    text_context_length: int = 13       # text_encoder.context_length: prompt tokens + learnable contexts
    vocab_size: int = 49408
    transformer_width: int = 512
    transformer_heads: int = 8
    transformer_layers: int = 12
    embed_dim: int = 512
    # context decoder: ContextDecoder (models.py:908-916)
    decoder_width: int = 256
    decoder_heads: int = 4
    decoder_layers: int = 3
    visual_dim: int = 512
    # segmentor (denseclip.py:31-53)
    context_length: int = 5             # tokens kept of every class name (tokenize(c, context_length))
    num_classes: int = 20               # BASELINE configs[4]: 20-class Pascal-VOC
    score_concat_index: int = 2
    tau: float = 0.07
    token_embed_dim: int = 512
    text_dim: int = 512

    @property
    def num_contexts(self) -> int:      # denseclip.py:104-105
        return self.text_context_length - self.context_length

    @property
    def grid(self) -> int:              # models.py:564 spatial_size
        return self.input_resolution // self.patch_size

    @classmethod
    def from_dict(cls, d: dict[str, Any]) -> "DenseCLIPConfig":
        d = {k: v for k, v in dict(d).items() if k in cls.__dataclass_fields__}
        if "out_indices" in d:
            d["out_indices"] = tuple(d["out_indices"])
        return cls(**d)

    def to_dict(self) -> dict[str, Any]:
        out = asdict(self)
        out["out_indices"] = list(out["out_indices"])
        return out

    @classmethod
    def vitb16_640(cls, num_classes: int = 20) -> "DenseCLIPConfig":
        return cls(num_classes=num_classes)

    @classmethod
    def tiny(cls, num_classes: int = 5) -> "DenseCLIPConfig":
        """Reduced geometry for the golden fixtures: every code path of the full model -- position table resized from a 4 x 4 grid
        to the image's own, four taps, both transposed convs + folded BatchNorm + GELU, max-pool, context splice, causal text tower,
        cross-attention over 1 + H*W memory rows, score map concatenated into the third map."""
        return cls(input_resolution=64, patch_size=16, width=64, layers=4, heads=4, output_dim=32, out_indices=(0, 1, 2, 3),
                   text_context_length=9, vocab_size=64, transformer_width=32, transformer_heads=2, transformer_layers=3, embed_dim=32,
                   decoder_width=32, decoder_heads=2, decoder_layers=2, visual_dim=32, context_length=4, num_classes=num_classes,
                   token_embed_dim=32, text_dim=32)
