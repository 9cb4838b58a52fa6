"""Backbone configuration for the CLIPSeg hot path.

Field names and defaults follow the un-vendored HuggingFace ``CLIPSegConfig``
that the reference loads through ``HFCLIPSegWrapper``
(reference ``src/models/components/hf_clipseg_wrapper.py:15-35``); the nets read
``config.reduce_dim``, ``config.projection_dim``, ``config.text_config.*`` and
``config.vision_config.*`` (reference ``coop_clipseg.py:31-37``,
``base_clipseg.py:61,65,97``), so those attribute paths are kept.
"""
from __future__ import annotations

from dataclasses import asdict, dataclass, field
from typing import Any


@dataclass
class TextConfig:
    vocab_size: int = 49408
    hidden_size: int = 512
    intermediate_size: int = 2048
    num_hidden_layers: int = 12
    num_attention_heads: int = 8
    max_position_embeddings: int = 77
    hidden_act: str = "quick_gelu"
    layer_norm_eps: float = 1e-5
    pad_token_id: int = 1
    bos_token_id: int = 49406
    eos_token_id: int = 49407
    # never consulted by the nets; kept so HF config dicts round-trip
    output_attentions: bool = False
    output_hidden_states: bool = False
    use_return_dict: bool = True


@dataclass
class VisionConfig:
    hidden_size: int = 768
    intermediate_size: int = 3072
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    num_channels: int = 3
    image_size: int = 224
    patch_size: int = 16
    hidden_act: str = "quick_gelu"
    layer_norm_eps: float = 1e-5
    output_attentions: bool = False
    output_hidden_states: bool = False
    use_return_dict: bool = True


@dataclass
class CLIPSegConfig:
    text_config: TextConfig = field(default_factory=TextConfig)
    vision_config: VisionConfig = field(default_factory=VisionConfig)
    projection_dim: int = 512
    extract_layers: tuple[int, ...] = (3, 6, 9)
    reduce_dim: int = 64
    decoder_num_attention_heads: int = 4
    decoder_intermediate_size: int = 2048
    conditional_layer: int = 0
    use_complex_transposed_convolution: bool = False
    use_return_dict: bool = True

    @classmethod
    def from_dict(cls, d: dict[str, Any]) -> "CLIPSegConfig":
        d = dict(d)
        known_t = TextConfig.__dataclass_fields__
        known_v = VisionConfig.__dataclass_fields__
        t = {k: v for k, v in (d.pop("text_config", None) or {}).items() if k in known_t}
        v = {k: v for k, v in (d.pop("vision_config", None) or {}).items() if k in known_v}
        top = {k: val for k, val in d.items() if k in cls.__dataclass_fields__}
        if "extract_layers" in top:
            top["extract_layers"] = tuple(top["extract_layers"])
        return cls(text_config=TextConfig(**t), vision_config=VisionConfig(**v), **top)

    def to_dict(self) -> dict[str, Any]:
        out = asdict(self)
        out["extract_layers"] = list(self.extract_layers)
        return out

    # ---- named presets -------------------------------------------------
    @classmethod
    def rd64(cls, eos_token_id: int = 2, image_size: int = 224) -> "CLIPSegConfig":
        """CIDAS/clipseg-rd64 geometry (ViT-B/16, reduce_dim 64)."""
        return cls(
            text_config=TextConfig(eos_token_id=eos_token_id),
            vision_config=VisionConfig(image_size=image_size, patch_size=16),
        )

    @classmethod
    def tiny(cls, eos_token_id: int = 2) -> "CLIPSegConfig":
        """Reduced-width geometry used by the golden fixtures.

        Keeps every code path of the full model: interpolated position
        embeddings (image_size 32 run at 64x64), early break after the last
        extract layer, FiLM at decoder layer 0, strip of CLS/prompt tokens.
        """
        return cls(
            text_config=TextConfig(
                vocab_size=64,
                hidden_size=32,
                intermediate_size=64,
                num_hidden_layers=3,
                num_attention_heads=2,
                max_position_embeddings=16,
                pad_token_id=1,
                bos_token_id=62,
                eos_token_id=eos_token_id,
            ),
            vision_config=VisionConfig(
                hidden_size=32,
                intermediate_size=64,
                num_hidden_layers=4,
                num_attention_heads=2,
                image_size=32,
                patch_size=16,
            ),
            projection_dim=32,
            extract_layers=(0, 1, 2),
            reduce_dim=16,
            decoder_num_attention_heads=2,
            decoder_intermediate_size=32,
        )
