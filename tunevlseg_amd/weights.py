"""Deterministic random weights for the CLIPSeg backbone, under HF key names.

Pretrained ``CIDAS/clipseg-rd64`` weights are not reachable offline
(SURVEY.md §8c), so tests, goldens and the benchmark all use seeded random
weights. The key names are exactly those of HF ``CLIPSegForImageSegmentation``
(what the reference loads in ``hf_clipseg_wrapper.py:38-58``), so the same dict
loads into the HF model (golden generation), into the CPU oracle and into the
HIP-backed backbone, and a real HF checkpoint loads by name.

The draw order is the order of :func:`clipseg_param_specs`; it is part of the
golden-fixture contract -- do not reorder.
"""
from __future__ import annotations

import math
from typing import Iterator

import torch

from .config import CLIPSegConfig


def _layer_specs(prefix: str, hidden: int, inter: int, n_layers_for_scale: int) -> Iterator[tuple[str, tuple[int, ...], str, float]]:
    attn_std = hidden**-0.5
    in_std = hidden**-0.5 * (2 * n_layers_for_scale) ** -0.5 * 2.0
    for p in ("k_proj", "v_proj", "q_proj"):
        yield f"{prefix}.self_attn.{p}.weight", (hidden, hidden), "normal", attn_std
        yield f"{prefix}.self_attn.{p}.bias", (hidden,), "normal", 0.02
    yield f"{prefix}.self_attn.out_proj.weight", (hidden, hidden), "normal", in_std
    yield f"{prefix}.self_attn.out_proj.bias", (hidden,), "normal", 0.02
    yield f"{prefix}.layer_norm1.weight", (hidden,), "ln_w", 0.05
    yield f"{prefix}.layer_norm1.bias", (hidden,), "normal", 0.02
    yield f"{prefix}.mlp.fc1.weight", (inter, hidden), "normal", (2 * hidden) ** -0.5 * 1.5
    yield f"{prefix}.mlp.fc1.bias", (inter,), "normal", 0.02
    yield f"{prefix}.mlp.fc2.weight", (hidden, inter), "normal", in_std * (hidden / inter) ** 0.5
    yield f"{prefix}.mlp.fc2.bias", (hidden,), "normal", 0.02
    yield f"{prefix}.layer_norm2.weight", (hidden,), "ln_w", 0.05
    yield f"{prefix}.layer_norm2.bias", (hidden,), "normal", 0.02


def clipseg_param_specs(cfg: CLIPSegConfig) -> Iterator[tuple[str, tuple[int, ...], str, float]]:
    """Yield ``(hf_key, shape, kind, scale)`` for every backbone tensor."""
    t, v = cfg.text_config, cfg.vision_config
    yield "clip.logit_scale", (), "const", 2.6592
    # text tower
    yield "clip.text_model.embeddings.token_embedding.weight", (t.vocab_size, t.hidden_size), "normal", 0.02 * 5
    yield "clip.text_model.embeddings.position_embedding.weight", (t.max_position_embeddings, t.hidden_size), "normal", 0.01 * 5
    for i in range(t.num_hidden_layers):
        yield from _layer_specs(f"clip.text_model.encoder.layers.{i}", t.hidden_size, t.intermediate_size, t.num_hidden_layers)
    yield "clip.text_model.final_layer_norm.weight", (t.hidden_size,), "ln_w", 0.05
    yield "clip.text_model.final_layer_norm.bias", (t.hidden_size,), "normal", 0.02
    # vision tower
    n_pos = (v.image_size // v.patch_size) ** 2 + 1
    yield "clip.vision_model.embeddings.class_embedding", (v.hidden_size,), "normal", v.hidden_size**-0.5 * 5
    yield "clip.vision_model.embeddings.patch_embedding.weight", (v.hidden_size, v.num_channels, v.patch_size, v.patch_size), "normal", 0.02
    yield "clip.vision_model.embeddings.position_embedding.weight", (n_pos, v.hidden_size), "normal", 0.1
    yield "clip.vision_model.pre_layrnorm.weight", (v.hidden_size,), "ln_w", 0.05
    yield "clip.vision_model.pre_layrnorm.bias", (v.hidden_size,), "normal", 0.02
    for i in range(v.num_hidden_layers):
        yield from _layer_specs(f"clip.vision_model.encoder.layers.{i}", v.hidden_size, v.intermediate_size, v.num_hidden_layers)
    yield "clip.vision_model.post_layernorm.weight", (v.hidden_size,), "ln_w", 0.05
    yield "clip.vision_model.post_layernorm.bias", (v.hidden_size,), "normal", 0.02
    yield "clip.visual_projection.weight", (cfg.projection_dim, v.hidden_size), "normal", v.hidden_size**-0.5
    yield "clip.text_projection.weight", (cfg.projection_dim, t.hidden_size), "normal", t.hidden_size**-0.5
    # decoder
    r = cfg.reduce_dim
    yield "decoder.film_mul.weight", (r, cfg.projection_dim), "normal", cfg.projection_dim**-0.5
    yield "decoder.film_mul.bias", (r,), "film_b", 0.1
    yield "decoder.film_add.weight", (r, cfg.projection_dim), "normal", cfg.projection_dim**-0.5
    yield "decoder.film_add.bias", (r,), "normal", 0.05
    yield "decoder.transposed_convolution.weight", (r, 1, v.patch_size, v.patch_size), "normal", 0.35 * r**-0.5
    yield "decoder.transposed_convolution.bias", (1,), "normal", 0.1
    for i in range(len(cfg.extract_layers)):
        yield f"decoder.reduces.{i}.weight", (r, v.hidden_size), "normal", v.hidden_size**-0.5
        yield f"decoder.reduces.{i}.bias", (r,), "normal", 0.02
    for i in range(len(cfg.extract_layers)):
        yield from _layer_specs(f"decoder.layers.{i}", r, cfg.decoder_intermediate_size, len(cfg.extract_layers))


def init_clipseg_state_dict(cfg: CLIPSegConfig, seed: int = 0, dtype: torch.dtype = torch.float32) -> dict[str, torch.Tensor]:
    """Seeded CPU draw of every backbone tensor (fp32), HF key names.

    Biases and LayerNorm affine terms are non-trivial on purpose so that every
    bias/affine code path of the kernels is exercised by parity tests.
    """
    g = torch.Generator(device="cpu")
    g.manual_seed(int(seed))
    out: dict[str, torch.Tensor] = {}
    for name, shape, kind, scale in clipseg_param_specs(cfg):
        if kind == "const":
            x = torch.tensor(scale, dtype=torch.float32)
        elif kind == "normal":
            x = torch.randn(shape, generator=g, dtype=torch.float32) * scale
        elif kind == "ln_w":
            x = 1.0 + torch.randn(shape, generator=g, dtype=torch.float32) * scale
        elif kind == "film_b":
            x = 1.0 + torch.randn(shape, generator=g, dtype=torch.float32) * scale
        else:  # pragma: no cover
            raise ValueError(kind)
        out[name] = x.to(dtype)
    return out


def count_params(cfg: CLIPSegConfig) -> int:
    return sum(math.prod(s) for _, s, _, _ in clipseg_param_specs(cfg))
