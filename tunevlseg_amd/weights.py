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
from .cris_config import CRISConfig


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


def _draw(specs, seed: int, dtype: torch.dtype) -> dict[str, torch.Tensor]:
    g = torch.Generator(device="cpu")
    g.manual_seed(int(seed))
    out: dict[str, torch.Tensor] = {}
    for name, shape, kind, scale in specs:
        if kind == "const":
            x = torch.tensor(scale, dtype=torch.float32)
        elif kind == "normal":
            x = torch.randn(shape, generator=g, dtype=torch.float32) * scale
        elif kind == "ln_w":
            x = 1.0 + torch.randn(shape, generator=g, dtype=torch.float32) * scale
        elif kind == "film_b":
            x = 1.0 + torch.randn(shape, generator=g, dtype=torch.float32) * scale
        elif kind == "bn_w":  # BatchNorm scale around `scale`
            x = scale * (1.0 + 0.1 * torch.randn(shape, generator=g, dtype=torch.float32))
        elif kind == "bn_var":  # running variance in [0.5, 1.5] * scale
            x = scale * (0.5 + torch.rand(shape, generator=g, dtype=torch.float32))
        else:  # pragma: no cover
            raise ValueError(kind)
        out[name] = x.to(dtype)
    return out


TAIL_LEVELS = {1: (4.0, 8.0, 8.0, 4.0), 2: (30.0, 100.0, 20.0, 8.0)}   # (gain lo, gain hi, row factor, |bias outlier|)


def heavy_tails(sd: dict[str, torch.Tensor], cfg: CLIPSegConfig, seed: int, level: int = 1) -> dict[str, torch.Tensor]:
    """Outlier structure of trained CLIP checkpoints laid over a seeded draw (in place): a handful of residual-stream channels
    that are "massive" in every layer.  Per tower, ``n_out`` fixed channels get large LayerNorm gains (both norms of every layer,
    random sign), the ``out_proj`` / ``fc2`` rows that write the first half of those channels are scaled up, and one ``q`` and one
    ``k`` bias entry per layer becomes an outlier.  Rows of the activations then span several decades and single columns dominate
    the row norms -- the regime that decides whether the scale bounds of the two-piece fp16 operand format hold.

    ``level`` 1: gains 4-8, rows x8, bias +-4 -- the strongest setting under which the REFERENCE's own fp32 run still agrees with
    its float64 run to 4e-5 in the logits on the full-size net, so the usual parity gates stay meaningful.  ``level`` 2: gains
    30-100, rows x20, bias +-8 -- there the reference's fp32 logits are 0.14 away from its float64 logits (attention saturates and
    the net is chaotic in fp32); fixtures drawn with it are gated against the float64 run, relative to the reference's own fp32
    deviation.  Pretrained weights are unreachable offline; the ``*_tails`` / ``*_tails2`` fixtures are drawn with these."""
    glo, ghi, rowf, biasv = TAIL_LEVELS[int(level)]
    g = torch.Generator(device="cpu")
    g.manual_seed(7919 * int(seed) + 17)
    for tower, tc in (("clip.vision_model", cfg.vision_config), ("clip.text_model", cfg.text_config)):
        D = tc.hidden_size
        n_out = 6 if D >= 256 else 2
        chans = torch.randperm(D, generator=g)[:n_out]
        rows = chans[: max(1, n_out // 2)]
        for i in range(tc.num_hidden_layers):
            p = f"{tower}.encoder.layers.{i}"
            for ln in ("layer_norm1", "layer_norm2"):
                gain = glo + (ghi - glo) * torch.rand(n_out, generator=g)
                sign = torch.where(torch.rand(n_out, generator=g) < 0.5, -1.0, 1.0)
                sd[f"{p}.{ln}.weight"][chans] = (gain * sign).to(sd[f"{p}.{ln}.weight"].dtype)
            for lin in ("self_attn.out_proj", "mlp.fc2"):
                sd[f"{p}.{lin}.weight"][rows] *= rowf
            for qk in ("q_proj", "k_proj"):
                j = int(torch.randint(0, D, (1,), generator=g))
                sd[f"{p}.self_attn.{qk}.bias"][j] = biasv if torch.rand(1, generator=g).item() < 0.5 else -biasv
    return sd


def init_clipseg_state_dict(cfg: CLIPSegConfig, seed: int = 0, dtype: torch.dtype = torch.float32, tails: int = 0) -> dict[str, torch.Tensor]:
    """Seeded CPU draw of every backbone tensor (fp32), HF key names.

    Biases and LayerNorm affine terms are non-trivial on purpose so that every
    bias/affine code path of the kernels is exercised by parity tests.
    ``tails`` = 1 / 2 lays the outlier channels of :func:`heavy_tails` (that level) over the draw.
    """
    sd = _draw(clipseg_param_specs(cfg), seed, dtype)
    return heavy_tails(sd, cfg, seed, tails) if tails else sd


def count_params(cfg: CLIPSegConfig) -> int:
    return sum(math.prod(s) for _, s, _, _ in clipseg_param_specs(cfg))


# ---------------------------------------------------------------------------------------------------------------------
# CRIS (reference src/models/components/cris_model/{__init__,clip,layers}.py); key names = the reference module tree
# ---------------------------------------------------------------------------------------------------------------------
def _bn_specs(prefix: str, c: int, w_scale: float = 1.0):
    yield f"{prefix}.weight", (c,), "bn_w", w_scale
    yield f"{prefix}.bias", (c,), "normal", 0.05
    yield f"{prefix}.running_mean", (c,), "normal", 0.1
    yield f"{prefix}.running_var", (c,), "bn_var", 1.0


def _conv_specs(name: str, cout: int, cin: int, k: int, gain: float = 1.0):
    yield name, (cout, cin, k, k), "normal", gain * (2.0 / (cin * k * k)) ** 0.5


def _conv_layer_specs(prefix: str, cin: int, cout: int, k: int):
    """``conv_layer`` = Conv(no bias) + BN + ReLU (layers.py:15-26)."""
    yield from _conv_specs(f"{prefix}.0.weight", cout, cin, k)
    yield from _bn_specs(f"{prefix}.1", cout)


def _mha_specs(prefix: str, d: int, out_gain: float = 1.0):
    yield f"{prefix}.in_proj_weight", (3 * d, d), "normal", d**-0.5
    yield f"{prefix}.in_proj_bias", (3 * d,), "normal", 0.02
    yield f"{prefix}.out_proj.weight", (d, d), "normal", out_gain * d**-0.5
    yield f"{prefix}.out_proj.bias", (d,), "normal", 0.02


def _ln_specs(prefix: str, d: int):
    yield f"{prefix}.weight", (d,), "ln_w", 0.05
    yield f"{prefix}.bias", (d,), "normal", 0.02


def cris_param_specs(cfg: CRISConfig) -> Iterator[tuple[str, tuple[int, ...], str, float]]:
    w = cfg.vision_width
    v = "backbone.visual"
    # ModifiedResNet stem (clip.py:205-229)
    yield from _conv_specs(f"{v}.conv1.weight", w // 2, 3, 3)
    yield from _bn_specs(f"{v}.bn1", w // 2)
    yield from _conv_specs(f"{v}.conv2.weight", w // 2, w // 2, 3)
    yield from _bn_specs(f"{v}.bn2", w // 2)
    yield from _conv_specs(f"{v}.conv3.weight", w, w // 2, 3)
    yield from _bn_specs(f"{v}.bn3", w)
    inplanes = w
    for li, (planes, blocks) in enumerate(zip((w, 2 * w, 4 * w, 8 * w), cfg.vision_layers), start=1):
        for bi in range(blocks):
            stride = 2 if (li > 1 and bi == 0) else 1
            p = f"{v}.layer{li}.{bi}"
            yield from _conv_specs(f"{p}.conv1.weight", planes, inplanes, 1)
            yield from _bn_specs(f"{p}.bn1", planes)
            yield from _conv_specs(f"{p}.conv2.weight", planes, planes, 3)
            yield from _bn_specs(f"{p}.bn2", planes)
            yield from _conv_specs(f"{p}.conv3.weight", 4 * planes, planes, 1)
            yield from _bn_specs(f"{p}.bn3", 4 * planes, 0.5)
            if stride > 1 or inplanes != 4 * planes:
                yield from _conv_specs(f"{p}.downsample.0.weight", 4 * planes, inplanes, 1, 0.7)
                yield from _bn_specs(f"{p}.downsample.1", 4 * planes)
            inplanes = 4 * planes
    e = cfg.vision_embed
    sp = cfg.image_resolution // 32
    yield f"{v}.attnpool.positional_embedding", (sp * sp + 1, e), "normal", e**-0.5 * 4
    for n in ("k_proj", "q_proj", "v_proj"):
        yield f"{v}.attnpool.{n}.weight", (e, e), "normal", e**-0.5
        yield f"{v}.attnpool.{n}.bias", (e,), "normal", 0.02
    yield f"{v}.attnpool.c_proj.weight", (cfg.embed_dim, e), "normal", e**-0.5
    yield f"{v}.attnpool.c_proj.bias", (cfg.embed_dim,), "normal", 0.02
    yield from _conv_specs(f"{v}.attnpool.connect.0.weight", cfg.embed_dim, e, 1, 0.7)
    yield from _bn_specs(f"{v}.attnpool.connect.1", cfg.embed_dim)
    # CLIP text tower (clip.py:296-325,460-470)
    d = cfg.transformer_width
    for i in range(cfg.transformer_layers):
        p = f"backbone.transformer.resblocks.{i}"
        yield from _mha_specs(f"{p}.attn", d, (2 * cfg.transformer_layers) ** -0.5 * 2.0)
        yield from _ln_specs(f"{p}.ln_1", d)
        yield f"{p}.mlp.c_fc.weight", (4 * d, d), "normal", (2 * d) ** -0.5 * 1.5
        yield f"{p}.mlp.c_fc.bias", (4 * d,), "normal", 0.02
        yield f"{p}.mlp.c_proj.weight", (d, 4 * d), "normal", (4 * d) ** -0.5 * (2 * cfg.transformer_layers) ** -0.5 * 2.0
        yield f"{p}.mlp.c_proj.bias", (d,), "normal", 0.02
        yield from _ln_specs(f"{p}.ln_2", d)
    yield "backbone.token_embedding.weight", (cfg.vocab_size, d), "normal", 0.1
    yield "backbone.positional_embedding", (cfg.context_length, d), "normal", 0.05
    yield from _ln_specs("backbone.ln_final", d)
    yield "backbone.text_projection", (d, cfg.embed_dim), "normal", d**-0.5
    yield "backbone.logit_scale", (), "const", 2.6592
    # FPN neck (layers.py:359-410)
    fi, fo = cfg.fpn_in, cfg.fpn_out
    yield "neck.txt_proj.0.weight", (fo[2], fi[2]), "normal", fi[2] ** -0.5 * 1.5
    yield from _bn_specs("neck.txt_proj.1", fo[2])
    yield from _conv_layer_specs("neck.f1_v_proj", fi[2], fo[2], 1)
    yield from _bn_specs("neck.norm_layer.0", fo[2])
    yield from _conv_layer_specs("neck.f2_v_proj", fi[1], fo[1], 3)
    yield from _conv_layer_specs("neck.f2_cat", fo[2] + fo[1], fo[1], 1)
    yield from _conv_layer_specs("neck.f3_v_proj", fi[0], fo[0], 3)
    yield from _conv_layer_specs("neck.f3_cat", fo[0] + fo[1], fo[1], 1)
    yield from _conv_layer_specs("neck.f4_proj5", fo[2], fo[1], 3)
    yield from _conv_layer_specs("neck.f4_proj4", fo[1], fo[1], 3)
    yield from _conv_layer_specs("neck.f4_proj3", fo[1], fo[1], 3)
    yield from _conv_layer_specs("neck.aggr", 3 * fo[1], fo[1], 1)
    yield from _conv_layer_specs("neck.coordconv.0.conv1", fo[1] + 2, fo[1], 3)
    yield from _conv_layer_specs("neck.coordconv.1", fo[1], fo[1], 3)
    # cross-attention decoder (layers.py:124-356)
    dm, ff = cfg.vis_dim, cfg.dim_ffn
    for i in range(cfg.num_layers):
        p = f"decoder.layers.{i}"
        yield from _ln_specs(f"{p}.self_attn_norm", dm)
        yield from _ln_specs(f"{p}.cross_attn_norm", dm)
        yield from _mha_specs(f"{p}.self_attn", dm)
        yield from _mha_specs(f"{p}.multihead_attn", dm)
        yield f"{p}.ffn.0.weight", (ff, dm), "normal", (2.0 / dm) ** 0.5
        yield f"{p}.ffn.0.bias", (ff,), "normal", 0.02
        yield from _ln_specs(f"{p}.ffn.3", ff)
        yield f"{p}.ffn.4.weight", (dm, ff), "normal", ff**-0.5
        yield f"{p}.ffn.4.bias", (dm,), "normal", 0.02
        yield from _ln_specs(f"{p}.norm1", dm)
        yield from _ln_specs(f"{p}.norm2", dm)
        yield from _ln_specs(f"{p}.norm3", dm)
    yield from _ln_specs("decoder.norm", dm)
    # projector (layers.py:71-94)
    c = cfg.vis_dim // 2
    yield from _conv_layer_specs("proj.vis.1", 2 * c, 2 * c, 3)
    yield from _conv_layer_specs("proj.vis.3", 2 * c, c, 3)
    yield "proj.vis.4.weight", (c, c, 1, 1), "normal", c**-0.5
    yield "proj.vis.4.bias", (c,), "normal", 0.02
    yield "proj.txt.weight", (c * 9 + 1, cfg.word_dim), "normal", cfg.word_dim**-0.5 * (c * 9) ** -0.5 * 3
    yield "proj.txt.bias", (c * 9 + 1,), "normal", 0.02


def init_cris_state_dict(cfg: CRISConfig, seed: int = 0, dtype: torch.dtype = torch.float32) -> dict[str, torch.Tensor]:
    """Seeded CPU draw of every CRIS tensor under the reference's module names (``CRIS.state_dict()`` keys minus
    ``num_batches_tracked``).  BatchNorm running statistics are non-trivial: the towers run in eval mode
    (reference coop_cris.py:66-68)."""
    return _draw(cris_param_specs(cfg), seed, dtype)


# ---------------------------------------------------------------------------------------------------------------------
# DenseCLIP (reference src/models/components/denseclip/{models,denseclip}.py); key names = the reference module tree
# ---------------------------------------------------------------------------------------------------------------------
def _resblock_specs(prefix: str, d: int, n_layers: int):
    """OpenAI-CLIP ``ResidualAttentionBlock`` (models.py:391-431): packed in-projection, QuickGELU MLP."""
    yield from _mha_specs(f"{prefix}.attn", d, (2 * n_layers) ** -0.5 * 2.0)
    yield from _ln_specs(f"{prefix}.ln_1", d)
    yield f"{prefix}.mlp.c_fc.weight", (4 * d, d), "normal", (2 * d) ** -0.5 * 1.5
    yield f"{prefix}.mlp.c_fc.bias", (4 * d,), "normal", 0.02
    yield f"{prefix}.mlp.c_proj.weight", (d, 4 * d), "normal", (4 * d) ** -0.5 * (2 * n_layers) ** -0.5 * 2.0
    yield f"{prefix}.mlp.c_proj.bias", (d,), "normal", 0.02
    yield from _ln_specs(f"{prefix}.ln_2", d)


def denseclip_param_specs(cfg) -> Iterator[tuple[str, tuple[int, ...], str, float]]:
    """``DenseCLIP.state_dict()`` keys of the parts on the hot path (``backbone``, ``text_encoder``, ``context_decoder``,
    ``contexts``, ``gamma``; denseclip.py:75-108) minus ``num_batches_tracked``.  The mmseg neck / decode head are not held."""
    w, p = cfg.width, cfg.patch_size
    b = "backbone"
    yield f"{b}.conv1.weight", (w, 3, p, p), "normal", 0.02
    yield f"{b}.class_embedding", (w,), "normal", w**-0.5 * 5
    yield f"{b}.positional_embedding", (cfg.grid**2 + 1, w), "normal", 0.1
    yield from _ln_specs(f"{b}.ln_pre", w)
    for i in range(cfg.layers):
        yield from _resblock_specs(f"{b}.transformer.resblocks.{i}", w, cfg.layers)
    yield from _ln_specs(f"{b}.ln_post", w)
    yield f"{b}.proj", (w, cfg.output_dim), "normal", w**-0.5
    if p != 16:
        raise NotImplementedError("DenseCLIP: the patch-16 FPN (models.py:582-600) is the one on the path; patch 8 (models.py:602-619) is not built")
    # fpn1: GroupNorm(1) -> ConvTranspose2d(k2, s2) -> SyncBatchNorm -> GELU -> ConvTranspose2d(k2, s2)   (models.py:583-589)
    yield from _ln_specs(f"{b}.fpn1.0", w)
    yield f"{b}.fpn1.1.weight", (w, w, 2, 2), "normal", w**-0.5
    yield f"{b}.fpn1.1.bias", (w,), "normal", 0.02
    yield from _bn_specs(f"{b}.fpn1.2", w)
    yield f"{b}.fpn1.4.weight", (w, w, 2, 2), "normal", w**-0.5 * 1.5
    yield f"{b}.fpn1.4.bias", (w,), "normal", 0.02
    # fpn2: GroupNorm(1) -> ConvTranspose2d ; fpn3: GroupNorm(1) ; fpn4: GroupNorm(1) -> MaxPool2d(2)   (models.py:591-600)
    yield from _ln_specs(f"{b}.fpn2.0", w)
    yield f"{b}.fpn2.1.weight", (w, w, 2, 2), "normal", w**-0.5
    yield f"{b}.fpn2.1.bias", (w,), "normal", 0.02
    yield from _ln_specs(f"{b}.fpn3", w)
    yield from _ln_specs(f"{b}.fpn4.0", w)
    # CLIPTextContextEncoder (models.py:805-841)
    d, t = cfg.transformer_width, "text_encoder"
    for i in range(cfg.transformer_layers):
        yield from _resblock_specs(f"{t}.transformer.resblocks.{i}", d, cfg.transformer_layers)
    yield f"{t}.token_embedding.weight", (cfg.vocab_size, d), "normal", 0.1
    yield f"{t}.positional_embedding", (cfg.text_context_length, d), "normal", 0.05
    yield from _ln_specs(f"{t}.ln_final", d)
    yield f"{t}.text_projection", (d, cfg.embed_dim), "normal", d**-0.5
    # ContextDecoder (models.py:907-946): memory_proj = LN, Linear, LN; text_proj = LN, Linear; out_proj = LN, Linear
    c, dw, vd = "context_decoder", cfg.decoder_width, cfg.visual_dim
    yield from _ln_specs(f"{c}.memory_proj.0", vd)
    yield f"{c}.memory_proj.1.weight", (dw, vd), "normal", vd**-0.5
    yield f"{c}.memory_proj.1.bias", (dw,), "normal", 0.02
    yield from _ln_specs(f"{c}.memory_proj.2", dw)
    yield from _ln_specs(f"{c}.text_proj.0", vd)
    yield f"{c}.text_proj.1.weight", (dw, vd), "normal", vd**-0.5
    yield f"{c}.text_proj.1.bias", (dw,), "normal", 0.02
    for i in range(cfg.decoder_layers):
        q = f"{c}.decoder.{i}"
        for a in ("self_attn", "cross_attn"):   # models.py:448-486: q/k/v projections without bias, output projection with
            for n in ("q_proj", "k_proj", "v_proj"):
                yield f"{q}.{a}.{n}.weight", (dw, dw), "normal", dw**-0.5 * 1.5
            yield f"{q}.{a}.proj.weight", (dw, dw), "normal", dw**-0.5
            yield f"{q}.{a}.proj.bias", (dw,), "normal", 0.02
        for n in ("norm1", "norm2", "norm3"):
            yield from _ln_specs(f"{q}.{n}", dw)
        yield f"{q}.mlp.0.weight", (4 * dw, dw), "normal", (2 * dw) ** -0.5 * 1.5
        yield f"{q}.mlp.0.bias", (4 * dw,), "normal", 0.02
        yield f"{q}.mlp.3.weight", (dw, 4 * dw), "normal", (4 * dw) ** -0.5
        yield f"{q}.mlp.3.bias", (dw,), "normal", 0.02
    yield from _ln_specs(f"{c}.out_proj.0", dw)
    yield f"{c}.out_proj.1.weight", (vd, dw), "normal", dw**-0.5
    yield f"{c}.out_proj.1.bias", (vd,), "normal", 0.02
    # the segmentor's own parameters (denseclip.py:104-108); trainable on the prompt-tuning path
    yield "contexts", (1, cfg.num_contexts, cfg.token_embed_dim), "normal", 0.1
    yield "gamma", (cfg.text_dim,), "ln_w", 0.0   # placeholder value 1.0; fixtures / callers set it (reference init: 1e-4)


def init_denseclip_state_dict(cfg, seed: int = 0, dtype: torch.dtype = torch.float32) -> dict[str, torch.Tensor]:
    """Seeded CPU draw of every DenseCLIP tensor on the hot path under the reference's module names; ``gamma`` starts at the
    reference's 1e-4 (denseclip.py:108).  BatchNorm running statistics are non-trivial: the frozen model runs in eval mode."""
    sd = _draw(denseclip_param_specs(cfg), seed, dtype)
    sd["gamma"] = torch.full_like(sd["gamma"], 1e-4)
    return sd
