"""CPU oracle for the DenseCLIP path (BASELINE configs[4]) -- TEST INFRASTRUCTURE ONLY.

Functional fp32 PyTorch-CPU restatement of the torch-only part of the reference's DenseCLIP
(``src/models/components/denseclip/models.py``: ``CLIPVisionTransformer`` :530-714, ``CLIPTextContextEncoder`` :805-903,
``ContextDecoder`` :907-960, ``TransformerDecoderLayer`` / ``Attention`` :448-526, ``ResidualAttentionBlock`` :391-431) and of the
segmentor's glue between them (``denseclip.py:140-169`` ``after_extract_feat``: text embeddings refined by the context decoder,
L2-normalised pixel-text score map, concatenation into one FPN map), over a flat state dict with the reference's module names
(``tunevlseg_amd.weights.init_denseclip_state_dict``).  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this module; the product path never does.

Pinned by ``tests/golden/denseclip_*.npz``: outputs of the reference's own three classes imported from ``/root/reference`` with a no-op
``mmseg.models.builder.BACKBONES`` registry (``tests/golden/make_denseclip_goldens.py``).  ``denseclip.py`` itself subclasses mmseg's
``BaseSegmentor`` and cannot be imported in this image (mmseg / mmengine absent): its ten glue lines are restated in the
generator and here, and the mmseg FPN neck / FPNHead behind them (``denseclip.py:171-205``, ``heads.py``) stay **unpinned and unbuilt**.
The whole model runs in eval mode (dropout and drop-path are identities, BatchNorm uses its running statistics).
"""
from __future__ import annotations

from typing import Mapping

import torch
import torch.nn.functional as F

from .clipseg_oracle import quick_gelu
from .cris_oracle import ln, mha

SD = Mapping[str, torch.Tensor]
EPS = 1e-5


def resblock(sd: SD, p: str, x: torch.Tensor, heads: int, causal: bool) -> torch.Tensor:
    """models.py:429-431 (batch-first): x + attn(ln_1 x); x + c_proj(QuickGELU(c_fc(ln_2 x)))."""
    h = ln(sd, f"{p}.ln_1", x)
    x = x + mha(sd, f"{p}.attn", h, h, h, heads, causal=causal)
    h = ln(sd, f"{p}.ln_2", x)
    return x + F.linear(quick_gelu(F.linear(h, sd[f"{p}.mlp.c_fc.weight"], sd[f"{p}.mlp.c_fc.bias"])), sd[f"{p}.mlp.c_proj.weight"], sd[f"{p}.mlp.c_proj.bias"])


def position_table(sd: SD, cfg, H: int, W: int) -> torch.Tensor:
    """models.py:682-692: [1 + H*W, C]; the spatial rows bilinearly resized (align_corners=False) from the checkpoint grid, and the class
    embedding added to the CLS row a SECOND time (it is already part of the token)."""
    pos = sd["backbone.positional_embedding"]
    C, g = pos.shape[1], cfg.grid
    cls_pos = pos[0] + sd["backbone.class_embedding"]
    sp = F.interpolate(pos[1:].reshape(1, g, g, C).permute(0, 3, 1, 2), size=(H, W), mode="bilinear")
    return torch.cat((cls_pos[None], sp.reshape(C, H * W).t()), 0)


def vision_forward(sd: SD, cfg, img: torch.Tensor):
    """``CLIPVisionTransformer.forward`` (models.py:660-714) -> ([fpn1..fpn4 maps, NCHW], global_embedding [B, E], visual_embedding [B, E, H, W])."""
    b = "backbone"
    x = F.conv2d(img, sd[f"{b}.conv1.weight"], stride=cfg.patch_size)
    B, C, H, W = x.shape
    x = x.flatten(2).transpose(1, 2)
    x = torch.cat((sd[f"{b}.class_embedding"].expand(B, 1, C), x), 1) + position_table(sd, cfg, H, W)
    x = ln(sd, f"{b}.ln_pre", x)
    taps = []
    for i in range(cfg.layers):
        x = resblock(sd, f"{b}.transformer.resblocks.{i}", x, cfg.heads, causal=False)
        if i in cfg.out_indices:
            taps.append(x[:, 1:].transpose(1, 2).reshape(B, C, H, W))
    gn = lambda p, t: F.group_norm(t, 1, sd[f"{p}.weight"], sd[f"{p}.bias"], EPS)  # noqa: E731
    tconv = lambda p, t: F.conv_transpose2d(t, sd[f"{p}.weight"], sd[f"{p}.bias"], stride=2)  # noqa: E731
    f1 = tconv(f"{b}.fpn1.1", gn(f"{b}.fpn1.0", taps[0]))
    f1 = F.batch_norm(f1, sd[f"{b}.fpn1.2.running_mean"], sd[f"{b}.fpn1.2.running_var"], sd[f"{b}.fpn1.2.weight"], sd[f"{b}.fpn1.2.bias"], False, 0.0, EPS)
    f1 = tconv(f"{b}.fpn1.4", F.gelu(f1))
    f2 = tconv(f"{b}.fpn2.1", gn(f"{b}.fpn2.0", taps[1]))
    f3 = gn(f"{b}.fpn3", taps[2])
    f4 = F.max_pool2d(gn(f"{b}.fpn4.0", taps[3]), 2, 2)
    x = ln(sd, f"{b}.ln_post", x) @ sd[f"{b}.proj"]
    return [f1, f2, f3, f4], x[:, 0], x[:, 1:].reshape(B, H, W, -1).permute(0, 3, 1, 2)


def text_context_encoder(sd: SD, cfg, text: torch.Tensor, contexts: torch.Tensor) -> torch.Tensor:
    """``CLIPTextContextEncoder.forward`` (models.py:878-903): [BOS, contexts, tokens 1..] + positions, causal tower, ln_final, the row at
    ``argmax(text) + n_contexts`` through the text projection -> [Bc, K, E] (Bc = contexts' batch, 1 in DenseCLIP)."""
    t = "text_encoder"
    emb = sd[f"{t}.token_embedding.weight"][text]                       # [K, N1, C]
    K, N1, C = emb.shape
    Bc, N2, _ = contexts.shape
    eos = (text.argmax(-1) + N2).reshape(1, K).expand(Bc, K).reshape(-1)
    x = torch.cat((emb[None, :, :1].expand(Bc, K, 1, C), contexts[:, None].expand(Bc, K, N2, C), emb[None, :, 1:].expand(Bc, K, N1 - 1, C)), 2)
    x = x.reshape(Bc * K, N1 + N2, C) + sd[f"{t}.positional_embedding"]
    for i in range(cfg.transformer_layers):
        x = resblock(sd, f"{t}.transformer.resblocks.{i}", x, cfg.transformer_heads, causal=True)
    x = ln(sd, f"{t}.ln_final", x)
    return (x[torch.arange(x.shape[0]), eos] @ sd[f"{t}.text_projection"]).reshape(Bc, K, -1)


def dc_attention(sd: SD, p: str, q, k, v, heads: int) -> torch.Tensor:
    """``Attention.forward`` (models.py:463-481): bias-free q / k / v projections, softmax(q k^T d_h^-1/2) v, output projection with bias."""
    B, N, C = q.shape
    M = k.shape[1]
    dh = C // heads
    qh = F.linear(q, sd[f"{p}.q_proj.weight"]).view(B, N, heads, dh).transpose(1, 2)
    kh = F.linear(k, sd[f"{p}.k_proj.weight"]).view(B, M, heads, dh).transpose(1, 2)
    vh = F.linear(v, sd[f"{p}.v_proj.weight"]).view(B, M, heads, dh).transpose(1, 2)
    o = ((qh @ kh.transpose(-1, -2)) * dh**-0.5).softmax(-1) @ vh
    return F.linear(o.transpose(1, 2).reshape(B, N, C), sd[f"{p}.proj.weight"], sd[f"{p}.proj.bias"])


def context_decoder(sd: SD, cfg, text: torch.Tensor, visual: torch.Tensor) -> torch.Tensor:
    """``ContextDecoder.forward`` (models.py:951-960) over ``TransformerDecoderLayer`` (models.py:508-526)."""
    c = "context_decoder"
    mem = ln(sd, f"{c}.memory_proj.2", F.linear(ln(sd, f"{c}.memory_proj.0", visual), sd[f"{c}.memory_proj.1.weight"], sd[f"{c}.memory_proj.1.bias"]))
    x = F.linear(ln(sd, f"{c}.text_proj.0", text), sd[f"{c}.text_proj.1.weight"], sd[f"{c}.text_proj.1.bias"])
    for i in range(cfg.decoder_layers):
        p = f"{c}.decoder.{i}"
        q = ln(sd, f"{p}.norm1", x)
        x = x + dc_attention(sd, f"{p}.self_attn", q, q, q, cfg.decoder_heads)
        x = x + dc_attention(sd, f"{p}.cross_attn", ln(sd, f"{p}.norm2", x), mem, mem, cfg.decoder_heads)
        h = F.gelu(F.linear(ln(sd, f"{p}.norm3", x), sd[f"{p}.mlp.0.weight"], sd[f"{p}.mlp.0.bias"]))
        x = x + F.linear(h, sd[f"{p}.mlp.3.weight"], sd[f"{p}.mlp.3.bias"])
    return F.linear(ln(sd, f"{c}.out_proj.0", x), sd[f"{c}.out_proj.1.weight"], sd[f"{c}.out_proj.1.bias"])


def after_extract_feat(sd: SD, cfg, feats, global_feat, visual_embeddings, texts, contexts, gamma):
    """``DenseCLIP.after_extract_feat`` (denseclip.py:140-169) -> (text_embeddings [B, K, C], x_orig, score_map [B, K, H, W])."""
    B, C, H, W = visual_embeddings.shape
    visual_context = torch.cat((global_feat.reshape(B, C, 1), visual_embeddings.reshape(B, C, H * W)), 2).permute(0, 2, 1)
    text_embeddings = text_context_encoder(sd, cfg, texts, contexts).expand(B, -1, -1)
    text_embeddings = text_embeddings + gamma * context_decoder(sd, cfg, text_embeddings, visual_context)
    score_map = torch.einsum("bchw,bkc->bkhw", F.normalize(visual_embeddings, dim=1, p=2), F.normalize(text_embeddings, dim=2, p=2))
    x_orig = list(feats)
    x_orig[cfg.score_concat_index] = torch.cat((x_orig[cfg.score_concat_index], score_map), 1)
    return text_embeddings, x_orig, score_map


def denseclip_forward(sd: SD, cfg, img, texts, contexts=None, gamma=None):
    """``extract_feat`` + ``after_extract_feat`` (denseclip.py:136-169): what reaches the mmseg neck / head."""
    feats, g, v = vision_forward(sd, cfg, img)
    return after_extract_feat(sd, cfg, feats, g, v, texts, sd["contexts"] if contexts is None else contexts, sd["gamma"] if gamma is None else gamma)
