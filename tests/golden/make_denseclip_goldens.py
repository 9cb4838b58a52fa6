#!/usr/bin/env python
"""Generate the DenseCLIP golden fixtures by running the REFERENCE classes (build container only).

Run:  python tests/golden/make_denseclip_goldens.py [name-prefix ...]          (needs /root/reference)

What it does
------------
* imports the reference's own ``CLIPVisionTransformer``, ``CLIPTextContextEncoder`` and ``ContextDecoder`` from
  ``/root/reference/src/models/components/denseclip/models.py`` (:530-714, :805-903, :907-960) -- nothing of the reference is
  copied.  ``models.py`` is torch-only apart from ``from mmseg.models.builder import BACKBONES`` (line 6, a class-registration
  decorator); mmseg is not in the image, so a no-op registry is put in ``sys.modules`` first -- the same standing as the
  transformers shim of ``make_goldens.py``.  The package ``__init__`` (which imports the mmseg-based segmentor) is bypassed by
  importing ``models.py`` through a path-only package object;
* ``denseclip.py`` (the segmentor, an mmseg ``BaseSegmentor``) and ``heads.py`` cannot be imported: the ten glue lines between the
  three modules (``denseclip.py:140-169``: visual context, text embeddings + gamma * context-decoder output, L2-normalised score map) are
  restated below from the reference text, and in ``oracle/denseclip_oracle.py``; the mmseg FPN neck / FPNHead stay unpinned;
* loads the seeded random weights of ``tunevlseg_amd.weights.init_denseclip_state_dict`` (strict), eval mode, strict fp32 on CPU;
* runs forward and the backward of  L = sum(score_map * Gs) + sum(text_embeddings * Gt)  (seeded cotangents) and writes
  ``tests/golden/denseclip_*.npz``: inputs (or their seed), the trainable tensors, every output of the path and the gradients w.r.t.
  ``contexts``, ``gamma`` and the context decoder's parameters (+ the same gradients from a float64 run of the same classes).
"""
from __future__ import annotations

import importlib
import json
import os
import sys
import types
from pathlib import Path

import numpy as np
import torch
import torch.nn.functional as F

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
REFERENCE = Path(os.environ.get("TVL_REFERENCE", "/root/reference"))

from tests.golden_util import denseclip_subsample, synth_denseclip_inputs  # noqa: E402
from tunevlseg_amd.denseclip_config import DenseCLIPConfig  # noqa: E402
from tunevlseg_amd.weights import init_denseclip_state_dict  # noqa: E402

OUT = Path(__file__).resolve().parent
ONLY = sys.argv[1:]
COMPACT_DECODER_GRADS = ("context_decoder.decoder.0.cross_attn.q_proj.weight", "context_decoder.out_proj.1.weight", "context_decoder.text_proj.1.weight")


def reference_models():
    class _Registry:   # mmseg.models.builder.BACKBONES: only ``@BACKBONES.register_module()`` is used (models.py:176,529,718,804,906)
        def register_module(self, *a, **k):
            return lambda cls: cls

    for name in ("mmseg", "mmseg.models", "mmseg.models.builder"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["mmseg.models.builder"].BACKBONES = _Registry()
    pkg = types.ModuleType("_ref_denseclip")
    pkg.__path__ = [str(REFERENCE / "src" / "models" / "components" / "denseclip")]
    sys.modules["_ref_denseclip"] = pkg
    return importlib.import_module("_ref_denseclip.models")


def build(M, cfg: DenseCLIPConfig, sd):
    bb = M.CLIPVisionTransformer(input_resolution=cfg.input_resolution, patch_size=cfg.patch_size, width=cfg.width, layers=cfg.layers,
                                 heads=cfg.heads, output_dim=cfg.output_dim, drop_path_rate=0.1, out_indices=list(cfg.out_indices),
                                 get_embeddings=True)
    te = M.CLIPTextContextEncoder(context_length=cfg.text_context_length, vocab_size=cfg.vocab_size, transformer_width=cfg.transformer_width,
                                  transformer_heads=cfg.transformer_heads, transformer_layers=cfg.transformer_layers, embed_dim=cfg.embed_dim)
    cd = M.ContextDecoder(transformer_width=cfg.decoder_width, transformer_heads=cfg.decoder_heads, transformer_layers=cfg.decoder_layers,
                          visual_dim=cfg.visual_dim, dropout=0.1)
    for mod, prefix in ((bb, "backbone."), (te, "text_encoder."), (cd, "context_decoder.")):
        sub = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
        missing, unexpected = mod.load_state_dict(sub, strict=False)
        assert not unexpected and all(m.endswith("num_batches_tracked") for m in missing), (prefix, missing, unexpected)
        mod.eval()
    return bb, te, cd


def glue(cfg, feats, text_encoder, context_decoder, texts, contexts, gamma):
    """``DenseCLIP.after_extract_feat`` restated (denseclip.py:140-169)."""
    x_orig = list(feats[:4])
    global_feat, visual_embeddings = feats[4]
    B, C, H, W = visual_embeddings.shape
    visual_context = torch.cat((global_feat.reshape(B, C, 1), visual_embeddings.reshape(B, C, H * W)), dim=2).permute(0, 2, 1)
    text_embeddings = text_encoder(texts, contexts).expand(B, -1, -1)
    text_diff = context_decoder(text_embeddings, visual_context)
    text_embeddings = text_embeddings + gamma * text_diff
    score_map = torch.einsum("bchw,bkc->bkhw", F.normalize(visual_embeddings, dim=1, p=2), F.normalize(text_embeddings, dim=2, p=2))
    x_orig[cfg.score_concat_index] = torch.cat([x_orig[cfg.score_concat_index], score_map], dim=1)
    return text_embeddings, x_orig, score_map


def run_case(M, name: str, cfg: DenseCLIPConfig, *, wseed: int, iseed: int, B: int, H: int, compact: bool, W: int | None = None):
    if ONLY and not any(name.startswith(o) for o in ONLY):
        return
    torch.set_float32_matmul_precision("highest")
    sd = init_denseclip_state_dict(cfg, wseed)
    bb, te, cd = build(M, cfg, sd)
    for mod in (bb, te):
        mod.requires_grad_(False)
    g = torch.Generator().manual_seed(3000 + iseed)
    contexts = (torch.randn(1, cfg.num_contexts, cfg.token_embed_dim, generator=g) * 0.1).requires_grad_(True)
    gamma = (0.3 + 0.1 * torch.randn(cfg.text_dim, generator=g)).requires_grad_(True)
    pix, texts, gs, gt = synth_denseclip_inputs(cfg, B, H, iseed, W)

    def run(dtype):
        c_, g_ = (contexts.detach().to(dtype).requires_grad_(True), gamma.detach().to(dtype).requires_grad_(True))
        mods = (bb, te, cd) if dtype == torch.float32 else tuple(__import__("copy").deepcopy(m).to(dtype) for m in (bb, te, cd))
        for p in mods[2].parameters():
            p.grad = None
        feats = mods[0](pix.to(dtype))
        text_embeddings, x_orig, score_map = glue(cfg, feats, mods[1], mods[2], texts, c_, g_)
        loss = (score_map * gs.to(dtype)).sum() + (text_embeddings * gt.to(dtype)).sum()
        loss.backward()
        grads = {"contexts": c_.grad, "gamma": g_.grad, **{f"context_decoder.{k}": p.grad for k, p in mods[2].named_parameters()}}
        return feats, text_embeddings, x_orig, score_map, loss, grads

    feats, text_embeddings, x_orig, score_map, loss, grads = run(torch.float32)
    assert torch.equal(x_orig[cfg.score_concat_index][:, : cfg.width], feats[cfg.score_concat_index])
    arrays = {"in.texts": texts.numpy(), "param.contexts": contexts.detach().numpy(), "param.gamma": gamma.detach().numpy(),
              "out.loss": loss.detach().numpy(), "out.score_map": score_map.detach().numpy(), "out.text_embeddings": text_embeddings.detach().numpy(),
              "out.global_embedding": feats[4][0].detach().numpy(),
              "out.visual_embedding": denseclip_subsample("visual_embedding", feats[4][1].detach(), compact).numpy()}
    if not compact:
        arrays["in.pixel_values"] = pix.numpy()
    for i in range(4):
        arrays[f"out.fpn{i + 1}"] = denseclip_subsample(f"fpn{i + 1}", feats[i].detach(), compact).numpy()
        arrays[f"out.fpn{i + 1}_abs_sum"] = feats[i].detach().double().abs().sum().numpy()   # whole-map checksum of the compact fixtures
    # compact fixtures keep the context decoder's small gradients (biases, norms) and three of its weight matrices; the rest (3.2 M floats
    # per run) would make the file 30 MB
    keep = lambda k, v: not compact or not k.startswith("context_decoder.") or v.numel() <= 2048 or k in COMPACT_DECODER_GRADS  # noqa: E731
    for k, v in grads.items():
        if keep(k, v):
            arrays["grad." + k] = v.numpy()
    # the reference's LayerNorm subclass pins its input to fp32 (models.py:363-369): lifted for the float64 run of the same classes
    ln_forward = M.LayerNorm.forward
    M.LayerNorm.forward = torch.nn.LayerNorm.forward
    try:
        _, te64, _, sm64, loss64, grads64 = run(torch.float64)
    finally:
        M.LayerNorm.forward = ln_forward
    for k, v in grads64.items():
        if keep(k, v):
            arrays["grad64." + k] = v.float().numpy()
    dev = max(float((grads[k].double() - grads64[k]).norm() / grads64[k].norm().clamp(min=1e-30)) for k in ("contexts", "gamma"))
    meta = {"name": name, "family": "denseclip", "compact": compact, "config": cfg.to_dict(), "weight_seed": wseed, "input_seed": iseed, "B": B, "H": H, "W": H if W is None else W,
            "weights_checksum": float(sum(v.double().abs().sum() for k, v in sd.items() if k not in ("contexts", "gamma"))), "torch": torch.__version__}
    arrays["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(OUT / f"{name}.npz", **arrays)
    print(f"{name}: loss {loss.item():.6f} (f64 {loss64.item():.6f}) score_map |max| {score_map.abs().max():.4f} std {score_map.std():.4f} "
          f"text_emb |max| {text_embeddings.abs().max():.3f} fpn1 |max| {feats[0].abs().max():.3f} grads "
          f"{[(k, float(v.abs().max())) for k, v in list(grads.items())[:2]]} fp32-vs-fp64 rel {dev:.2e}; "
          f"score fp32-vs-fp64 {float((score_map.double() - sm64).abs().max()):.2e}")


def main():
    M = reference_models()
    tiny = DenseCLIPConfig.tiny()
    # reduced width; the 96 x 96 image makes the position table resize 4 x 4 -> 6 x 6 (models.py:684-690)
    run_case(M, "denseclip_tiny_b2_96", tiny, wseed=51, iseed=51, B=2, H=96, compact=False)
    # the checkpoint's own grid (identity resize), odd batch, more classes than decoder heads
    run_case(M, "denseclip_tiny_b3_64_k7", DenseCLIPConfig.tiny(num_classes=7), wseed=52, iseed=52, B=3, H=64, compact=False)
    # a non-square image: the position table resized 4 x 4 -> 4 x 6 (models.py:684-690 takes (H, W)), every map H != W
    run_case(M, "denseclip_tiny_b2_64x96", tiny, wseed=53, iseed=53, B=2, H=64, W=96, compact=False)
    # BASELINE configs[4] geometry: ViT-B/16 at 640 x 640, 20 classes, 8 contexts (compact: large maps subsampled)
    run_case(M, "denseclip_vitb16_640_b1", DenseCLIPConfig.vitb16_640(), wseed=61, iseed=61, B=1, H=640, compact=True)
    # ... and with two samples (the per-sample context decoder / score map over a batch)
    run_case(M, "denseclip_vitb16_640_b2", DenseCLIPConfig.vitb16_640(), wseed=61, iseed=62, B=2, H=640, compact=True)


if __name__ == "__main__":
    main()
