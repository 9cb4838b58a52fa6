"""reference ``src/models/components/denseclip/denseclip.py:21-169`` (the segmentor up to the mmseg neck / head) over
``models.py``'s ``CLIPVisionTransformer`` (:530-714), ``CLIPTextContextEncoder`` (:805-903) and ``ContextDecoder`` (:907-960)
-- BASELINE configs[4].

Same constructor keywords, attribute names (``backbone``, ``text_encoder``, ``context_decoder``, ``contexts``, ``gamma``, ``texts``,
``num_classes``, ``context_length``, ``score_concat_index``, ``tau``) and ``state_dict`` keys as the reference segmentor; the arithmetic is HIP
launches.  Feature maps live as NHWC pixel matrices and are handed out as ``[B, C, H, W]``-shaped views of them (the values and shapes of
the reference's tensors, channels-last strides).  The CLIP towers are frozen (prompt tuning): the image side runs without an autograd tape;
``contexts`` and ``gamma`` train, and the context decoder's parameters when ``train_context_decoder`` (DenseCLIP's own recipe trains it).

What is NOT here: the mmseg ``FPN`` neck, ``FPNHead`` / ``IdentityHead`` and their losses (``denseclip.py:171-290``, ``heads.py``) -- mmseg and
mmengine are absent from the image and those classes have no source under the reference tree; ``forward`` returns what
``after_extract_feat`` hands to them.  Passing ``neck`` / ``decode_head`` / ``auxiliary_head`` / ``identity_head`` raises.
"""
from __future__ import annotations

import os
from collections.abc import Iterable, Mapping
from typing import Any

import torch
from torch import nn

from .. import cris_ops as C
from .. import hip, ops
from ..denseclip_backbone import DenseCLIPWeights
from ..denseclip_config import DenseCLIPConfig
from .towers import SideStream

LN_EPS = 1e-5
GN_EPS = 1e-5


class ColScaleAddFn(ops.Fn):
    """``a + g * b`` with a trainable per-column ``g`` (``text_embeddings + self.gamma * text_diff``, denseclip.py:157)."""

    @staticmethod
    def forward(ctx, a, b, g):
        a, b, g = ops._c(a), ops._c(b), ops._c(g)
        ctx.save_for_backward(b, g)
        return hip.colscale_add(a, b, g)

    @staticmethod
    def backward(ctx, d):
        b, g = ctx.saved_tensors
        d = ops._c(d)
        db, dg = hip.colscale_bwd(d, b, g, ctx.needs_input_grad[1], ctx.needs_input_grad[2])
        return (d if ctx.needs_input_grad[0] else None), db, dg


class ScoreMapFn(ops.Fn):
    """``einsum("bchw,bkc->bkhw", normalize(visual), normalize(text))`` (denseclip.py:162-165) as one [H*W, C] x [K, C]^T GEMM per sample over
    the pixel matrix -> [B*H*W, K].  The image side is frozen: only the text operand has a gradient, dT[b] = dS[b]^T V[b]."""

    @staticmethod
    def forward(ctx, v_hat, t_hat, B, HW, skip_rows):
        K, Cc = t_hat.shape[-2:]
        t_hat = ops._c(t_hat).view(B, K, Cc)
        T = skip_rows + HW
        if B > 1 and B * K <= 4096:
            # ONE GEMM of every sample's rows against every sample's class vectors ([B*T, C] x [B*K, C]^T: B times the FLOPs of the diagonal blocks, all of them
            # small) + one gather of the diagonal blocks, instead of B skinny launches of 25 workgroups each (16 x 56 us at 640^2)
            full = torch.empty((B * T, B * K), device=v_hat.device, dtype=torch.float32)
            hip.gemm(hip.NT, B * T, B * K, Cc, v_hat, Cc, t_hat.view(B * K, Cc), Cc, full, B * K)
            out = hip.blockdiag_gather(full, B, T, skip_rows, HW, K)
        else:
            out = torch.empty((B * HW, K), device=v_hat.device, dtype=torch.float32)
            for b in range(B):
                vb = v_hat[b * T + skip_rows: (b + 1) * T]
                hip.gemm(hip.NT, HW, K, Cc, vb, Cc, t_hat[b], Cc, out[b * HW: (b + 1) * HW], K)
        ctx.save_for_backward(v_hat)
        ctx.geom = (B, HW, skip_rows, K, Cc)
        return out

    @staticmethod
    def backward(ctx, d):
        (v_hat,) = ctx.saved_tensors
        B, HW, skip_rows, K, Cc = ctx.geom
        T = skip_rows + HW
        if d.dim() != 2 or d.stride(1) != 1:
            d = d.contiguous()
        dt = hip.score_map_text_grad(d, v_hat, B, T, skip_rows, HW, K) if d.is_contiguous() else None   # one launch (up to 32 classes); else a TN GEMM per sample
        if dt is None:
            dt = torch.empty((B, K, Cc), device=d.device, dtype=torch.float32)
            for b in range(B):
                hip.gemm(hip.TN, K, Cc, HW, d[b * HW: (b + 1) * HW], d.stride(0), v_hat[b * T + skip_rows: (b + 1) * T], Cc, dt[b], Cc)
        return None, dt, None, None, None


class DenseCLIP(nn.Module):
    def __init__(self, backbone: Mapping[str, Any] | None = None, text_encoder: Mapping[str, Any] | None = None,
                 context_decoder: Mapping[str, Any] | None = None, decode_head: Any = None, class_names: Iterable[str] | None = None,
                 context_length: int = 5, context_feature: str = "attention", score_concat_index: int = 3, text_head: bool = False, neck: Any = None,
                 tau: float = 0.07, auxiliary_head: Any = None, identity_head: Any = None, train_cfg: Any = None, test_cfg: Any = None,
                 pretrained: Any = None, init_cfg: Any = None, token_embed_dim: int = 512, text_dim: int = 1024, *,
                 texts: torch.Tensor | None = None, train_context_decoder: bool = False, bpe_path: str | os.PathLike | None = None, **args) -> None:
        super().__init__()
        for name, val in (("neck", neck), ("decode_head", decode_head), ("auxiliary_head", auxiliary_head), ("identity_head", identity_head)):
            if val is not None:
                raise NotImplementedError(
                    f"DenseCLIP({name}=...): the mmseg neck / decode heads are outside the hot path (mmseg is not importable here and their source "
                    "is not part of the reference tree); this module ends at after_extract_feat")
        if context_feature not in ("attention", "backbone"):
            raise AssertionError(context_feature)   # denseclip.py:81
        bb, te, cd = dict(backbone or {}), dict(text_encoder or {}), dict(context_decoder or {})
        if isinstance(pretrained, DenseCLIPWeights):
            weights = pretrained
        else:
            cfg = DenseCLIPConfig(
                input_resolution=bb.get("input_resolution", 224), patch_size=bb.get("patch_size", 32), width=bb.get("width", 768),
                layers=bb.get("layers", 12), heads=bb.get("heads", 12), output_dim=bb.get("output_dim", 512),
                out_indices=tuple(bb.get("out_indices") or (3, 5, 7, 11)),
                text_context_length=te.get("context_length", 22), vocab_size=te.get("vocab_size", 49408), transformer_width=te.get("transformer_width", 512),
                transformer_heads=te.get("transformer_heads", 8), transformer_layers=te.get("transformer_layers", 12), embed_dim=te.get("embed_dim", 1024),
                decoder_width=cd.get("transformer_width", 256), decoder_heads=cd.get("transformer_heads", 4), decoder_layers=cd.get("transformer_layers", 6),
                visual_dim=cd.get("visual_dim", 1024), context_length=context_length, score_concat_index=score_concat_index, tau=tau,
                token_embed_dim=token_embed_dim, text_dim=text_dim)
            if not bb.get("get_embeddings", True):
                raise NotImplementedError("DenseCLIP needs backbone.get_embeddings=True (the score map reads the projected tokens, models.py:703-712)")
            weights = DenseCLIPWeights.from_spec(pretrained if pretrained is not None else "random:vitb16_640", cfg)
        cfg = weights.config
        if texts is None:
            if class_names is None:
                raise ValueError("DenseCLIP needs class_names (tokenised like the reference, untils.py:173-221) or pre-tokenised texts [K, context_length]")
            texts = tokenize(class_names, cfg.context_length, bpe_path)
        texts = torch.as_tensor(texts, dtype=torch.long)
        if texts.dim() != 2 or texts.shape[1] != cfg.context_length:
            raise ValueError(f"texts must be [num_classes, context_length={cfg.context_length}], got {tuple(texts.shape)}")
        cfg.num_classes = int(texts.shape[0])
        self.config = cfg
        self._weights = [weights]   # not a submodule: its three subtrees are registered under the reference's names below
        self.backbone, self.text_encoder, self.context_decoder = weights.backbone, weights.text_encoder, weights.context_decoder
        self.context_length = cfg.context_length
        self.score_concat_index = cfg.score_concat_index
        self.context_feature = context_feature
        self.text_head = text_head
        self.tau = cfg.tau
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.register_buffer("texts", texts, persistent=False)
        self.num_classes = cfg.num_classes
        self.requires_grad_(False)
        self.eval()
        self.context_decoder.requires_grad_(bool(train_context_decoder))
        # denseclip.py:104-108
        self.contexts = nn.Parameter(torch.randn(1, cfg.num_contexts, cfg.token_embed_dim))
        nn.init.trunc_normal_(self.contexts)
        self.gamma = nn.Parameter(torch.ones(cfg.text_dim) * 1e-4)

    @property
    def weights(self) -> DenseCLIPWeights:
        return self._weights[0]

    def _apply(self, fn, *a, **k):
        w = self.weights
        w._prep, w._plist, w._const = None, None, {}
        return super()._apply(fn, *a, **k)

    def train(self, mode: bool = True):
        """The frozen towers and the context decoder stay in eval mode (dropout / drop-path off, BatchNorm on running statistics) whatever the
        caller asks: that is the arithmetic the kernels implement."""
        return super().train(False)

    # ------------------------------------------------------------------ CLIPVisionTransformer.forward (models.py:660-714), no autograd tape
    def _vision_tokens(self, inputs: torch.Tensor):
        """-> (taps: four token matrices [B, 1 + H*W, C] after the ``out_indices`` blocks, projected tokens [B*(1 + H*W), E], H, W)."""
        cfg, prep = self.config, self.weights.prepared()
        B, _, Hi, Wi = inputs.shape
        ps = cfg.patch_size
        if Hi % ps or Wi % ps:
            raise ValueError(f"image size {Hi}x{Wi} is not a multiple of the patch size {ps}")
        H, W = Hi // ps, Wi // ps
        with torch.no_grad():
            cols = hip.im2col_patch(ops._c(inputs.to(torch.float32)), ps)
            patch = hip.linear_fwd(cols, prep["patch_w"]).view(B, H * W, cfg.width)
            del cols
            x = hip.vision_assemble(patch.view(B * H * W, cfg.width), self.backbone.class_embedding.detach(), self.weights.position_table(H, W), None, 0, B, H * W, 0, cfg.width)
            del patch
            T = 1 + H * W
            x = hip.layernorm_fwd(x.view(B * T, cfg.width), *prep["ln_pre"], LN_EPS, want_stats=False)[0].view(B, T, cfg.width)
            spec = ops.AttnSpec(cfg.heads, hip.ACT_QUICK_GELU, LN_EPS)
            taps = []
            for i, lw in enumerate(prep["vision_layers"]):
                x = ops.encoder_layer(x, lw, spec)
                if i in cfg.out_indices:
                    taps.append(x)
            xe = C.flinear(hip.layernorm_fwd(x.view(B * T, cfg.width), *prep["ln_post"], LN_EPS, want_stats=False)[0], prep["proj"])
        return taps, xe, H, W

    def _fpn(self, taps, B: int, H: int, W: int, fpn3_out: torch.Tensor | None = None):
        """fpn1 .. fpn4 (models.py:581-600) as NHWC pixel matrices.  A ``ConvTranspose2d(k=2, s=2)`` is a GEMM into (dy, dx, co) columns; what follows
        it in fpn1 (BatchNorm, folded; GELU; the second transposed conv) is per pixel, so the pixels stay in the GEMM's blocked order until one
        final pass puts them back in raster order."""
        prep, Cc = self.weights.prepared(), self.config.width
        with torch.no_grad():
            x = hip.groupnorm_nhwc(taps[0], 1, H, W, *prep["fpn1_gn"], GN_EPS)
            x = C.flinear(x, prep["fpn1_t1"], hip.ACT_GELU)
            x = C.flinear(x.view(-1, Cc), prep["fpn1_t2"])
            f1 = hip.tconv2x2_unshuffle(x, B, H, W, Cc, 2)
            x = C.flinear(hip.groupnorm_nhwc(taps[1], 1, H, W, *prep["fpn2_gn"], GN_EPS), prep["fpn2_t"])
            f2 = hip.tconv2x2_unshuffle(x, B, H, W, Cc, 1)
            del x
            f3 = hip.groupnorm_nhwc(taps[2], 1, H, W, *prep["fpn3_gn"], GN_EPS, out=fpn3_out)
            f4 = hip.groupnorm_nhwc(taps[3], 1, H, W, *prep["fpn4_gn"], GN_EPS, pool=2)
        return f1, f2, f3, f4

    @staticmethod
    def _nchw(x2d: torch.Tensor, B: int, H: int, W: int) -> torch.Tensor:
        return x2d.view(B, H, W, x2d.shape[-1]).permute(0, 3, 1, 2)

    def extract_feat(self, inputs: torch.Tensor):
        """``self.backbone(inputs)`` (denseclip.py:136-138): (fpn1, fpn2, fpn3, fpn4, [global_embedding, visual_embedding])."""
        B = inputs.shape[0]
        taps, xe, H, W = self._vision_tokens(inputs)
        f1, f2, f3, f4 = self._fpn(taps, B, H, W)
        E = xe.shape[1]
        xe3 = xe.view(B, 1 + H * W, E)
        return (self._nchw(f1, B, 4 * H, 4 * W), self._nchw(f2, B, 2 * H, 2 * W), self._nchw(f3, B, H, W), self._nchw(f4, B, H // 2, W // 2),
                [xe3[:, 0], xe3[:, 1:].view(B, H, W, E).permute(0, 3, 1, 2)])

    # ------------------------------------------------------------------ CLIPTextContextEncoder.forward (models.py:878-903)
    def encode_text(self) -> torch.Tensor:
        """-> [K, E]: one row per class, the same for every sample (the reference expands it over the batch, denseclip.py:152-154)."""
        cfg, prep = self.config, self.weights.prepared()
        te = self.text_encoder
        dev = te.positional_embedding.device
        K, N1 = self.texts.shape
        N2 = cfg.num_contexts
        tmap = hip.const_i32([0, *[-(j + 1) for j in range(N2)], *range(1, N1)], dev)
        x = ops.TextAssembleFn.apply(self.texts.contiguous(), tmap, te.token_embedding.weight.detach(), self.contexts[0], te.positional_embedding.detach(), N2)
        spec = ops.AttnSpec(cfg.transformer_heads, hip.ACT_QUICK_GELU, LN_EPS, causal=True)
        for lw in prep["text_layers"]:
            x = ops.EncoderLayerFn.apply(x, lw, spec)
        x = ops.layer_norm(x, te.ln_final.weight.detach(), te.ln_final.bias.detach(), LN_EPS)
        eos = (self.texts.argmax(dim=-1) + N2).to(torch.int32)
        return C.flinear_g(ops.GatherRowsFn.apply(x, eos), prep["text_projection"])

    # ------------------------------------------------------------------ ContextDecoder.forward (models.py:951-960)
    def _attention(self, node, q_in, kv_in, B: int, Tq: int, Tk: int):
        """``Attention.forward`` (models.py:463-481): bias-free q / k / v projections, output projection with bias (added by the caller with the residual)."""
        heads = self.config.decoder_heads
        D = q_in.shape[-1]
        q = ops.linear(q_in, node.q_proj.weight).view(B * Tq, D)
        k = ops.linear(kv_in, node.k_proj.weight).view(B * Tk, D)
        v = ops.linear(kv_in, node.v_proj.weight).view(B * Tk, D)
        return C.CrossAttnFn.apply(q, k, v, None, B, Tq, Tk, heads, D // heads)

    def decode_context(self, text: torch.Tensor, visual_context: torch.Tensor, B: int) -> torch.Tensor:
        """text [B, K, C], visual_context [B*(1 + H*W), C] -> text_diff [B, K, C]."""
        cd = self.context_decoder
        K = text.shape[1]
        Tm = visual_context.shape[0] // B
        ln = lambda node, t: ops.layer_norm(t, node.weight, node.bias, LN_EPS)  # noqa: E731
        mp, tp, op = cd.memory_proj, cd.text_proj, cd.out_proj
        mem = ln(mp[2], ops.linear(ln(mp[0], visual_context), mp[1].weight, mp[1].bias))
        x = ops.linear(ln(tp[0], text), tp[1].weight, tp[1].bias)
        for layer in cd.decoder:
            q = ln(layer.norm1, x)
            o = self._attention(layer.self_attn, q, q, B, K, K)
            x = ops.linear(o.view(B, K, -1), layer.self_attn.proj.weight, layer.self_attn.proj.bias, residual=x)
            o = self._attention(layer.cross_attn, ln(layer.norm2, x), mem, B, K, Tm)
            x = ops.linear(o.view(B, K, -1), layer.cross_attn.proj.weight, layer.cross_attn.proj.bias, residual=x)
            h = ops.linear(ln(layer.norm3, x), layer.mlp[0].weight, layer.mlp[0].bias, act=hip.ACT_GELU)
            x = ops.linear(h, layer.mlp[3].weight, layer.mlp[3].bias, residual=x)
        return ops.linear(ln(op[0], x), op[1].weight, op[1].bias)

    # ------------------------------------------------------------------ DenseCLIP.after_extract_feat (denseclip.py:140-169)
    def forward(self, inputs: torch.Tensor):
        """-> ``(text_embeddings [B, K, C], x_orig, score_map [B, K, H, W])`` with ``x_orig`` the four FPN maps, the score map concatenated behind the
        channels of map ``score_concat_index``: what the reference hands to its neck / decode head."""
        cfg = self.config
        B = inputs.shape[0]
        K = self.num_classes
        dev = inputs.device
        side = SideStream(dev)
        taps, xe, H, W = self._vision_tokens(inputs)
        with side:   # the text encoder sees parameters only: K x 13 rows of launch-bound kernels beside the chip-filling vision tower
            te = self.encode_text()
        side.join(te)
        HW, Cc = H * W, cfg.width
        concat = None
        if cfg.score_concat_index == 2:   # fpn3 and the score map share one [B*H*W, C + K] matrix: the GroupNorm writes its first C columns in place
            concat = torch.empty((B * HW, Cc + K), device=dev, dtype=torch.float32)
        # The FPN taps (chip-filling GEMMs, GroupNorm, unshuffles: no tape, nothing below reads them) on the second stream, beside the context decoder's
        # launch-bound chain over K rows per sample; joined before the maps are handed out.
        fpn_side = SideStream(dev) if FPN_SIDE_STREAM else None
        if fpn_side is not None and fpn_side.on:
            for t in taps:
                t.record_stream(fpn_side.side)   # allocated on this stream, last read on that one
            with fpn_side:
                f1, f2, f3, f4 = self._fpn(taps, B, H, W, None if concat is None else concat[:, :Cc])
        else:
            fpn_side = None
            f1, f2, f3, f4 = self._fpn(taps, B, H, W, None if concat is None else concat[:, :Cc])
        del taps
        text = ops.OuterAddFn.apply(hip_zeros(B, te.shape[1], dev), te)                         # expand(B, -1, -1)
        diff = self.decode_context(text, xe, B)                                                  # visual_context = [global | pixels] = the projected tokens as they stand
        text_embeddings = ColScaleAddFn.apply(text, diff, self.gamma)
        with torch.no_grad():
            v_hat, _ = hip.l2norm_fwd(xe)                                                        # F.normalize(visual_embeddings, dim=1): per pixel row
        t_hat = ops.L2NormFn.apply(text_embeddings.view(B * K, -1))
        score = ScoreMapFn.apply(v_hat, t_hat.view(B, K, -1), B, HW, 1)
        score_map = score.view(B, H, W, K).permute(0, 3, 1, 2)
        if fpn_side is not None:
            fpn_side.join(f1, f2, f3, f4)
        maps = [self._nchw(f1, B, 4 * H, 4 * W), self._nchw(f2, B, 2 * H, 2 * W), self._nchw(f3, B, H, W), self._nchw(f4, B, H // 2, W // 2)]
        i = cfg.score_concat_index
        if concat is not None:
            hip.copy2d(score.detach(), concat[:, Cc:])
            maps[2] = self._nchw(concat, B, H, W)
        else:
            if maps[i].shape[-2:] != score_map.shape[-2:]:
                raise ValueError(f"score_concat_index={i}: map {tuple(maps[i].shape)} and score map {tuple(score_map.shape)} differ in size")
            maps[i] = torch.cat((maps[i], score_map.detach()), 1)
        return text_embeddings, maps, score_map

    def after_extract_feat(self, x):  # pragma: no cover - documented non-entry
        raise NotImplementedError("call forward(inputs): the FPN taps, the text tower and the score map are produced in one pass (the fpn3 map and the "
                                  "score map share one buffer)")


_ZEROS: dict = {}
FPN_SIDE_STREAM = __import__("os").environ.get("TVL_DENSECLIP_FPN_STREAM", "1") != "0"   # A/B switch: the FPN taps beside the context decoder


def hip_zeros(B: int, Cc: int, device) -> torch.Tensor:
    """[B, C] zeros, kept per (shape, device): the bias operand of the broadcast node."""
    key = (B, Cc, str(device))
    if key not in _ZEROS:
        _ZEROS[key] = torch.zeros((B, Cc), device=device, dtype=torch.float32)
    return _ZEROS[key]


def tokenize(class_names: Iterable[str], context_length: int, bpe_path=None, vendored_ids: bool = True) -> torch.Tensor:
    """``torch.cat([tokenize(c, context_length=...) for c in class_names])`` (denseclip.py:99-101, untils.py:173-221): [SOT, ids, EOT] zero-padded;
    a name that does not fit raises like the reference (``truncate=False``).

    ``vendored_ids`` (default): the ids of the reference's own ``SimpleTokenizer`` (untils.py:96-104), whose vocabulary lists the two specials
    right AFTER the 512 byte symbols -- SOT = 512, EOT = 513, every merge two places later than in OpenAI's order -- so that the rows of
    ``text_encoder.token_embedding`` a class name selects, and the row ``text.argmax(-1)`` pools, are the ones the reference selects
    (a quirk that is behaviour: with it EOT is NOT the largest id of a row).  False: OpenAI / HF order (specials last)."""
    from ..data.tokenizer import ClipBpeTokenizer

    tok = ClipBpeTokenizer(bpe_path)
    bos, eos = tok.bos_token_id, tok.eos_token_id
    rows = []
    for name in class_names:
        ids = tok(name)["input_ids"]
        if len(ids) > context_length:
            raise RuntimeError(f"Input {name} is too long for context length {context_length}")
        if vendored_ids:
            ids = [512 if i == bos else 513 if i == eos else (i if i < 512 else i + 2) for i in ids]
        rows.append(ids + [0] * (context_length - len(ids)))
    return torch.tensor(rows, dtype=torch.long)
