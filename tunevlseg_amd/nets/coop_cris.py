"""reference ``src/models/core_models/coop/coop_cris.py:21-242`` (+ the vendored CRIS model it drives,
``src/models/components/cris_model/{__init__,clip,layers}.py``) -- BASELINE configs[2].

Same constructor surface, attribute names (``backbone``, ``neck``, ``decoder``, ``proj``, ``context_learner``,
``additive_decoder_layer``, ``residual_ratio``) and ``state_dict`` keys as the reference net; the arithmetic is HIP
launches over NHWC pixel matrices.  The image tower and the text-independent part of the neck run without an autograd
tape (nothing trainable upstream); everything downstream of the prompts carries a data-gradient-only backward.
"""
from __future__ import annotations

from collections.abc import Mapping
from typing import Any

import torch
from torch import nn

from .. import cris_ops as C
from .. import hip, ops
from ..cris_backbone import CRISWeights
from .context_learner import CoCoOpContextLearner, CoOpContextLearner

RELU, NONE = hip.ACT_RELU, hip.ACT_NONE
LN_EPS = 1e-5


class COOPCRIS(nn.Module):
    max_length = 77  # CRIS.max_length (cris_model/__init__.py:21)

    def __init__(self, model_cfg: Mapping[str, Any], context_learner, freeze_all: bool = True, no_freeze_last_layer: bool = False,
                 use_new_last_layer: bool = False, new_last_layer_kernel_size=5, residual_ratio: float = 0.5) -> None:
        super().__init__()
        cfg = dict(model_cfg)
        weights = CRISWeights.from_spec(cfg.get("clip_pretrain"), overrides=cfg)
        if cfg.get("cris_pretrain") is not None and not isinstance(cfg.get("clip_pretrain"), (CRISWeights, Mapping)):
            sd = cfg["cris_pretrain"] if isinstance(cfg["cris_pretrain"], Mapping) else torch.load(cfg["cris_pretrain"], map_location="cpu", weights_only=False)
            # strict, as the reference loads it (cris_model/__init__.py:66-70); BatchNorm's num_batches_tracked counters are the only
            # entries this (eval-only, BN-folded) parameter tree does not hold
            sd = {k: v for k, v in sd.items() if not k.endswith("num_batches_tracked")}
            weights.load_state_dict(sd, strict=True)
        self._weights = [weights]  # not a submodule: its four subtrees are registered under the reference's names below
        self.config = weights.config
        self.img_size = int(cfg.get("img_size", weights.config.img_size))
        self.word_dim = weights.config.word_dim
        self.backbone, self.neck, self.decoder, self.proj = weights.backbone, weights.neck, weights.decoder, weights.proj
        self.backbone.requires_grad_(not cfg.get("freeze_encoder", True))
        self.assign_model_learnability(freeze_all, no_freeze_last_layer, use_new_last_layer, new_last_layer_kernel_size, residual_ratio)
        self.context_learner = context_learner(
            max_network_depth=self.config.transformer_layers,
            visual_dim=self.config.embed_dim,
            context_dim=self.word_dim,  # as the reference (coop_cris.py:45): only a context_initializer makes the width right
            embedding_layer=self.backbone.token_embedding,
        )
        if self.context_learner.context_vectors.shape[-1] != self.config.transformer_width:
            raise ValueError(
                f"context vectors have width {self.context_learner.context_vectors.shape[-1]} but the CLIP text transformer is "
                f"{self.config.transformer_width} wide: COOPCRIS passes context_dim=word_dim ({self.word_dim}); use a "
                "context_initializer (reference configs/model/coop/cris.yaml:25), which takes its width from the embedding")
        self._is_cocoop = isinstance(self.context_learner, CoCoOpContextLearner)

    # ------------------------------------------------------------------ learnability (coop_cris.py:58-94)
    def assign_model_learnability(self, freeze_all, no_freeze_last_layer, use_new_last_layer, new_last_layer_kernel_size, residual_ratio):
        if not freeze_all:
            raise NotImplementedError(
                "only the prompt-tuning path is implemented for CRIS: frozen model (freeze_all=True) with the prompts and the "
                "optional new last layer / unfrozen projector head trainable (SURVEY.md §8)")
        self.eval()
        self.requires_grad_(False)
        self.additive_decoder_layer = None
        self._train_proj_head = False
        if use_new_last_layer:
            k = new_last_layer_kernel_size
            if not isinstance(k, int):
                if k[0] != k[1]:
                    raise NotImplementedError("square new_last_layer_kernel_size only")
                k = k[0]
            # parameter holders under the reference's keys (additive_decoder_layer.{0,2}.*); the HIP path does the math
            self.additive_decoder_layer = nn.Sequential(
                nn.Conv2d(self.proj.in_dim * 2, 64, 1, bias=False),
                nn.Upsample(size=self.img_size, mode="bilinear"),
                nn.Conv2d(64, 1, kernel_size=k, padding="same", padding_mode="replicate"),
            )
            self.residual_ratio = nn.Parameter(torch.tensor(residual_ratio))
        elif no_freeze_last_layer:
            # the text alignment layer and the last 1x1 conv of the visual branch train (coop_cris.py:88-94)
            self.proj.txt.requires_grad_(True)
            self.proj.vis._modules["4"].requires_grad_(True)
            self._train_proj_head = True

    @property
    def weights(self) -> CRISWeights:
        return self._weights[0]

    def _apply(self, fn, *a, **k):
        self.weights._prep = None
        self.weights._const = {}
        return super()._apply(fn, *a, **k)

    # ------------------------------------------------------------------ masks (cris_model/__init__.py:79-86, coop_cris.py:101-113)
    def get_pad_mask(self, input_ids: torch.Tensor, attention_mask: torch.Tensor | None) -> torch.Tensor:
        pad_mask = ~(attention_mask.bool()) if attention_mask is not None else input_ids == 0
        return self.context_learner.update_pad_mask_for_context(pad_mask=pad_mask, max_length=self.max_length)

    # ------------------------------------------------------------------ CLIP-RN50 (clip.py:185-274), no autograd tape
    def encode_image(self, image: torch.Tensor):
        """-> ((C3, H3, W3), (C4, H4, W4), (C5, H5, W5)) pixel matrices [B*H*W, C]."""
        prep = self.weights.prepared()
        B, _, H, W = image.shape
        with torch.no_grad():
            image = image.contiguous()
            s1, s2, s3 = prep["stem"]
            cols = hip.im2col3x3_nchw(image, 2)
            H, W = (H - 1) // 2 + 1, (W - 1) // 2 + 1
            x = torch.empty((cols.shape[0], s1.cout), device=image.device, dtype=torch.float32)
            hip.gemm(hip.NT, cols.shape[0], s1.cout, cols.shape[1], cols, cols.shape[1], s1.Wm, cols.shape[1], x, s1.cout, bias=s1.b, act=RELU)
            del cols
            x = C.fconv3(x, s2, B, H, W, RELU)
            x = C.fconv3(x, s3, B, H, W, RELU)
            x = hip.avgpool_fwd(x, B, H, W, 2)
            H, W = H // 2, W // 2
            feats = []
            for blk in prep["blocks"]:
                o = C.flinear(x, blk["c1"], RELU)
                o = C.fconv3(o, blk["c2"], B, H, W, RELU)
                idt = x
                if blk["stride"] > 1:
                    o = hip.avgpool_fwd(o, B, H, W, blk["stride"])
                    idt = hip.avgpool_fwd(x, B, H, W, blk["stride"])
                    H, W = H // blk["stride"], W // blk["stride"]
                if blk["down"] is not None:
                    idt = C.flinear(idt, blk["down"])
                x = C.flinear(o, blk["c3"], RELU | hip.ACT_POST_RESIDUAL, residual=idt)  # relu(bn3(conv3) + identity)
                if blk["stage_end"]:
                    feats.append((x, H, W))
            # attention pool that keeps the map (clip.py:148-182)
            ap = prep["attnpool"]
            x4, H4, W4 = feats[3]
            E = x4.shape[1]
            res = C.flinear(x4, ap["connect"])
            xp = hip.bias_act(x4.view(B, H4 * W4 * E), self.weights.attnpool_pos(H4, W4).view(-1), NONE).view(B * H4 * W4, E)
            qkv = C.flinear(xp, ap["qkv"])
            heads = self.config.vision_heads
            o, _ = hip.attn_fwd_packed(qkv, B, H4 * W4, heads, E // heads, (E // heads) ** -0.5, want_lse=False)
            x5 = C.flinear(o, ap["c_proj"], RELU | hip.ACT_POST_RESIDUAL, residual=res)
        return feats[1], feats[2], (x5, H4, W4)

    # ------------------------------------------------------------------ CLIP text tower with prompts (coop_cris.py:115-183)
    def encode_text(self, text: torch.Tensor, image_features: torch.Tensor | None = None, key_padding_mask: torch.Tensor | None = None):
        prep = self.weights.prepared()
        cfg = self.config
        learner = self.context_learner
        bb = self.backbone
        dev = bb.positional_embedding.device
        text = text.view(-1, text.shape[-1]).to(dev)
        B, L = text.shape
        n, depth = learner.num_context, learner.prompt_depth
        tmap_list = learner.splice_map(L, self.max_length)
        T = len(tmap_list)
        tmap = hip.const_i32(tmap_list, dev)   # cached: a per-step torch.tensor(list, device=...) is a pageable copy the host blocks on
        ctx0 = learner.get_textual_context(image_features=image_features, index=0)
        x = ops.TextAssembleFn.apply(text.contiguous(), tmap, bb.token_embedding.weight.detach(), ctx0, bb.positional_embedding.detach(), n)
        key_mask = None
        if key_padding_mask is not None:
            key_mask = (~key_padding_mask.to(dev).bool()).to(torch.int32).contiguous()
            if key_mask.shape[1] != T:
                raise ValueError(f"pad mask covers {key_mask.shape[1]} tokens but the prompted sequence has {T}")
        spec = ops.AttnSpec(cfg.transformer_heads, hip.ACT_QUICK_GELU, LN_EPS, causal=True, key_mask=key_mask)
        for idx in range(cfg.transformer_layers):
            x = ops.EncoderLayerFn.apply(x, prep["text_layers"][idx], spec)
            if idx < depth:  # 0-based: block 0 re-writes ctx[0] (coop_cris.py:128-143)
                x = learner.mutate_text_hidden_states(x, index=idx, image_features=image_features)
        x = ops.layer_norm(x, bb.ln_final.weight.detach(), bb.ln_final.bias.detach(), LN_EPS)
        pool = torch.clamp(text.argmax(dim=-1) + n, max=self.max_length - 1).to(torch.int32)
        pooled = ops.GatherRowsFn.apply(x, pool)
        state = C.flinear_g(pooled, prep["text_projection"])
        return x, state

    # ------------------------------------------------------------------ FPN neck (layers.py:412-445)
    def neck_visual(self, vis, B: int):
        """The text-independent branches of the neck (``f1/f2/f3_v_proj``, layers.py:414-427 before the text gate): no tape, and -- being
        independent of the text tower -- enqueued while that tower runs on the side stream (forward)."""
        nk = self.weights.prepared()["neck"]
        (v3, H3, W3), (v4, H4, W4), (v5, H5, W5) = vis
        with torch.no_grad():
            f5pre = C.flinear(v5, nk["f1_v_proj"], RELU)
            Cn = f5pre.shape[1]
            scale_b = nk["norm_scale"].expand(B, Cn).contiguous()
            f5pre = hip.film_fwd(f5pre.view(B, H5 * W5, Cn), scale_b, torch.zeros_like(scale_b))  # BN scale of norm_layer
            f4a = C.fconv3(v4, nk["f2_v_proj"], B, H4, W4, RELU)
            f3a = hip.avgpool_fwd(C.fconv3(v3, nk["f3_v_proj"], B, H3, W3, RELU), B, H3, W3, 2)
        return f5pre, f4a, f3a

    def neck_forward(self, vis, state: torch.Tensor, visual=None) -> tuple[torch.Tensor, int, int]:
        prep = self.weights.prepared()
        nk = prep["neck"]
        (v3, H3, W3), (v4, H4, W4), (v5, H5, W5) = vis
        B = state.shape[0]
        f5pre, f4a, f3a = visual if visual is not None else self.neck_visual(vis, B)
        Cn = f5pre.shape[-1]
        s = C.flinear_g(state, nk["txt_proj"], RELU)  # [B, C5]
        f5 = C.ReluFn.apply(ops.FilmFn.apply(f5pre, s, nk["norm_shift"].expand(B, Cn).contiguous())).view(B * H5 * W5, Cn)
        f5u = C.BilinearUpFn.apply(f5, B, H5, W5, 2)
        f4 = C.flinear_g(C.CatColsFn.apply(f4a, f5u), nk["f2_cat"], RELU)
        f3 = C.flinear_g(C.CatColsFn.apply(f3a, f4), nk["f3_cat"], RELU)
        fq5 = C.BilinearUpFn.apply(C.fconv3_g(f5, nk["f4_proj5"], B, H5, W5), B, H5, W5, 2)
        fq4 = C.fconv3_g(f4, nk["f4_proj4"], B, H4, W4)
        fq3 = C.fconv3_g(f3, nk["f4_proj3"], B, H4, W4)
        fq = C.flinear_g(C.CatColsFn.apply(fq3, fq4, fq5), nk["aggr"], RELU)
        fq = C.fconv3_g(C.CatColsFn.apply(fq, self.weights.coords(B, H4, W4)), nk["coord0"], B, H4, W4)
        fq = C.fconv3_g(fq, nk["coord1"], B, H4, W4)
        return fq, H4, W4

    # ------------------------------------------------------------------ cross-attention decoder (layers.py:238-356)
    def decoder_forward(self, fq: torch.Tensor, H: int, W: int, words: torch.Tensor, pad_mask: torch.Tensor) -> torch.Tensor:
        prep = self.weights.prepared()
        cfg = self.config
        B, L, Dt = words.shape
        D, heads = cfg.vis_dim, cfg.num_head
        T = H * W
        dh = D // heads
        vis_pos = self.weights.pos2d(D, H, W)
        txt_pos = self.weights.pos1d(Dt, L)
        key_mask = (~pad_mask.to(words.device).bool()).to(torch.int32).contiguous()
        vis = fq.view(B, T, D)
        kin = C.BcastAddFn.apply(words, txt_pos, B)  # key input = words + 1-D code, shared by the layers
        for lw in prep["decoder_layers"]:
            v2 = ops.layer_norm(vis, *lw["norm1"], LN_EPS)
            o = C.self_attn_block(C.BcastAddFn.apply(v2, vis_pos, B), v2, lw["sa_qk"], lw["sa_v"], lw["sa_o"], B, T, heads, dh).view(B, T, D)
            vis = ops.add(vis, ops.layer_norm(o, *lw["self_attn_norm"], LN_EPS))
            v2 = ops.layer_norm(vis, *lw["norm2"], LN_EPS)
            q = C.flinear_g(C.BcastAddFn.apply(v2, vis_pos, B), lw["ca_q"]).view(B * T, D)
            k = C.flinear_g(kin, lw["ca_k"]).view(B * L, D)
            vv = C.flinear_g(words, lw["ca_v"]).view(B * L, D)
            o = C.flinear_g(C.CrossAttnFn.apply(q, k, vv, key_mask, B, T, L, heads, dh), lw["ca_o"]).view(B, T, D)
            vis = ops.add(vis, ops.layer_norm(o, *lw["cross_attn_norm"], LN_EPS))
            if C.FFNBlockFn.takes(B * T, D, lw["ffn0"].W.shape[0], lw["ffn0"], lw["ffn4"]):
                vis = C.FFNBlockFn.apply(vis, *lw["norm3"], lw["ffn0"], *lw["ffn_norm"], lw["ffn4"], LN_EPS)
            else:
                v2 = ops.layer_norm(vis, *lw["norm3"], LN_EPS)
                v2 = ops.layer_norm(C.flinear_g(v2, lw["ffn0"], RELU), *lw["ffn_norm"], LN_EPS)
                vis = ops.add(vis, C.flinear_g(v2, lw["ffn4"]))
        return ops.layer_norm(vis, *prep["decoder_norm"], LN_EPS).view(B * T, D)

    # ------------------------------------------------------------------ projector (layers.py:96-119)
    def proj_forward(self, fq: torch.Tensor, H: int, W: int, state: torch.Tensor) -> torch.Tensor:
        pj = self.weights.prepared()["proj"]
        B = state.shape[0]
        x = C.up_conv3_g(fq, pj["vis1"], B, H, W, 2)            # Upsample(x2) -> conv 3x3 + BN + ReLU
        x = C.up_conv3_g(x, pj["vis3"], B, 2 * H, 2 * W, 2)
        if self._train_proj_head:   # trainable copies live in the parameter tree, not in the prepared (frozen) matrices
            v4 = self.proj.vis._modules["4"]
            x = ops.linear(x, v4.weight.view(v4.weight.shape[0], -1), v4.bias)
            word = ops.linear(state, self.proj.txt.weight, self.proj.txt.bias)
        else:
            x = C.flinear_g(x, pj["vis4"])
            word = C.flinear_g(state, pj["txt"])
        return C.DynConvFn.apply(x, word, B, 4 * H, 4 * W)  # [B, 4H, 4W]

    def encode_image_features(self, image_input: torch.Tensor):
        vis = self.encode_image(image_input)
        image_features = None
        if self._is_cocoop:  # C5.mean((2, 3)) (coop_cris.py:96-99) == one avg-pool window covering the whole map
            x5, H5, W5 = vis[2]
            if H5 != W5:
                raise NotImplementedError("square inputs only")
            with torch.no_grad():
                image_features = hip.avgpool_fwd(x5, image_input.shape[0], H5, W5, H5)
        return vis, image_features

    def get_unimodal_outputs(self, image_input: torch.Tensor, input_ids: torch.Tensor, *args, **kwargs):
        vis, image_features = self.encode_image_features(image_input)
        words, state = self.encode_text(input_ids, image_features, *args, **kwargs)
        return vis, words, state

    def forward(self, text_input: Mapping[str, torch.Tensor], image_input: torch.Tensor):
        input_ids = text_input["input_ids"]
        attention_mask = text_input.get("attention_mask")
        B = image_input.shape[0]
        if image_input.shape[-1] != self.img_size or image_input.shape[-2] != self.img_size:
            raise ValueError(f"COOPCRIS was built for img_size={self.img_size}, got {tuple(image_input.shape[-2:])}")
        pad_mask = self.get_pad_mask(input_ids.to(image_input.device), None if attention_mask is None else attention_mask.to(image_input.device))
        # (Measured in round 4 and dropped: the text tower on a side stream beside the neck's text-independent convs -- 35.3-35.5 ms against
        # 35.1 ms on one box: those convs fill the chip, the tower's latency-bound launches only get in their way.)
        vis, words, state = self.get_unimodal_outputs(image_input, input_ids, key_padding_mask=pad_mask)
        fq, H, W = self.neck_forward(vis, state)
        fq = self.decoder_forward(fq, H, W, words, pad_mask)
        pred = self.proj_forward(fq, H, W, state)
        logits = C.BicubicFn.apply(pred, self.img_size, self.img_size)
        if self.additive_decoder_layer is not None:
            if self.img_size != 16 * H:
                raise NotImplementedError("the new last layer assumes the stride-16 C4 map (img_size == 16 * H)")
            w1 = self.additive_decoder_layer[0].weight
            z = ops.linear(fq, w1.view(w1.shape[0], -1))
            conv = self.additive_decoder_layer[2]
            extra = C.UpconvFn.apply(z, conv.weight, conv.bias, B, H, 16)
            logits = C.MixFn.apply(logits, extra, self.residual_ratio)
        return logits.view(B, 1, self.img_size, self.img_size)
