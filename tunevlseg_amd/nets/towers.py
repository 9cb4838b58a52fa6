"""The three sub-networks of CLIPSeg, orchestrated over HIP kernels.

Each function restates the *orchestration* the reference keeps in its nets (layer loops, prompt
splicing, pooling, decoder loop) and issues HIP launches for all arithmetic:

* :func:`vision_tower`  -- reference ``vpt_clipseg.py:36-235`` / ``base_multimodal_clipseg.py:310-484`` (prompt path)
                           and the HF vision model as used by ``coop_clipseg.py:341-371`` (full path)
* :func:`text_tower`    -- reference ``coop_clipseg.py:40-339`` / ``base_multimodal_clipseg.py:24-300``;
                           without a learner: HF ``get_text_features`` (modeling_clipseg.py:594-686)
* :func:`decoder_tokens` / :func:`seg_head` -- reference ``base_clipseg.py:82-172``, ``vpt_clipseg.py:237-319``,
                           HF ``CLIPSegDecoder.forward`` (modeling_clipseg.py:549-586)
"""
from __future__ import annotations

import math

import torch

from .. import hip, ops
from ..backbone import CLIPSegBackbone

_ACT = hip.ACT_IDS

import os  # noqa: E402

TEXT_SIDE_STREAM = os.environ.get("TVL_TEXT_STREAM", "1") != "0"   # VPT: the gradient-free text tower on a second HIP stream
_SIDE: dict = {}


def side_stream(device) -> torch.cuda.Stream:
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=key)
    return _SIDE[key]


class SideStream:
    """``with SideStream(device) as s: y = f(...)`` runs ``f`` on the second HIP stream of the device (after everything already enqueued
    on the current one); ``s.join(y)`` makes the current stream wait for it and tells the allocator that ``y`` lives on.  The text tower
    (M = B * L rows: launch-latency-bound kernels on a fraction of the CUs) and the vision tower (chip-filling kernels with bubbles at
    every round's tail) share nothing until the decoder; autograd replays each node on its forward stream, so the backward overlaps too.
    The fork point is where the object is CREATED: the side stream waits for what the current stream held then, not for what is enqueued
    between creation and ``with`` -- so the caller may enqueue the vision tower first and the text tower after it (the big kernels start at
    once when the host is not ahead of the GPU -- first step after a synchronisation, profiled runs, a busy host; and autograd, which
    replays nodes in reverse creation order, then enqueues the text tower's backward BEFORE the vision tower's).  The block may be
    entered more than once (MaPLe: prompt projections before the vision tower, text tower after it).
    No-op on CPU tensors or with TVL_TEXT_STREAM=0."""

    def __init__(self, device):
        self.on = TEXT_SIDE_STREAM and torch.device(device).type == "cuda"
        self.device = device
        self._ctx = None
        if self.on:
            self.cur = torch.cuda.current_stream(self.device)
            self.side = side_stream(self.device)
            self.fork = self.cur.record_event()

    def __enter__(self):
        if self.on:
            self.side.wait_event(self.fork)
            self._ctx = torch.cuda.stream(self.side)
            self._ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self._ctx is not None:
            self._ctx.__exit__(*exc)
            self._ctx = None
        return False

    def join(self, *tensors):
        if self.on:
            self.cur.wait_stream(self.side)
            for t in tensors:
                if t is not None:
                    t.record_stream(self.cur)

    def mark(self):
        """An event after everything enqueued on the side stream so far (call inside the ``with`` block); None when off."""
        return self.side.record_event() if self.on else None


def patch_embeddings(model: CLIPSegBackbone, pixel_values: torch.Tensor) -> torch.Tensor:
    """16x16/s16 patch conv as im2col + GEMM (HF:195-197).  Frozen, image carries no grad -> no autograd node."""
    prep = model.prepared()
    B, _, H, W = pixel_values.shape
    ps = prep["patch_size"]
    if H % ps or W % ps:
        raise ValueError(f"image size {H}x{W} is not a multiple of the patch size {ps}")
    with torch.no_grad():
        cols = hip.im2col_patch(ops._c(pixel_values.to(torch.float32)), ps)
        patch = hip.linear_fwd(cols, prep["patch_w"])
    return patch.view(B, (H // ps) * (W // ps), -1)


def vision_tower(model: CLIPSegBackbone, pixel_values: torch.Tensor, learner=None, full: bool = False, visual_contexts=None, ready=None):
    """Returns ``(activations at extract_layers, pooled)``; ``pooled`` = visual_projection(post_layernorm(CLS)) when ``full``.
    ``visual_contexts``: the learner's visual prompts of every depth, computed ahead (they depend on parameters only) -- e.g. on the side
    stream, in which case ``ready`` is the event after them (or one event per depth, awaited in front of that depth's row overwrite): the tower then
    launches nothing between its layers but the row overwrite."""
    cfg = model.config
    v = cfg.vision_config
    prep = model.prepared()
    vm = model.clip.vision_model
    B, _, H, W = pixel_values.shape
    patch = patch_embeddings(model, pixel_values)
    pos = model.vision_pos(H, W)
    prompts = None
    depth = 1
    if learner is not None and not full:
        depth = learner.prompt_depth
        if visual_contexts is not None:
            if ready is not None:
                ready = list(ready) if isinstance(ready, (list, tuple)) else [ready]   # one event per depth, or one behind all of them
                torch.cuda.current_stream(pixel_values.device).wait_event(ready[0])
                for c in visual_contexts:
                    c.record_stream(torch.cuda.current_stream(pixel_values.device))
            prompts = visual_contexts[0]
        else:
            prompts = learner.get_visual_context(index=0)
    x = ops.VisionAssembleFn.apply(patch, vm.embeddings.class_embedding.detach(), pos, prompts)
    # NB concat happens BEFORE pre_layrnorm (vpt_clipseg.py:178-181)
    x = ops.layer_norm(x, vm.pre_layrnorm.weight.detach(), vm.pre_layrnorm.bias.detach(), v.layer_norm_eps)
    spec = ops.AttnSpec(v.num_attention_heads, _ACT[v.hidden_act], v.layer_norm_eps)
    max_idx = max(model.extract_layers)
    states = [x]
    T_all = x.shape[1]
    for idx in range(1, v.num_hidden_layers + 1):
        # the first layer's input is LayerNorm(frozen patches | CLS | positions, prompts): only the prompt rows (the last n) carry a gradient
        first_rows = (T_all - learner.num_context, learner.num_context) if (idx == 1 and prompts is not None) else None
        x = ops.encoder_layer(x, prep["vision_layers"][idx - 1], spec, first_rows)
        if prompts is not None and idx < depth:
            if visual_contexts is not None:   # learner.mutate_image_hidden_states with the context computed ahead
                if ready is not None and idx < len(ready):
                    torch.cuda.current_stream(pixel_values.device).wait_event(ready[idx])
                x = ops.RowsOverwriteFn.apply(x, visual_contexts[idx], x.shape[1] - learner.num_context)
            else:
                x = learner.mutate_image_hidden_states(x, index=idx)
        states.append(x)
        if not full and idx > max_idx:  # "No need to run the vision transformer for more layers" (vpt_clipseg.py:129-131)
            break
    acts = tuple(states[i + 1] for i in model.extract_layers)
    pooled = None
    if full:
        cls_idx = torch.zeros(B, dtype=torch.int32, device=x.device)
        cls = ops.GatherRowsFn.apply(x, cls_idx)
        cls = ops.layer_norm(cls, vm.post_layernorm.weight.detach(), vm.post_layernorm.bias.detach(), v.layer_norm_eps)
        pooled = ops.linear(cls, model.clip.visual_projection.weight.detach())
    return acts, pooled


def text_tower(model: CLIPSegBackbone, input_ids: torch.Tensor, attention_mask: torch.Tensor | None, learner=None,
               image_features: torch.Tensor | None = None) -> torch.Tensor:
    """``text_projection(pooled EOS state)`` -> [B, projection_dim]."""
    cfg = model.config
    t = cfg.text_config
    prep = model.prepared()
    tm = model.clip.text_model
    dev = tm.embeddings.token_embedding.weight.device
    input_ids = input_ids.view(-1, input_ids.shape[-1]).to(dev)
    B, L = input_ids.shape
    n, depth, ctx0 = 0, 1, None
    if learner is not None:
        n, depth = learner.num_context, learner.prompt_depth
        tmap_list = learner.splice_map(L, t.max_position_embeddings)
        ctx0 = learner.get_textual_context(image_features=image_features, index=0)
    else:
        tmap_list = list(range(L))
    T = len(tmap_list)
    if T > t.max_position_embeddings:
        raise ValueError(f"Sequence length must be less than max_position_embeddings (got {T} > {t.max_position_embeddings})")
    tmap = hip.const_i32(tmap_list, dev)
    x = ops.TextAssembleFn.apply(input_ids.contiguous(), tmap, tm.embeddings.token_embedding.weight.detach(), ctx0,
                                 tm.embeddings.position_embedding.weight.detach(), n)
    key_mask = None
    if attention_mask is not None:
        am = attention_mask.to(dev)
        if learner is not None:
            am = learner.update_attention_mask_for_context(am, t.max_position_embeddings)
        key_mask = am.to(torch.int32).contiguous()
    spec = ops.AttnSpec(t.num_attention_heads, _ACT[t.hidden_act], t.layer_norm_eps, causal=True, key_mask=key_mask)
    for idx in range(1, t.num_hidden_layers + 1):
        x = ops.encoder_layer(x, prep["text_layers"][idx - 1], spec)
        if n and idx < depth:
            x = learner.mutate_text_hidden_states(x, index=idx, image_features=image_features)
    x = ops.layer_norm(x, tm.final_layer_norm.weight.detach(), tm.final_layer_norm.bias.detach(), t.layer_norm_eps)
    ids32 = input_ids.to(torch.int)
    if t.eos_token_id == 2:
        pre = ids32  # legacy checkpoints: EOT is the highest id in each row (HF:630-642)
    else:
        pre = (ids32 == t.eos_token_id).int()
    pool = torch.clamp(pre.argmax(dim=-1) + n, max=t.max_position_embeddings - 1).to(torch.int32)
    pooled = ops.GatherRowsFn.apply(x, pool)
    return ops.linear(pooled, model.clip.text_projection.weight.detach())


def decoder_tokens(model: CLIPSegBackbone, activations, conditional_embeddings: torch.Tensor) -> torch.Tensor:
    """3 x [reduce + running sum; FiLM at conditional_layer; post-LN layer] -> tokens [B, T, reduce_dim]."""
    cfg = model.config
    dec = model.decoder
    prep = model.prepared()
    spec = ops.AttnSpec(cfg.decoder_num_attention_heads, hip.ACT_RELU, cfg.vision_config.layer_norm_eps)
    out = None
    for i, act in enumerate(activations[::-1]):
        rw, rb, rwt = prep["reduces"][i]
        out = ops.linear(act, rw, rb, hip.ACT_NONE, out, rwt)
        if i == dec.conditional_layer:
            mul = ops.linear(conditional_embeddings, dec.film_mul.weight.detach(), dec.film_mul.bias.detach())
            add = ops.linear(conditional_embeddings, dec.film_add.weight.detach(), dec.film_add.bias.detach())
            out = ops.FilmFn.apply(out, mul, add)
        out = ops.DecoderLayerFn.apply(out, prep["decoder_layers"][i], spec)
    return out


def seg_head(model: CLIPSegBackbone, tokens: torch.Tensor, n_strip: int, additive_layer=None, residual_ratio=None, mix: int = 0):
    """Drop CLS (+ the trailing prompt tokens), ConvTranspose2d, optional new last layer -> logits [B, H, W]."""
    B, T, _ = tokens.shape
    N = T - 1 - n_strip
    G = math.isqrt(N)
    if G * G != N:
        raise ValueError(f"{N} patch tokens do not form a square grid")
    tc = model.decoder.transposed_convolution
    ps = model.config.vision_config.patch_size
    if additive_layer is None:
        mix = 0
    conv = additive_layer[1] if additive_layer is not None else None
    return ops.SegHeadFn.apply(tokens, tc.weight, tc.bias, conv.weight if conv is not None else None,
                               conv.bias if conv is not None else None, residual_ratio, mix, G, ps)
