"""CPU ORACLE -- test infrastructure only.  Never imported by ``tunevlseg_amd``.

A plain fp32 PyTorch-on-CPU restatement of the reference's prompt-tuned CLIPSeg
forward (autograd supplies the backward), written functionally over an
HF-named state dict.  It follows, line by line:

* reference ``src/models/core_models/coop/vpt_clipseg.py:36-395``      (VPT)
* reference ``src/models/core_models/coop/coop_clipseg.py:40-484``     (CoOp / CoCoOp)
* reference ``src/models/core_models/coop/base_multimodal_clipseg.py`` (MaPLe)
* reference ``src/models/core_models/coop/base_clipseg.py:82-199``     (decoder, boundary)
* reference ``context_learner/*.py``                                   (learners)
* HF ``transformers/models/clipseg/modeling_clipseg.py`` (un-vendored dependency,
  ``requirements.txt:26``): embeddings 126-206, attention 247-339, encoder layer
  341-371, decoder layer 374-410, decoder 501-586, text pooling 630-655.

Pinning: ``tests/golden/*.npz`` hold outputs of the *reference classes
themselves* (run in the build container through the API-drift shim of
``tests/golden/make_goldens.py``); ``tests/test_oracle_golden.py`` checks this
file against them.  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import it.
"""
from __future__ import annotations

import math
from typing import Any, Mapping, Sequence

import torch
import torch.nn.functional as F

SD = Mapping[str, torch.Tensor]


# ----------------------------------------------------------------------------
# primitives (HF modeling_clipseg.py)
# ----------------------------------------------------------------------------
def quick_gelu(x: torch.Tensor) -> torch.Tensor:
    return x * torch.sigmoid(1.702 * x)


def _act(name: str):
    return {"quick_gelu": quick_gelu, "relu": F.relu, "gelu": F.gelu}[name]


def layer_norm(sd: SD, p: str, x: torch.Tensor, eps: float) -> torch.Tensor:
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], eps)


def linear(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def attention(sd: SD, p: str, x: torch.Tensor, heads: int, add_mask: torch.Tensor | None) -> torch.Tensor:
    """HF CLIPSegAttention + eager_attention_forward (HF:247-339)."""
    B, T, D = x.shape
    dh = D // heads
    q = linear(sd, p + ".q_proj", x).view(B, T, heads, dh).transpose(1, 2)
    k = linear(sd, p + ".k_proj", x).view(B, T, heads, dh).transpose(1, 2)
    v = linear(sd, p + ".v_proj", x).view(B, T, heads, dh).transpose(1, 2)
    w = torch.matmul(q, k.transpose(-1, -2)) * dh**-0.5
    if add_mask is not None:
        w = w + add_mask
    w = F.softmax(w, dim=-1, dtype=torch.float32)
    o = torch.matmul(w, v).transpose(1, 2).reshape(B, T, D)
    return linear(sd, p + ".out_proj", o)


def encoder_layer(sd: SD, p: str, x: torch.Tensor, heads: int, act: str, eps: float, add_mask=None) -> torch.Tensor:
    """Pre-LN layer (HF:341-371)."""
    x = x + attention(sd, p + ".self_attn", layer_norm(sd, p + ".layer_norm1", x, eps), heads, add_mask)
    h = layer_norm(sd, p + ".layer_norm2", x, eps)
    h = linear(sd, p + ".mlp.fc2", _act(act)(linear(sd, p + ".mlp.fc1", h)))
    return x + h


def decoder_layer(sd: SD, p: str, x: torch.Tensor, heads: int, eps: float) -> torch.Tensor:
    """Post-LN layer, ReLU MLP (HF:374-410)."""
    x = layer_norm(sd, p + ".layer_norm1", x + attention(sd, p + ".self_attn", x, heads, None), eps)
    h = linear(sd, p + ".mlp.fc2", F.relu(linear(sd, p + ".mlp.fc1", x)))
    return layer_norm(sd, p + ".layer_norm2", x + h, eps)


def vision_position_embedding(sd: SD, cfg, height: int, width: int) -> torch.Tensor:
    """HF interpolate_pos_encoding (HF:149-188): bicubic, align_corners=False."""
    pos = sd["clip.vision_model.embeddings.position_embedding.weight"]
    ps = cfg.vision_config.patch_size
    nh, nw = height // ps, width // ps
    n_pos = pos.shape[0] - 1
    if nh * nw == n_pos and height == width:
        return pos.unsqueeze(0)
    side = int(n_pos**0.5)
    dim = pos.shape[-1]
    patch = pos[1:].reshape(1, side, side, dim).permute(0, 3, 1, 2)
    patch = F.interpolate(patch, size=(nh, nw), mode="bicubic", align_corners=False)
    patch = patch.permute(0, 2, 3, 1).reshape(1, -1, dim)
    return torch.cat((pos[:1].unsqueeze(0), patch), dim=1)


def vision_embeddings(sd: SD, cfg, pixel_values: torch.Tensor) -> torch.Tensor:
    """HF CLIPSegVisionEmbeddings.forward (HF:190-206)."""
    B, _, H, W = pixel_values.shape
    ps = cfg.vision_config.patch_size
    w = sd["clip.vision_model.embeddings.patch_embedding.weight"]
    patch = F.conv2d(pixel_values, w, stride=ps).flatten(2).transpose(1, 2)
    cls = sd["clip.vision_model.embeddings.class_embedding"].expand(B, 1, -1)
    return torch.cat((cls, patch), dim=1) + vision_position_embedding(sd, cfg, H, W)


def text_additive_mask(attention_mask: torch.Tensor | None, B: int, T: int, dtype=torch.float32) -> torch.Tensor:
    """Causal + key-padding additive mask, clamped at finfo.min.

    The reference adds a causal 4-D mask and a padding 4-D mask, both filled
    with finfo.min (``coop_clipseg.py:229-246``); HF sums them and the fp32 sum
    saturates to -inf for doubly-masked entries, whose softmax weight is 0 either
    way.  Every query row keeps key 0 (BOS) visible, so no row is fully masked.
    """
    neg = torch.finfo(dtype).min
    allowed = torch.ones(T, T, dtype=torch.bool).tril().expand(B, 1, T, T)
    if attention_mask is not None:
        allowed = allowed & attention_mask.bool()[:, None, None, :]
    return torch.where(allowed, torch.zeros((), dtype=dtype), torch.full((), neg, dtype=dtype))


def eos_pool_index(cfg, input_ids: torch.Tensor) -> torch.Tensor:
    """HF:630-655 / coop_clipseg.py:261-283 -- argmax(ids) if eos_token_id == 2 else first eos."""
    ids = input_ids.to(torch.int)
    if cfg.text_config.eos_token_id == 2:
        return ids.argmax(dim=-1)
    return (ids == cfg.text_config.eos_token_id).int().argmax(dim=-1)


# ----------------------------------------------------------------------------
# learner maths (reference context_learner/*.py) over a plain description:
#   learner = {"kind": "vpt"|"coop"|"cocoop"|"maple", "ctx": Tensor[depth,n,dim],
#              "proj": [ per-depth list of ops ], "norm_image_features": bool}
#   op = ("linear", W, b|None) | ("relu",) | ("layernorm", w, b|None, eps)
# ----------------------------------------------------------------------------
def run_projection(ops: Sequence[tuple], x: torch.Tensor) -> torch.Tensor:
    for op in ops:
        if op[0] == "linear":
            x = F.linear(x, op[1], op[2])
        elif op[0] == "relu":
            x = F.relu(x)
        elif op[0] == "layernorm":
            x = F.layer_norm(x, (x.shape[-1],), op[1], op[2], op[3] if len(op) > 3 else 1e-5)
        else:  # pragma: no cover
            raise ValueError(op[0])
    return x


def transformer_encoder_layer(tp: Mapping[str, Any], x: torch.Tensor) -> torch.Tensor:
    """torch.nn.TransformerEncoderLayer (batch_first=False, eval/dropout-free), restated for x [S, N, E].

    tp: in_proj_weight/bias, out_proj.weight/bias, linear1/2.weight/bias, norm1/2.weight/bias, nhead, norm_first, eps.
    """
    S, N, E = x.shape
    H = tp["nhead"]
    dh = E // H

    def sa(t):
        qkv = F.linear(t, tp["self_attn.in_proj_weight"], tp["self_attn.in_proj_bias"])
        q, k, v = qkv.chunk(3, dim=-1)
        q = q.reshape(S, N * H, dh).transpose(0, 1)
        k = k.reshape(S, N * H, dh).transpose(0, 1)
        v = v.reshape(S, N * H, dh).transpose(0, 1)
        w = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(dh), dim=-1)
        o = (w @ v).transpose(0, 1).reshape(S, N, E)
        return F.linear(o, tp["self_attn.out_proj.weight"], tp["self_attn.out_proj.bias"])

    def ff(t):
        return F.linear(F.relu(F.linear(t, tp["linear1.weight"], tp["linear1.bias"])), tp["linear2.weight"], tp["linear2.bias"])

    n1 = lambda t: F.layer_norm(t, (E,), tp["norm1.weight"], tp["norm1.bias"], tp.get("eps", 1e-5))  # noqa: E731
    n2 = lambda t: F.layer_norm(t, (E,), tp["norm2.weight"], tp["norm2.bias"], tp.get("eps", 1e-5))  # noqa: E731
    if tp.get("norm_first", False):
        x = x + sa(n1(x))
        return x + ff(n2(x))
    x = n1(x + sa(x))
    return n2(x + ff(x))


def shared_attn_halves(learner: Mapping[str, Any], index: int):
    """shared_attn_learner.py:49-104: ctx[index].unsqueeze(0) is fed to a batch_first=False layer, i.e. as a length-1
    sequence of n_ctx batch items; the result is split into (textual, visual) column blocks."""
    out = transformer_encoder_layer(learner["tlayers"][index], learner["ctx"][index].unsqueeze(0)).squeeze(0)
    td = learner["textual_dim"]
    return out[:, :td], out[:, td:]


def textual_context(learner: Mapping[str, Any], index: int, image_features: torch.Tensor | None) -> torch.Tensor:
    """coop_context_learner.py:115-122 / cocoop_context_learner.py:33-58 / shared_*_learner.py."""
    ctx = learner["ctx"][index]
    if learner["kind"] == "shared_separate":
        return run_projection(learner["tproj"][index], ctx)
    if learner["kind"] == "shared_attn":
        return shared_attn_halves(learner, index)[0]
    if learner["kind"] == "cocoop":
        feats = image_features
        if learner.get("norm_image_features", True):
            feats = feats / feats.norm(dim=-1, keepdim=True)
        bias = run_projection(learner["proj"][index], feats).unsqueeze(1)
        return bias + ctx
    return ctx


def visual_context(learner: Mapping[str, Any], index: int) -> torch.Tensor:
    """vpt_context_learner.py:41-44 / maple_context_learner.py:19-20."""
    if learner["kind"] == "maple":
        return run_projection(learner["proj"][index], learner["ctx"][index])
    if learner["kind"] == "shared_separate":
        return run_projection(learner["vproj"][index], learner["ctx"][index])
    if learner["kind"] == "shared_attn":
        return shared_attn_halves(learner, index)[1]
    return learner["ctx"][index]


def coop_splice(emb: torch.Tensor, ctx: torch.Tensor, max_length: int) -> torch.Tensor:
    """CoOpContextLearner.forward (coop_context_learner.py:136-181)."""
    n = ctx.shape[-2]
    first = emb[:, :1]
    last_idx = min(max_length - n, emb.size(1)) - 1
    mid = emb[:, 1:last_idx]
    last = emb[:, -1:]
    if ctx.dim() == 2:
        ctx = ctx.expand(emb.size(0), -1, -1)
    return torch.cat((first, ctx, mid, last), dim=1)


# ----------------------------------------------------------------------------
# towers
# ----------------------------------------------------------------------------
def text_features(sd: SD, cfg, input_ids, attention_mask, learner=None, image_features=None) -> torch.Tensor:
    """Text tower -> text_projection(pooled).

    learner None: HF ``get_text_features`` (HF:594-686) as called under no_grad by
    ``vpt_clipseg.py:347-354``.  Otherwise ``coop_clipseg.py:188-339`` /
    ``base_multimodal_clipseg.py:166-300``.
    """
    t = cfg.text_config
    B = input_ids.shape[0]
    emb = F.embedding(input_ids, sd["clip.text_model.embeddings.token_embedding.weight"])
    n = 0
    if learner is not None and learner["kind"] in ("coop", "cocoop", "maple", "shared_separate", "shared_attn"):
        n = learner["ctx"].shape[1]
        emb = coop_splice(emb, textual_context(learner, 0, image_features), t.max_position_embeddings)
        if attention_mask is not None:
            ones = torch.ones(B, n, dtype=attention_mask.dtype)
            attention_mask = torch.cat((ones, attention_mask), dim=1)[:, : t.max_position_embeddings]
    T = emb.shape[1]
    # NB the reference takes position rows [:L+n] even when the splice truncated to
    # max_position_embeddings (coop_clipseg.py:66-73); L+n <= 77 whenever shapes agree.
    x = emb + sd["clip.text_model.embeddings.position_embedding.weight"][:T]
    mask = text_additive_mask(attention_mask, B, T)
    depth = 1 if learner is None else learner["ctx"].shape[0]
    for idx in range(1, t.num_hidden_layers + 1):
        x = encoder_layer(sd, f"clip.text_model.encoder.layers.{idx - 1}", x, t.num_attention_heads, t.hidden_act, t.layer_norm_eps, mask)
        if n and idx < depth:
            x = x.clone()
            x[:, 1 : n + 1] = textual_context(learner, idx, image_features)
    x = layer_norm(sd, "clip.text_model.final_layer_norm", x, t.layer_norm_eps)
    pool = torch.minimum(eos_pool_index(cfg, input_ids) + n, torch.tensor(t.max_position_embeddings - 1))
    pooled = x[torch.arange(B), pool]
    return F.linear(pooled, sd["clip.text_projection.weight"])


def vision_tower(sd: SD, cfg, pixel_values, learner=None, full: bool = False):
    """Returns (activations at extract layers, pooled visual_projection(CLS) or None).

    full=False: prompt path of ``vpt_clipseg.py:151-235`` /
    ``base_multimodal_clipseg.py:310-484`` (prompts appended at the END, concat
    BEFORE pre_layrnorm, early break after max(extract_layers)+1 layers).
    full=True:  HF vision model as used by ``coop_clipseg.py:341-371`` (all layers,
    post_layernorm on CLS, visual_projection).
    """
    v = cfg.vision_config
    x = vision_embeddings(sd, cfg, pixel_values)
    n, depth = 0, 1
    if learner is not None and learner["kind"] in ("vpt", "maple", "shared_separate", "shared_attn") and not full:
        n = learner["ctx"].shape[1]
        depth = learner["ctx"].shape[0]
        x = torch.cat((x, visual_context(learner, 0).expand(x.shape[0], -1, -1)), dim=1)
    x = layer_norm(sd, "clip.vision_model.pre_layrnorm", x, v.layer_norm_eps)
    states = [x]
    max_idx = max(cfg.extract_layers)
    for idx in range(1, v.num_hidden_layers + 1):
        x = encoder_layer(sd, f"clip.vision_model.encoder.layers.{idx - 1}", x, v.num_attention_heads, v.hidden_act, v.layer_norm_eps)
        if n and idx < depth:
            x = x.clone()
            x[:, -n:] = visual_context(learner, idx)
        states.append(x)
        if not full and idx > max_idx:
            break
    acts = tuple(states[i + 1] for i in cfg.extract_layers)
    pooled = None
    if full:
        cls = layer_norm(sd, "clip.vision_model.post_layernorm", x[:, 0], v.layer_norm_eps)
        pooled = F.linear(cls, sd["clip.visual_projection.weight"])
    return acts, pooled


def additive_layer(out: torch.Tensor, conv_w: torch.Tensor, conv_b: torch.Tensor, scale: int) -> torch.Tensor:
    """``base_clipseg.py:58-71``: Upsample(xP, bilinear) -> Conv2d(k, same, replicate)."""
    up = F.interpolate(out, scale_factor=float(scale), mode="bilinear")
    k = conv_w.shape[-1]
    pad = k // 2
    up = F.pad(up, (pad, pad, pad, pad), mode="replicate")
    return F.conv2d(up, conv_w, conv_b)


def decoder(sd: SD, cfg, acts, cond, n_strip: int = 0, new_last=None, mix: str = "none") -> torch.Tensor:
    """``base_clipseg.py:82-172`` / ``vpt_clipseg.py:237-319`` / HF:549-586.

    mix: "none" (HF decoder, CoOp), "add" (VPT: logits += f(out)),
    "ratio" (Base/MaPLe: (1-r)*logits + r*f(out)).  new_last = (conv_w, conv_b, r).
    Returns logits [B, H, W].
    """
    eps = cfg.vision_config.layer_norm_eps
    out = None
    for i, act in enumerate(acts[::-1]):
        red = linear(sd, f"decoder.reduces.{i}", act)
        out = red if out is None else red + out
        if i == cfg.conditional_layer:
            out = linear(sd, "decoder.film_mul", cond) * out.permute(1, 0, 2) + linear(sd, "decoder.film_add", cond)
            out = out.permute(1, 0, 2)
        out = decoder_layer(sd, f"decoder.layers.{i}", out, cfg.decoder_num_attention_heads, eps)
    out = out[:, 1 : (-n_strip if n_strip else None), :].permute(0, 2, 1)
    B, C, N = out.shape
    size = math.isqrt(N)
    out = out.reshape(B, C, size, size)
    ps = cfg.vision_config.patch_size
    logits = F.conv_transpose2d(out, sd["decoder.transposed_convolution.weight"], sd["decoder.transposed_convolution.bias"], stride=ps)
    if new_last is not None and mix != "none":
        conv_w, conv_b, r = new_last
        extra = additive_layer(out, conv_w, conv_b, ps)
        logits = logits + extra if mix == "add" else (1 - r) * logits + r * extra
    return logits[:, 0]


# ----------------------------------------------------------------------------
# nets (the boundary call ``net(text_input, image_input) -> [B,1,H,W]``)
# ----------------------------------------------------------------------------
def vpt_forward(sd, cfg, learner, pixel_values, input_ids, attention_mask, new_last=None) -> torch.Tensor:
    """``VPTCLIPSeg.model_forward`` (vpt_clipseg.py:321-395) + ``BaseCLIPSeg.forward``."""
    B, _, H, W = pixel_values.shape
    with torch.no_grad():
        cond = text_features(sd, cfg, input_ids, attention_mask)
    acts, _ = vision_tower(sd, cfg, pixel_values, learner)
    logits = decoder(sd, cfg, acts, cond, n_strip=learner["ctx"].shape[1], new_last=new_last, mix="add")
    return logits.reshape(B, 1, H, W)


def coop_forward(sd, cfg, learner, pixel_values, input_ids, attention_mask) -> torch.Tensor:
    """``COOPCLIPSeg.model_forward`` (coop_clipseg.py:418-484); HF decoder, no new last layer."""
    B, _, H, W = pixel_values.shape
    acts, pooled = vision_tower(sd, cfg, pixel_values, None, full=True)
    cond = text_features(sd, cfg, input_ids, attention_mask, learner, image_features=pooled)
    logits = decoder(sd, cfg, acts, cond)
    return logits.reshape(B, 1, H, W)


def maple_forward(sd, cfg, learner, pixel_values, input_ids, attention_mask, new_last=None) -> torch.Tensor:
    """``BaseMultimodalCLIPSeg.model_forward`` (base_multimodal_clipseg.py:552-629)."""
    B, _, H, W = pixel_values.shape
    acts, _ = vision_tower(sd, cfg, pixel_values, learner)
    cond = text_features(sd, cfg, input_ids, attention_mask, learner)
    logits = decoder(sd, cfg, acts, cond, n_strip=learner["ctx"].shape[1], new_last=new_last, mix="ratio")
    return logits.reshape(B, 1, H, W)


# ----------------------------------------------------------------------------
# loss / metrics (third-party semantics restated from their published source;
# monai and torchmetrics are not installed here => "parity unpinned" for these)
# ----------------------------------------------------------------------------
def dice_ce_loss(logits: torch.Tensor, target: torch.Tensor, lambda_dice: float = 1.0, lambda_ce: float = 0.2,
                 smooth_nr: float = 1e-5, smooth_dr: float = 1e-5) -> torch.Tensor:
    """monai.losses.DiceCELoss(sigmoid=True) for one channel
    (call site ``configs/model/vpt_clipseg.yaml:21-25``)."""
    p = torch.sigmoid(logits)
    axes = tuple(range(2, logits.dim()))
    inter = (p * target).sum(axes)
    denom = p.sum(axes) + target.sum(axes)
    dice = (1.0 - (2.0 * inter + smooth_nr) / (denom + smooth_dr)).mean()
    bce = F.binary_cross_entropy_with_logits(logits, target)
    return lambda_dice * dice + lambda_ce * bce


def confusion_counts(preds: torch.Tensor, targets: torch.Tensor, threshold: float = 0.5):
    """Per-sample integer TP/FP/FN/TN of ``(preds > threshold)`` vs ``targets`` (int64)."""
    lab = (preds > threshold).flatten(1)
    tgt = targets.flatten(1).bool()
    tp = (lab & tgt).sum(1)
    fp = (lab & ~tgt).sum(1)
    fn = (~lab & tgt).sum(1)
    tn = (~lab & ~tgt).sum(1)
    return tp, fp, fn, tn


def dice_samples(tp, fp, fn, zero_division: float = 1.0) -> torch.Tensor:
    """torchmetrics.Dice(average="samples"): mean over samples of 2TP/(2TP+FP+FN)."""
    den = (2 * tp + fp + fn).double()
    val = torch.where(den > 0, 2 * tp.double() / den.clamp(min=1), torch.full_like(den, zero_division))
    return val.mean()


def jaccard_binary(tp, fp, fn, zero_division: float = 1.0) -> torch.Tensor:
    """torchmetrics.JaccardIndex(task="binary"): dataset-global TP/(TP+FP+FN)."""
    tp, fp, fn = tp.sum().double(), fp.sum().double(), fn.sum().double()
    den = tp + fp + fn
    return tp / den if den > 0 else torch.tensor(zero_division, dtype=torch.float64)
