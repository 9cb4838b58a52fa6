#!/usr/bin/env python
"""Generate golden fixtures by running the REFERENCE classes (build container only).

Run:  python tests/golden/make_goldens.py            (needs /root/reference)

What it does
------------
* imports the reference's own nets / learners from ``/root/reference``
  (``src.models.core_models.coop``) -- nothing of the reference is copied;
* bridges transformers-4.4x -> 5.x API drift with a small in-memory shim
  (SURVEY.md §8c): the two mask helpers the reference imports, and the
  ``(hidden, attention_mask, causal_attention_mask, output_attentions=)``
  calling convention of ``CLIPSegEncoderLayer`` / ``CLIPSegDecoderLayer``;
* replaces ``from_pretrained`` (no network) by a local constructor that loads the
  seeded random weights of ``tunevlseg_amd.weights.init_clipseg_state_dict``;
* runs forward + backward in strict fp32 on CPU and writes ``tests/golden/*.npz``
  holding inputs, trainable parameters, logits, loss and parameter gradients.

The fixtures are data only.  Backbone weights are NOT stored; they are re-drawn
from ``(config preset, seed)`` and guarded by a checksum stored in the fixture.
"""
from __future__ import annotations

import json
import os
import sys
from functools import partial
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
REFERENCE = Path(os.environ.get("TVL_REFERENCE", "/root/reference"))
sys.path.insert(0, str(REFERENCE))

from tunevlseg_amd.config import CLIPSegConfig  # noqa: E402
from tunevlseg_amd.cris_config import CRISConfig  # noqa: E402
from tunevlseg_amd.weights import init_clipseg_state_dict, init_cris_state_dict  # noqa: E402
from tests.golden_util import synth_cris_inputs, synth_inputs  # noqa: E402  (input recipes shared with the tests)

OUT = Path(__file__).resolve().parent


# ----------------------------------------------------------------------------
# shim: transformers 5.x  <->  reference written for 4.4x
# ----------------------------------------------------------------------------
def install_shim():
    from transformers.models.clipseg import modeling_clipseg as M

    def _create_4d_causal_attention_mask(shape, dtype, device=None):
        B, T = shape
        m = torch.full((T, T), torch.finfo(dtype).min, dtype=dtype, device=device).triu(1)
        return m[None, None].expand(B, 1, T, T)

    def _prepare_4d_attention_mask(mask, dtype, tgt_len=None):
        B, S = mask.shape
        T = tgt_len or S
        inv = 1.0 - mask[:, None, None, :].to(dtype).expand(B, 1, T, S)
        return inv.masked_fill(inv.bool(), torch.finfo(dtype).min)

    M._create_4d_causal_attention_mask = _create_4d_causal_attention_mask
    M._prepare_4d_attention_mask = _prepare_4d_attention_mask

    def wrap(cls):
        orig = cls.forward

        def forward(self, hidden_states, attention_mask=None, *args, **kwargs):
            ref_style = len(args) >= 1 or "causal_attention_mask" in kwargs
            if not ref_style:
                return orig(self, hidden_states, attention_mask, **kwargs)
            causal = args[0] if args else kwargs.pop("causal_attention_mask", None)
            kwargs.pop("output_attentions", None)
            mask = attention_mask
            if causal is not None:
                mask = causal if mask is None else (mask + causal).clamp(min=torch.finfo(hidden_states.dtype).min)
            return (orig(self, hidden_states, mask),)

        cls.forward = forward

    wrap(M.CLIPSegEncoderLayer)
    wrap(M.CLIPSegDecoderLayer)
    return M


def hf_model_from_state(cfg: CLIPSegConfig, sd):
    from transformers import CLIPSegConfig as HFConfig
    from transformers import CLIPSegForImageSegmentation

    d = cfg.to_dict()
    hf_cfg = HFConfig(
        text_config={k: v for k, v in d["text_config"].items() if not k.startswith(("output_", "use_return"))},
        vision_config={k: v for k, v in d["vision_config"].items() if not k.startswith(("output_", "use_return"))},
        projection_dim=cfg.projection_dim,
        extract_layers=list(cfg.extract_layers),
        reduce_dim=cfg.reduce_dim,
        decoder_num_attention_heads=cfg.decoder_num_attention_heads,
        decoder_intermediate_size=cfg.decoder_intermediate_size,
        conditional_layer=cfg.conditional_layer,
    )
    hf_cfg._attn_implementation = "eager"
    hf_cfg.text_config._attn_implementation = "eager"
    hf_cfg.vision_config._attn_implementation = "eager"
    model = CLIPSegForImageSegmentation(hf_cfg)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    missing = [m for m in missing if "position_ids" not in m]
    assert not missing and not unexpected, (missing, unexpected)
    for c in (model.config, model.config.text_config, model.config.vision_config):
        if not hasattr(c, "use_return_dict"):
            try:
                c.use_return_dict = True
            except Exception:
                pass
    return model.eval()


class StubTokenizer:
    """The learner only reads ``.input_ids`` (coop_context_learner.py:71-77)."""

    def __init__(self, ids):
        self.ids = ids

    def __call__(self, text, **kw):
        class R:
            pass

        r = R()
        n = 1 if isinstance(text, str) else len(text)
        r.input_ids = torch.tensor([self.ids] * n, dtype=torch.long)
        return r


def state_checksum(sd) -> float:
    return float(sum(v.double().abs().sum() for v in sd.values()))


# ----------------------------------------------------------------------------


def compact_outputs(logits, mask):
    """What a full-batch fixture keeps instead of the 16 MB logit map: every 11th pixel, and the integer / per-sample
    statistics of the reference's own logits (plain counting of ``sigmoid(logits) > 0.5`` against ``mask.long()``)."""
    lab = (torch.sigmoid(logits.detach()) > 0.5).flatten(1)
    tgt = mask.long().flatten(1).bool()
    tp, fp = (lab & tgt).sum(1), (lab & ~tgt).sum(1)
    fn, tn = (~lab & tgt).sum(1), (~lab & ~tgt).sum(1)
    den = (2 * tp + fp + fn).double()
    dice = torch.where(den > 0, 2 * tp.double() / den.clamp(min=1), torch.ones_like(den))
    # full label map as packed bits + the pixels whose logit is within 1e-4 of the threshold (any fp32 evaluation, the reference's
    # own included, may put those on either side): bit-exactness is asserted everywhere else
    flat = logits.detach().flatten()
    amb = torch.nonzero(flat.abs() < 1e-4).flatten()
    return {"out.label_bits": np.packbits(lab.numpy().astype(np.uint8)), "out.ambiguous_idx": amb.numpy(),
            "out.ambiguous_logits": flat[amb].numpy(),
            "out.logits_s11": logits.detach()[..., ::11, ::11].contiguous().numpy(),
            "out.counts": torch.stack((tp, fp, fn, tn), 1).numpy(), "out.dice_per_sample": dice.numpy(),
            "out.logits_absmax": logits.detach().abs().amax().numpy(),
            "out.min_abs_logit": logits.detach().abs().amin().numpy()}


def run_case(name: str, *, preset: str, eos: int, wseed: int, net_kind: str, learner_kw: dict, net_kw: dict,
             B: int, H: int, L: int, iseed: int, M, compact: bool = False, with_f64: bool = False, tails: int = 0, big_by_seed: bool = False):
    from src.models.components.hf_clipseg_wrapper import HFCLIPSegWrapper
    from src.models.core_models import coop as R
    from src.models.core_models.coop import context_learner as CL

    sys.path.insert(0, str(ROOT))
    from oracle.clipseg_oracle import dice_ce_loss

    cfg = CLIPSegConfig.tiny(eos_token_id=eos) if preset == "tiny" else CLIPSegConfig.rd64(eos_token_id=eos)
    sd = init_clipseg_state_dict(cfg, wseed, tails=tails)
    HFCLIPSegWrapper.get_pretrained_model = staticmethod(lambda *a, **k: hf_model_from_state(cfg, sd))

    if ONLY and not any(name.startswith(o) for o in ONLY):
        return
    net_cls = {"vpt": R.VPTCLIPSeg, "coop": R.COOPCLIPSeg, "cocoop": R.COOPCLIPSeg, "maple": R.MapleCLIPSeg,
               "shared_separate": R.SharedSeparateCLIPSeg, "shared_attn": R.SharedAttnCLIPSeg}[net_kind]
    learner_cls = {"vpt": CL.VPTContextLearner, "coop": CL.CoOpContextLearner,
                   "cocoop": CL.CoCoOpContextLearner, "maple": CL.MapleContextLearner,
                   "shared_separate": CL.SharedSeparateLearner, "shared_attn": CL.SharedAttnLearner}[net_kind]
    lkw = dict(learner_kw)
    if "_tlayer" in lkw:  # SharedAttn: the config's partial(nn.TransformerEncoderLayer, ...)
        lkw["unified_projector"] = partial(torch.nn.TransformerEncoderLayer, **lkw.pop("_tlayer"))
    if lkw.get("context_initializer") is not None:
        lkw["tokenizer"] = StubTokenizer(lkw.pop("_init_ids"))
    torch.manual_seed(1000 + iseed)
    net = net_cls(context_learner=partial(learner_cls, **lkw),
                  model_cfg={"pretrained_model_name_or_path": None, "freeze_encoder": False, "freeze_decoder": False},
                  **net_kw)
    torch.set_float32_matmul_precision("highest")  # reference sets "medium" on import; goldens are strict fp32
    # give every trainable tensor an O(1)-visible, seeded value (biases too)
    g = torch.Generator().manual_seed(2000 + iseed)
    params = {k: p for k, p in net.named_parameters() if p.requires_grad}
    with torch.no_grad():
        for k, p in params.items():
            if k.endswith("context_vectors") and lkw.get("context_initializer") is None:
                p.copy_(torch.randn(p.shape, generator=g) * 0.02)
            elif k == "residual_ratio":
                pass
            elif p.dim() == 1 and "projection_layers" in k and not k.endswith(".bias") and "in_proj" not in k:
                p.copy_(1 + 0.05 * torch.randn(p.shape, generator=g))
            elif k.endswith(".bias"):
                p.copy_(0.05 * torch.randn(p.shape, generator=g))

    # ``big_by_seed``: trainable tensors above 200 k elements (the SharedAttn learner's 1280-wide TransformerEncoderLayers: 10 M parameters a
    # layer) are re-drawn from a per-tensor seed, and the fixture keeps (seed, std, shape) instead of the values and every 61st gradient
    # entry + the gradient's absolute sum instead of the gradient (tests/golden_util.py trainable_of / grad_of rebuild / compare)
    by_seed = {}
    if big_by_seed:
        import zlib

        from tests.golden_util import redraw

        with torch.no_grad():
            for k, p in params.items():
                if p.numel() > 200_000:
                    seed, std = (zlib.crc32(k.encode()) ^ (7919 * iseed)) & 0x7FFFFFFF, float(f"{float(p.std()):.3g}")
                    p.copy_(redraw(seed, std, tuple(p.shape)))
                    by_seed[k] = (seed, std)
    pix, ids, am, mask = synth_inputs(cfg, B, H, L, iseed)
    logits = net(text_input={"input_ids": ids, "attention_mask": am}, image_input=pix)
    loss = dice_ce_loss(logits, mask)
    loss.backward()

    if compact:  # inputs are re-drawn from the seed by the test (tests/golden_util.py synth_inputs)
        arrays = {"out.loss": loss.detach().numpy(), **compact_outputs(logits, mask)}
    else:
        arrays = {"in.pixel_values": pix.numpy(), "in.input_ids": ids.numpy(), "in.attention_mask": am.numpy(),
                  "in.mask": mask.numpy(), "out.logits": logits.detach().numpy(), "out.loss": loss.detach().numpy()}
    grads_none = []
    for k, p in params.items():
        if k in by_seed:
            arrays["paramseed." + k] = np.array([by_seed[k][0], by_seed[k][1], *p.shape], dtype=np.float64)
        else:
            arrays["param." + k] = p.detach().numpy()
        if p.grad is None:
            grads_none.append(k)
        elif k in by_seed:
            arrays["gradsub." + k] = p.grad.flatten()[::61].contiguous().numpy()
            arrays["gradabs." + k] = p.grad.double().abs().sum().numpy()
        else:
            arrays["grad." + k] = p.grad.numpy()
    if with_f64:
        # The same reference classes run in float64 ("grad64.*", stored rounded to fp32): deep prompts give gradients that are
        # small differences of larger terms, and on such fixtures the reference's own fp32 gradient is ~1e-3 away from the exact
        # one.  The tests then hold the HIP path to the exact gradient, no looser than the reference's own fp32 deviation.
        import copy

        import torch.nn.functional as TF

        net64 = copy.deepcopy(net).double()
        for p_ in net64.parameters():
            p_.grad = None
        orig_softmax = TF.softmax
        TF.softmax = lambda x, dim=-1, dtype=None, **kw: orig_softmax(x, dim=dim)  # HF's eager attention pins the softmax to fp32
        try:
            logits64 = net64(text_input={"input_ids": ids, "attention_mask": am}, image_input=pix.double())
            dice_ce_loss(logits64, mask.double()).backward()
        finally:
            TF.softmax = orig_softmax
        for k, p_ in net64.named_parameters():
            if p_.requires_grad and p_.grad is not None:
                arrays["grad64." + k] = p_.grad.float().numpy()
        if tails and not compact:
            arrays["out.logits64"] = logits64.detach().float().numpy()
        elif tails:
            arrays["out.logits64_s11"] = logits64.detach().float()[..., ::11, ::11].contiguous().numpy()
        print(f"   f64: logits fp32-vs-fp64 {float((logits.detach().double() - logits64.detach()).abs().max()):.2e}; worst fp32 gradient "
              f"deviation {max(float((params[k].grad.double() - p_.grad).abs().max() / p_.grad.abs().max()) for k, p_ in net64.named_parameters() if p_.requires_grad and p_.grad is not None):.2e}")
    meta = {"name": name, "compact": compact, "preset": preset, "eos_token_id": eos, "weight_seed": wseed, "tails": tails, "net": net_kind,
            "learner_kw": {k: v for k, v in learner_kw.items()}, "net_kw": net_kw, "B": B, "H": H, "L": L,
            "input_seed": iseed, "weights_checksum": state_checksum(sd), "grads_none": grads_none,
            "torch": torch.__version__}
    arrays["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(OUT / f"{name}.npz", **arrays)
    print(f"{name}: logits mean {logits.mean():+.4f} std {logits.std():.4f} |max| {logits.abs().max():.3f} "
          f"pos-frac {(logits > 0).float().mean():.3f} loss {loss.item():.6f} "
          f"grads {[(k, float(p.grad.abs().max())) for k, p in params.items() if p.grad is not None][:3]} none={grads_none}")




def run_cris_case(name: str, *, preset: str, wseed: int, learner_kind: str, learner_kw: dict, net_kw: dict,
                  B: int, L: int, iseed: int, with_attention_mask: bool = True, img_size: int | None = None, compact: bool = False):
    """COOPCRIS (reference coop_cris.py) with seeded random weights: ``CRIS.get_backbone`` (which wants pretrain/RN50.pt)
    is replaced by a local ``CLIP(...)`` constructor of the same geometry."""
    if ONLY and not any(name.startswith(o) for o in ONLY):
        return
    from src.models.components.cris_model import CRIS
    from src.models.components.cris_model.clip import CLIP
    from src.models.core_models import coop as R
    from src.models.core_models.coop import context_learner as CL

    from oracle.clipseg_oracle import dice_ce_loss

    cfg = CRISConfig.tiny() if preset == "tiny" else CRISConfig.rn50()
    if img_size:
        cfg.img_size = img_size
    sd = init_cris_state_dict(cfg, wseed)

    def get_backbone(_path):
        clip = CLIP(cfg.embed_dim, cfg.image_resolution, cfg.vision_layers, cfg.vision_width, None, cfg.context_length,
                    cfg.vocab_size, cfg.transformer_width, cfg.transformer_heads, cfg.transformer_layers).float()
        sub = {k[len("backbone."):]: v for k, v in sd.items() if k.startswith("backbone.")}
        missing, unexpected = clip.load_state_dict(sub, strict=False)
        assert not unexpected and all(m.endswith("num_batches_tracked") for m in missing), (missing, unexpected)
        return clip.eval()

    CRIS.get_backbone = staticmethod(get_backbone)
    learner_cls = {"coop": CL.CoOpContextLearner, "cocoop": CL.CoCoOpContextLearner}[learner_kind]
    lkw = dict(learner_kw)
    if lkw.get("context_initializer") is not None:
        lkw["tokenizer"] = StubTokenizer(lkw.pop("_init_ids"))
    torch.manual_seed(1000 + iseed)
    net = R.COOPCRIS(
        model_cfg=dict(clip_pretrain=None, fpn_in=list(cfg.fpn_in), fpn_out=list(cfg.fpn_out), vis_dim=cfg.vis_dim,
                       word_dim=cfg.word_dim, num_layers=cfg.num_layers, num_head=cfg.num_head, dim_ffn=cfg.dim_ffn,
                       dropout=cfg.dropout, return_intermediate=False, img_size=cfg.img_size, freeze_encoder=True,
                       cris_pretrain=None),
        context_learner=partial(learner_cls, **lkw), **net_kw)
    missing, unexpected = net.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    bad = [m for m in missing if not (m.endswith("num_batches_tracked") or m.startswith(("context_learner.", "additive_decoder_layer."))
                                      or m == "residual_ratio")]
    assert not bad, bad
    assert not net.training or True
    net.neck.eval(), net.decoder.eval(), net.proj.eval(), net.backbone.eval()
    torch.set_float32_matmul_precision("highest")
    g = torch.Generator().manual_seed(2000 + iseed)
    params = {k: p for k, p in net.named_parameters() if p.requires_grad}
    with torch.no_grad():
        for k, p in params.items():
            if k.endswith("context_vectors") and lkw.get("context_initializer") is None:
                p.copy_(torch.randn(p.shape, generator=g) * 0.02)
            elif k == "residual_ratio":
                pass
            elif p.dim() == 1 and "projection_layers" in k and not k.endswith(".bias") and "in_proj" not in k:
                p.copy_(1 + 0.05 * torch.randn(p.shape, generator=g))
            elif k.endswith(".bias"):
                p.copy_(0.05 * torch.randn(p.shape, generator=g))

    pix, ids, am, mask = synth_cris_inputs(cfg, B, L, iseed, with_attention_mask)
    text_input = {"input_ids": ids} if am is None else {"input_ids": ids, "attention_mask": am}
    logits = net(text_input=text_input, image_input=pix)
    loss = dice_ce_loss(logits, mask)
    loss.backward()
    if compact:
        arrays = {"out.loss": loss.detach().numpy(), **compact_outputs(logits, mask)}
    else:
        arrays = {"in.pixel_values": pix.numpy(), "in.input_ids": ids.numpy(), "in.mask": mask.numpy(),
                  "out.logits": logits.detach().numpy(), "out.loss": loss.detach().numpy()}
        if am is not None:
            arrays["in.attention_mask"] = am.numpy()
    grads_none = []
    for k, p in params.items():
        arrays["param." + k] = p.detach().numpy()
        if p.grad is None:
            grads_none.append(k)
        else:
            arrays["grad." + k] = p.grad.numpy()
    meta = {"name": name, "family": "cris", "compact": compact, "with_attention_mask": with_attention_mask, "preset": preset,
            "weight_seed": wseed, "net": learner_kind,
            "learner_kw": dict(learner_kw), "net_kw": net_kw, "B": B, "L": L, "img_size": cfg.img_size,
            "input_seed": iseed, "weights_checksum": state_checksum(sd), "grads_none": grads_none, "torch": torch.__version__}
    arrays["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(OUT / f"{name}.npz", **arrays)
    print(f"{name}: logits mean {logits.mean():+.4f} std {logits.std():.4f} |max| {logits.abs().max():.3f} "
          f"pos-frac {(logits > 0).float().mean():.3f} loss {loss.item():.6f} "
          f"grads {[(k, float(p.grad.abs().max())) for k, p in params.items() if p.grad is not None][:4]} none={grads_none}")


ONLY = sys.argv[1:]  # optional name prefixes: regenerate only those fixtures


def main():
    M = install_shim()
    T = dict(preset="tiny", B=2, H=64, L=6, M=M)
    base_new = dict(use_new_last_layer=True, new_last_layer_kernel_size=5, residual_ratio=0.5)
    base_old = dict(use_new_last_layer=False)
    # --- tiny: VPT ------------------------------------------------------------
    run_case("tiny_vpt_n4_d1", eos=2, wseed=11, net_kind="vpt", iseed=1,
             learner_kw=dict(prompt_depth=1, num_context=4, vector_std=0.02), net_kw=base_old, **T)
    run_case("tiny_vpt_n1_d3_newlast", eos=2, wseed=11, net_kind="vpt", iseed=2,
             learner_kw=dict(prompt_depth=3, num_context=1, vector_std=0.02), net_kw=base_new, **T)
    run_case("tiny_vpt_n10_d2_eos", eos=63, wseed=12, net_kind="vpt", iseed=3,
             learner_kw=dict(prompt_depth=2, num_context=10, vector_std=0.02), net_kw=base_old, **T)
    # --- tiny: CoOp / CoCoOp --------------------------------------------------
    run_case("tiny_coop_n4_d1", eos=2, wseed=11, net_kind="coop", iseed=4,
             learner_kw=dict(prompt_depth=1, num_context=4, vector_std=0.02), net_kw=base_old, **T)
    run_case("tiny_coop_init_d3_eos", eos=63, wseed=12, net_kind="coop", iseed=5,
             learner_kw=dict(prompt_depth=3, num_context=4, context_initializer="a photo of a", _init_ids=[5, 9, 7, 5],
                             vector_std=0.02), net_kw=base_old, **T)
    run_case("tiny_coop_trunc_n12", eos=2, wseed=11, net_kind="coop", iseed=6,
             learner_kw=dict(prompt_depth=1, num_context=12, vector_std=0.02), net_kw=base_old, **T)
    run_case("tiny_cocoop_d2_i8_norm", eos=2, wseed=11, net_kind="cocoop", iseed=7,
             learner_kw=dict(prompt_depth=2, num_context=4, vector_std=0.02, use_unified_projection=False,
                             intermediate_dim=8, use_proj_norm=True, use_lora_proj=False, norm_image_features=True),
             net_kw=base_old, **T)
    run_case("tiny_cocoop_d3_unified_lora", eos=2, wseed=11, net_kind="cocoop", iseed=8,
             learner_kw=dict(prompt_depth=3, num_context=2, vector_std=0.02, use_unified_projection=True,
                             intermediate_dim=8, use_proj_norm=False, use_lora_proj=True, norm_image_features=False),
             net_kw=base_old, **T)
    # --- tiny: MaPLe ----------------------------------------------------------
    run_case("tiny_maple_d3_n2_newlast", eos=2, wseed=11, net_kind="maple", iseed=9,
             learner_kw=dict(prompt_depth=3, num_context=2, vector_std=0.02, use_unified_projection=False,
                             intermediate_dim=8, use_proj_norm=True, use_lora_proj=False), net_kw=base_new, **T)
    run_case("tiny_maple_d1_init", eos=63, wseed=12, net_kind="maple", iseed=10,
             learner_kw=dict(prompt_depth=1, num_context=4, context_initializer="a photo of a", _init_ids=[5, 9, 7, 5],
                             vector_std=0.02, use_unified_projection=True, intermediate_dim=None, use_proj_norm=False),
             net_kw=base_old, **T)
    # --- tiny: shared learners (row A16) ----------------------------------------
    run_case("tiny_sharedsep_d2_i8", eos=2, wseed=11, net_kind="shared_separate", iseed=11,
             learner_kw=dict(prompt_depth=2, num_context=3, vector_std=0.02, shared_dim=8, use_unified_projection=False,
                             intermediate_dim=8, use_proj_norm=True, use_lora_proj=False), net_kw=base_new, **T)
    run_case("tiny_sharedattn_d2", eos=2, wseed=11, net_kind="shared_attn", iseed=12,
             learner_kw=dict(prompt_depth=2, num_context=3, vector_std=0.02, use_unified_projection=False,
                             _tlayer=dict(nhead=4, dim_feedforward=48, dropout=0.0, norm_first=True)), net_kw=base_old, **T)
    run_case("tiny_sharedattn_d3_unified_postnorm", eos=63, wseed=12, net_kind="shared_attn", iseed=13,
             learner_kw=dict(prompt_depth=3, num_context=2, vector_std=0.02, use_unified_projection=True,
                             _tlayer=dict(nhead=2, dim_feedforward=32, dropout=0.0, norm_first=False)), net_kw=base_new, **T)
    # --- full size (ViT-B/16 rd64 geometry), B=1 ------------------------------
    F_ = dict(preset="rd64", B=1, H=352, L=8, M=M)
    run_case("rd64_vpt_n10_d1", eos=2, wseed=21, net_kind="vpt", iseed=21,
             learner_kw=dict(prompt_depth=1, num_context=10, vector_std=0.02), net_kw=base_old, **F_)
    run_case("rd64_coop_n4_d1", eos=2, wseed=21, net_kind="coop", iseed=22,
             learner_kw=dict(prompt_depth=1, num_context=4, vector_std=0.02), net_kw=base_old, **F_)
    run_case("rd64_maple_n4_d9_newlast", eos=2, wseed=21, net_kind="maple", iseed=23,
             learner_kw=dict(prompt_depth=9, num_context=4, vector_std=0.02, use_unified_projection=False,
                             intermediate_dim=64, use_proj_norm=True, use_lora_proj=False), net_kw=base_new, **F_)

    # --- unfrozen last transposed conv (row A2, base_clipseg.py:74-80) ---------------------------------------------
    run_case("tiny_vpt_n4_d2_nofreeze_last", eos=2, wseed=11, net_kind="vpt", iseed=14,
             learner_kw=dict(prompt_depth=2, num_context=4, vector_std=0.02),
             net_kw=dict(use_new_last_layer=False, no_freeze_last_layer=True), **T)
    run_case("tiny_maple_d2_nofreeze_last", eos=2, wseed=11, net_kind="maple", iseed=15,
             learner_kw=dict(prompt_depth=2, num_context=2, vector_std=0.02, use_unified_projection=False,
                             intermediate_dim=8, use_proj_norm=True, use_lora_proj=False),
             net_kw=dict(use_new_last_layer=False, no_freeze_last_layer=True), **T)
    # --- ends of the Optuna ranges (prompt_depth 1-10, intermediate_dim 32-128, LoRA on/off), B=1 -------
    run_case("rd64_vpt_n10_d10", eos=2, wseed=21, net_kind="vpt", iseed=24,
             learner_kw=dict(prompt_depth=10, num_context=10, vector_std=0.02), net_kw=base_old, with_f64=True, **F_)
    run_case("rd64_maple_n4_d10_i32_lora", eos=2, wseed=21, net_kind="maple", iseed=25,
             learner_kw=dict(prompt_depth=10, num_context=4, vector_std=0.02, use_unified_projection=False,
                             intermediate_dim=32, use_proj_norm=False, use_lora_proj=True), net_kw=base_new, with_f64=True, **F_)
    run_case("rd64_cocoop_n4_d10_i128", eos=2, wseed=21, net_kind="cocoop", iseed=26,
             learner_kw=dict(prompt_depth=10, num_context=4, vector_std=0.02, use_unified_projection=False,
                             intermediate_dim=128, use_proj_norm=True, use_lora_proj=False, norm_image_features=True),
             net_kw=base_old, with_f64=True, **F_)
    # --- row A16 at full size: the reference's own SharedAttn / SharedSeparate settings (configs/model/shared_attn_clipseg.yaml:9-27: 4 context
    # tokens, 1280-wide shared prompt through TransformerEncoderLayer(nhead 16, ff 1536, norm_first), new last layer; dropout 0 for a
    # deterministic fixture; depth 2 so that the deep-prompt path of both towers is in it; shared_separate_clipseg.yaml at shared_dim 64)
    run_case("rd64_sharedattn_n4_d2_newlast", eos=2, wseed=21, net_kind="shared_attn", iseed=29,
             learner_kw=dict(prompt_depth=2, num_context=4, vector_std=0.02, use_unified_projection=False,
                             _tlayer=dict(nhead=16, dim_feedforward=1536, dropout=0.0, norm_first=True)), net_kw=base_new, big_by_seed=True, **F_)
    run_case("rd64_sharedsep_n4_d2_i64_newlast", eos=2, wseed=21, net_kind="shared_separate", iseed=30,
             learner_kw=dict(prompt_depth=2, num_context=4, vector_std=0.02, shared_dim=64, use_unified_projection=False,
                             intermediate_dim=64, use_proj_norm=True, use_lora_proj=False), net_kw=base_new, **F_)
    # --- BASELINE configs[1] exactly: VPT-10 shallow, 352x352, B = 32 (seed 1).  Compact fixture: the inputs
    # are re-drawn from the seed; kept are loss, the 7 680-float prompt gradient, per-sample integer counts / Dice and
    # every 11th logit.  This is the case whose M = 15 840 rows select the large GEMM tiles of the benchmarked step.
    FB = dict(preset="rd64", B=32, H=352, L=8, M=M, compact=True)
    run_case("rd64_vpt_n10_d1_b32", eos=2, wseed=21, net_kind="vpt", iseed=1,
             learner_kw=dict(prompt_depth=1, num_context=10, vector_std=0.02), net_kw=base_old, **FB)

    # --- BASELINE configs[3], the per-GPU step of the 8-GPU config: MaPLe depth 9, 4 context tokens, new last layer, B = 32 (compact)
    run_case("rd64_maple_n4_d9_newlast_b32", eos=2, wseed=21, net_kind="maple", iseed=100,
             learner_kw=dict(prompt_depth=9, num_context=4, vector_std=0.02, use_unified_projection=False,
                             intermediate_dim=64, use_proj_norm=True, use_lora_proj=False), net_kw=base_new, **FB)
    # --- heavy-tailed stress fixtures (weights.heavy_tails: outlier LayerNorm gains, out_proj / fc2 rows, q / k bias outliers),
    # run through the same reference classes: the regime of trained checkpoints that loosens the scale bounds of the two-piece
    # fp16 operand format.  The B = 1 ones also carry the float64 run (the reference's own fp32 noise grows with the tails);
    # level 2 (`*_tails2`) is the regime where the reference's fp32 run is itself 0.14 away from float64 in the logits.
    run_case("tiny_vpt_n4_d2_tails", eos=2, wseed=11, net_kind="vpt", iseed=16, tails=1,
             learner_kw=dict(prompt_depth=2, num_context=4, vector_std=0.02), net_kw=base_old, with_f64=True, **T)
    run_case("rd64_vpt_n10_d1_tails", eos=2, wseed=21, net_kind="vpt", iseed=27, tails=1,
             learner_kw=dict(prompt_depth=1, num_context=10, vector_std=0.02), net_kw=base_old, with_f64=True, **F_)
    run_case("rd64_maple_n4_d9_newlast_tails", eos=2, wseed=21, net_kind="maple", iseed=28, tails=1,
             learner_kw=dict(prompt_depth=9, num_context=4, vector_std=0.02, use_unified_projection=False,
                             intermediate_dim=64, use_proj_norm=True, use_lora_proj=False), net_kw=base_new, with_f64=True, **F_)
    run_case("rd64_vpt_n10_d1_tails2", eos=2, wseed=21, net_kind="vpt", iseed=27, tails=2,
             learner_kw=dict(prompt_depth=1, num_context=10, vector_std=0.02), net_kw=base_old, with_f64=True, **F_)
    run_case("rd64_vpt_n10_d1_b32_tails", eos=2, wseed=21, net_kind="vpt", iseed=3, tails=1,
             learner_kw=dict(prompt_depth=1, num_context=10, vector_std=0.02), net_kw=base_old, with_f64=True, **FB)

    # --- CRIS (BASELINE configs[2]; reference coop_cris.py) ---------------------------------------------------------
    init = dict(context_initializer="a photo of a", _init_ids=[5, 9, 7, 5])
    run_cris_case("cris_tiny_coop_n4_d1", preset="tiny", wseed=31, learner_kind="coop", iseed=31, B=2, L=7,
                  learner_kw=dict(prompt_depth=1, num_context=4, vector_std=0.02), net_kw=base_old)
    run_cris_case("cris_tiny_coop_init_d3_newlast_idpad", preset="tiny", wseed=31, learner_kind="coop", iseed=32, B=2, L=8,
                  learner_kw=dict(prompt_depth=3, num_context=4, vector_std=0.02, **init), net_kw=base_new,
                  with_attention_mask=False)
    run_cris_case("cris_tiny_cocoop_d2_i8_newlast", preset="tiny", wseed=32, learner_kind="cocoop", iseed=33, B=3, L=6,
                  learner_kw=dict(prompt_depth=2, num_context=4, vector_std=0.02, use_unified_projection=False,
                                  intermediate_dim=8, use_proj_norm=True, use_lora_proj=False, norm_image_features=False),
                  net_kw=base_new)
    run_cris_case("cris_tiny_cocoop_d1_unified_norm", preset="tiny", wseed=32, learner_kind="cocoop", iseed=34, B=2, L=6,
                  learner_kw=dict(prompt_depth=1, num_context=2, vector_std=0.02, use_unified_projection=True,
                                  intermediate_dim=None, use_proj_norm=False, use_lora_proj=False, norm_image_features=True),
                  net_kw=base_old, img_size=128)
    run_cris_case("cris_tiny_coop_trunc_n12_b1", preset="tiny", wseed=31, learner_kind="coop", iseed=35, B=1, L=70,
                  learner_kw=dict(prompt_depth=2, num_context=12, vector_std=0.02), net_kw=base_new)
    # unfrozen projector head (coop_cris.py:88-94: proj.txt and proj.vis[-1] train): weight / bias gradients of both
    run_cris_case("cris_tiny_coop_n4_d2_nofreeze_last", preset="tiny", wseed=31, learner_kind="coop", iseed=36, B=2, L=7,
                  learner_kw=dict(prompt_depth=2, num_context=4, vector_std=0.02),
                  net_kw=dict(use_new_last_layer=False, no_freeze_last_layer=True))
    run_cris_case("cris_rn50_cocoop_n4_d1_newlast", preset="rn50", wseed=41, learner_kind="cocoop", iseed=41, B=1, L=8,
                  learner_kw=dict(prompt_depth=1, num_context=4, vector_std=0.02, use_unified_projection=False,
                                  intermediate_dim=64, use_proj_norm=True, use_lora_proj=False, norm_image_features=False,
                                  context_initializer="a photo of a", _init_ids=[320, 1125, 539, 320]), net_kw=base_new)
    # BASELINE configs[2] geometry at B = 8 (compact): large-M implicit-conv tiles
    run_cris_case("cris_rn50_cocoop_n4_d1_newlast_b8", preset="rn50", wseed=41, learner_kind="cocoop", iseed=2, B=8, L=8,
                  learner_kw=dict(prompt_depth=1, num_context=4, vector_std=0.02, use_unified_projection=False,
                                  intermediate_dim=64, use_proj_norm=True, use_lora_proj=False, norm_image_features=False,
                                  context_initializer="a photo of a", _init_ids=[320, 1125, 539, 320]), net_kw=base_new,
                  compact=True)
    # ... and at the BASELINE batch itself, B = 32 (compact)
    run_cris_case("cris_rn50_cocoop_n4_d1_newlast_b32", preset="rn50", wseed=41, learner_kind="cocoop", iseed=2, B=32, L=8,
                  learner_kw=dict(prompt_depth=1, num_context=4, vector_std=0.02, use_unified_projection=False,
                                  intermediate_dim=64, use_proj_norm=True, use_lora_proj=False, norm_image_features=False,
                                  context_initializer="a photo of a", _init_ids=[320, 1125, 539, 320]), net_kw=base_new,
                  compact=True)


if __name__ == "__main__":
    main()
