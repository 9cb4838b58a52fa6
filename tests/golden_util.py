"""Helpers shared by the parity tests: load golden fixtures, rebuild learner parameters."""
from __future__ import annotations

import json
import re
from pathlib import Path

import numpy as np
import torch

from tunevlseg_amd.config import CLIPSegConfig
from tunevlseg_amd.cris_config import CRISConfig
from tunevlseg_amd.denseclip_config import DenseCLIPConfig
from tunevlseg_amd.weights import init_clipseg_state_dict, init_cris_state_dict, init_denseclip_state_dict

GOLDEN_DIR = Path(__file__).resolve().parent / "golden"


def golden_names(prefix: str = "") -> list[str]:
    return sorted(p.stem for p in GOLDEN_DIR.glob(f"{prefix}*.npz"))


def load_golden(name: str) -> dict:
    z = np.load(GOLDEN_DIR / f"{name}.npz")
    fx = {k: z[k] for k in z.files if k != "meta"}
    fx["meta"] = json.loads(bytes(z["meta"]).decode())
    return fx


def config_of(fx) -> CLIPSegConfig:
    m = fx["meta"]
    return CLIPSegConfig.tiny(m["eos_token_id"]) if m["preset"] == "tiny" else CLIPSegConfig.rd64(m["eos_token_id"])


def state_of(fx) -> dict[str, torch.Tensor]:
    sd = init_clipseg_state_dict(config_of(fx), fx["meta"]["weight_seed"], tails=fx["meta"].get("tails", 0))
    chk = float(sum(v.double().abs().sum() for v in sd.values()))
    assert abs(chk - fx["meta"]["weights_checksum"]) <= 1e-6 * abs(chk), "seeded weight draw drifted from the fixture"
    return sd


def redraw(seed: int, std: float, shape) -> torch.Tensor:
    """A large trainable tensor of a fixture that keeps (seed, std, shape) instead of its values (make_goldens.py ``big_by_seed``)."""
    return torch.randn(tuple(int(d) for d in shape), generator=torch.Generator().manual_seed(int(seed))) * float(std)


def trainable_of(fx, requires_grad: bool = True) -> dict[str, torch.Tensor]:
    out = {}
    for k, v in fx.items():
        if k.startswith("param."):
            t = torch.from_numpy(np.array(v)).clone()
            out[k[len("param."):]] = t.requires_grad_(requires_grad)
        elif k.startswith("paramseed."):
            out[k[len("paramseed."):]] = redraw(v[0], v[1], v[2:]).requires_grad_(requires_grad)
    return out


def grad_of(fx, k: str, g: torch.Tensor | None = None):
    """(reference gradient, this path's gradient) in comparable form: whole tensors, or -- for the by-seed tensors, whose fixture keeps every
    61st entry and the absolute sum -- the same subsample of ``g`` (the absolute sum is checked here, to 1e-3)."""
    if "grad." + k in fx:
        return torch.from_numpy(fx["grad." + k]), g
    ref = torch.from_numpy(fx["gradsub." + k])
    if g is None:
        return ref, None
    s = float(fx["gradabs." + k])
    assert abs(float(g.detach().double().abs().sum()) - s) <= 1e-3 * max(s, 1e-12), (k, float(g.detach().double().abs().sum()), s)
    return ref, g.detach().flatten()[::61]


def oracle_learner(fx, params: dict[str, torch.Tensor]) -> dict:
    """Describe the fixture's learner the way ``oracle.clipseg_oracle`` expects."""
    m = fx["meta"]
    kind = m["net"]
    ctx = params["context_learner.context_vectors"]
    depth = ctx.shape[0]
    learner = {"kind": kind, "ctx": ctx}
    pat = re.compile(r"context_learner\.projection_layers\.(\d+)\.(?:(\d+)\.)?(weight|bias)$")
    layers: dict[int, dict[int, dict[str, torch.Tensor]]] = {}
    for k, t in params.items():
        mm = pat.match(k)
        if mm:
            li, sub, wb = int(mm.group(1)), int(mm.group(2) or 0), mm.group(3)
            layers.setdefault(li, {}).setdefault(sub, {})[wb] = t
    def ops_list(prefix):
        pat2 = re.compile(re.escape(prefix) + r"\.(\d+)\.(?:(\d+)\.)?(weight|bias)$")
        lay: dict[int, dict[int, dict[str, torch.Tensor]]] = {}
        for k, t in params.items():
            mm = pat2.match(k)
            if mm:
                lay.setdefault(int(mm.group(1)), {}).setdefault(int(mm.group(2) or 0), {})[mm.group(3)] = t
        out = []
        for d in range(depth):
            src = lay[d if d in lay else 0]
            ops = []
            for sub in range(max(src) + 1):
                if sub not in src:
                    ops.append(("relu",))
                    continue
                w, b = src[sub]["weight"], src[sub].get("bias")
                ops.append(("linear", w, b) if w.dim() == 2 else ("layernorm", w, b, 1e-5))
            out.append(ops)
        return out

    if kind == "shared_separate":
        learner["tproj"] = ops_list("context_learner.textual_projection_layers")
        learner["vproj"] = ops_list("context_learner.visual_projection_layers")
        return learner
    if kind == "shared_attn":
        tl = m["learner_kw"]["_tlayer"]
        cfg = config_of(fx)
        tlayers = []
        for d in range(depth):
            dd = d if f"context_learner.projection_layers.{d}.linear1.weight" in params else 0
            pre = f"context_learner.projection_layers.{dd}."
            tp = {k[len(pre):]: v for k, v in params.items() if k.startswith(pre)}
            tp.update(nhead=tl["nhead"], norm_first=tl.get("norm_first", False), eps=1e-5)
            tlayers.append(tp)
        learner["tlayers"] = tlayers
        learner["textual_dim"] = cfg.text_config.hidden_size
        return learner
    if layers:
        proj = []
        for d in range(depth):
            src = layers[d if d in layers else 0]  # unified projection: one shared module
            ops = []
            for sub in range(max(src) + 1):
                if sub not in src:
                    ops.append(("relu",))
                    continue
                w, b = src[sub]["weight"], src[sub].get("bias")
                ops.append(("linear", w, b) if w.dim() == 2 else ("layernorm", w, b, 1e-5))
            proj.append(ops)
        learner["proj"] = proj
    if kind == "cocoop":
        learner["norm_image_features"] = m["learner_kw"].get("norm_image_features", True)
    return learner


def new_last_of(fx, params):
    if "additive_decoder_layer.1.weight" not in params:
        return None
    return (params["additive_decoder_layer.1.weight"], params["additive_decoder_layer.1.bias"], params["residual_ratio"])


def inputs_of(fx):
    am = torch.from_numpy(fx["in.attention_mask"]) if "in.attention_mask" in fx else None
    return (torch.from_numpy(fx["in.pixel_values"]), torch.from_numpy(fx["in.input_ids"]), am, torch.from_numpy(fx["in.mask"]))


# ---- CRIS fixtures (meta["family"] == "cris") -----------------------------------------------------------------------
def cris_config_of(fx) -> CRISConfig:
    m = fx["meta"]
    cfg = CRISConfig.tiny() if m["preset"] == "tiny" else CRISConfig.rn50()
    cfg.img_size = m["img_size"]
    return cfg


def cris_state_of(fx) -> dict[str, torch.Tensor]:
    sd = init_cris_state_dict(cris_config_of(fx), fx["meta"]["weight_seed"])
    chk = float(sum(v.double().abs().sum() for v in sd.values()))
    assert abs(chk - fx["meta"]["weights_checksum"]) <= 1e-6 * abs(chk), "seeded weight draw drifted from the fixture"
    return sd


def cris_new_last_of(params):
    if "additive_decoder_layer.0.weight" not in params:
        return None
    return {"w1": params["additive_decoder_layer.0.weight"], "w": params["additive_decoder_layer.2.weight"],
            "b": params["additive_decoder_layer.2.bias"], "ratio": params["residual_ratio"]}


# ---- seeded synthetic inputs; the compact full-batch fixtures store only the seed ----------------------
def synth_inputs(cfg: CLIPSegConfig, B: int, H: int, L: int, seed: int, pad: bool = True):
    """img N(0,1); ids rows padded to L; mask = U(0,1) > 0.7."""
    g = torch.Generator().manual_seed(seed)
    t = cfg.text_config
    pix = torch.randn(B, 3, H, H, generator=g)
    ids = torch.full((B, L), t.pad_token_id, dtype=torch.long)
    am = torch.zeros(B, L, dtype=torch.long)
    eos = 49407 if t.vocab_size > 49407 else t.vocab_size - 1
    if t.eos_token_id != 2:
        eos = t.eos_token_id
    for b in range(B):
        n_words = (L - 2) if not pad else max(1, L - 2 - (b % 3) - (1 if B == 1 else 0))
        words = torch.randint(2, min(t.vocab_size - 2, 40000), (n_words,), generator=g)
        if t.eos_token_id != 2:
            words = words.masked_fill(words == t.eos_token_id, 3)
        row = [t.bos_token_id, *words.tolist(), eos]
        ids[b, : len(row)] = torch.tensor(row)
        am[b, : len(row)] = 1
    mask = (torch.rand(B, 1, H, H, generator=g) > 0.7).float()
    return pix, ids, am, mask


def synth_cris_inputs(cfg: CRISConfig, B: int, L: int, seed: int, with_attention_mask: bool):
    """img N(0,1) at cfg.img_size; ids = [BOS, words, EOS(highest id), 0-pads]; CRIS derives its pad mask either from
    the attention mask or from ``ids == 0`` (cris_model/__init__.py:79-86) -- both branches are exercised."""
    g = torch.Generator().manual_seed(seed)
    H = cfg.img_size
    pix = torch.randn(B, 3, H, H, generator=g)
    ids = torch.zeros(B, L, dtype=torch.long)
    am = torch.zeros(B, L, dtype=torch.long)
    eos, bos = cfg.vocab_size - 1, cfg.vocab_size - 2
    for b in range(B):
        n_words = max(1, L - 2 - (b % 3) - (1 if B == 1 else 0))
        words = torch.randint(1, min(cfg.vocab_size - 2, 40000), (n_words,), generator=g)
        row = [bos, *words.tolist(), eos]
        ids[b, : len(row)] = torch.tensor(row)
        am[b, : len(row)] = 1
    mask = (torch.rand(B, 1, H, H, generator=g) > 0.7).float()
    return pix, ids, (am if with_attention_mask else None), mask


def check_compact_labels(fx, logits: torch.Tensor, isum: torch.Tensor, mask: torch.Tensor, logit_tol: float = 1e-3) -> int:
    """Integer parity of a compact (full-batch) fixture.  The label map ``sigmoid(logit) > 0.5`` must equal the reference's bit
    for bit, except at the pixels the fixture lists as ambiguous (|reference logit| < 1e-4: within fp32 evaluation noise of the
    threshold, so the reference's own fp32 run could land on either side); there the logit itself must agree to ``logit_tol``.
    The kernel's integer TP/FP/FN/TN must be exactly the counts of the label map this path produced, and differ from the
    reference's counts only by the ambiguous pixels that flipped.  Returns the number of flipped pixels."""
    lg = logits.detach().cpu()
    B = lg.shape[0]
    lab = (torch.sigmoid(lg) > 0.5).flatten()
    ref_lab = torch.from_numpy(np.unpackbits(fx["out.label_bits"])[: lab.numel()].astype(bool))
    amb = torch.from_numpy(fx["out.ambiguous_idx"]).long()
    flips = torch.nonzero(lab != ref_lab).flatten()
    outside = set(flips.tolist()) - set(amb.tolist())
    assert not outside, f"{len(outside)} label pixels differ from the reference away from the threshold"
    if amb.numel():
        assert (lg.flatten()[amb] - torch.from_numpy(fx["out.ambiguous_logits"])).abs().max().item() <= logit_tol
    tgt = mask.detach().cpu().long().flatten(1).bool()
    labs = lab.view(B, -1)
    own = torch.stack(((labs & tgt).sum(1), (labs & ~tgt).sum(1), (~labs & tgt).sum(1), (~labs & ~tgt).sum(1)), 1)
    # the kernel thresholds its own fp32 sigmoid: a pixel whose logit is within ~1e-6 of zero may round to exactly 0.5 in one
    # sigmoid implementation and not in another, so allow that many pixels per sample between the two countings
    knife_edge = (lg.flatten(1).abs() < 1e-6).sum(1)
    assert ((isum.cpu() - own).abs().sum(1) <= 2 * knife_edge).all(), "integer TP/FP/FN/TN are not the counts of this path's own label map"
    ref_counts = torch.from_numpy(fx["out.counts"])
    per_sample_flips = torch.zeros(B, dtype=torch.long).index_add_(0, flips // labs.shape[1], torch.ones_like(flips))
    assert ((isum.cpu() - ref_counts).abs().sum(1) <= 2 * (per_sample_flips + knife_edge)).all(), \
        "integer counts differ from the reference beyond the ambiguous pixels"
    return int(flips.numel())


# ---- DenseCLIP fixtures (meta["family"] == "denseclip") ---------------------------------------------------------------
def denseclip_config_of(fx) -> DenseCLIPConfig:
    return DenseCLIPConfig.from_dict(fx["meta"]["config"])


def denseclip_state_of(fx) -> dict[str, torch.Tensor]:
    sd = init_denseclip_state_dict(denseclip_config_of(fx), fx["meta"]["weight_seed"])
    chk = float(sum(v.double().abs().sum() for k, v in sd.items() if k not in ("contexts", "gamma")))
    assert abs(chk - fx["meta"]["weights_checksum"]) <= 1e-6 * abs(chk), "seeded weight draw drifted from the fixture"
    return sd


def synth_denseclip_inputs(cfg: DenseCLIPConfig, B: int, H: int, seed: int, W: int | None = None):
    """img N(0,1) [B, 3, H, W] (W defaults to H); class-name token rows as the reference's ``tokenize(name, context_length)`` lays them out
    (untils.py:173-221): [SOT, words, EOT, 0-pads] with EOT the highest id, 1 .. context_length - 2 words per class; the two seeded
    cotangents of the fixture scalar  L = sum(score_map * Gs) + sum(text_embeddings * Gt)."""
    g = torch.Generator().manual_seed(seed)
    W = H if W is None else W
    pix = torch.randn(B, 3, H, W, generator=g)
    K, N1 = cfg.num_classes, cfg.context_length
    sot, eot = cfg.vocab_size - 2, cfg.vocab_size - 1
    texts = torch.zeros(K, N1, dtype=torch.long)
    for k in range(K):
        n_words = 1 + k % (N1 - 2)
        row = [sot, *torch.randint(1, min(cfg.vocab_size - 2, 40000), (n_words,), generator=g).tolist(), eot]
        texts[k, : len(row)] = torch.tensor(row)
    gs = torch.randn(B, K, H // cfg.patch_size, W // cfg.patch_size, generator=g)
    gt = torch.randn(B, K, cfg.embed_dim, generator=g) * 0.1
    return pix, texts, gs, gt


def denseclip_subsample(name: str, t: torch.Tensor, compact: bool) -> torch.Tensor:
    """What a compact (full-size) fixture keeps of an output: every ``DENSECLIP_STRIDE[name]``-th pixel of the large maps."""
    if not compact or name not in DENSECLIP_STRIDE:
        return t
    s = DENSECLIP_STRIDE[name]
    return t[..., ::s, ::s].contiguous()


DENSECLIP_STRIDE = {"fpn1": 9, "fpn2": 5, "fpn3": 3, "fpn4": 2, "visual_embedding": 3}
