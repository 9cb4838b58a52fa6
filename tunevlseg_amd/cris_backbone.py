"""Frozen CRIS weights: a parameter tree under the reference's module names + device-resident GEMM-ready forms.

The reference ``CRIS`` module (src/models/components/cris_model/__init__.py:20-77) owns ``backbone`` (CLIP-RN50 + text
transformer), ``neck`` (FPN), ``decoder`` (TransformerDecoder) and ``proj`` (Projector); checkpoints carry those names.
This module keeps the names (``state_dict`` drop-in) and contains no forward math.  ``prepared()`` builds what the HIP path
consumes: eval-mode BatchNorm folded into the preceding conv / linear (the model is frozen and in eval mode, reference
coop_cris.py:66-68), 3x3 kernels laid out as im2col GEMM matrices (+ the tap-flipped transpose for data gradients), 1x1
convs as plain matrices (+ transposes), attention projections packed per use.
"""
from __future__ import annotations

import os

import math
from typing import Any, Mapping

import torch
import torch.nn.functional as F
from torch import nn

from . import hip
from .backbone import _Embedding, _Node
from .cris_config import CRISConfig
from .cris_ops import FrozenConv3, FrozenLinear
from .ops import LayerWeights
from .weights import cris_param_specs, init_cris_state_dict

BN_EPS = 1e-5


def _pad_cols(m: torch.Tensor, mult: int = 4) -> torch.Tensor:
    k = m.shape[1]
    kp = (k + mult - 1) // mult * mult
    if kp == k:
        return m.contiguous()
    out = torch.zeros((m.shape[0], kp), device=m.device, dtype=m.dtype)
    out[:, :k] = m
    return out


COORD_PAD = int(os.environ.get("TVL_CRIS_COORD_PAD", "30"))   # zero channels behind CoordConv's (x, y): vis_dim + 2 + 30 is a multiple of 32 for vis_dim % 32 == 0

TRAINABLE_HEAD = ("proj.txt.weight", "proj.txt.bias", "proj.vis.4.weight", "proj.vis.4.bias")   # no_freeze_last_layer


def conv3_matrices(w4: torch.Tensor, b: torch.Tensor | None, need_dgrad: bool) -> FrozenConv3:
    """[Cout, Cin, 3, 3] -> GEMM operands.  Forward columns are (ky, kx, ci); the data-gradient matrix is
    Wd[ci, (ky, kx, co)] = w[co, ci, 2-ky, 2-kx] so that dX = im2col(dY) . Wd^T (transposed conv, stride 1)."""
    cout, cin = w4.shape[:2]
    wm = _pad_cols(w4.permute(0, 2, 3, 1).reshape(cout, 9 * cin))
    wd = _pad_cols(w4.flip(2, 3).permute(1, 2, 3, 0).reshape(cin, 9 * cout)) if need_dgrad else None
    # persistent, never written again: hip.conv3x3 keeps their two-piece fp16 images
    return FrozenConv3(hip.mark_frozen(wm), None if b is None else b.contiguous(), None if wd is None else hip.mark_frozen(wd), cin, cout)


def linear_matrices(w2: torch.Tensor, b: torch.Tensor | None) -> FrozenLinear:
    w2 = w2.contiguous()
    # persistent, never written again: lets hip.gemm keep their two-piece fp16 images (hip.weight_h2_cached)
    return FrozenLinear(hip.mark_frozen(w2), None if b is None else b.contiguous(), hip.mark_frozen(w2.t().contiguous()))


def _numel(shape) -> int:
    n = 1
    for d in shape:
        n *= int(d)
    return n


class CRISWeights(_Node):
    def __init__(self, config: CRISConfig, state_dict: Mapping[str, torch.Tensor] | None = None, seed: int = 0):
        super().__init__()
        self.config = config
        sd = state_dict if state_dict is not None else init_cris_state_dict(config, seed)
        for name, shape, _, _ in cris_param_specs(config):
            parts = name.split(".")
            node: nn.Module = self
            for part in parts[:-1]:
                if part not in node._modules:
                    node.add_module(part, _Embedding() if part == "token_embedding" else _Node())
                node = node._modules[part]
            t = sd[name].detach().to(torch.float32).reshape(shape).clone()
            if parts[-1] in ("running_mean", "running_var"):
                node.register_buffer(parts[-1], t)
            else:
                node.register_parameter(parts[-1], nn.Parameter(t))
        # attributes the reference net reads (coop_cris.py:42-47, 128, 154-181)
        self.backbone.transformer.layers = config.transformer_layers
        self.backbone.transformer.width = config.transformer_width
        self.backbone.visual.output_dim = config.embed_dim
        self.proj.in_dim = config.vis_dim // 2
        self._prep: dict[str, Any] | None = None
        self._prep_key = None
        self._plist: list | None = None   # flat parameter list (walking the module tree per prepared() call cost ~1 ms)
        self._const: dict[tuple, torch.Tensor] = {}

    # ------------------------------------------------------------------ construction
    @classmethod
    def from_spec(cls, spec: Any, overrides: Mapping[str, Any] | None = None) -> "CRISWeights":
        """``spec``: a ``CRISWeights``; a mapping {preset|config, seed, state_dict}; ``"random:<rn50|tiny>[:seed=N]"``; or a
        path to a ``torch.save``d state dict under the reference's names (``pretrain/cris_best_single.pth`` after
        ``scripts/process_cris_checkpoint.py``).  ``overrides``: CRISConfig fields from ``model_cfg`` (img_size, fpn_in ...)."""
        if isinstance(spec, CRISWeights):
            return spec
        ov = {k: v for k, v in dict(overrides or {}).items() if k in CRISConfig.__dataclass_fields__ and v is not None}
        for k in ("vision_layers", "fpn_in", "fpn_out"):
            if k in ov:
                ov[k] = tuple(ov[k])
        if isinstance(spec, Mapping):
            cfg = spec["config"] if "config" in spec else _preset(spec.get("preset", "rn50"))
            cfg = cfg if isinstance(cfg, CRISConfig) else CRISConfig.from_dict(cfg)
            for k, v in ov.items():
                setattr(cfg, k, v)
            return cls(cfg, spec.get("state_dict"), seed=int(spec.get("seed", 0)))
        if isinstance(spec, str) and spec.startswith("random:"):
            parts = spec.split(":")[1:]
            opts = dict(p.split("=") for p in parts[1:])
            cfg = _preset(parts[0])
            for k, v in ov.items():
                setattr(cfg, k, v)
            return cls(cfg, None, seed=int(opts.get("seed", 0)))
        cfg = (overrides or {}).get("config") if isinstance((overrides or {}).get("config"), CRISConfig) else _preset("rn50")
        for k, v in ov.items():
            setattr(cfg, k, v)
        return cls(cfg, cls.read_checkpoint(spec, cfg))

    @staticmethod
    def read_checkpoint(path, cfg: CRISConfig, seed: int = 0) -> dict[str, torch.Tensor]:
        """A weight file under the reference's names.  Two forms reach ``clip_pretrain`` (``configs/model/coop/cris.yaml``):
        OpenAI's TorchScript CLIP archive ``pretrain/RN50.pt`` (the reference reads it with ``torch.jit.load(...).state_dict()`` and
        ``build_model``, ``cris_model/__init__.py:66-70``) -- keys ``visual.*``, ``transformer.*``, ``token_embedding.weight`` ...
        without the ``backbone.`` prefix and without neck / decoder / projector -- or a full CRIS state dict.  CLIP-only files are
        prefixed; what they do not contain starts from the seeded initialisation and is expected from ``cris_pretrain``."""
        try:
            sd = torch.jit.load(path, map_location="cpu").state_dict()
        except Exception:
            try:
                sd = torch.load(path, map_location="cpu", weights_only=False)
            except Exception as e:  # pragma: no cover - needs a corrupt file
                raise RuntimeError(f"cannot load CRIS / CLIP weights from {path!r} ({type(e).__name__}: {e}). Pass the CLIP RN50 archive, a "
                                   "state-dict file under the reference's module names, or 'random:rn50:seed=0'.") from e
        if isinstance(sd, Mapping) and "state_dict" in sd and not any(k.startswith(("backbone.", "visual.")) for k in sd):
            sd = sd["state_dict"]
        sd = {k[len("module."):] if k.startswith("module.") else k: v for k, v in sd.items()}
        if not any(k.startswith("backbone.") for k in sd):  # a bare CLIP model
            skip = ("input_resolution", "context_length", "vocab_size")  # scalars of the TorchScript archive
            sd = {f"backbone.{k}": v for k, v in sd.items() if k not in skip}
        full = init_cris_state_dict(cfg, seed)
        wanted = {name: tuple(shape) for name, shape, _, _ in cris_param_specs(cfg)}
        bad = [k for k, v in sd.items() if k in wanted and tuple(v.shape) != wanted[k] and v.numel() != _numel(wanted[k])]
        if bad:
            raise RuntimeError(f"{path}: tensors do not match the {type(cfg).__name__} geometry: {bad[:4]}")
        unknown = [k for k in sd if k not in wanted and not k.endswith("num_batches_tracked")]
        if unknown:
            raise RuntimeError(f"{path}: unexpected keys {unknown[:4]} (+{max(0, len(unknown) - 4)} more)")
        full.update({k: v.float() for k, v in sd.items() if k in wanted})
        return full

    # ------------------------------------------------------------------ GEMM-ready frozen weights
    def prepared(self) -> dict[str, Any]:
        dev = self.backbone.logit_scale.device
        # the projector head may train (no_freeze_last_layer, coop_cris.py:88-94): the net reads those four tensors from the parameter
        # tree, so they neither invalidate nor enter the prepared matrices
        if self._plist is None:
            self._plist = list(self.parameters())
        key = (dev, sum(p._version for p in self._plist if not p.requires_grad))
        if self._prep is not None and self._prep_key == key:
            return self._prep
        loose = [n for n, p in self.named_parameters() if p.requires_grad and n not in TRAINABLE_HEAD]
        if loose:
            raise NotImplementedError(
                "only the prompt-tuning path is implemented: the CRIS model must be frozen (freeze_all=True) apart from the projector "
                f"head; fine-tuning (e2e_cris) is outside the hot path (SURVEY.md §8): {loose[:3]}")
        P = {k: v.detach() for k, v in self.state_dict().items()}
        cfg = self.config

        def fold(w: torch.Tensor, bn: str):
            s = P[f"{bn}.weight"] / torch.sqrt(P[f"{bn}.running_var"] + BN_EPS)
            return w * s.view(-1, *([1] * (w.dim() - 1))), P[f"{bn}.bias"] - P[f"{bn}.running_mean"] * s

        def conv1_bn(conv: str, bn: str) -> FrozenLinear:
            w, b = fold(P[conv], bn)
            return linear_matrices(w.reshape(w.shape[0], -1), b)

        def conv3_bn(conv: str, bn: str, dgrad: bool, pad_cin: int = 0) -> FrozenConv3:
            w, b = fold(P[conv], bn)
            if pad_cin:
                w = torch.cat((w, torch.zeros((w.shape[0], pad_cin, 3, 3), device=w.device)), 1)
            return conv3_matrices(w, b, dgrad)

        def conv_layer(p: str, k: int, dgrad: bool, pad_cin: int = 0):
            return conv3_bn(f"{p}.0.weight", f"{p}.1", dgrad, pad_cin) if k == 3 else conv1_bn(f"{p}.0.weight", f"{p}.1")

        v = "backbone.visual"
        prep: dict[str, Any] = {}
        prep["stem"] = [conv3_bn(f"{v}.conv{i}.weight", f"{v}.bn{i}", False) for i in (1, 2, 3)]
        blocks = []
        for li, n in enumerate(cfg.vision_layers, start=1):
            for bi in range(n):
                p = f"{v}.layer{li}.{bi}"
                blocks.append({
                    "stride": 2 if (li > 1 and bi == 0) else 1, "stage_end": bi == n - 1,
                    "c1": conv1_bn(f"{p}.conv1.weight", f"{p}.bn1"), "c2": conv3_bn(f"{p}.conv2.weight", f"{p}.bn2", False),
                    "c3": conv1_bn(f"{p}.conv3.weight", f"{p}.bn3"),
                    "down": conv1_bn(f"{p}.downsample.0.weight", f"{p}.downsample.1") if f"{p}.downsample.0.weight" in P else None})
        prep["blocks"] = blocks
        ap = f"{v}.attnpool"
        prep["attnpool"] = {
            "qkv": linear_matrices(torch.cat((P[f"{ap}.q_proj.weight"], P[f"{ap}.k_proj.weight"], P[f"{ap}.v_proj.weight"]), 0),
                                   torch.cat((P[f"{ap}.q_proj.bias"], P[f"{ap}.k_proj.bias"], P[f"{ap}.v_proj.bias"]), 0)),
            "c_proj": linear_matrices(P[f"{ap}.c_proj.weight"], P[f"{ap}.c_proj.bias"]),
            "connect": conv1_bn(f"{ap}.connect.0.weight", f"{ap}.connect.1"),
        }
        # text tower: same layer node as the CLIPSeg towers (pre-LN, QuickGELU, packed in_proj)
        tl = []
        for i in range(cfg.transformer_layers):
            p = f"backbone.transformer.resblocks.{i}"
            c = lambda k: P[k].contiguous()  # noqa: E731
            tr = lambda k: P[k].t().contiguous()  # noqa: E731
            tl.append(LayerWeights(
                ln1_w=c(f"{p}.ln_1.weight"), ln1_b=c(f"{p}.ln_1.bias"), wqkv=c(f"{p}.attn.in_proj_weight"), bqkv=c(f"{p}.attn.in_proj_bias"),
                wo=c(f"{p}.attn.out_proj.weight"), bo=c(f"{p}.attn.out_proj.bias"), ln2_w=c(f"{p}.ln_2.weight"), ln2_b=c(f"{p}.ln_2.bias"),
                w1=c(f"{p}.mlp.c_fc.weight"), b1=c(f"{p}.mlp.c_fc.bias"), w2=c(f"{p}.mlp.c_proj.weight"), b2=c(f"{p}.mlp.c_proj.bias"),
                wqkv_t=tr(f"{p}.attn.in_proj_weight"), wo_t=tr(f"{p}.attn.out_proj.weight"), w1_t=tr(f"{p}.mlp.c_fc.weight"),
                w2_t=tr(f"{p}.mlp.c_proj.weight")))
        prep["text_layers"] = tl
        prep["text_projection"] = linear_matrices(P["backbone.text_projection"].t(), None)  # x @ P == x (P^T)^T
        # neck (layers.py:359-445); only the text-dependent convs need data gradients
        w, b = fold(P["neck.txt_proj.0.weight"], "neck.txt_proj.1")
        nl_s = P["neck.norm_layer.0.weight"] / torch.sqrt(P["neck.norm_layer.0.running_var"] + BN_EPS)
        prep["neck"] = {
            "txt_proj": linear_matrices(w, b),
            "f1_v_proj": conv_layer("neck.f1_v_proj", 1, False),
            "norm_scale": nl_s.contiguous(), "norm_shift": (P["neck.norm_layer.0.bias"] - P["neck.norm_layer.0.running_mean"] * nl_s).contiguous(),
            "f2_v_proj": conv_layer("neck.f2_v_proj", 3, False), "f2_cat": conv_layer("neck.f2_cat", 1, True),
            "f3_v_proj": conv_layer("neck.f3_v_proj", 3, False), "f3_cat": conv_layer("neck.f3_cat", 1, True),
            "f4_proj5": conv_layer("neck.f4_proj5", 3, True), "f4_proj4": conv_layer("neck.f4_proj4", 3, True),
            "f4_proj3": conv_layer("neck.f4_proj3", 3, True), "aggr": conv_layer("neck.aggr", 1, True),
            # CoordConv: the two coordinate channels ride along as real channels, + zero channels up to a multiple of 32 (514 -> 544) so
            # that the conv and its data gradient (N = 544) take the two-fp16-piece implicit GEMM (hip.conv3x3_takes_h2)
            "coord0": conv_layer("neck.coordconv.0.conv1", 3, True, pad_cin=COORD_PAD), "coord1": conv_layer("neck.coordconv.1", 3, True),
        }
        # decoder (layers.py:278-356)
        D = cfg.vis_dim
        dl = []
        for i in range(cfg.num_layers):
            p = f"decoder.layers.{i}"
            sw, sb = P[f"{p}.self_attn.in_proj_weight"], P[f"{p}.self_attn.in_proj_bias"]
            cw, cb = P[f"{p}.multihead_attn.in_proj_weight"], P[f"{p}.multihead_attn.in_proj_bias"]
            ln = lambda n: (P[f"{p}.{n}.weight"].contiguous(), P[f"{p}.{n}.bias"].contiguous())  # noqa: E731
            dl.append({
                "norm1": ln("norm1"), "norm2": ln("norm2"), "norm3": ln("norm3"), "self_attn_norm": ln("self_attn_norm"),
                "cross_attn_norm": ln("cross_attn_norm"), "ffn_norm": ln("ffn.3"),
                "sa_qk": linear_matrices(sw[:2 * D], sb[:2 * D]), "sa_v": linear_matrices(sw[2 * D:], sb[2 * D:]),
                "sa_o": linear_matrices(P[f"{p}.self_attn.out_proj.weight"], P[f"{p}.self_attn.out_proj.bias"]),
                "ca_q": linear_matrices(cw[:D], cb[:D]), "ca_k": linear_matrices(cw[D:2 * D], cb[D:2 * D]),
                "ca_v": linear_matrices(cw[2 * D:], cb[2 * D:]),
                "ca_o": linear_matrices(P[f"{p}.multihead_attn.out_proj.weight"], P[f"{p}.multihead_attn.out_proj.bias"]),
                "ffn0": linear_matrices(P[f"{p}.ffn.0.weight"], P[f"{p}.ffn.0.bias"]),
                "ffn4": linear_matrices(P[f"{p}.ffn.4.weight"], P[f"{p}.ffn.4.bias"]),
            })
        prep["decoder_layers"] = dl
        prep["decoder_norm"] = (P["decoder.norm.weight"].contiguous(), P["decoder.norm.bias"].contiguous())
        # projector (layers.py:71-119)
        prep["proj"] = {
            "vis1": conv_layer("proj.vis.1", 3, True), "vis3": conv_layer("proj.vis.3", 3, True),
            "vis4": linear_matrices(P["proj.vis.4.weight"].reshape(P["proj.vis.4.weight"].shape[0], -1), P["proj.vis.4.bias"]),
            "txt": linear_matrices(P["proj.txt.weight"], P["proj.txt.bias"]),
        }
        self._prep, self._prep_key = prep, key
        self._const.clear()
        hip._built(device=dev)   # built on whichever stream asked first; other streams (the text encoder's side stream) read it next
        return prep

    # ------------------------------------------------------------------ input-independent constants (cached per size)
    def attnpool_pos(self, H: int, W: int) -> torch.Tensor:
        """Bicubic-resized attention-pool position table [H*W, C] (clip.py:102-146); frozen, so computed once per size."""
        pos = self.backbone.visual.attnpool.positional_embedding.detach()
        key = ("appos", H, W, pos.device, pos._version)
        if key not in self._const:
            sp = self.config.image_resolution // 32
            t = pos[-sp * sp:].reshape(1, sp, sp, -1).permute(0, 3, 1, 2)
            t = F.interpolate(t, size=(H, W), mode="bicubic", align_corners=False)
            self._const[key] = t.flatten(2)[0].t().contiguous()
        return self._const[key]

    def pos2d(self, d: int, H: int, W: int) -> torch.Tensor:
        """Fixed 2-D sin/cos code [H*W, d] (layers.py:187-236): x in the first half of the channels, y in the second."""
        dev = self.backbone.logit_scale.device
        key = ("pos2d", d, H, W, dev)
        if key not in self._const:
            pe = torch.zeros(d, H, W)
            half = d // 2
            mul = 1e-4 ** (torch.arange(0, half, 2, dtype=torch.float32) / half)
            aw = torch.arange(W, dtype=torch.float32)[:, None] * mul
            ah = torch.arange(H, dtype=torch.float32)[:, None] * mul
            pe[0:half:2] = torch.sin(aw).t()[:, None, :].expand(-1, H, -1)
            pe[1:half:2] = torch.cos(aw).t()[:, None, :].expand(-1, H, -1)
            pe[half::2] = torch.sin(ah).t()[:, :, None].expand(-1, -1, W)
            pe[half + 1::2] = torch.cos(ah).t()[:, :, None].expand(-1, -1, W)
            self._const[key] = pe.reshape(d, H * W).t().contiguous().to(dev)
        return self._const[key]

    def pos1d(self, d: int, length: int) -> torch.Tensor:
        """Fixed 1-D sin/cos code [length, d] (layers.py:148-185)."""
        dev = self.backbone.logit_scale.device
        key = ("pos1d", d, length, dev)
        if key not in self._const:
            pe = torch.zeros(length, d)
            ang = torch.arange(length, dtype=torch.float32)[:, None] * (1e-4 ** (torch.arange(0, d, 2, dtype=torch.float32) / d))
            pe[:, 0::2], pe[:, 1::2] = torch.sin(ang), torch.cos(ang)
            self._const[key] = pe.contiguous().to(dev)
        return self._const[key]

    def coords(self, B: int, H: int, W: int) -> torch.Tensor:
        """CoordConv channels (layers.py:52-64) as a [B*H*W, 2 + COORD_PAD] matrix: (x, y, 0, ..., 0), linspace(-1, 1)."""
        dev = self.backbone.logit_scale.device
        key = ("coords", B, H, W, dev)
        if key not in self._const:
            yy, xx = torch.meshgrid(torch.linspace(-1, 1, H), torch.linspace(-1, 1, W), indexing="ij")
            m = torch.zeros(H * W, 2 + COORD_PAD)
            m[:, 0], m[:, 1] = xx.reshape(-1), yy.reshape(-1)
            self._const[key] = m.repeat(B, 1).contiguous().to(dev)
        return self._const[key]

    def _apply(self, fn, *a, **k):
        self._prep = None
        self._plist = None
        self._const = {}
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._prep = None
        self._plist = None
        self._const = {}
        return super().load_state_dict(*a, **k)


def _preset(name: str) -> CRISConfig:
    if name == "tiny":
        return CRISConfig.tiny()
    if name in ("rn50", "cris"):
        return CRISConfig.rn50()
    raise ValueError(f"unknown CRIS preset {name!r}")


def count_params(cfg: CRISConfig) -> int:
    return sum(math.prod(s) for _, s, _, _ in cris_param_specs(cfg))
