"""Frozen DenseCLIP weights: a parameter tree under the reference's module names + device-resident GEMM-ready forms.

The reference segmentor (``src/models/components/denseclip/denseclip.py:75-108``) owns ``backbone`` (``CLIPVisionTransformer``),
``text_encoder`` (``CLIPTextContextEncoder``), ``context_decoder`` (``ContextDecoder``) and the parameters ``contexts`` / ``gamma``;
checkpoints carry those names.  This module keeps them (``state_dict`` drop-in) and contains no forward math.  ``prepared()`` builds
what the HIP path consumes: the OpenAI-CLIP residual blocks as packed ``LayerWeights`` (same layer node as the CLIPSeg / CRIS towers),
the patch conv as a GEMM matrix, every ``ConvTranspose2d(k=2, s=2)`` of the FPN as a ``[(dy, dx, co), ci]`` GEMM matrix with the eval-mode
``SyncBatchNorm`` behind the first one folded in (models.py:583-589), ``proj`` / ``text_projection`` transposed for NT GEMMs.
"""
from __future__ import annotations

from typing import Any, Mapping

import torch
import torch.nn.functional as F
from torch import nn

from . import hip
from .backbone import _Embedding, _Node
from .cris_backbone import linear_matrices
from .denseclip_config import DenseCLIPConfig
from .ops import LayerWeights
from .weights import denseclip_param_specs, init_denseclip_state_dict

BN_EPS = 1e-5
TOP_LEVEL = ("contexts", "gamma")   # owned by the segmentor itself (denseclip.py:106-108): registered on the net, not here


def resblock_weights(P: Mapping[str, torch.Tensor], p: str) -> LayerWeights:
    """``ResidualAttentionBlock`` (models.py:391-431): nn.MultiheadAttention's packed in-projection is the packed QKV of the layer node."""
    c = lambda k: P[k].contiguous()  # noqa: E731
    tr = lambda k: P[k].t().contiguous()  # noqa: E731
    return LayerWeights(
        ln1_w=c(f"{p}.ln_1.weight"), ln1_b=c(f"{p}.ln_1.bias"), wqkv=c(f"{p}.attn.in_proj_weight"), bqkv=c(f"{p}.attn.in_proj_bias"),
        wo=c(f"{p}.attn.out_proj.weight"), bo=c(f"{p}.attn.out_proj.bias"), ln2_w=c(f"{p}.ln_2.weight"), ln2_b=c(f"{p}.ln_2.bias"),
        w1=c(f"{p}.mlp.c_fc.weight"), b1=c(f"{p}.mlp.c_fc.bias"), w2=c(f"{p}.mlp.c_proj.weight"), b2=c(f"{p}.mlp.c_proj.bias"),
        wqkv_t=tr(f"{p}.attn.in_proj_weight"), wo_t=tr(f"{p}.attn.out_proj.weight"), w1_t=tr(f"{p}.mlp.c_fc.weight"), w2_t=tr(f"{p}.mlp.c_proj.weight"))


def tconv_matrices(w: torch.Tensor, b: torch.Tensor, bn: tuple[torch.Tensor, ...] | None = None):
    """``nn.ConvTranspose2d(Ci, Co, 2, 2)`` weight [Ci, Co, 2, 2] -> FrozenLinear over W[(dy, dx, co), ci] (kernel == stride: every output pixel
    (2y + dy, 2x + dx) is one row of the input times one [Ci, Co] slice); ``bn`` = (weight, bias, running_mean, running_var) of an eval-mode
    BatchNorm behind it, folded into rows and bias."""
    ci, co = w.shape[:2]
    wm = w.permute(2, 3, 1, 0).reshape(4 * co, ci)
    bias = b.repeat(4)
    if bn is not None:
        g, beta, mean, var = bn
        s = g / torch.sqrt(var + BN_EPS)
        wm = wm * s.repeat(4)[:, None]
        bias = (bias - mean.repeat(4)) * s.repeat(4) + beta.repeat(4)
    return linear_matrices(wm, bias)


class DenseCLIPWeights(_Node):
    def __init__(self, config: DenseCLIPConfig, state_dict: Mapping[str, torch.Tensor] | None = None, seed: int = 0):
        super().__init__()
        self.config = config
        sd = state_dict if state_dict is not None else init_denseclip_state_dict(config, seed)
        for name, shape, _, _ in denseclip_param_specs(config):
            if name in TOP_LEVEL:
                continue
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
        self.text_encoder.context_length = config.text_context_length   # read by the segmentor (denseclip.py:104)
        self.text_encoder.embed_dim = config.embed_dim
        self._prep: dict[str, Any] | None = None
        self._prep_key = None
        self._plist: list | None = None
        self._const: dict[tuple, torch.Tensor] = {}

    @classmethod
    def from_spec(cls, spec: Any, config: DenseCLIPConfig | None = None) -> "DenseCLIPWeights":
        """``spec``: a ``DenseCLIPWeights``; a mapping {preset|config, seed, state_dict}; ``"random:<vitb16_640|tiny>[:seed=N]"``; or a path to a
        ``torch.save``d state dict under the reference's names (an mmseg DenseCLIP checkpoint's ``state_dict``: the neck / head entries are
        ignored, they are outside the path)."""
        if isinstance(spec, DenseCLIPWeights):
            return spec
        if isinstance(spec, Mapping):
            cfg = spec.get("config", config) or _preset(spec.get("preset", "vitb16_640"))
            cfg = cfg if isinstance(cfg, DenseCLIPConfig) else DenseCLIPConfig.from_dict(cfg)
            return cls(cfg, spec.get("state_dict"), seed=int(spec.get("seed", 0)))
        if isinstance(spec, str) and spec.startswith("random:"):
            parts = spec.split(":")[1:]
            opts = dict(p.split("=") for p in parts[1:])
            return cls(config or _preset(parts[0]), None, seed=int(opts.get("seed", 0)))
        cfg = config or _preset("vitb16_640")
        sd = torch.load(spec, map_location="cpu", weights_only=False)
        if isinstance(sd, Mapping) and "state_dict" in sd:
            sd = sd["state_dict"]
        wanted = {name for name, _, _, _ in denseclip_param_specs(cfg)}
        full = init_denseclip_state_dict(cfg, 0)
        missing = sorted(wanted - set(sd) - set(TOP_LEVEL))
        if missing:
            raise RuntimeError(f"{spec}: missing DenseCLIP tensors {missing[:4]} (+{max(0, len(missing) - 4)} more)")
        full.update({k: v.float() for k, v in sd.items() if k in wanted})
        return cls(cfg, full)

    # ------------------------------------------------------------------ GEMM-ready frozen weights
    def prepared(self) -> dict[str, Any]:
        dev = self.backbone.class_embedding.device
        if self._plist is None:
            self._plist = list(self.parameters())
        key = (dev, sum(p._version for p in self._plist if not p.requires_grad))
        if self._prep is not None and self._prep_key == key:
            return self._prep
        loose = [n for n, p in self.named_parameters() if p.requires_grad and not n.startswith("context_decoder.")]
        if loose:
            raise NotImplementedError(
                "only the prompt-tuning path is implemented: the CLIP towers of DenseCLIP must be frozen (contexts, gamma and optionally the "
                f"context decoder train); fine-tuning the backbone is outside the hot path (SURVEY.md §8): {loose[:3]}")
        P = {k: v.detach() for k, v in self.state_dict().items()}
        cfg = self.config
        b = "backbone"
        prep: dict[str, Any] = {
            "patch_w": hip.mark_frozen(P[f"{b}.conv1.weight"].reshape(cfg.width, -1).contiguous()),
            "ln_pre": (P[f"{b}.ln_pre.weight"].contiguous(), P[f"{b}.ln_pre.bias"].contiguous()),
            "ln_post": (P[f"{b}.ln_post.weight"].contiguous(), P[f"{b}.ln_post.bias"].contiguous()),
            "vision_layers": [resblock_weights(P, f"{b}.transformer.resblocks.{i}") for i in range(cfg.layers)],
            "proj": linear_matrices(P[f"{b}.proj"].t(), None),                                         # x @ proj == x (proj^T)^T
            "fpn1_gn": (P[f"{b}.fpn1.0.weight"].contiguous(), P[f"{b}.fpn1.0.bias"].contiguous()),
            "fpn1_t1": tconv_matrices(P[f"{b}.fpn1.1.weight"], P[f"{b}.fpn1.1.bias"],
                                      (P[f"{b}.fpn1.2.weight"], P[f"{b}.fpn1.2.bias"], P[f"{b}.fpn1.2.running_mean"], P[f"{b}.fpn1.2.running_var"])),
            "fpn1_t2": tconv_matrices(P[f"{b}.fpn1.4.weight"], P[f"{b}.fpn1.4.bias"]),
            "fpn2_gn": (P[f"{b}.fpn2.0.weight"].contiguous(), P[f"{b}.fpn2.0.bias"].contiguous()),
            "fpn2_t": tconv_matrices(P[f"{b}.fpn2.1.weight"], P[f"{b}.fpn2.1.bias"]),
            "fpn3_gn": (P[f"{b}.fpn3.weight"].contiguous(), P[f"{b}.fpn3.bias"].contiguous()),
            "fpn4_gn": (P[f"{b}.fpn4.0.weight"].contiguous(), P[f"{b}.fpn4.0.bias"].contiguous()),
            "text_layers": [resblock_weights(P, f"text_encoder.transformer.resblocks.{i}") for i in range(cfg.transformer_layers)],
            "text_projection": linear_matrices(P["text_encoder.text_projection"].t(), None),
        }
        self._prep, self._prep_key = prep, key
        self._const.clear()
        hip._built(device=dev)   # built on whichever stream asked first; other streams (the text encoder's side stream) read it next
        return prep

    def position_table(self, H: int, W: int) -> torch.Tensor:
        """[1 + H*W, C] (models.py:682-692): the spatial rows bilinearly resized from the checkpoint grid (align_corners=False), the class
        embedding added to the CLS row (the token holds it once already).  Frozen, so computed once per size."""
        pos = self.backbone.positional_embedding.detach()
        cls = self.backbone.class_embedding.detach()
        key = ("pos", H, W, pos.device, pos._version, cls._version)
        if key not in self._const:
            g, Cc = self.config.grid, pos.shape[1]
            sp = pos[1:]
            if (H, W) != (g, g):
                sp = F.interpolate(sp.reshape(1, g, g, Cc).permute(0, 3, 1, 2), size=(H, W), mode="bilinear").reshape(Cc, H * W).t()
            self._const[key] = torch.cat(((pos[0] + cls)[None], sp), 0).contiguous()
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


def _preset(name: str) -> DenseCLIPConfig:
    if name == "tiny":
        return DenseCLIPConfig.tiny()
    if name in ("vitb16_640", "vit-b", "vitb16"):
        return DenseCLIPConfig.vitb16_640()
    raise ValueError(f"unknown DenseCLIP preset {name!r}")
