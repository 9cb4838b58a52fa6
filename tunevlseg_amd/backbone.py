"""Frozen CLIPSeg backbone: a parameter tree with HF attribute paths + packed per-layer weights.

The reference holds an HF ``CLIPSegForImageSegmentation`` in ``self.model`` and reaches into it by
attribute (``model.clip.text_model.embeddings.token_embedding``, ``model.decoder.reduces`` ... --
SURVEY.md Appendix B).  This module keeps those paths and the HF ``state_dict`` key names (so HF
checkpoints and reference Lightning checkpoints load by name) but contains NO forward math: the
arithmetic is the HIP kernels driven from ``tunevlseg_amd.nets``.
"""
from __future__ import annotations

import math
from typing import Any, Mapping

import torch
import torch.nn.functional as F
from torch import nn

from . import hip
from .config import CLIPSegConfig
from .ops import LayerWeights
from .weights import clipseg_param_specs, init_clipseg_state_dict


class _Node(nn.Module):
    """Container whose numeric children behave like an ``nn.ModuleList``."""

    def __getitem__(self, i: int):
        return getattr(self, str(i))

    def __len__(self) -> int:
        return sum(1 for k in self._modules if k.isdigit())

    def __iter__(self):
        return (self[i] for i in range(len(self)))


class _Embedding(_Node):
    """Callable like ``nn.Embedding`` (used at construction time by ``context_initializer``)."""

    def forward(self, ids: torch.Tensor) -> torch.Tensor:
        return F.embedding(ids, self.weight)


class CLIPSegBackbone(_Node):
    def __init__(self, config: CLIPSegConfig, state_dict: Mapping[str, torch.Tensor] | None = None, seed: int = 0, tails: int = 0):
        super().__init__()
        self.config = config
        self.extract_layers = tuple(config.extract_layers)
        sd = state_dict if state_dict is not None else init_clipseg_state_dict(config, seed, tails=tails)
        for name, shape, _, _ in clipseg_param_specs(config):
            parts = name.split(".")
            node: nn.Module = self
            for i, part in enumerate(parts[:-1]):
                if part not in node._modules:
                    is_emb = part in ("token_embedding", "position_embedding")
                    node.add_module(part, _Embedding() if is_emb else _Node())
                node = node._modules[part]
            t = sd[name].detach().to(torch.float32).reshape(shape).clone()
            node.register_parameter(parts[-1], nn.Parameter(t))
        # attributes the reference nets read
        self.clip.text_model.eos_token_id = config.text_config.eos_token_id
        self.clip.text_model.config = config.text_config
        self.clip.vision_model.config = config.vision_config
        self.clip.text_model.encoder.config = config.text_config
        self.clip.vision_model.encoder.config = config.vision_config
        self.clip.text_model.encoder.gradient_checkpointing = False
        self.clip.vision_model.encoder.gradient_checkpointing = False
        self.decoder.conditional_layer = config.conditional_layer
        self.clip.text_model.embeddings.register_buffer(
            "position_ids", torch.arange(config.text_config.max_position_embeddings).expand((1, -1)), persistent=False)
        self._prep: dict[str, Any] | None = None
        self._prep_key = None
        self._plist: list | None = None
        self._pos_cache: dict[tuple, torch.Tensor] = {}

    # ------------------------------------------------------------------ construction helpers
    @classmethod
    def from_spec(cls, spec: Any) -> "CLIPSegBackbone":
        """``spec``: a backbone, a mapping {preset|config, seed, eos_token_id}, ``"random:<preset>[:seed=N][:eos=N][:tails=1]"``
        (``tails``: the outlier-channel preset of ``weights.heavy_tails``),
        or a local HF checkpoint directory (``CLIPSegForImageSegmentation.from_pretrained`` layout)."""
        if isinstance(spec, CLIPSegBackbone):
            return spec
        if isinstance(spec, Mapping):
            cfg = _config_from_mapping(spec)
            return cls(cfg, spec.get("state_dict"), seed=int(spec.get("seed", 0)), tails=int(spec.get("tails", 0)))
        if isinstance(spec, str) and spec.startswith("random:"):
            parts = spec.split(":")[1:]
            opts = dict(p.split("=") for p in parts[1:])
            cfg = _config_from_mapping({"preset": parts[0], "eos_token_id": int(opts.get("eos", 2))})
            return cls(cfg, None, seed=int(opts.get("seed", 0)), tails=int(opts.get("tails", 0)))
        return cls._from_hf(spec)

    @classmethod
    def _from_hf(cls, path) -> "CLIPSegBackbone":
        try:
            from transformers import CLIPSegForImageSegmentation

            hf = CLIPSegForImageSegmentation.from_pretrained(path, local_files_only=True)
        except Exception as e:  # pragma: no cover - needs a real checkpoint
            raise RuntimeError(
                f"cannot load CLIPSeg weights from {path!r} offline ({type(e).__name__}: {e}). Pass a local checkpoint "
                "directory, or 'random:rd64:seed=0' for seeded random weights.") from e
        cfg = CLIPSegConfig.from_dict(hf.config.to_dict())
        sd = {k: v for k, v in hf.state_dict().items() if "position_ids" not in k}
        return cls(cfg, sd)

    # ------------------------------------------------------------------ packed weights
    def _layer_weights(self, node) -> LayerWeights:
        a = node.self_attn
        c = lambda t: t.detach().contiguous()  # noqa: E731
        tr = lambda t: t.detach().t().contiguous()  # noqa: E731
        wqkv = torch.cat((a.q_proj.weight, a.k_proj.weight, a.v_proj.weight), 0).detach().contiguous()
        return LayerWeights(
            wqkv_t=tr(wqkv), wo_t=tr(a.out_proj.weight), w1_t=tr(node.mlp.fc1.weight), w2_t=tr(node.mlp.fc2.weight),
            ln1_w=c(node.layer_norm1.weight), ln1_b=c(node.layer_norm1.bias),
            wqkv=wqkv,
            bqkv=torch.cat((a.q_proj.bias, a.k_proj.bias, a.v_proj.bias), 0).detach().contiguous(),
            wo=c(a.out_proj.weight), bo=c(a.out_proj.bias),
            ln2_w=c(node.layer_norm2.weight), ln2_b=c(node.layer_norm2.bias),
            w1=c(node.mlp.fc1.weight), b1=c(node.mlp.fc1.bias), w2=c(node.mlp.fc2.weight), b2=c(node.mlp.fc2.bias))

    def prepared(self) -> dict[str, Any]:
        """Packed, device-resident views of the frozen weights; rebuilt if any parameter changed or moved."""
        dev = self.clip.logit_scale.device
        # (the flat parameter list is cached: walking the module tree for it cost ~1 ms per call, several calls per step)
        if self._plist is None:
            self._plist = list(self.parameters())
        key = (dev, sum(p._version for p in self._plist))
        if self._prep is None or self._prep_key != key:
            if any(p.requires_grad for n, p in self.named_parameters() if not n.startswith("decoder.transposed_convolution")):
                raise NotImplementedError(
                    "only the prompt-tuning path is implemented: towers and decoder must be frozen (freeze_all=True); "
                    "full fine-tuning (e2e_* configs) is outside the hot path (SURVEY.md §8)")
            v, t = self.clip.vision_model, self.clip.text_model
            ps = self.config.vision_config.patch_size
            self._prep = {
                "vision_layers": [self._layer_weights(l) for l in v.encoder.layers],
                "text_layers": [self._layer_weights(l) for l in t.encoder.layers],
                "decoder_layers": [self._layer_weights(l) for l in self.decoder.layers],
                # the decoder's reduce Linears (768 -> 64) with their transposes: the data gradient into the tower runs as an NT GEMM
                "reduces": [(r.weight.detach().contiguous(), r.bias.detach().contiguous(), r.weight.detach().t().contiguous()) for r in self.decoder.reduces],
                # (marked frozen: at M >= 2048 rows the patch GEMM packs its im2col rows and runs on two fp16 pieces, hip._gemm_takes_h2)
                "patch_w": hip.mark_frozen(v.embeddings.patch_embedding.weight.detach().reshape(self.config.vision_config.hidden_size, -1).contiguous()),
                "patch_size": ps,
            }
            self._prep_key = key
            self._pos_cache.clear()
            hip._built(device=dev)   # built on whichever stream asked first; the text tower's side stream reads it next
        return self._prep

    def vision_pos(self, height: int, width: int) -> torch.Tensor:
        """Position embedding for an HxW input, bicubic-interpolated exactly like HF ``interpolate_pos_encoding``
        (modeling_clipseg.py:149-188).  Frozen, so computed once per input size and cached."""
        pos = self.clip.vision_model.embeddings.position_embedding.weight.detach()
        key = (height, width, pos.device, pos._version)
        if key not in self._pos_cache:
            ps = self.config.vision_config.patch_size
            nh, nw = height // ps, width // ps
            n_pos = pos.shape[0] - 1
            if nh * nw == n_pos and height == width:
                out = pos
            else:
                side = int(n_pos**0.5)
                dim = pos.shape[-1]
                patch = pos[1:].reshape(1, side, side, dim).permute(0, 3, 1, 2)
                patch = F.interpolate(patch, size=(nh, nw), mode="bicubic", align_corners=False)
                out = torch.cat((pos[:1], patch.permute(0, 2, 3, 1).reshape(-1, dim)), 0)
            self._pos_cache[key] = out.contiguous()
            hip._built(self._pos_cache[key])
        return self._pos_cache[key]

    def _apply(self, fn, *a, **k):
        self._prep = None
        self._plist = None
        self._pos_cache = {}
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._prep = None
        self._plist = None
        self._pos_cache = {}
        return super().load_state_dict(*a, **k)


def _config_from_mapping(spec: Mapping[str, Any]) -> CLIPSegConfig:
    if "config" in spec:
        c = spec["config"]
        return c if isinstance(c, CLIPSegConfig) else CLIPSegConfig.from_dict(c)
    preset = spec.get("preset", "rd64")
    eos = int(spec.get("eos_token_id", 2))
    if preset == "tiny":
        return CLIPSegConfig.tiny(eos)
    if preset == "rd64":
        return CLIPSegConfig.rd64(eos, image_size=int(spec.get("image_size", 224)))
    raise ValueError(f"unknown backbone preset {preset!r}")
