"""reference ``src/models/core_models/coop/base_clipseg.py:24-199``"""
from __future__ import annotations

from abc import ABC, abstractmethod
from collections.abc import Mapping, Sequence
from typing import Any

import torch
from torch import nn

from . import towers
from .context_learner import BaseVisualLearner
from .hf_clipseg_wrapper import HFCLIPSegWrapper, SegOutput


class BaseCLIPSeg(HFCLIPSegWrapper, ABC):
    # how the new last layer is mixed in decoder_forward: 2 = (1-r)*logits + r*f(out)  (base_clipseg.py:152-155)
    LAST_LAYER_MIX = 2

    def __init__(self, model_cfg: Mapping[str, Any], freeze_all: bool = True, no_freeze_last_layer: bool = False,
                 use_new_last_layer: bool = False, new_last_layer_kernel_size=5, residual_ratio: float = 0.5) -> None:
        super().__init__(**model_cfg)
        self.assign_model_learnability(freeze_all, no_freeze_last_layer, use_new_last_layer, new_last_layer_kernel_size, residual_ratio)

    def assign_model_learnability(self, freeze_all: bool, no_freeze_last_layer: bool, use_new_last_layer: bool,
                                  new_last_layer_kernel_size, residual_ratio: float):
        if not freeze_all and any(p.requires_grad for p in self.model.parameters()):
            raise NotImplementedError(
                "freeze_all=False would fine-tune the CLIPSeg towers; only the prompt-tuning path (frozen towers, "
                "dgrad-only backward) is implemented (SURVEY.md §8)")
        if freeze_all:
            self.eval()
            self.requires_grad_(False)
        self.additive_decoder_layer = None
        if use_new_last_layer:
            k = new_last_layer_kernel_size
            if not isinstance(k, int):
                if k[0] != k[1]:
                    raise NotImplementedError("square new_last_layer_kernel_size only")
                k = k[0]
            # parameter holders with the reference's state_dict keys (additive_decoder_layer.1.{weight,bias});
            # the fused HIP path (upconv taps) does the math
            self.additive_decoder_layer = nn.Sequential(
                nn.Upsample(scale_factor=float(self.model.config.vision_config.patch_size), mode="bilinear"),
                nn.Conv2d(self.model.config.reduce_dim, 1, kernel_size=k, padding="same", padding_mode="replicate"),
            )
            self.residual_ratio = nn.Parameter(torch.tensor(residual_ratio))
        elif no_freeze_last_layer:
            self.model.decoder.transposed_convolution.requires_grad_(True)

    def decoder_forward(self, hidden_states: Sequence[torch.Tensor], conditional_embeddings: torch.Tensor, **_unused) -> SegOutput:
        tokens = towers.decoder_tokens(self.model, tuple(hidden_states), conditional_embeddings)
        n_strip = self.context_learner.num_context if isinstance(self.context_learner, BaseVisualLearner) else 0
        ratio = getattr(self, "residual_ratio", None)
        logits = towers.seg_head(self.model, tokens, n_strip, self.additive_decoder_layer, ratio, self.LAST_LAYER_MIX)
        return SegOutput(logits=logits)

    @abstractmethod
    def model_forward(self, input_ids=None, pixel_values=None, attention_mask=None, **kwargs) -> SegOutput: ...

    def forward(self, text_input: Mapping[str, torch.Tensor], image_input: torch.Tensor):
        B, _, H, W = image_input.shape
        text_input = {k: v for k, v in text_input.items() if k in ("input_ids", "attention_mask", "position_ids")}
        outputs = self.model_forward(**text_input, pixel_values=image_input)
        return outputs.logits.view(B, 1, H, W)
