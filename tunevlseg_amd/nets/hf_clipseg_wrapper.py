"""reference ``src/models/components/hf_clipseg_wrapper.py:15-74``"""
from __future__ import annotations

from collections.abc import Mapping
from dataclasses import dataclass

import torch
from torch import nn

from ..backbone import CLIPSegBackbone
from . import towers


@dataclass
class SegOutput:
    """The fields of HF ``CLIPSegImageSegmentationOutput`` the callers read."""
    logits: torch.Tensor
    conditional_embeddings: torch.Tensor | None = None
    pooled_output: torch.Tensor | None = None
    loss: torch.Tensor | None = None


class HFCLIPSegWrapper(nn.Module):
    def __init__(self, pretrained_model_name_or_path=None, freeze_encoder: bool = False, freeze_decoder: bool = False, *args, **kwargs) -> None:
        super().__init__()
        model = self.get_pretrained_model(pretrained_model_name_or_path, *args, **kwargs)
        model.clip.requires_grad_(not freeze_encoder)
        model.decoder.requires_grad_(not freeze_decoder)
        self.model = model

    @staticmethod
    def get_pretrained_model(pretrained_model_name_or_path, *args, **kwargs) -> CLIPSegBackbone:
        return CLIPSegBackbone.from_spec(pretrained_model_name_or_path)

    def forward(self, text_input: Mapping[str, torch.Tensor], image_input: torch.Tensor):
        """Plain CLIPSeg (HF ``CLIPSegForImageSegmentation.forward``, modeling_clipseg.py:1040-1110), frozen."""
        B, _, H, W = image_input.shape
        with torch.no_grad():
            acts, _ = towers.vision_tower(self.model, image_input, None, full=True)
            cond = towers.text_tower(self.model, text_input["input_ids"], text_input.get("attention_mask"))
            logits = towers.seg_head(self.model, towers.decoder_tokens(self.model, acts, cond), 0)
        return logits.view(B, 1, H, W)
