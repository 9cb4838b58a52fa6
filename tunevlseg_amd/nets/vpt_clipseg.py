"""reference ``src/models/core_models/coop/vpt_clipseg.py:22-395``"""
from __future__ import annotations

import torch

from . import towers
from .base_clipseg import BaseCLIPSeg
from .hf_clipseg_wrapper import SegOutput


class VPTCLIPSeg(BaseCLIPSeg):
    # VPT's own decoder_forward adds the new last layer plainly: logits += f(out)  (vpt_clipseg.py:301-302)
    LAST_LAYER_MIX = 1

    def __init__(self, context_learner, *args, **kwargs) -> None:
        super().__init__(*args, **kwargs)
        cfg = self.model.config
        self.context_learner = context_learner(
            max_network_depth=min(cfg.text_config.num_hidden_layers, cfg.vision_config.num_hidden_layers),
            context_dim=cfg.vision_config.hidden_size,
        )

    def get_vision_outputs(self, pixel_values: torch.Tensor):
        acts, _ = towers.vision_tower(self.model, pixel_values, self.context_learner)
        return acts

    def model_forward(self, input_ids=None, pixel_values=None, attention_mask=None, position_ids=None,
                      conditional_embeddings=None, **_unused) -> SegOutput:
        if pixel_values is None:
            raise ValueError("You have to specify pixel_values to use `CLIPSegForImageSegmentation`")
        # step 1: conditional embeddings from the frozen text tower, no grad (HF get_conditional_embeddings, HF:972-999)
        if conditional_embeddings is None:
            if input_ids is None:
                raise ValueError("Invalid conditional, should be either provided as `input_ids` or `conditional_pixel_values`")
            if len(input_ids) != pixel_values.shape[0]:
                raise ValueError("Make sure to pass as many prompt texts as there are query images")
            with torch.no_grad():
                conditional_embeddings = towers.text_tower(self.model, input_ids, attention_mask)
        elif conditional_embeddings.shape[0] != pixel_values.shape[0]:
            raise ValueError("Make sure to pass as many conditional embeddings as there are query images in the batch")
        # step 2: vision tower with the visual prompts appended
        activations = self.get_vision_outputs(pixel_values)
        # step 3: decoder
        out = self.decoder_forward(activations, conditional_embeddings)
        out.conditional_embeddings = conditional_embeddings
        return out
