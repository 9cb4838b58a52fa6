"""reference ``src/models/core_models/coop/coop_clipseg.py:21-484``"""
from __future__ import annotations

import torch

from . import towers
from .base_clipseg import BaseCLIPSeg
from .hf_clipseg_wrapper import SegOutput


class COOPCLIPSeg(BaseCLIPSeg):
    def __init__(self, context_learner, *args, **kwargs) -> None:
        super().__init__(*args, **kwargs)
        cfg = self.model.config
        self.context_learner = context_learner(
            visual_dim=cfg.projection_dim,
            max_network_depth=min(cfg.text_config.num_hidden_layers, cfg.vision_config.num_hidden_layers),
            context_dim=cfg.text_config.hidden_size,
            embedding_layer=self.model.clip.text_model.embeddings.token_embedding,
        )

    def get_vision_outputs(self, pixel_values: torch.Tensor):
        """Frozen HF vision model, all layers, + visual_projection(post_layernorm(CLS)) (coop_clipseg.py:341-371)."""
        with torch.no_grad():
            return towers.vision_tower(self.model, pixel_values, None, full=True)

    def get_text_features(self, input_ids, attention_mask=None, image_features=None):
        return towers.text_tower(self.model, input_ids, attention_mask, self.context_learner, image_features)

    def model_forward(self, input_ids=None, pixel_values=None, attention_mask=None, position_ids=None,
                      conditional_embeddings=None, **_unused) -> SegOutput:
        if pixel_values is None:
            raise ValueError("You have to specify pixel_values to use `CLIPSegForImageSegmentation`")
        from .context_learner import CoCoOpContextLearner

        if conditional_embeddings is None and input_ids is not None and len(input_ids) == pixel_values.shape[0] \
                and not isinstance(self.context_learner, CoCoOpContextLearner):
            # plain CoOp: the text tower does not read the image features -> it runs beside the (gradient-free) vision tower
            side = towers.SideStream(pixel_values.device)
            with side:
                conditional_embeddings = self.get_text_features(input_ids, attention_mask, image_features=None)
            activations, pooled_output = self.get_vision_outputs(pixel_values)
            side.join(conditional_embeddings)
        else:
            activations, pooled_output = self.get_vision_outputs(pixel_values)
        if conditional_embeddings is None:
            if input_ids is None:
                raise ValueError("Invalid conditional, should be either provided as `input_ids` or `conditional_pixel_values`")
            if len(input_ids) != pixel_values.shape[0]:
                raise ValueError("Make sure to pass as many prompt texts as there are query images")
            conditional_embeddings = self.get_text_features(input_ids, attention_mask, image_features=pooled_output)
        elif conditional_embeddings.shape[0] != pixel_values.shape[0]:
            raise ValueError("Make sure to pass as many conditional embeddings as there are query images in the batch")
        # HF's decoder, NOT self.decoder_forward: the new last layer is unused on this path (coop_clipseg.py:462-468)
        tokens = towers.decoder_tokens(self.model, activations, conditional_embeddings)
        logits = towers.seg_head(self.model, tokens, 0)
        return SegOutput(logits=logits, conditional_embeddings=conditional_embeddings, pooled_output=pooled_output)
