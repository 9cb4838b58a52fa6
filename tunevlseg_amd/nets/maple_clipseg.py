"""reference ``src/models/core_models/coop/base_multimodal_clipseg.py:21-629`` and ``maple_clipseg.py:11-25``"""
from __future__ import annotations

import torch

from . import towers
from .base_clipseg import BaseCLIPSeg
from .hf_clipseg_wrapper import SegOutput


READY_PER_DEPTH = __import__("os").environ.get("TVL_MAPLE_READY_PER_DEPTH", "1") != "0"   # A/B switch (0 = one event behind all depths' projections)


class BaseMultimodalCLIPSeg(BaseCLIPSeg):
    # MaPLe's two towers read the shared prompts independently (text: ctx[idx]; vision: proj_idx(ctx[idx])): the text tower may run
    # beside the vision tower on the side stream.  The shared-attention learners hand a cached half from the vision pass to the text
    # pass (shared_attn_learner.py:89-90): they keep the reference's sequential order.
    TEXT_BESIDE_VISION = False

    def get_vision_outputs(self, pixel_values: torch.Tensor, visual_contexts=None, ready=None):
        acts, _ = towers.vision_tower(self.model, pixel_values, self.context_learner, visual_contexts=visual_contexts, ready=ready)
        return acts

    def get_conditional_embeddings(self, batch_size, input_ids, attention_mask):
        if input_ids is None:
            raise ValueError("Invalid conditional, should be either provided as `input_ids` or `conditional_pixel_values`")
        if len(input_ids) != batch_size:
            raise ValueError("Make sure to pass as many prompt texts as there are query images")
        return towers.text_tower(self.model, input_ids, attention_mask, self.context_learner)

    def model_forward(self, input_ids=None, pixel_values=None, attention_mask=None, position_ids=None,
                      conditional_embeddings=None, **_unused) -> SegOutput:
        if pixel_values is None:
            raise ValueError("You have to specify pixel_values to use `CLIPSegForImageSegmentation`")
        # vision first, then text (base_multimodal_clipseg.py:577-596)
        if conditional_embeddings is None and self.TEXT_BESIDE_VISION:
            side = towers.SideStream(pixel_values.device)
            with side:
                # the visual prompts of every depth are projections of parameters (maple_context_learner.py:7-20): computed ahead of the
                # text tower on the side stream, they (and their weight gradients in the backward) leave the vision tower's critical path
                learner = self.context_learner
                # ... one event per depth: the tower's first layer waits for the first projection only, layer i for the i-th (all of them
                # behind ONE event cost the tower ~0.3 ms of idle chip at the start of every step: 2 launches x 25 us per depth)
                vis, ready = None, None
                if side.on:
                    vis, marks = [], []
                    for i in range(learner.prompt_depth):
                        vis.append(learner.get_visual_context(index=i))
                        marks.append(side.mark())
                    ready = marks if READY_PER_DEPTH else marks[-1]
            activations = self.get_vision_outputs(pixel_values, vis, ready)
            with side:   # the text tower: enqueued after the vision tower (whose kernels go first when the host is not ahead), replayed before it in the backward
                conditional_embeddings = self.get_conditional_embeddings(pixel_values.shape[0], input_ids, attention_mask)
            side.join(conditional_embeddings)
            out = self.decoder_forward(activations, conditional_embeddings)
            out.conditional_embeddings = conditional_embeddings
            return out
        activations = self.get_vision_outputs(pixel_values)
        if conditional_embeddings is None:
            conditional_embeddings = self.get_conditional_embeddings(pixel_values.shape[0], input_ids, attention_mask)
        elif conditional_embeddings.shape[0] != pixel_values.shape[0]:
            raise ValueError("Make sure to pass as many conditional embeddings as there are query images in the batch")
        out = self.decoder_forward(activations, conditional_embeddings)
        out.conditional_embeddings = conditional_embeddings
        return out


class MapleCLIPSeg(BaseMultimodalCLIPSeg):
    TEXT_BESIDE_VISION = True

    def __init__(self, context_learner, *args, **kwargs) -> None:
        super().__init__(*args, **kwargs)
        cfg = self.model.config
        self.context_learner = context_learner(
            visual_dim=cfg.vision_config.hidden_size,
            max_network_depth=min(cfg.text_config.num_hidden_layers, cfg.vision_config.num_hidden_layers),
            context_dim=cfg.text_config.hidden_size,
            embedding_layer=self.model.clip.text_model.embeddings.token_embedding,
        )


class _SharedCLIPSeg(BaseMultimodalCLIPSeg):
    """reference ``shared_attn_learner_clipseg.py:11-25`` / ``shared_separate_learner_clipseg.py:11-26``"""

    def __init__(self, context_learner, *args, **kwargs) -> None:
        super().__init__(*args, **kwargs)
        cfg = self.model.config
        self.context_learner = context_learner(
            textual_dim=cfg.text_config.hidden_size,
            visual_dim=cfg.vision_config.hidden_size,
            max_network_depth=min(cfg.text_config.num_hidden_layers, cfg.vision_config.num_hidden_layers),
            context_dim=cfg.text_config.hidden_size,
        )


class SharedAttnCLIPSeg(_SharedCLIPSeg):
    pass


class SharedSeparateCLIPSeg(_SharedCLIPSeg):
    pass
