"""reference ``src/models/core_models/coop/vpt_clipseg.py:22-395``"""
from __future__ import annotations

import torch

from . import towers
from .base_clipseg import BaseCLIPSeg
from .hf_clipseg_wrapper import SegOutput


class VPTCLIPSeg(BaseCLIPSeg):
    # VPT's own decoder_forward adds the new last layer plainly: logits += f(out)  (vpt_clipseg.py:301-302)
    LAST_LAYER_MIX = 1

    def __init__(self, context_learner, *args, cache_text_features: bool = False, **kwargs) -> None:
        """``cache_text_features`` (extension, default off): VPT trains nothing upstream of the conditional embeddings -- they are a
        function of the token row and the frozen text tower only -- so they can be kept per distinct (input_ids, attention_mask) row
        and the text tower runs only for rows not seen before (a dataset has a handful of distinct phrases).  It SKIPS work: numbers
        measured with it are reported separately (``bench.py --cond-cache``)."""
        super().__init__(*args, **kwargs)
        self.cache_text_features = bool(cache_text_features)
        self._text_cache: dict[bytes, torch.Tensor] = {}
        self._copy_stream = None
        cfg = self.model.config
        self.context_learner = context_learner(
            max_network_depth=min(cfg.text_config.num_hidden_layers, cfg.vision_config.num_hidden_layers),
            context_dim=cfg.vision_config.hidden_size,
        )

    def clear_text_cache(self) -> None:
        """Call after changing the text tower's weights (loading another backbone): the cache is keyed by token rows only."""
        self._text_cache.clear()

    def _apply(self, fn, *args, **kwargs):  # .to() / .cuda(): cached rows live on the old device
        self._text_cache.clear()
        return super()._apply(fn, *args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        self._text_cache.clear()
        return super().load_state_dict(*args, **kwargs)

    def cached_conditional_embeddings(self, input_ids: torch.Tensor, attention_mask: torch.Tensor | None) -> torch.Tensor:
        """Conditional embeddings [B, proj] with the text tower run only on rows whose key is new.  The keys need the token rows on
        the host: they are copied on a side stream so that the copy does not wait for the compute already enqueued."""
        am = attention_mask if attention_mask is not None else torch.ones_like(input_ids)
        rows = torch.cat([input_ids, am.to(input_ids.dtype)], dim=1)
        if rows.is_cuda:
            if self._copy_stream is None:
                self._copy_stream = torch.cuda.Stream(device=rows.device)
            self._copy_stream.wait_stream(torch.cuda.current_stream(rows.device))   # rows was just built on the current stream
            with torch.cuda.stream(self._copy_stream):
                host = rows.to("cpu", non_blocking=False)
        else:
            host = rows
        L = input_ids.shape[1]
        keys = []
        for i in range(host.shape[0]):   # trailing padding (mask 0) does not change the embedding: it is not part of the key
            n = L
            while n > 1 and host[i, L + n - 1] == 0:
                n -= 1
            keys.append(host[i, :n].numpy().tobytes() + host[i, L:L + n].numpy().tobytes())
        missing = [i for i in dict.fromkeys(i for i, k in enumerate(keys) if k not in self._text_cache)]
        if missing:
            first = {}
            for i in missing:
                first.setdefault(keys[i], i)
            idx = torch.tensor(list(first.values()), device=input_ids.device)
            with torch.no_grad():
                feats = towers.text_tower(self.model, input_ids.index_select(0, idx), None if attention_mask is None else attention_mask.index_select(0, idx))
            for j, k in enumerate(first):
                self._text_cache[k] = feats[j].clone()
        return torch.stack([self._text_cache[k] for k in keys])

    def get_vision_outputs(self, pixel_values: torch.Tensor):
        acts, _ = towers.vision_tower(self.model, pixel_values, self.context_learner)
        return acts

    def model_forward(self, input_ids=None, pixel_values=None, attention_mask=None, position_ids=None,
                      conditional_embeddings=None, **_unused) -> SegOutput:
        if pixel_values is None:
            raise ValueError("You have to specify pixel_values to use `CLIPSegForImageSegmentation`")
        # step 1: conditional embeddings from the frozen text tower, no grad (HF get_conditional_embeddings, HF:972-999)
        side = None
        if conditional_embeddings is None:
            if input_ids is None:
                raise ValueError("Invalid conditional, should be either provided as `input_ids` or `conditional_pixel_values`")
            if len(input_ids) != pixel_values.shape[0]:
                raise ValueError("Make sure to pass as many prompt texts as there are query images")
            if self.cache_text_features:
                conditional_embeddings = self.cached_conditional_embeddings(input_ids, attention_mask)
            else:
                # the frozen text tower (84 launch-latency-bound kernels at M = B * L rows, no gradient) shares nothing with the vision
                # tower until the decoder's FiLM: it runs on a side stream, in the bubbles of the vision tower's chip-filling kernels.
                # Forked HERE, enqueued after the vision tower: the chip-filling kernels go first when the host is not ahead of the GPU
                side = towers.SideStream(pixel_values.device)
        elif conditional_embeddings.shape[0] != pixel_values.shape[0]:
            raise ValueError("Make sure to pass as many conditional embeddings as there are query images in the batch")
        # step 2: vision tower with the visual prompts appended
        activations = self.get_vision_outputs(pixel_values)
        if side is not None:   # the decoder is the first consumer of the side stream's result
            with side, torch.no_grad():
                conditional_embeddings = towers.text_tower(self.model, input_ids, attention_mask)
            side.join(conditional_embeddings)
        # step 3: decoder
        out = self.decoder_forward(activations, conditional_embeddings)
        out.conditional_embeddings = conditional_embeddings
        return out
