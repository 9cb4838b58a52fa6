"""Prompt learners with the reference's public surface
(reference ``src/models/core_models/coop/context_learner/*.py``).

Same class names, constructor keywords, attributes (``context_vectors``, ``num_context``,
``context_dim``, ``prompt_depth``, ``projection_layers``), methods and ``state_dict`` keys.  The
``nn.Linear`` / ``nn.LayerNorm`` objects inside ``projection_layers`` are parameter holders only:
their math runs through the HIP ops of ``tunevlseg_amd.ops``.
"""
from __future__ import annotations

import copy
from abc import ABC, abstractmethod
from dataclasses import dataclass
from collections.abc import Iterable, Sequence
from typing import Final, Literal

import torch
from torch import nn

from .. import hip, ops


@dataclass(frozen=True)
class ProjectionPlan:
    """What a prompt projection is made of, decided before any module exists.

    ``steps`` is a tuple of ``("linear", in, out, bias, kaiming)`` / ``("relu",)`` / ``("norm", dim, bias)`` entries.  The builder
    turns it into the ``nn.Linear`` / ``nn.Sequential`` parameter holder whose ``state_dict`` keys the reference checkpoints use
    (``projection_layers.<depth>[.<position>].weight``); :func:`run_projection` evaluates the holder with HIP ops.
    Layout rules (reference ``base_projector_learner.py:65-139``): no hidden width -> one bare Linear; an MLP ends in a Linear that
    drops its bias when a LayerNorm follows; hidden Linears are He-initialised; the low-rank form is two bias-free-then-biased
    Linears through ``min(out, rank)`` and degenerates to a single Linear when the rank exceeds the output width.
    """

    steps: tuple

    @classmethod
    def mlp(cls, in_dim: int, out_dim: int, hidden, final_norm: bool, final_bias: bool = True) -> "ProjectionPlan":
        if hidden is None:
            return cls((("linear", in_dim, out_dim, True, False),))
        widths = (in_dim, *((hidden,) if isinstance(hidden, int) else tuple(hidden)))
        steps = []
        for fan_in, fan_out in zip(widths, widths[1:]):
            steps += [("linear", fan_in, fan_out, True, True), ("relu",)]
        steps.append(("linear", widths[-1], out_dim, final_bias and not final_norm, False))
        if final_norm:
            steps.append(("norm", out_dim, final_bias))
        return cls(tuple(steps))

    @classmethod
    def low_rank(cls, in_dim: int, out_dim: int, rank: int, final_norm: bool, final_bias: bool = True) -> "ProjectionPlan":
        steps = [("linear", in_dim, min(out_dim, rank), False, False)]
        if rank <= out_dim:
            steps.append(("linear", rank, out_dim, final_bias and not final_norm, False))
        if final_norm:
            steps.append(("norm", out_dim, final_bias))
        return cls(tuple(steps))

    def build(self) -> nn.Module:
        made, hidden = [], []
        for step in self.steps:
            if step[0] == "linear":
                _, fan_in, fan_out, bias, kaiming = step
                if not kaiming and hidden:
                    # all hidden Linears exist, the output Linear does not yet: He-initialise them now, in order -- the point and
                    # the order at which the reference draws these random numbers
                    for h in hidden:
                        nn.init.kaiming_normal_(h.weight.data, nonlinearity="relu")
                    hidden = []
                made.append(nn.Linear(fan_in, fan_out, bias=bias))
                if kaiming:
                    hidden.append(made[-1])
            elif step[0] == "relu":
                made.append(nn.ReLU(inplace=True))
            else:
                made.append(nn.LayerNorm(step[1], bias=step[2]))
        if len(self.steps) == 1 and self.steps[0][0] == "linear" and self.steps[0][3]:
            return made[0]  # the bare nn.Linear of ``intermediate_dim=None``: keys without a position index
        return nn.Sequential(*made)


def per_depth(make_module, depth: int, shared: bool, clone_first: bool = False) -> nn.ModuleList:
    """One projection per prompt depth: the SAME object ``depth`` times when shared (unified projection), otherwise independent
    modules -- freshly built, or deep copies of the first one (``clone_first``: every depth then starts from identical weights)."""
    if shared:
        return nn.ModuleList([make_module()] * depth)
    if clone_first:
        first = make_module()
        return nn.ModuleList([copy.deepcopy(first) for _ in range(depth)])
    return nn.ModuleList([make_module() for _ in range(depth)])


def run_projection(module: nn.Module, x: torch.Tensor) -> torch.Tensor:
    """Evaluate a projection holder (bare ``nn.Linear`` or ``nn.Sequential`` of Linear / ReLU / LayerNorm) with HIP kernels; a
    Linear directly followed by a ReLU runs as one fused launch."""
    chain = [module] if isinstance(module, nn.Linear) else list(module)
    pos = 0
    while pos < len(chain):
        holder = chain[pos]
        if isinstance(holder, nn.Linear):
            fuse_relu = pos + 1 < len(chain) and isinstance(chain[pos + 1], nn.ReLU)
            x = ops.linear(x, holder.weight, holder.bias, hip.ACT_RELU if fuse_relu else hip.ACT_NONE)
            pos += 2 if fuse_relu else 1
        elif isinstance(holder, nn.LayerNorm):
            x = ops.layer_norm(x, holder.weight, holder.bias, holder.eps)
            pos += 1
        else:  # pragma: no cover
            raise TypeError(f"unsupported projection layer {type(holder).__name__}")
    return x


class BaseUnimodalLearner(nn.Module, ABC):
    """reference ``base_unimodal_learner.py:17-99``"""

    MIN_PROMPT_DEPTH: Final = 1

    def __init__(self, *, max_network_depth: int, prompt_depth: int = MIN_PROMPT_DEPTH, num_context: int | None = None,
                 context_dim: int | None = None, context_initializer: str | list[str] | None = None, tokenizer=None,
                 embedding_layer=None, vector_std: float = 0.02, **kwargs) -> None:
        self.verify_prompt_depth(prompt_depth=prompt_depth, max_network_depth=max_network_depth)
        start = self.get_context_vectors(num_context=num_context, context_dim=context_dim, context_initializer=context_initializer,
                                         tokenizer=tokenizer, embedding_layer=embedding_layer, prompt_depth=prompt_depth,
                                         vector_std=vector_std)
        # the drawn / looked-up tensor decides the shape (an initializer phrase overrides num_context and context_dim)
        if start.ndim != 3:
            raise ValueError("The number of dimensions of `context_vectors` must be 3")
        if start.shape[0] != prompt_depth:
            raise ValueError("The number of rows of `context_vectors` must be `prompt_depth`")
        super().__init__()
        self.prompt_depth = prompt_depth
        _, self.num_context, self.context_dim = start.shape
        self.context_vectors = nn.Parameter(start.detach().clone().to(torch.float32))

    @classmethod
    def verify_prompt_depth(cls, prompt_depth: int, max_network_depth: int) -> None:
        if prompt_depth < cls.MIN_PROMPT_DEPTH:
            raise ValueError(f"{prompt_depth=} must be at least {cls.MIN_PROMPT_DEPTH=}")
        if prompt_depth > max_network_depth:
            raise ValueError(f"{prompt_depth=} must be at most {max_network_depth=} for the used network.")

    @staticmethod
    def init_random_context_vectors(shape: Sequence[int], std: float = 0.02) -> torch.Tensor:
        context_vectors = torch.empty(tuple(shape))
        nn.init.normal_(context_vectors, std=std)
        return context_vectors

    @abstractmethod
    def get_context_vectors(self, num_context=None, context_dim=None, prompt_depth: int = MIN_PROMPT_DEPTH,
                            context_initializer=None, tokenizer=None, embedding_layer=None, vector_std: float = 0.02) -> torch.Tensor: ...


class BaseVisualLearner(BaseUnimodalLearner):
    """reference ``base_visual_learner.py:12-23``"""

    @abstractmethod
    def get_visual_context(self, in_context: torch.Tensor | None = None, index: int = 0) -> torch.Tensor: ...

    def mutate_image_hidden_states(self, hidden_states: torch.Tensor, index: int) -> torch.Tensor:
        # hidden_states[:, -num_context:] = visual context, in place
        return ops.RowsOverwriteFn.apply(hidden_states, self.get_visual_context(index=index), hidden_states.shape[1] - self.num_context)


class VPTContextLearner(BaseVisualLearner):
    """reference ``vpt_context_learner.py:15-64``"""

    def __init__(self, **kwargs) -> None:
        kwargs["context_initializer"] = None
        kwargs["tokenizer"] = None
        kwargs["embedding_layer"] = None
        super().__init__(**kwargs)

    def get_context_vectors(self, num_context=None, context_dim=None, prompt_depth: int = BaseVisualLearner.MIN_PROMPT_DEPTH,
                            context_initializer=None, tokenizer=None, embedding_layer=None, vector_std: float = 0.02) -> torch.Tensor:
        if num_context is None or context_dim is None:
            raise ValueError("`num_context` and `context_dim` must be specified for VPT")
        return self.init_random_context_vectors((prompt_depth, num_context, context_dim), std=vector_std)

    def get_visual_context(self, in_context: torch.Tensor | None = None, index: int = 0) -> torch.Tensor:
        return self.context_vectors[index]

    def forward(self, *, input_embeddings: torch.Tensor, max_length: int | None = None, image_features=None,
                context_vectors: torch.Tensor | None = None, index: int = 0) -> torch.Tensor:
        if context_vectors is None:
            context_vectors = self.context_vectors[index]
        return ops.concat_rows(input_embeddings, context_vectors)


class CoOpContextLearner(BaseUnimodalLearner):
    """reference ``coop_context_learner.py:15-181``"""

    def get_context_vectors(self, num_context=None, context_dim=None, prompt_depth: int = BaseUnimodalLearner.MIN_PROMPT_DEPTH,
                            context_initializer=None, tokenizer=None, embedding_layer=None, vector_std: float = 0.02) -> torch.Tensor:
        if context_initializer is None:
            if num_context is None or context_dim is None:
                raise ValueError("`num_context` and `context_dim` must be specified if `context_initializer` is None")
            return self.init_random_context_vectors((prompt_depth, num_context, context_dim), std=vector_std)
        if tokenizer is None or embedding_layer is None:
            raise ValueError("If `context_initializer` is not None, `tokenizer` and `embedding_layer` must be specified")
        truncated = context_initializer if isinstance(context_initializer, str) else context_initializer[:prompt_depth]
        initialized = self.get_context_vectors_from_initializer(truncated, embedding_layer, tokenizer)
        initialized_depth, num_context, context_dim = initialized.shape
        remaining = prompt_depth - initialized_depth
        if remaining == 0:
            return initialized
        random_vectors = self.init_random_context_vectors((remaining, num_context, context_dim), std=vector_std)
        return torch.cat((initialized.cpu(), random_vectors))

    @staticmethod
    def get_context_vectors_from_initializer(context_initializer, embedding_layer, tokenizer) -> torch.Tensor:
        input_ids = tokenizer(context_initializer, return_tensors="pt", return_attention_mask=False, truncation=True,
                              add_special_tokens=False).input_ids
        with torch.no_grad():
            w = getattr(embedding_layer, "weight", None)
            if w is not None:
                input_ids = input_ids.to(w.device)
            return embedding_layer(input_ids)

    def _update_mask_for_context(self, mask: torch.Tensor, constructor: Literal["zeros", "ones"], max_length: int | None = None):
        extra = getattr(torch, constructor)(mask.shape[0], self.num_context, dtype=mask.dtype, device=mask.device)
        return torch.cat((extra, mask), dim=1)[:, :max_length]

    def update_attention_mask_for_context(self, attention_mask: torch.Tensor, max_length: int | None = None) -> torch.Tensor:
        return self._update_mask_for_context(attention_mask, "ones", max_length)

    def update_pad_mask_for_context(self, pad_mask: torch.Tensor, max_length: int | None = None) -> torch.Tensor:
        return self._update_mask_for_context(pad_mask, "zeros", max_length)

    def get_textual_context(self, in_context: torch.Tensor | None = None, image_features: torch.Tensor | None = None,
                            index: int = 0) -> torch.Tensor:
        return self.context_vectors[index]

    def mutate_text_hidden_states(self, hidden_states: torch.Tensor, index: int, image_features: torch.Tensor | None = None):
        # hidden_states[:, 1:num_context+1] = textual context, in place
        return ops.RowsOverwriteFn.apply(hidden_states, self.get_textual_context(image_features=image_features, index=index), 1)

    def splice_map(self, seq_len: int, max_length: int | None) -> list[int]:
        """Row map of ``forward``: entry >= 0 = source token position, -(j+1) = context vector j."""
        n = self.num_context
        last_idx = -1 if max_length is None else min(max_length - n, seq_len) - 1
        mid = list(range(seq_len))[1:last_idx]
        return [0, *[-(j + 1) for j in range(n)], *mid, seq_len - 1]

    def forward(self, *, input_embeddings: torch.Tensor, max_length: int | None = None, image_features: torch.Tensor | None = None,
                context_vectors: torch.Tensor | None = None, index: int = 0) -> torch.Tensor:
        if context_vectors is None:
            context_vectors = self.get_textual_context(image_features=image_features, index=index)
        return ops.splice_rows(input_embeddings, context_vectors, self.splice_map(input_embeddings.size(1), max_length))


class BaseProjectorLearner(CoOpContextLearner):
    """reference ``base_projector_learner.py:10-139``: prompts pushed through one projection per depth (see ProjectionPlan)."""

    def __init__(self, *, proj_in_dim: int | None, proj_out_dim: int | None, prompt_depth: int = CoOpContextLearner.MIN_PROMPT_DEPTH,
                 use_unified_projection: bool = True, intermediate_dim: int | Iterable[int] | None = None, use_proj_norm: bool = False,
                 use_lora_proj: bool = False, use_final_bias: bool = True, **kwargs) -> None:
        low_rank = self._wants_low_rank(use_lora_proj, intermediate_dim)
        super().__init__(prompt_depth=prompt_depth, **kwargs)
        fan_in = self.context_dim if proj_in_dim is None else proj_in_dim
        fan_out = self.context_dim if proj_out_dim is None else proj_out_dim
        plan = (ProjectionPlan.low_rank if low_rank else ProjectionPlan.mlp)(fan_in, fan_out, intermediate_dim, use_proj_norm, use_final_bias)
        self.projection_layers = per_depth(plan.build, prompt_depth, shared=use_unified_projection)

    @staticmethod
    def _wants_low_rank(use_lora_proj: bool, intermediate_dim) -> bool:
        if use_lora_proj and intermediate_dim is not None and not isinstance(intermediate_dim, int):
            raise ValueError("Lora projection is only available for a single layer.")
        return bool(use_lora_proj) and intermediate_dim is not None

    def get_transformed_context(self, in_context: torch.Tensor | None = None, index: int = 0) -> torch.Tensor:
        source = self.context_vectors[index] if in_context is None else in_context
        return run_projection(self.projection_layers[index], source)

    # the two factory names of the reference's public surface, kept as thin fronts of ProjectionPlan
    @staticmethod
    def get_lora_projection(in_dim: int, out_dim: int, intermediate_dim: int, use_final_norm: bool, use_final_bias: bool = True) -> nn.Module:
        return ProjectionPlan.low_rank(in_dim, out_dim, intermediate_dim, use_final_norm, use_final_bias).build()

    @staticmethod
    def get_mlp_projection(in_dim: int, out_dim: int, intermediate_dim, use_final_norm: bool, use_final_bias: bool = True) -> nn.Module:
        return ProjectionPlan.mlp(in_dim, out_dim, intermediate_dim, use_final_norm, use_final_bias).build()


class CoCoOpContextLearner(BaseProjectorLearner):
    """reference ``cocoop_context_learner.py:7-77``"""

    def __init__(self, *, visual_dim: int, norm_image_features: bool = True, **kwargs) -> None:
        kwargs["proj_in_dim"] = visual_dim
        kwargs["proj_out_dim"] = None
        kwargs["use_final_bias"] = False
        super().__init__(**kwargs)
        self.norm_image_features = norm_image_features
        self.image_features_normalizer_or_identity = self._normalize_features if norm_image_features else nn.Identity()

    @staticmethod
    def _normalize_features(features: torch.Tensor, p="fro", dim: int = -1) -> torch.Tensor:
        return ops.L2NormFn.apply(features)

    def get_textual_context(self, in_context: torch.Tensor | None = None, image_features: torch.Tensor | None = None,
                            index: int = 0) -> torch.Tensor:
        if image_features is None:
            raise ValueError("`image_features` must be provided when `context_vectors` is None for CoCoOp")
        image_features = self.image_features_normalizer_or_identity(image_features)
        bias = self.get_transformed_context(image_features, index)  # (batch, context_dim)
        if in_context is None:
            in_context = self.context_vectors[index]
        return ops.OuterAddFn.apply(bias, in_context)  # (batch, num_context, context_dim)

    def forward(self, *, input_embeddings: torch.Tensor, max_length: int | None = None, image_features: torch.Tensor | None = None,
                context_vectors: torch.Tensor | None = None, index: int = 0) -> torch.Tensor:
        context_vectors = self.get_textual_context(in_context=context_vectors, image_features=image_features, index=index)
        return super().forward(input_embeddings=input_embeddings, max_length=max_length, context_vectors=context_vectors)


class MapleContextLearner(BaseProjectorLearner, BaseVisualLearner):
    """reference ``maple_context_learner.py:7-20``"""

    def __init__(self, *, visual_dim: int, **kwargs) -> None:
        kwargs["proj_in_dim"] = None
        kwargs["proj_out_dim"] = visual_dim
        super().__init__(**kwargs)

    def get_visual_context(self, *args, **kwargs) -> torch.Tensor:
        return self.get_transformed_context(*args, **kwargs)


class BaseSharedLearner(CoOpContextLearner, BaseVisualLearner):
    """reference ``base_shared_learner.py:5-11``"""

    def __init__(self, **kwargs):
        kwargs["context_initializer"] = None
        kwargs["tokenizer"] = None
        kwargs["embedding_layer"] = None
        super().__init__(**kwargs)


class SharedSeparateLearner(BaseSharedLearner):
    """reference ``shared_separate_learner.py:11-98``: one shared prompt, two projections per depth (textual / visual)."""

    def __init__(self, *, textual_dim: int, visual_dim: int, shared_dim: int = 64, prompt_depth: int = BaseSharedLearner.MIN_PROMPT_DEPTH,
                 use_unified_projection: bool = True, intermediate_dim: int | Iterable[int] | None = None, use_proj_norm: bool = False,
                 use_lora_proj: bool = False, **kwargs) -> None:
        low_rank = BaseProjectorLearner._wants_low_rank(use_lora_proj, intermediate_dim)
        kwargs["context_dim"] = shared_dim
        super().__init__(prompt_depth=prompt_depth, **kwargs)
        plan_for = lambda width: (ProjectionPlan.low_rank if low_rank else ProjectionPlan.mlp)(  # noqa: E731
            shared_dim, width, intermediate_dim, use_proj_norm)
        # independent depths are deep copies of ONE freshly built module per modality (all depths start from the same weights)
        self.textual_projection_layers = per_depth(plan_for(textual_dim).build, prompt_depth, use_unified_projection, clone_first=True)
        self.visual_projection_layers = per_depth(plan_for(visual_dim).build, prompt_depth, use_unified_projection, clone_first=True)

    @staticmethod
    def get_projection_layers(single_layer: nn.Module, prompt_depth: int, use_unified_projection) -> nn.ModuleList:
        return per_depth(lambda: single_layer, prompt_depth, use_unified_projection, clone_first=True)

    def _through(self, stack: nn.ModuleList, in_context: torch.Tensor | None, index: int) -> torch.Tensor:
        return run_projection(stack[index], self.context_vectors[index] if in_context is None else in_context)

    def get_textual_context(self, in_context: torch.Tensor | None = None, image_features: torch.Tensor | None = None, index: int = 0):
        return self._through(self.textual_projection_layers, in_context, index)

    def get_visual_context(self, in_context: torch.Tensor | None = None, index: int = 0) -> torch.Tensor:
        return self._through(self.visual_projection_layers, in_context, index)


class SharedAttnLearner(BaseSharedLearner):
    """reference ``shared_attn_learner.py:9-104``: shared prompt (textual_dim + visual_dim wide) through an
    ``nn.TransformerEncoderLayer``, split into its textual / visual column blocks; the cross-modal half is cached between
    the vision pass and the text pass (kept on the device here; the reference parks it on the CPU).

    The reference feeds ``ctx[index].unsqueeze(0)`` = [1, n, E] to a ``batch_first=False`` layer, i.e. n independent
    length-1 sequences: softmax over one key is 1, so self-attention reduces exactly to ``out_proj(v_proj(x))`` (the q/k
    rows of ``in_proj`` receive zero gradient, as in the reference).  The torch layer object only holds the parameters.
    """

    def __init__(self, *, textual_dim: int, visual_dim: int, unified_projector, prompt_depth: int = BaseSharedLearner.MIN_PROMPT_DEPTH,
                 use_unified_projection: bool = True, **kwargs) -> None:
        if unified_projector is None:
            raise NotImplementedError("You need to provide a transformer encoder layer for the unified projection layer from the config.")
        kwargs["context_dim"] = textual_dim + visual_dim
        super().__init__(prompt_depth=prompt_depth, **kwargs)
        self.textual_dim, self.visual_dim = textual_dim, visual_dim
        self.projection_layers = per_depth(lambda: self._checked_layer(unified_projector(d_model=self.context_dim)), prompt_depth,
                                           use_unified_projection, clone_first=True)
        # each depth is evaluated once per step; the half the OTHER tower will ask for waits here, keyed by (depth, modality)
        self._handoff: dict[tuple[int, str], torch.Tensor] = {}

    @staticmethod
    def _checked_layer(layer: nn.Module) -> nn.Module:
        if getattr(layer.self_attn, "batch_first", False):
            raise NotImplementedError("batch_first=True (real attention over the prompt tokens) is not used by the reference configs")
        if getattr(getattr(layer, "activation", None), "__name__", "relu") != "relu":
            raise NotImplementedError("only the default relu activation of nn.TransformerEncoderLayer is implemented")
        return layer

    def _run_layer(self, layer: nn.TransformerEncoderLayer, x: torch.Tensor) -> torch.Tensor:
        E = x.shape[-1]
        attn = layer.self_attn
        wv, bv = attn.in_proj_weight[2 * E:], (attn.in_proj_bias[2 * E:] if attn.in_proj_bias is not None else None)
        tr = layer.training

        def sa(t):
            v = ops.linear(t, wv, bv)
            return ops.dropout(ops.linear(v, attn.out_proj.weight, attn.out_proj.bias), layer.dropout1.p, tr)

        def ff(t):
            h1 = ops.dropout(ops.linear(t, layer.linear1.weight, layer.linear1.bias, hip.ACT_RELU), layer.dropout.p, tr)
            return ops.dropout(ops.linear(h1, layer.linear2.weight, layer.linear2.bias), layer.dropout2.p, tr)

        n1 = lambda t: ops.layer_norm(t, layer.norm1.weight, layer.norm1.bias, layer.norm1.eps)  # noqa: E731
        n2 = lambda t: ops.layer_norm(t, layer.norm2.weight, layer.norm2.bias, layer.norm2.eps)  # noqa: E731
        if layer.norm_first:
            x = ops.add(x, sa(n1(x)))
            return ops.add(x, ff(n2(x)))
        x = n1(ops.add(x, sa(x)))
        return n2(ops.add(x, ff(x)))

    def _half(self, want: str, in_context: torch.Tensor | None = None, index: int = 0) -> torch.Tensor:
        """The ``want`` ("text" | "vision") column block of depth ``index``'s transformed prompt.  Whichever tower asks first runs
        the layer and leaves the other block for its sibling (``_handoff``); the sibling's request consumes it."""
        waiting = self._handoff.pop((index, want), None)
        if waiting is not None:
            return waiting
        tokens = self.context_vectors[index].unsqueeze(0) if in_context is None else in_context
        if tokens.ndim != 3:
            raise ValueError("The tensor needs to have 3 dimensions: (batch, context_len, hidden_dim)")
        text_block, vision_block = ops.split_cols(self._run_layer(self.projection_layers[index], tokens.squeeze(0)), self.textual_dim)
        mine, theirs, other = (text_block, vision_block, "vision") if want == "text" else (vision_block, text_block, "text")
        self._handoff[(index, other)] = theirs
        return mine

    def _get_combined_transformed_context(self, is_curr_branch_textual: bool, in_context: torch.Tensor | None = None, index: int = 0):
        return self._half("text" if is_curr_branch_textual else "vision", in_context, index)

    def get_textual_context(self, image_features: torch.Tensor | None = None, *args, **kwargs) -> torch.Tensor:
        return self._half("text", *args, **kwargs)

    def get_visual_context(self, *args, **kwargs) -> torch.Tensor:
        return self._half("vision", *args, **kwargs)
