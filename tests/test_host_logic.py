"""Host-side mirror of the reference interface (CPU, no HIP compute): names, constructor keywords, error behaviour,
state_dict keys, optimiser groups, metric arithmetic, DDP batch split."""
from functools import partial

import pytest
import torch
from torch import nn

from oracle import clipseg_oracle as O
from tunevlseg_amd import nets
from tunevlseg_amd import dist as tdist
from tunevlseg_amd.config import CLIPSegConfig
from tunevlseg_amd.nets import context_learner as CL
from tunevlseg_amd.task import DiceCELoss, DiceSamples, ImageTextMaskModule, JaccardBinary, ReduceLROnPlateau
from tunevlseg_amd.weights import count_params, init_clipseg_state_dict

TINY = {"pretrained_model_name_or_path": "random:tiny:seed=11", "freeze_encoder": False, "freeze_decoder": False}


def test_rd64_geometry_matches_hf_parameter_count():
    # 150.7 M params (vision 85.8 M, text 63.2 M, decoder 1.07 M) for patch 16
    n = count_params(CLIPSegConfig.rd64())
    assert abs(n - 150_747_746) < 400_000, n  # position embedding rows differ with image_size only


def test_state_dict_keys_follow_reference_checkpoint_layout():
    net = nets.MapleCLIPSeg(context_learner=partial(CL.MapleContextLearner, prompt_depth=2, num_context=2, intermediate_dim=8,
                                                    use_proj_norm=True, use_unified_projection=False), model_cfg=TINY, use_new_last_layer=True)
    keys = set(net.state_dict())
    for k in ("model.clip.vision_model.encoder.layers.0.self_attn.q_proj.weight", "model.clip.vision_model.pre_layrnorm.weight",
              "model.clip.text_model.embeddings.token_embedding.weight", "model.decoder.reduces.2.bias",
              "model.decoder.transposed_convolution.weight", "context_learner.context_vectors",
              "context_learner.projection_layers.1.0.weight", "context_learner.projection_layers.1.3.bias",
              "additive_decoder_layer.1.weight", "additive_decoder_layer.1.bias", "residual_ratio"):
        assert k in keys, k
    trainable = {k for k, p in net.named_parameters() if p.requires_grad}
    assert all(k.startswith(("context_learner.", "additive_decoder_layer.", "residual_ratio")) for k in trainable)
    assert not net.training or True
    assert not net.model.training  # frozen towers stay in eval mode (base_clipseg.py:53)


def test_trainable_parameter_counts():
    # SURVEY.md §2b payloads: VPT-10 shallow 7 680 floats (+1 602 with the new last layer), CoOp-4 2 048
    cfg = {"pretrained_model_name_or_path": {"preset": "rd64", "seed": 0, "state_dict": _Lazy(CLIPSegConfig.rd64())}}
    tiny_vpt = nets.VPTCLIPSeg(context_learner=partial(CL.VPTContextLearner, prompt_depth=1, num_context=10), model_cfg=TINY,
                               use_new_last_layer=True)
    n = sum(p.numel() for p in tiny_vpt.parameters() if p.requires_grad)
    assert n == 10 * 32 + (16 * 25 + 1) + 1
    del cfg


class _Lazy(dict):
    def __init__(self, cfg):
        super().__init__()


def test_prompt_depth_validation_and_errors():
    with pytest.raises(ValueError):
        CL.VPTContextLearner(max_network_depth=3, prompt_depth=0, num_context=4, context_dim=8)
    with pytest.raises(ValueError):
        CL.VPTContextLearner(max_network_depth=3, prompt_depth=4, num_context=4, context_dim=8)
    with pytest.raises(ValueError):
        CL.VPTContextLearner(max_network_depth=3, prompt_depth=1, context_dim=8)  # num_context missing
    with pytest.raises(ValueError):
        CL.CoOpContextLearner(max_network_depth=3, prompt_depth=1, context_initializer="a photo")  # no tokenizer
    with pytest.raises(ValueError):
        CL.MapleContextLearner(visual_dim=8, max_network_depth=3, num_context=2, context_dim=4, use_lora_proj=True, intermediate_dim=(4, 4))
    with pytest.raises(NotImplementedError):
        nets.VPTCLIPSeg(context_learner=partial(CL.VPTContextLearner, num_context=2), model_cfg=TINY, freeze_all=False)


def test_initializer_overrides_context_shape():
    class Tok:
        def __call__(self, text, **kw):
            class R:
                input_ids = torch.tensor([[5, 9, 7]])
            return R()

    emb = nn.Embedding(20, 6)
    lrn = CL.CoOpContextLearner(max_network_depth=4, prompt_depth=3, num_context=99, context_dim=77, context_initializer="x y z",
                                tokenizer=Tok(), embedding_layer=emb)
    assert lrn.context_vectors.shape == (3, 3, 6) and lrn.num_context == 3 and lrn.context_dim == 6
    assert torch.equal(lrn.context_vectors[0], emb.weight[torch.tensor([5, 9, 7])].detach())


@pytest.mark.parametrize("L,n,maxlen", [(6, 4, 16), (6, 12, 16), (8, 4, 77), (77, 4, 77), (3, 2, None)])
def test_splice_map_matches_reference_concat(L, n, maxlen):
    lrn = CL.CoOpContextLearner(max_network_depth=4, prompt_depth=1, num_context=n, context_dim=5)
    emb = torch.arange(L, dtype=torch.float32).view(1, L, 1).expand(2, L, 5)
    ctx = -torch.arange(1, n + 1, dtype=torch.float32).view(n, 1).expand(n, 5)
    if maxlen is None:  # reference: mid = emb[1:-1]
        ref = torch.cat((emb[:, :1], ctx.expand(2, n, 5), emb[:, 1:-1], emb[:, -1:]), 1)
    else:
        ref = O.coop_splice(emb, ctx, maxlen)
    tmap = lrn.splice_map(L, maxlen)
    got = torch.stack([emb[0, m] if m >= 0 else ctx[-m - 1] for m in tmap])
    assert torch.equal(got, ref[0])
    mask = torch.ones(2, L, dtype=torch.long)
    assert lrn.update_attention_mask_for_context(mask, maxlen).shape[1] == (len(tmap) if maxlen else L + n)
    assert lrn.update_pad_mask_for_context(mask, maxlen)[:, :n].sum() == 0


def test_optim_groups_split():
    net = nets.MapleCLIPSeg(context_learner=partial(CL.MapleContextLearner, prompt_depth=1, num_context=2, intermediate_dim=8,
                                                    use_proj_norm=True), model_cfg=TINY, use_new_last_layer=True)
    mod = ImageTextMaskModule(net, DiceCELoss(sigmoid=True, lambda_dice=1, lambda_ce=0.2), weight_decay=0.01)
    decay, no_decay = mod.get_optim_groups()
    names = {id(p): k for k, p in mod.named_parameters()}
    d = {names[id(p)] for p in decay["params"] if p.requires_grad}
    nd = {names[id(p)] for p in no_decay["params"] if p.requires_grad}
    # Linear / Conv weights decay; prompts, residual_ratio, norms and biases do not (image_text_mask_module.py:304-361)
    assert d == {"net.context_learner.projection_layers.0.0.weight", "net.context_learner.projection_layers.0.2.weight",
                 "net.additive_decoder_layer.1.weight"}
    assert "net.context_learner.context_vectors" in nd and "net.residual_ratio" in nd
    assert "net.context_learner.projection_layers.0.3.weight" in nd  # LayerNorm weight
    assert ImageTextMaskModule(net, DiceCELoss(), weight_decay=0.0).get_optim_groups() is not None


def test_metrics_match_oracle_definitions():
    g = torch.Generator().manual_seed(0)
    preds, tgt = torch.rand(5, 1, 9, 9, generator=g), (torch.rand(5, 1, 9, 9, generator=g) > 0.6).long()
    tgt[4] = 0
    preds[4] = 0.1  # empty prediction and empty target -> zero_division = 1
    tp, fp, fn, tn = O.confusion_counts(preds, tgt)
    counts = torch.stack((tp, fp, fn, tn), 1)
    d, j = DiceSamples(), JaccardBinary()
    d.update(counts[:2]); d.update(counts[2:])
    j.update(counts[:2]); j.update(counts[2:])
    assert abs(d.compute() - O.dice_samples(tp, fp, fn).item()) < 1e-12
    assert abs(j.compute() - O.jaccard_binary(tp, fp, fn).item()) < 1e-12
    lab = (preds > 0.5)
    per = [2 * (lab[i] & tgt[i].bool()).sum().item() / max(1, (lab[i].sum() + tgt[i].sum()).item()) for i in range(4)] + [1.0]
    assert abs(d.compute() - sum(per) / 5) < 1e-12


def test_reduce_lr_on_plateau_matches_torch():
    class Opt:
        param_groups = [{"lr": 1.0}]

    p = nn.Parameter(torch.zeros(1))
    topt = torch.optim.SGD([p], lr=1.0)
    ts = torch.optim.lr_scheduler.ReduceLROnPlateau(topt, mode="min", factor=0.2, patience=2)
    mine = ReduceLROnPlateau(Opt, mode="min", factor=0.2, patience=2)
    for v in [1.0, 0.9, 0.95, 0.95, 0.95, 0.95, 0.5, 0.6, 0.6, 0.6, 0.6]:
        ts.step(v)
        mine.step(v)
        assert abs(Opt.param_groups[0]["lr"] - topt.param_groups[0]["lr"]) < 1e-12


def test_global_batch_split():
    assert tdist.per_device_batch_size(256, 8) == 32
    with pytest.raises(ValueError):
        tdist.per_device_batch_size(30, 8)


def test_weight_draw_is_deterministic():
    a, b = init_clipseg_state_dict(CLIPSegConfig.tiny(), 5), init_clipseg_state_dict(CLIPSegConfig.tiny(), 5)
    assert all(torch.equal(a[k], b[k]) for k in a)
    c = init_clipseg_state_dict(CLIPSegConfig.tiny(), 6)
    assert not torch.equal(a["decoder.film_mul.weight"], c["decoder.film_mul.weight"])


def _kav():
    import json
    from pathlib import Path

    return json.loads((Path(__file__).resolve().parent / "golden" / "loss_metric_kav.json").read_text())


@pytest.mark.parametrize("case", [c for c in _kav()["cases"] if c["asserted"]], ids=lambda c: c["name"])
def test_oracle_loss_and_metrics_match_hand_derived_vectors(case):
    """Rows L1 / L2: the oracle's DiceCE / Dice(samples) / binary IoU against vectors derived by hand from the published monai /
    torchmetrics definitions (tests/golden/make_loss_kav.py) -- the pin that does not pass through this repo's own restatement."""
    kav = _kav()
    x = torch.tensor(case["logits"], dtype=torch.float32)[:, None, :, None]   # [B, 1, N, 1]
    t = torch.tensor(case["mask"], dtype=torch.float32)[:, None, :, None]
    e = case["expect"]
    assert abs(O.dice_ce_loss(x, t, kav["lambda_dice"], kav["lambda_ce"]).item() - e["loss"]) < 2e-6
    tp, fp, fn, tn = O.confusion_counts(torch.sigmoid(x), t.long(), kav["threshold"])
    assert torch.stack((tp, fp, fn, tn), 1).tolist() == e["counts_tp_fp_fn_tn"]
    assert abs(O.dice_samples(tp, fp, fn).item() - e["metric_dice_samples"]) < 1e-12
    assert abs(O.jaccard_binary(tp, fp, fn).item() - e["metric_iou"]) < 1e-12
    d, j = DiceSamples(), JaccardBinary()
    counts = torch.stack((tp, fp, fn, tn), 1)
    d.update(counts); j.update(counts)
    assert abs(d.compute() - e["metric_dice_samples"]) < 1e-12 and abs(j.compute() - e["metric_iou"]) < 1e-12


def test_projection_plan_builds_reference_shaped_holders():
    """ProjectionPlan -> holder: state_dict keys as in reference checkpoints (``<pos>.weight`` / bare Linear), bias rules, and the
    He-initialised hidden Linears (drawn after all hidden layers exist, before the output Linear)."""
    from tunevlseg_amd.nets.context_learner import ProjectionPlan

    bare = ProjectionPlan.mlp(12, 10, None, False).build()
    assert isinstance(bare, nn.Linear) and set(bare.state_dict()) == {"weight", "bias"}
    mlp = ProjectionPlan.mlp(12, 10, 8, True, final_bias=False).build()  # CoCoOp meta-net: Linear, ReLU, Linear(no bias), LayerNorm(no bias)
    assert set(mlp.state_dict()) == {"0.weight", "0.bias", "2.weight", "3.weight"}
    deep = ProjectionPlan.mlp(12, 10, (8, 6), False).build()
    assert [type(m).__name__ for m in deep] == ["Linear", "ReLU", "Linear", "ReLU", "Linear"] and deep[4].bias is not None
    lora = ProjectionPlan.low_rank(12, 10, 4, True).build()
    assert set(lora.state_dict()) == {"0.weight", "1.weight", "2.weight", "2.bias"} and lora[0].weight.shape == (4, 12)
    wide = ProjectionPlan.low_rank(12, 10, 16, False).build()  # rank above the output width: a single bias-free Linear
    assert len(wide) == 1 and wide[0].weight.shape == (10, 12) and wide[0].bias is None
    torch.manual_seed(0)
    a = ProjectionPlan.mlp(64, 10, 256, False).build()
    assert abs(a[0].weight.std().item() - (2.0 / 64) ** 0.5) < 0.02  # kaiming_normal_(nonlinearity="relu"): std = sqrt(2 / fan_in)


def test_graphed_step_keys_batches_by_tensor_shapes_only():
    """tunevlseg_amd/graph.py: one captured graph per padded text length -- the key is (name, shape, dtype) of the tensors, nothing else."""
    from tunevlseg_amd.graph import _key

    def batch(L, extra=None):
        b = {"image": torch.zeros(2, 3, 8, 8), "input_ids": torch.zeros(2, L, dtype=torch.long), "attention_mask": torch.ones(2, L, dtype=torch.long),
             "mask": torch.zeros(2, 1, 8, 8)}
        if extra is not None:
            b["mask_name"] = extra
        return b

    assert _key(batch(6)) == _key(batch(6, extra=["a", "b"]))        # non-tensor entries do not count
    assert _key(batch(6)) != _key(batch(7))                            # another padded length, another graph
    other = batch(6)
    other["image"] = other["image"].double()
    assert _key(batch(6)) != _key(other)


def test_ranks_sharing_one_device_are_refused_without_the_debug_flag(monkeypatch):
    """One process per GPU is a requirement: TVL_DIST_BACKEND=gloo, which puts every rank on cuda:0, needs
    TVL_ALLOW_SHARED_DEVICE=1 to say that it is a debugging rehearsal."""
    from tunevlseg_amd import dist as tdist

    for k, v in (("RANK", "1"), ("LOCAL_RANK", "1"), ("WORLD_SIZE", "4"), ("TVL_DIST_BACKEND", "gloo")):
        monkeypatch.setenv(k, v)
    monkeypatch.delenv("TVL_ALLOW_SHARED_DEVICE", raising=False)
    with pytest.raises(RuntimeError, match="TVL_ALLOW_SHARED_DEVICE"):
        tdist.init_distributed("cuda")
