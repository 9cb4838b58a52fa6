"""The DenseCLIP CPU oracle against the reference's own outputs (tests/golden/denseclip_*.npz, made by running the reference's
``CLIPVisionTransformer`` / ``CLIPTextContextEncoder`` / ``ContextDecoder``, tests/golden/make_denseclip_goldens.py): pins
``oracle/denseclip_oracle.py``.  Tolerance: fp32 round-off of two op orders on CPU (2e-5 on O(1) outputs, gradients 2e-4 of their scale)."""
import pytest
import torch

from oracle import denseclip_oracle as OD
from tests.golden_util import (denseclip_config_of, denseclip_state_of, denseclip_subsample, golden_names, load_golden,
                               synth_denseclip_inputs)


def run_oracle(fx, train_decoder: bool = True):
    cfg, sd = denseclip_config_of(fx), denseclip_state_of(fx)
    m = fx["meta"]
    pix, texts, gs, gt = synth_denseclip_inputs(cfg, m["B"], m["H"], m["input_seed"], m.get("W"))
    if "in.pixel_values" in fx:
        assert torch.equal(pix, torch.from_numpy(fx["in.pixel_values"]))
    assert torch.equal(texts, torch.from_numpy(fx["in.texts"]))
    contexts = torch.from_numpy(fx["param.contexts"]).clone().requires_grad_(True)
    gamma = torch.from_numpy(fx["param.gamma"]).clone().requires_grad_(True)
    if train_decoder:
        sd = {k: (v.clone().requires_grad_(True) if k.startswith("context_decoder.") else v) for k, v in sd.items()}
    feats, g, v = OD.vision_forward(sd, cfg, pix)
    text_embeddings, x_orig, score_map = OD.after_extract_feat(sd, cfg, feats, g, v, texts, contexts, gamma)
    loss = (score_map * gs).sum() + (text_embeddings * gt).sum()
    loss.backward()
    grads = {"contexts": contexts.grad, "gamma": gamma.grad, **{k: t.grad for k, t in sd.items() if k.startswith("context_decoder.") and t.requires_grad}}
    out = {"fpn1": feats[0], "fpn2": feats[1], "fpn3": feats[2], "fpn4": feats[3], "global_embedding": g, "visual_embedding": v,
           "text_embeddings": text_embeddings, "score_map": score_map, "loss": loss}
    assert torch.equal(x_orig[cfg.score_concat_index], torch.cat((feats[cfg.score_concat_index], score_map), 1))
    return {k: t.detach() for k, t in out.items()}, grads


def check(fx):
    out, grads = run_oracle(fx)
    compact = fx["meta"]["compact"]
    for k, t in out.items():
        ref = torch.from_numpy(fx["out." + k])
        got = denseclip_subsample(k, t, compact)
        assert got.shape == ref.shape, k
        scale = max(1.0, ref.abs().max().item())
        assert (got - ref).abs().max().item() <= 2e-5 * scale, (k, (got - ref).abs().max().item())
        if f"out.{k}_abs_sum" in fx:   # whole-map checksum (compact fixtures keep a subsample of the large maps)
            s = float(fx[f"out.{k}_abs_sum"])
            assert abs(t.double().abs().sum().item() - s) <= 1e-5 * s, k
    for k, g in grads.items():
        if "grad." + k not in fx:   # compact fixtures hold the context decoder's small gradients + three weight matrices only
            assert fx["meta"]["compact"] and k.startswith("context_decoder.")
            continue
        g_ref = torch.from_numpy(fx["grad." + k])
        scale = g_ref.abs().max().item() + 1e-12
        # the reference's own fp32 gradient vs its float64 run bounds what two fp32 evaluations can agree to
        noise = (g_ref - torch.from_numpy(fx["grad64." + k])).abs().max().item()
        assert (g - g_ref).abs().max().item() <= max(2e-4 * scale, 3.0 * noise) + 1e-9, (k, (g - g_ref).abs().max().item(), scale)


@pytest.mark.parametrize("name", golden_names("denseclip_tiny"))
def test_denseclip_oracle_matches_reference_tiny(name):
    check(load_golden(name))


@pytest.mark.slow
@pytest.mark.parametrize("name", golden_names("denseclip_vitb16"))
def test_denseclip_oracle_matches_reference_full_size(name):
    check(load_golden(name))
