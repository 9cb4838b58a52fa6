"""CPU-only checks of the CRIS host logic: weight preparation (eval-BatchNorm folding, im2col-ordered conv matrices and
their tap-flipped transposes for data gradients), constants, constructor surface.  No HIP calls."""
from functools import partial

import pytest
import torch
import torch.nn.functional as F

from oracle import cris_oracle as OC
from tunevlseg_amd.cris_backbone import CRISWeights, conv3_matrices
from tunevlseg_amd.cris_config import CRISConfig
from tunevlseg_amd.weights import cris_param_specs, init_cris_state_dict


def im2col_ref(x):  # [B,C,H,W] -> [B*H*W, 9*C] in (ky, kx, c) column order, pad 1, stride 1
    B, C, H, W = x.shape
    xp = F.pad(x, (1, 1, 1, 1))
    cols = [xp[:, :, ky:ky + H, kx:kx + W].permute(0, 2, 3, 1).reshape(B * H * W, C) for ky in range(3) for kx in range(3)]
    return torch.cat(cols, 1)


@pytest.fixture(scope="module")
def tiny():
    cfg = CRISConfig.tiny()
    sd = init_cris_state_dict(cfg, 31)
    w = CRISWeights(cfg, sd)
    w.requires_grad_(False)
    return cfg, sd, w, w.prepared()


def test_state_dict_round_trips_reference_names(tiny):
    cfg, sd, w, _ = tiny
    own = w.state_dict()
    assert set(own) == set(sd) == {n for n, *_ in cris_param_specs(cfg)}
    assert all(torch.equal(own[k], sd[k]) for k in sd)


def test_folded_conv_bn_relu_matches_oracle(tiny):
    cfg, sd, _, prep = tiny
    fc = prep["neck"]["f4_proj4"]
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, fc.cin, 6, 6, generator=g)
    ref = OC.conv_layer(sd, "neck.f4_proj4", x, 1)  # conv -> eval BN -> ReLU
    cols = im2col_ref(x)
    y = torch.relu(cols @ fc.Wm[:, : cols.shape[1]].t() + fc.b)
    got = y.reshape(2, 6, 6, fc.cout).permute(0, 3, 1, 2)
    assert (got - ref).abs().max().item() <= 2e-5


def test_dgrad_matrix_is_the_transposed_conv():
    g = torch.Generator().manual_seed(1)
    w4 = torch.randn(5, 4, 3, 3, generator=g)
    fc = conv3_matrices(w4, None, True)
    x = torch.randn(2, 4, 5, 7, generator=g, requires_grad=True)
    y = F.conv2d(x, w4, padding=1)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    dx = im2col_ref(dy) @ fc.Wd[:, : 9 * 5].t()  # [B*H*W, Cin]
    got = dx.reshape(2, 5, 7, 4).permute(0, 3, 1, 2)
    assert (got - x.grad).abs().max().item() <= 2e-5


def test_folded_text_gate_and_position_constants(tiny):
    cfg, sd, w, prep = tiny
    g = torch.Generator().manual_seed(2)
    state = torch.randn(3, cfg.word_dim, generator=g)
    tp = prep["neck"]["txt_proj"]
    ref = F.relu(OC.bn(sd, "neck.txt_proj.1", F.linear(state, sd["neck.txt_proj.0.weight"])))
    assert (F.relu(F.linear(state, tp.W, tp.b)) - ref).abs().max().item() <= 2e-5
    assert torch.allclose(w.pos2d(cfg.vis_dim, 6, 6), OC.pos2d(cfg.vis_dim, 6, 6))
    assert torch.allclose(w.pos1d(cfg.transformer_width, 11), OC.pos1d(cfg.transformer_width, 11))
    co = w.coords(2, 3, 4)
    assert co.shape == (24, 32) and co[:, 2:].abs().max() == 0 and co[0, 0] == -1 and co[3, 0] == 1 and co[11, 1] == 1


def test_constructor_surface_and_errors():
    from tunevlseg_amd.nets import COOPCRIS
    from tunevlseg_amd.nets.context_learner import CoCoOpContextLearner, CoOpContextLearner

    mk = lambda **kw: COOPCRIS(model_cfg={"clip_pretrain": "random:tiny:seed=31", "img_size": 96, "fpn_in": [64, 128, 32], "dropout": 0},  # noqa: E731
                               context_learner=partial(CoOpContextLearner, prompt_depth=2, num_context=3), **kw)
    net = mk(use_new_last_layer=True)
    assert {k for k, p in net.named_parameters() if p.requires_grad} == {
        "context_learner.context_vectors", "additive_decoder_layer.0.weight", "additive_decoder_layer.2.weight",
        "additive_decoder_layer.2.bias", "residual_ratio"}
    assert not net.training or not any(m.training for m in (net.backbone, net.neck, net.decoder, net.proj))
    keys = set(net.state_dict())
    assert {"backbone.visual.layer1.0.bn1.running_mean", "neck.f2_cat.0.weight", "decoder.layers.0.self_attn.in_proj_weight",
            "proj.txt.weight", "context_learner.context_vectors"} <= keys
    with pytest.raises(NotImplementedError):
        mk(freeze_all=False)
    net2 = COOPCRIS(model_cfg={"clip_pretrain": "random:tiny:seed=31", "img_size": 96},
                    context_learner=partial(CoCoOpContextLearner, prompt_depth=1, num_context=2, intermediate_dim=8))
    assert net2.context_learner.visual_dim == net2.config.embed_dim if hasattr(net2.context_learner, "visual_dim") else True
    pm = net2.get_pad_mask(torch.tensor([[5, 3, 9, 0, 0]]), None)
    assert pm.tolist() == [[False, False, False, False, False, True, True]]  # 2 zeros prepended (coop_context_learner.py:82-114)
    with pytest.raises(ValueError):  # wrong image size is refused up front, before any kernel runs
        net2(text_input={"input_ids": torch.tensor([[5, 3, 9, 0, 0]])}, image_input=torch.zeros(1, 3, 64, 64))


def test_checkpoints_under_the_reference_file_layouts(tmp_path):
    """``clip_pretrain`` is a bare CLIP state dict (the RN50.pt archive has no ``backbone.`` prefix and no neck /
    decoder / projector) and ``cris_pretrain`` the full model, loaded strictly (only BatchNorm's num_batches_tracked may be extra)."""
    cfg = CRISConfig.tiny()
    full = init_cris_state_dict(cfg, 77)
    clip_only = {k[len("backbone."):]: v for k, v in full.items() if k.startswith("backbone.")}
    clip_only["input_resolution"] = torch.tensor(224)  # scalars the TorchScript archive carries besides the weights
    clip_file, cris_file = tmp_path / "RN50_like.pt", tmp_path / "cris_best_single.pth"
    torch.save(clip_only, clip_file)
    other = init_cris_state_dict(cfg, 78)
    cris_sd = {**other, "neck.f1_v_proj.1.num_batches_tracked": torch.tensor(5)}
    torch.save(cris_sd, cris_file)
    w = CRISWeights.from_spec(str(clip_file), overrides={"config": cfg})
    own = w.state_dict()
    assert all(torch.equal(own[k], full[k]) for k in full if k.startswith("backbone."))       # the CLIP part came from the file
    seeded = init_cris_state_dict(cfg, 0)
    assert all(torch.equal(own[k], seeded[k]) for k in full if not k.startswith("backbone."))  # the rest awaits cris_pretrain
    from tunevlseg_amd.nets import COOPCRIS
    from tunevlseg_amd.nets.context_learner import CoOpContextLearner

    net = COOPCRIS(model_cfg={"clip_pretrain": str(clip_file), "cris_pretrain": str(cris_file), "config": cfg, "img_size": cfg.img_size},
                   context_learner=partial(CoOpContextLearner, prompt_depth=1, num_context=2))
    got = {k: v for k, v in net.state_dict().items() if k in other}
    assert got and all(torch.equal(got[k], other[k]) for k in got)  # strict load of the full CRIS checkpoint over it
    bad = dict(other)
    bad.pop("proj.txt.weight")
    torch.save(bad, cris_file)
    with pytest.raises(RuntimeError):  # a checkpoint with a missing tensor is an error, not a silent partial load
        COOPCRIS(model_cfg={"clip_pretrain": str(clip_file), "cris_pretrain": str(cris_file), "config": cfg, "img_size": cfg.img_size},
                 context_learner=partial(CoOpContextLearner, prompt_depth=1, num_context=2))
    clip_only["visual.bogus.weight"] = torch.zeros(1)
    torch.save(clip_only, clip_file)
    with pytest.raises(RuntimeError):
        CRISWeights.from_spec(str(clip_file), overrides={"config": cfg})
