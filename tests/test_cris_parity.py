"""End-to-end parity of the HIP CRIS path (BASELINE configs[2]) against the REFERENCE's own outputs (golden fixtures made
by running reference ``COOPCRIS``, tests/golden/make_goldens.py).

Boundary: ``COOPCRIS(text_input, image_input) -> logits[B,1,H,W]`` + DiceCE loss + gradients of every trainable tensor
(prompts, meta-net, new last layer, residual ratio).  Tolerances as for the CLIPSeg nets: |logits - ref| <= 1e-3,
loss 1e-5, gradients 1e-3 relative to the largest reference entry, thresholded label map bit-exact.
"""
from functools import partial

import pytest
import torch

from tests.golden_util import cris_config_of, cris_state_of, golden_names, inputs_of, load_golden, trainable_of

pytestmark = pytest.mark.gpu

LOGIT_TOL = 1e-3
GRAD_RTOL = 1e-3


def build_net(fx, device="cuda"):
    from tunevlseg_amd.nets import COOPCRIS
    from tunevlseg_amd.nets import context_learner as CL

    m = fx["meta"]
    lkw = dict(m["learner_kw"])
    if lkw.get("context_initializer") is not None:
        ids = lkw.pop("_init_ids")

        class Tok:
            def __call__(self, text, **kw):
                class R:
                    pass

                r = R()
                r.input_ids = torch.tensor([ids] * (1 if isinstance(text, str) else len(text)), dtype=torch.long)
                return r

        lkw["tokenizer"] = Tok()
    learner_cls = {"coop": CL.CoOpContextLearner, "cocoop": CL.CoCoOpContextLearner}[m["net"]]
    cfg = cris_config_of(fx)
    net = COOPCRIS(model_cfg={"clip_pretrain": {"config": cfg, "state_dict": cris_state_of(fx)}, "img_size": cfg.img_size,
                              "freeze_encoder": True, "cris_pretrain": None},
                   context_learner=partial(learner_cls, **lkw), **m["net_kw"])
    params = trainable_of(fx, requires_grad=False)
    own = dict(net.named_parameters())
    trainable = {k for k, p in own.items() if p.requires_grad}
    assert trainable == set(params), (sorted(trainable), sorted(params))  # same trainable surface as the reference
    with torch.no_grad():
        for k, v in params.items():
            own[k].copy_(v)
    return net.to(device)


def run_case(name):
    from tunevlseg_amd import hip, ops

    fx = load_golden(name)
    net = build_net(fx)
    pix, ids, am, mask = inputs_of(fx)
    text_input = {"input_ids": ids.cuda()}
    if am is not None:
        text_input["attention_mask"] = am.cuda()
    logits = net(text_input=text_input, image_input=pix.cuda())
    ref = torch.from_numpy(fx["out.logits"])
    assert logits.shape == ref.shape
    err = (logits.detach().cpu() - ref).abs().max().item()
    assert err <= LOGIT_TOL, f"{name}: logits max abs err {err:.3e}"
    loss, _ = ops.DiceCELossFn.apply(logits, mask.cuda(), 1.0, 0.2, 0.5)
    assert abs(loss.item() - float(fx["out.loss"])) <= 1e-5, (loss.item(), float(fx["out.loss"]))
    lab_ref = torch.sigmoid(ref) > 0.5
    lab = torch.sigmoid(logits.detach().cpu()) > 0.5
    flips = (lab != lab_ref).sum().item()
    if hip.GEMM_MODE in ("f32", "bf16x6"):
        assert flips == 0, f"{name}: {flips} label pixels differ"
    loss.backward()
    worst = 0.0
    for k, p in net.named_parameters():
        if not p.requires_grad:
            continue
        if k in fx["meta"]["grads_none"]:
            assert p.grad is None or p.grad.abs().max().item() == 0, k
            continue
        g_ref = torch.from_numpy(fx["grad." + k])
        assert p.grad is not None, f"{name}: no grad for {k}"
        scale = g_ref.abs().max().item() + 1e-12
        gerr = (p.grad.cpu() - g_ref).abs().max().item()
        assert gerr <= GRAD_RTOL * scale + 1e-9, f"{name}: grad {k} err {gerr:.3e} scale {scale:.3e}"
        worst = max(worst, gerr / scale)
    print(f"PARITY mode={hip.GEMM_MODE} case={name} logit_err={err:.3e} loss_err={abs(loss.item() - float(fx['out.loss'])):.2e} "
          f"worst_grad_rel={worst:.3e} label_flips={flips}")


@pytest.mark.parametrize("name", golden_names("cris_tiny_"))
def test_hip_cris_matches_reference_tiny(name):
    run_case(name)


@pytest.mark.parametrize("name", [n for n in golden_names("cris_rn50_") if not n.endswith(("_b8", "_b32"))])
def test_hip_cris_matches_reference_full_size(name):
    run_case(name)


@pytest.mark.parametrize("name", ["cris_rn50_cocoop_n4_d1_newlast_b8", "cris_rn50_cocoop_n4_d1_newlast_b32"])
def test_hip_cris_matches_reference_batch(name):
    """BASELINE configs[2] geometry (RN50, 416x416) at B = 8 (M = B*H*W rows that select the large implicit-conv tiles) and at the
    BASELINE batch itself, B = 32.  Compact fixtures: inputs re-drawn from the seed; compared are loss, all gradients, per-sample
    integer counts / Dice, every 11th logit."""
    import numpy as np

    from tests.golden_util import check_compact_labels, synth_cris_inputs
    from tunevlseg_amd import hip, ops

    fx = load_golden(name)
    m = fx["meta"]
    assert m["compact"]
    net = build_net(fx)
    pix, ids, am, mask = synth_cris_inputs(cris_config_of(fx), m["B"], m["L"], m["input_seed"], m["with_attention_mask"])
    text_input = {"input_ids": ids.cuda()}
    if am is not None:
        text_input["attention_mask"] = am.cuda()
    hip.gemm_profile_start()
    logits = net(text_input=text_input, image_input=pix.cuda())
    loss, isum = ops.DiceCELossFn.apply(logits, mask.cuda(), 1.0, 0.2, 0.5)
    loss.backward()
    prof = hip.gemm_profile_stop()
    err = (logits.detach()[..., ::11, ::11].cpu() - torch.from_numpy(fx["out.logits_s11"])).abs().max().item()
    assert err <= LOGIT_TOL, f"{name}: strided logits max abs err {err:.3e}"
    assert abs(loss.item() - float(fx["out.loss"])) <= 1e-5, (loss.item(), float(fx["out.loss"]))
    flips = check_compact_labels(fx, logits, isum, mask)
    tp, fp, fn = (isum[:, i].double().cpu() for i in range(3))
    den = 2 * tp + fp + fn
    assert np.allclose(torch.where(den > 0, 2 * tp / den.clamp(min=1), torch.ones_like(den)).numpy(), fx["out.dice_per_sample"], atol=1e-3 if flips else 1e-12)
    worst = 0.0
    for k, p in net.named_parameters():
        if not p.requires_grad or k in m["grads_none"]:
            continue
        g_ref = torch.from_numpy(fx["grad." + k])
        scale = g_ref.abs().max().item() + 1e-12
        gerr = (p.grad.cpu() - g_ref).abs().max().item()
        assert gerr <= GRAD_RTOL * scale + 1e-9, f"{name}: grad {k} err {gerr:.3e} scale {scale:.3e}"
        worst = max(worst, gerr / scale)
    assert any("192" in k for k in prof), sorted(prof)  # the large-tile kernels ran
    print(f"PARITY(compact) case={name} label_flips_at_ambiguous_pixels={flips} logit_err={err:.3e} worst_grad_rel={worst:.3e} kernels={sorted(prof)}")
