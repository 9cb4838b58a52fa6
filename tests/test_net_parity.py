"""End-to-end parity of the HIP path against the REFERENCE's own outputs (golden fixtures) and the CPU oracle.

Boundary under test: ``net(text_input, image_input) -> logits[B,1,H,W]`` + DiceCE loss + prompt gradients
(reference image_text_mask_module.py:87-107,257-265).  Tolerances (north_star): |logits - ref| <= 1e-3 absolute
(achieved: ~1e-5), loss 1e-5, grads 1e-3 relative to the largest reference gradient entry, thresholded label map
bit-exact, Dice/IoU 1e-3.
"""
from functools import partial

import pytest
import torch

from tests.golden_util import config_of, golden_names, grad_of, inputs_of, load_golden, trainable_of

pytestmark = pytest.mark.gpu

LOGIT_TOL = 1e-3
GRAD_RTOL = 1e-3


def build_net(fx, device="cuda"):
    from tunevlseg_amd import nets
    from tunevlseg_amd.nets import context_learner as CL

    m = fx["meta"]
    kind = m["net"]
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
    if "_tlayer" in lkw:
        lkw["unified_projector"] = partial(torch.nn.TransformerEncoderLayer, **lkw.pop("_tlayer"))
    net_cls = {"vpt": nets.VPTCLIPSeg, "coop": nets.COOPCLIPSeg, "cocoop": nets.COOPCLIPSeg, "maple": nets.MapleCLIPSeg,
               "shared_separate": nets.SharedSeparateCLIPSeg, "shared_attn": nets.SharedAttnCLIPSeg}[kind]
    learner_cls = {"vpt": CL.VPTContextLearner, "coop": CL.CoOpContextLearner, "cocoop": CL.CoCoOpContextLearner,
                   "maple": CL.MapleContextLearner, "shared_separate": CL.SharedSeparateLearner, "shared_attn": CL.SharedAttnLearner}[kind]
    spec = f"random:{m['preset']}:seed={m['weight_seed']}:eos={m['eos_token_id']}:tails={m.get('tails', 0)}"
    net = net_cls(context_learner=partial(learner_cls, **lkw),
                  model_cfg={"pretrained_model_name_or_path": spec, "freeze_encoder": False, "freeze_decoder": False}, **m["net_kw"])
    params = trainable_of(fx, requires_grad=False)
    own = dict(net.named_parameters())
    trainable = {k for k, p in own.items() if p.requires_grad}
    assert trainable == set(params), (sorted(trainable), sorted(params))  # same trainable surface as the reference
    with torch.no_grad():
        for k, v in params.items():
            own[k].copy_(v)
    return net.to(device)


def run_case(name):
    from tunevlseg_amd import ops

    fx = load_golden(name)
    net = build_net(fx)
    pix, ids, am, mask = (t.cuda() for t in inputs_of(fx))
    logits = net(text_input={"input_ids": ids, "attention_mask": am}, image_input=pix)
    ref = torch.from_numpy(fx["out.logits"])
    assert logits.shape == ref.shape
    err = (logits.detach().cpu() - ref).abs().max().item()
    assert err <= LOGIT_TOL, f"{name}: logits max abs err {err:.3e}"
    loss, isum = ops.DiceCELossFn.apply(logits, mask, 1.0, 0.2, 0.5)
    assert abs(loss.item() - float(fx["out.loss"])) <= 1e-5, (loss.item(), float(fx["out.loss"]))
    # integer label map: bit-exact against the reference's sigmoid(logits) > 0.5
    lab_ref = torch.sigmoid(ref) > 0.5
    lab = torch.sigmoid(logits.detach().cpu()) > 0.5
    from oracle import clipseg_oracle as O
    from tunevlseg_amd import hip as _hip

    exact_mode = _hip.GEMM_MODE in ("f32", "bf16x6")  # fp32-equivalent arithmetic: label map must be bit-exact
    flips = (lab != lab_ref).sum().item()
    # heavy-tailed fixtures carry the reference's float64 logits: where the reference's OWN fp32 logit is within its fp32-vs-float64
    # deviation of the threshold, its label is a coin flip of that rounding noise and is not held against this path
    noise = (ref - torch.from_numpy(fx["out.logits64"])).abs().max().item() if "out.logits64" in fx else 0.0
    if exact_mode and noise:
        assert ((lab != lab_ref) & (ref.abs() > 3.0 * noise)).sum().item() == 0, f"{name}: label pixels differ away from the threshold"
        exact_mode = flips == 0   # the count comparison against the reference below presumes identical label maps
    elif exact_mode:
        assert flips == 0, f"{name}: {flips} label pixels differ"
    else:  # reduced-precision modes: only pixels whose reference logit is within the logit tolerance may flip
        assert ((lab != lab_ref) & (ref.abs() > LOGIT_TOL)).sum().item() == 0
    # integer statistics are exact given the logits this path produced
    tp, fp, fn, tn = O.confusion_counts(torch.sigmoid(logits.detach().cpu()), torch.from_numpy(fx["in.mask"]).long())
    assert torch.equal(isum.cpu(), torch.stack((tp, fp, fn, tn), 1))
    if exact_mode:
        tp, fp, fn, tn = O.confusion_counts(torch.sigmoid(ref), torch.from_numpy(fx["in.mask"]).long())
        assert torch.equal(isum.cpu(), torch.stack((tp, fp, fn, tn), 1))
    loss.backward()
    worst = 0.0
    agg = [0.0, 0.0, 0.0]  # squared error of HIP / of the reference's fp32 run vs the float64 gradients, and their squared norm
    for k, p in net.named_parameters():
        if not p.requires_grad:
            continue
        if k in fx["meta"]["grads_none"]:
            assert p.grad is None or p.grad.abs().max().item() == 0, k
            continue
        assert p.grad is not None, f"{name}: no grad for {k}"
        g_ref, g_own = grad_of(fx, k, p.grad.cpu())
        scale = g_ref.abs().max().item() + 1e-12
        if "grad64." + k in fx:
            # ill-conditioned fixture (deep prompts: the gradient is a small difference of larger terms).  The fixture also holds
            # the reference run in float64: its own fp32 gradient is up to 1.2e-3 (per tensor, relative to the largest entry) away
            # from that exact value on the CoCoOp depth-10 fixture, so "within 1e-3 of the reference" is not a meaningful gate
            # there.  Gate instead against the EXACT gradient: per tensor within 1e-3, or within 5x the reference's own fp32
            # deviation (individual tensors scatter: measured up to 3.3x); over all tensors together (relative L2, below) the
            # HIP path must carry the same order of rounding noise as the reference (<= 2.5x; measured ~1.7x).
            truth = torch.from_numpy(fx["grad64." + k])
            ref_noise = (g_ref - truth).abs().max().item()
            gerr = (p.grad.cpu() - truth).abs().max().item()
            # Heavy-tailed fixtures amplify rounding noise further, and every fp32-equivalent evaluation is ONE draw of it: on
            # rd64_maple_n4_d9_newlast_tails the all-gradient rel-L2 error vs float64 measured 2.4e-4 (32x32x16 MFMA ring), 1.0e-3
            # (16x16x32 ring: same per-instruction accuracy, tools/gemm_accuracy_h2.py), 2.8e-3 (three bf16 pieces) against the
            # reference's own 1.4e-3; single small tensors scatter up to 7x the reference's deviation.  Per tensor they get 10x;
            # the all-tensor gate below (2.5x) is unchanged -- a scale bound that broke would be off by orders of magnitude.
            per_tensor = 10.0 if fx["meta"].get("tails") else 5.0
            assert gerr <= max(GRAD_RTOL * scale, per_tensor * ref_noise) + 1e-9, f"{name}: grad {k} err vs fp64 {gerr:.3e} (reference fp32: {ref_noise:.3e}) scale {scale:.3e}"
            agg[0] += (p.grad.cpu().double() - truth.double()).pow(2).sum().item()
            agg[1] += (g_ref.double() - truth.double()).pow(2).sum().item()
            agg[2] += truth.double().pow(2).sum().item()
        else:
            gerr = (g_own - g_ref).abs().max().item()
            assert gerr <= GRAD_RTOL * scale + 1e-9, f"{name}: grad {k} err {gerr:.3e} scale {scale:.3e}"
        worst = max(worst, gerr / scale)
    from tunevlseg_amd import hip
    if agg[2] > 0:
        ours, theirs = (agg[0] / agg[2]) ** 0.5, (agg[1] / agg[2]) ** 0.5
        print(f"PARITY(fp64) case={name} rel-L2 error of all gradients vs the float64 reference: HIP {ours:.3e}, reference fp32 {theirs:.3e}")
        assert ours <= max(GRAD_RTOL, 2.5 * theirs), (ours, theirs)
    print(f"PARITY mode={hip.GEMM_MODE} case={name} logit_err={err:.3e} loss_err={abs(loss.item() - float(fx['out.loss'])):.2e} "
          f"worst_grad_rel={worst:.3e} label_flips={(lab != lab_ref).sum().item()}")
    return err


@pytest.mark.parametrize("name", golden_names("tiny_"))
def test_hip_net_matches_reference_tiny(name):
    run_case(name)


FULL_B1 = [n for n in golden_names("rd64_") if "_b32" not in n and not n.endswith("_tails2")]


@pytest.mark.parametrize("name", FULL_B1)
def test_hip_net_matches_reference_full_size(name):
    run_case(name)


@pytest.mark.parametrize("name", FULL_B1)
def test_hip_net_matches_reference_full_size_tp3(name, monkeypatch):
    """The same B = 1 fixtures forced onto the tp3 kernels (LDS-DMA GEMM ring, tp3-writing LayerNorm / attention) that the
    benchmarked batch size selects by itself."""
    from tunevlseg_amd import hip

    monkeypatch.setattr(hip, "TP3_MIN_ROWS", 1)
    hip.gemm_profile_start()
    run_case(name)
    prof = hip.gemm_profile_stop()
    assert any(k.startswith(("gemm_tp3_kernel", "gemm_h2m_kernel")) for k in prof), sorted(prof)


def run_compact_case(name):
    """Full-batch fixtures (compact form): inputs re-drawn from the seed; compared are the loss, every trainable gradient, the
    per-sample integer TP/FP/FN/TN counts (bit-exact), per-sample Dice and every 11th logit."""
    import numpy as np

    from tests.golden_util import check_compact_labels, synth_inputs
    from tunevlseg_amd import hip, ops

    fx = load_golden(name)
    m = fx["meta"]
    assert m["compact"]
    net = build_net(fx)
    pix, ids, am, mask = (t.cuda() for t in synth_inputs(config_of(fx), m["B"], m["H"], m["L"], m["input_seed"]))
    hip.gemm_profile_start()
    logits = net(text_input={"input_ids": ids, "attention_mask": am}, image_input=pix)
    loss, isum = ops.DiceCELossFn.apply(logits, mask, 1.0, 0.2, 0.5)
    loss.backward()
    prof = hip.gemm_profile_stop()
    ref_s = torch.from_numpy(fx["out.logits_s11"])
    err = (logits.detach()[..., ::11, ::11].cpu() - ref_s).abs().max().item()
    assert err <= LOGIT_TOL, f"{name}: strided logits max abs err {err:.3e}"
    assert abs(loss.item() - float(fx["out.loss"])) <= 1e-5, (loss.item(), float(fx["out.loss"]))
    flips = check_compact_labels(fx, logits, isum, mask)
    tp, fp, fn = (isum[:, i].double().cpu() for i in range(3))
    den = 2 * tp + fp + fn
    dice = torch.where(den > 0, 2 * tp / den.clamp(min=1), torch.ones_like(den))
    assert np.allclose(dice.numpy(), fx["out.dice_per_sample"], atol=1e-3 if flips else 1e-12)
    worst = 0.0
    for k, p in net.named_parameters():
        if not p.requires_grad or k in m["grads_none"]:
            continue
        g_ref, g_own = grad_of(fx, k, p.grad.cpu())
        scale = g_ref.abs().max().item() + 1e-12
        if "grad64." + k in fx:   # heavy-tailed batch fixture: anchored at the reference's float64 gradient, as in run_case
            truth = torch.from_numpy(fx["grad64." + k])
            ref_noise = (g_ref - truth).abs().max().item()
            gerr = (p.grad.cpu() - truth).abs().max().item()
            assert gerr <= max(GRAD_RTOL * scale, 10.0 * ref_noise) + 1e-9, f"{name}: grad {k} err vs fp64 {gerr:.3e} (reference fp32: {ref_noise:.3e}) scale {scale:.3e}"
            ours, theirs = ((p.grad.cpu().double() - truth.double()).norm() / truth.double().norm()).item(), ((g_ref.double() - truth.double()).norm() / truth.double().norm()).item()
            print(f"PARITY(fp64, compact) case={name} {k}: rel-L2 vs float64 HIP {ours:.3e}, reference fp32 {theirs:.3e}")
            assert ours <= max(GRAD_RTOL, 2.5 * theirs), (ours, theirs)
        else:
            gerr = (g_own - g_ref).abs().max().item()
            assert gerr <= GRAD_RTOL * scale + 1e-9, f"{name}: grad {k} err {gerr:.3e} scale {scale:.3e}"
        worst = max(worst, gerr / scale)
    print(f"PARITY(compact) case={name} label_flips_at_ambiguous_pixels={flips} logit_err={err:.3e} loss_err={abs(loss.item() - float(fx['out.loss'])):.2e} worst_grad_rel={worst:.3e} "
          f"gemm kernels={sorted(prof)}")
    return prof


def test_hip_net_matches_reference_headline_batch():
    """BASELINE configs[1] exactly (VPT-10 shallow, 352x352, B = 32): the configuration bench.py times.  Its M = 15 840 rows must
    have gone through the large-tile GEMM (the 192-row tp3 ring), which no B = 1 fixture reaches by itself."""
    prof = run_compact_case("rd64_vpt_n10_d1_b32")
    big = [k for k in prof if k.startswith(("gemm_tp3_kernel<192", "gemm_h2m_kernel<192")) or "gemm_bf16s_kernel<192" in k]
    assert big, sorted(prof)


def test_eval_forward_without_a_tape_matches_the_headline_fixture():
    """The eval / predict forward (``torch.no_grad()``, what validation, test and predict run under: no saved activations, ``want_stats`` off,
    fc1 without its pre-activation output) at the headline batch against the same reference logits, label map and integer counts."""
    from tests.golden_util import check_compact_labels, synth_inputs
    from tunevlseg_amd import ops

    fx = load_golden("rd64_vpt_n10_d1_b32")
    m = fx["meta"]
    net = build_net(fx)
    pix, ids, am, mask = (t.cuda() for t in synth_inputs(config_of(fx), m["B"], m["H"], m["L"], m["input_seed"]))
    with torch.no_grad():
        logits = net(text_input={"input_ids": ids, "attention_mask": am}, image_input=pix)
        assert not logits.requires_grad
        loss, isum = ops.DiceCELossFn.apply(logits, mask, 1.0, 0.2, 0.5)
    err = (logits[..., ::11, ::11].cpu() - torch.from_numpy(fx["out.logits_s11"])).abs().max().item()
    assert err <= LOGIT_TOL, err
    assert abs(loss.item() - float(fx["out.loss"])) <= 1e-5
    check_compact_labels(fx, logits, isum, mask)
    # ... and it is the same arithmetic as the taped forward: bit for bit
    taped = net(text_input={"input_ids": ids, "attention_mask": am}, image_input=pix)
    assert torch.equal(taped.detach(), logits)


def test_hip_net_matches_reference_maple_per_gpu_batch():
    """BASELINE configs[3]'s per-GPU step (MaPLe depth 9, 4 context tokens, new last layer, B = 32): both towers train, the deep-prompt
    overwrites and the coupling MLPs run at the batch the 8-GPU config gives every rank."""
    prof = run_compact_case("rd64_maple_n4_d9_newlast_b32")
    assert any(k.startswith("gemm_h2m_kernel<") or (k.startswith("gemm_tp3_kernel<") and ", 2, " in k) for k in prof), sorted(prof)


def test_hip_net_matches_reference_headline_batch_heavy_tails():
    """The headline batch with the outlier-channel weight preset (weights.heavy_tails level 1): the h2 operand images' scale bounds
    (Cauchy-Schwarz row bounds, one scale per tensor for QKV / dO) under rows that span decades."""
    run_compact_case("rd64_vpt_n10_d1_b32_tails")


@pytest.mark.parametrize("force_tp3", [False, True])
def test_hip_net_in_the_chaotic_tail_regime_is_no_further_from_float64_than_the_reference(force_tp3, monkeypatch):
    """weights.heavy_tails level 2 (LayerNorm gains 30-100, out_proj / fc2 rows x20, q / k bias +-8): attention saturates and the
    REFERENCE's own fp32 logits are 0.14 away from its float64 logits, so "within 1e-3 of the reference" has no meaning here.  The
    fixture carries the float64 run of the reference classes; the HIP path must be no further from it than a small multiple of the
    reference's own fp32 deviation -- a format whose scale bounds broke under the outliers would be orders of magnitude off."""
    from tunevlseg_amd import hip, ops

    if force_tp3:
        monkeypatch.setattr(hip, "TP3_MIN_ROWS", 1)
    fx = load_golden("rd64_vpt_n10_d1_tails2")
    net = build_net(fx)
    pix, ids, am, mask = (t.cuda() for t in inputs_of(fx))
    logits = net(text_input={"input_ids": ids, "attention_mask": am}, image_input=pix)
    l64, l32 = torch.from_numpy(fx["out.logits64"]), torch.from_numpy(fx["out.logits"])
    ref_dev = (l32 - l64).abs().max().item()
    own_dev = (logits.detach().cpu() - l64).abs().max().item()
    loss, _ = ops.DiceCELossFn.apply(logits, mask, 1.0, 0.2, 0.5)
    loss.backward()
    k = "context_learner.context_vectors"
    g64, g32 = torch.from_numpy(fx["grad64." + k]).double(), torch.from_numpy(fx["grad." + k]).double()
    g = dict(net.named_parameters())[k].grad.cpu().double()
    ref_g = ((g32 - g64).norm() / g64.norm()).item()
    own_g = ((g - g64).norm() / g64.norm()).item()
    print(f"PARITY(tails2, forced_tp3={force_tp3}) logits vs fp64: HIP {own_dev:.3e}, reference fp32 {ref_dev:.3e}; gradient rel-L2 vs fp64: HIP {own_g:.3e}, reference fp32 {ref_g:.3e}")
    assert own_dev <= max(LOGIT_TOL, 3.0 * ref_dev)
    # the gradient is reported, not gated: in this regime the reference's own fp32 gradient is 57 % (rel-L2) away from its float64 one,
    # and the amplification is heavy-tailed -- measured here between 0.3x and 34x that deviation across fp32-equivalent arithmetic
    # variants (three bf16 pieces in-kernel, two fp16 pieces on either MFMA shape).  It has to exist and be finite.
    assert torch.isfinite(g).all() and own_g == own_g
    # labels: bit-equal to the float64 run wherever its logit is further from the threshold than the reference's own fp32 deviation
    lab, lab64 = torch.sigmoid(logits.detach().cpu()) > 0.5, l64 > 0
    assert ((lab != lab64) & (l64.abs() > 3.0 * ref_dev)).sum().item() == 0


def test_vpt_conditional_embedding_cache_skips_the_text_tower_and_keeps_the_logits(monkeypatch):
    """Opt-in extension (VPTCLIPSeg(cache_text_features=True)): rows seen before do not run the text tower again; logits equal the
    uncached net's (the frozen text tower is a pure function of the token row; trailing padding is not part of the key)."""
    from functools import partial

    from tunevlseg_amd import nets
    from tunevlseg_amd.nets import towers
    from tunevlseg_amd.nets.context_learner import VPTContextLearner

    def make(cache):
        torch.manual_seed(0)
        return nets.VPTCLIPSeg(context_learner=partial(VPTContextLearner, prompt_depth=2, num_context=4), cache_text_features=cache,
                               model_cfg={"pretrained_model_name_or_path": "random:tiny:seed=5"}).cuda()

    plain, cached = make(False), make(True)
    g = torch.Generator().manual_seed(1)
    pix = torch.randn(3, 3, 64, 64, generator=g).cuda()
    ids = torch.tensor([[62, 5, 9, 63, 1, 1], [62, 7, 11, 13, 63, 1], [62, 5, 9, 63, 1, 1]]).cuda()
    am = (ids != 1).long()
    calls = []
    real = towers.text_tower
    monkeypatch.setattr(towers, "text_tower", lambda model, i, a, *r, **k: (calls.append(i.shape[0]), real(model, i, a, *r, **k))[1])
    ref = plain(text_input={"input_ids": ids, "attention_mask": am}, image_input=pix)
    calls.clear()
    a = cached(text_input={"input_ids": ids, "attention_mask": am}, image_input=pix)
    assert calls == [2], calls   # rows 0 and 2 are the same phrase: one text-tower call on the two distinct rows
    # the same phrases, padded to a longer row: no text-tower call at all
    ids2 = torch.cat([ids, torch.ones(3, 2, dtype=ids.dtype, device=ids.device)], dim=1)
    b = cached(text_input={"input_ids": ids2, "attention_mask": (ids2 != 1).long()}, image_input=pix)
    assert calls == [2], calls
    for out in (a, b):
        assert (out - ref).abs().max().item() <= 1e-5
    cached.clear_text_cache()
    cached(text_input={"input_ids": ids, "attention_mask": am}, image_input=pix)
    assert calls == [2, 2]


def test_shared_attn_learner_dropout_is_on_in_fit_and_off_in_eval():
    """SharedAttn with the reference's dropout 0.25 (configs/model/shared_attn_clipseg.yaml:21; the fixtures use 0): in train mode
    two forward passes differ and both differ from eval; in eval mode (what validation / test / predict run under,
    trainer.evaluation_mode) the net is deterministic and equals the dropout-free fixture; averaging train-mode passes moves towards
    the eval logits (inverted dropout keeps the expectation of the dropped activations; the LayerNorms behind them keep it from
    converging all the way)."""
    from tunevlseg_amd.trainer import evaluation_mode

    fx = load_golden("tiny_sharedattn_d2")
    fx["meta"]["learner_kw"] = dict(fx["meta"]["learner_kw"])
    fx["meta"]["learner_kw"]["_tlayer"] = dict(fx["meta"]["learner_kw"]["_tlayer"], dropout=0.25)
    net = build_net(fx)
    pix, ids, am, _ = (t.cuda() for t in inputs_of(fx))
    call = lambda: net(text_input={"input_ids": ids, "attention_mask": am}, image_input=pix).detach()  # noqa: E731
    net.context_learner.train()
    a, b = call(), call()
    assert not torch.equal(a, b)
    with evaluation_mode(net):
        e1, e2 = call(), call()
    assert torch.equal(e1, e2) and not torch.equal(e1, a)
    assert net.context_learner.training  # the flag came back
    ref = torch.from_numpy(fx["out.logits"]).cuda()
    assert (e1 - ref).abs().max().item() <= LOGIT_TOL   # eval == the dropout-free fixture
    mean = torch.stack([call() for _ in range(64)]).mean(0)
    spread = (a - e1).abs().mean().item()
    assert (mean - e1).abs().mean().item() < 0.8 * spread, ((mean - e1).abs().mean().item(), spread)
