"""DenseCLIP (BASELINE configs[4]) on the HIP path against the reference's own outputs (tests/golden/denseclip_*.npz: the reference's
``CLIPVisionTransformer`` / ``CLIPTextContextEncoder`` / ``ContextDecoder`` run on CPU, tests/golden/make_denseclip_goldens.py) and, kernel by
kernel, against plain torch on the same inputs.  Gates: outputs 1e-3 absolute on O(1) values (north_star's fp32 tolerance; measured ~1e-5),
gradients 1e-3 relative L2 per tensor."""
import pytest
import torch
import torch.nn.functional as F

from tests.golden_util import (denseclip_config_of, denseclip_state_of, denseclip_subsample, golden_names, load_golden,
                               synth_denseclip_inputs)

pytestmark = pytest.mark.gpu


def build(fx, train_decoder: bool):
    from tunevlseg_amd.denseclip_backbone import DenseCLIPWeights
    from tunevlseg_amd.nets import DenseCLIP

    cfg = denseclip_config_of(fx)
    net = DenseCLIP(pretrained=DenseCLIPWeights(cfg, denseclip_state_of(fx)), texts=torch.from_numpy(fx["in.texts"]), train_context_decoder=train_decoder).cuda()
    with torch.no_grad():
        net.contexts.copy_(torch.from_numpy(fx["param.contexts"]))
        net.gamma.copy_(torch.from_numpy(fx["param.gamma"]))
    return net, cfg


def rel_l2(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp(min=1e-30))


@pytest.mark.parametrize("name", golden_names("denseclip_"))
def test_denseclip_matches_reference(name):
    fx = load_golden(name)
    m = fx["meta"]
    net, cfg = build(fx, train_decoder=True)
    pix, texts, gs, gt = synth_denseclip_inputs(cfg, m["B"], m["H"], m["input_seed"], m.get("W"))
    text_embeddings, maps, score_map = net(pix.cuda())
    loss = (score_map * gs.cuda()).sum() + (text_embeddings * gt.cuda()).sum()
    loss.backward()
    K, Cw = cfg.num_classes, cfg.width
    assert maps[cfg.score_concat_index].shape[1] == Cw + K
    assert torch.equal(maps[cfg.score_concat_index][:, Cw:], score_map.detach())
    feats = net.extract_feat(pix.cuda())
    out = {"fpn1": maps[0], "fpn2": maps[1], "fpn3": maps[2][:, :Cw], "fpn4": maps[3], "global_embedding": feats[4][0], "visual_embedding": feats[4][1],
           "text_embeddings": text_embeddings, "score_map": score_map}
    for i in range(4):   # extract_feat hands out the same maps (before the score map is concatenated)
        assert torch.equal(feats[i], out[f"fpn{i + 1}"]), i
    worst = {}
    for k, t in out.items():
        ref = torch.from_numpy(fx["out." + k])
        got = denseclip_subsample(k, t.detach().cpu(), m["compact"])
        assert got.shape == ref.shape, (k, got.shape, ref.shape)
        worst[k] = (got - ref).abs().max().item()
        assert worst[k] <= 1e-3 * max(1.0, ref.abs().max().item()), (k, worst[k])
        if f"out.{k}_abs_sum" in fx:
            s = float(fx[f"out.{k}_abs_sum"])
            assert abs(t.detach().double().abs().sum().item() - s) <= 1e-4 * s, k
    assert abs(loss.item() - float(fx["out.loss"])) <= 1e-3 * max(1.0, abs(float(fx["out.loss"])))
    grads = {"contexts": net.contexts.grad, "gamma": net.gamma.grad, **{"context_decoder." + k: p.grad for k, p in net.context_decoder.named_parameters()}}
    gworst = {}
    for k, g in grads.items():
        if "grad." + k not in fx:
            continue
        assert g is not None, k
        ref64 = torch.from_numpy(fx["grad64." + k])
        # anchored at the reference's float64 gradient: no further from it than 1e-3, or 5x the reference's own fp32 distance
        own = rel_l2(torch.from_numpy(fx["grad." + k]), ref64)
        gworst[k] = rel_l2(g.detach().cpu().view(ref64.shape), ref64)
        assert gworst[k] <= max(1e-3, 5.0 * own), (k, gworst[k], own)
    print(f"{name}: worst output errors {({k: f'{v:.1e}' for k, v in worst.items()})}; worst gradient rel-L2 {max(gworst.values()):.2e} ({max(gworst, key=gworst.get)})")


def test_frozen_context_decoder_gives_the_same_prompt_gradients():
    fx = load_golden("denseclip_tiny_b2_96")
    m = fx["meta"]
    net, cfg = build(fx, train_decoder=False)
    pix, texts, gs, gt = synth_denseclip_inputs(cfg, m["B"], m["H"], m["input_seed"], m.get("W"))
    text_embeddings, maps, score_map = net(pix.cuda())
    ((score_map * gs.cuda()).sum() + (text_embeddings * gt.cuda()).sum()).backward()
    assert all(p.grad is None for p in net.context_decoder.parameters())
    for k, g in (("contexts", net.contexts.grad), ("gamma", net.gamma.grad)):
        assert rel_l2(g.cpu().view(fx["grad." + k].shape), torch.from_numpy(fx["grad." + k])) <= 1e-3, k
    assert sorted(n for n, p in net.named_parameters() if p.requires_grad) == ["contexts", "gamma"]


def test_repeated_steps_without_a_host_sync_are_bit_identical():
    """Forward + backward eight times on one input, no parameter update, no host sync in between: the FPN maps (second stream, beside the context
    decoder), score map, text embeddings and both gradients must be the same bits every time -- a buffer reused across streams while it is still
    read, or a missing stream dependency, shows up as a step that differs (tools/denseclip_soak.py is the long version at 640 x 640)."""
    fx = load_golden("denseclip_tiny_b2_64x96")
    m = fx["meta"]
    net, cfg = build(fx, train_decoder=False)
    pix, texts, gs, gt = synth_denseclip_inputs(cfg, m["B"], m["H"], m["input_seed"], m.get("W"))
    pix, gs, gt = pix.cuda(), gs.cuda(), gt.cuda()
    sums = torch.zeros(8, 8, device="cuda", dtype=torch.float64)
    for i in range(8):
        net.contexts.grad = None
        net.gamma.grad = None
        te, maps, score = net(pix)
        ((score * gs).sum() + (te * gt).sum()).backward()
        sums[i] = torch.stack([t.detach().double().abs().sum() for t in (*maps, score, te, net.contexts.grad, net.gamma.grad)])
        del te, maps, score
    s = sums.cpu()
    assert torch.isfinite(s).all()
    for i in range(1, 8):
        assert torch.equal(s[i], s[0]), (i, s[i].tolist(), s[0].tolist())


def test_state_dict_keys_are_the_reference_segmentors():
    fx = load_golden("denseclip_tiny_b2_96")
    net, cfg = build(fx, train_decoder=False)
    keys = set(net.state_dict())
    from tunevlseg_amd.weights import denseclip_param_specs

    assert keys == {n for n, _, _, _ in denseclip_param_specs(cfg)}


# ---- kernels against plain torch -------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,H,W,C,pool", [(2, 6, 6, 64, 1), (3, 4, 4, 20, 2), (2, 40, 40, 768, 1), (1, 40, 40, 768, 2), (2, 5, 7, 6, 1)])
def test_groupnorm_nhwc_matches_torch(B, H, W, C, pool):
    from tunevlseg_amd import hip

    if pool == 2 and (H % 2 or W % 2):
        pytest.skip("pool needs even sizes")
    g = torch.Generator().manual_seed(B * 100 + C)
    tok = (torch.randn(B, 1 + H * W, C, generator=g) * 3 + 0.7).cuda()
    gamma, beta = torch.randn(C, generator=g).cuda(), torch.randn(C, generator=g).cuda()
    y = hip.groupnorm_nhwc(tok, 1, H, W, gamma, beta, 1e-5, pool=pool)
    x = tok[:, 1:].transpose(1, 2).reshape(B, C, H, W).double()
    ref = F.group_norm(x, 1, gamma.double(), beta.double(), 1e-5)
    if pool == 2:
        ref = F.max_pool2d(ref, 2, 2)
    ref = ref.permute(0, 2, 3, 1).reshape(-1, C)
    assert (y.double() - ref).abs().max().item() <= 2e-5
    wide = torch.zeros((y.shape[0], C + 8), device="cuda")   # into a column range of a wider matrix (the fpn3 | score map concat)
    hip.groupnorm_nhwc(tok, 1, H, W, gamma, beta, 1e-5, pool=pool, out=wide[:, :C])
    assert torch.equal(wide[:, :C], y) and float(wide[:, C:].abs().max()) == 0.0


@pytest.mark.parametrize("levels", [1, 2])
def test_transposed_conv_as_gemm_and_unshuffle_matches_torch(levels):
    from tunevlseg_amd import cris_ops as C
    from tunevlseg_amd import hip
    from tunevlseg_amd.denseclip_backbone import tconv_matrices

    B, H, W, Cc = 2, 5, 6, 32
    g = torch.Generator().manual_seed(levels)
    x = torch.randn(B, Cc, H, W, generator=g)
    ws = [torch.randn(Cc, Cc, 2, 2, generator=g) * 0.2 for _ in range(levels)]
    bs = [torch.randn(Cc, generator=g) for _ in range(levels)]
    ref = x.double()
    for w, b in zip(ws, bs):
        ref = F.conv_transpose2d(ref, w.double(), b.double(), stride=2)
    t = x.permute(0, 2, 3, 1).reshape(B * H * W, Cc).contiguous().cuda()
    for w, b in zip(ws, bs):
        t = C.flinear(t.view(-1, Cc), tconv_matrices(w.cuda(), b.cuda()))
    y = hip.tconv2x2_unshuffle(t, B, H, W, Cc, levels)
    f = 1 << levels
    got = y.view(B, H * f, W * f, Cc).permute(0, 3, 1, 2).cpu().double()
    assert (got - ref).abs().max().item() <= 1e-4


def test_gelu_activation_and_its_derivative_match_torch():
    from tunevlseg_amd import hip

    x = torch.linspace(-6, 6, 4001).view(1, -1).cuda()
    y = hip.bias_act(x, None, hip.ACT_GELU)
    assert (y.cpu().double() - F.gelu(x.cpu().double())).abs().max().item() <= 1e-6
    xd = x.cpu().double().requires_grad_(True)
    F.gelu(xd).sum().backward()
    d = hip.dact_mul(torch.ones_like(x), x, hip.ACT_GELU)
    assert (d.cpu().double() - xd.grad).abs().max().item() <= 1e-6


def test_colscale_add_and_backward():
    from tunevlseg_amd.nets.denseclip import ColScaleAddFn

    g = torch.Generator().manual_seed(5)
    a, b, gm = (torch.randn(6, 5, 32, generator=g).cuda().requires_grad_(True), torch.randn(6, 5, 32, generator=g).cuda().requires_grad_(True),
                torch.randn(32, generator=g).cuda().requires_grad_(True))
    w = torch.randn(6, 5, 32, generator=g).cuda()
    (ColScaleAddFn.apply(a, b, gm) * w).sum().backward()
    a2, b2, g2 = (t.detach().clone().requires_grad_(True) for t in (a, b, gm))
    ((a2 + g2 * b2) * w).sum().backward()
    for x, y in ((a, a2), (b, b2), (gm, g2)):
        assert (x.grad - y.grad).abs().max().item() <= 1e-5


@pytest.mark.parametrize("B,H,Tq,Tk,dh", [(3, 4, 20, 1601, 64), (2, 2, 7, 300, 16), (1, 4, 32, 257, 32), (16, 4, 20, 1601, 64)])
def test_few_query_cross_attention_matches_torch(B, H, Tq, Tk, dh):
    """csrc/attention_fq.hip through the autograd node the context decoder uses: output and all three gradients against float64 torch; repeated calls
    bit-identical (every reduction has a fixed order); the dispatch takes the few-query kernels for these shapes and leaves dK / dV out when unasked."""
    from tunevlseg_amd import cris_ops as C
    from tunevlseg_amd import hip

    assert hip.fq_attention_ok(Tq, Tk, dh, False, None) and not hip.fq_attention_ok(Tq, Tk, dh, False, torch.ones(1))
    D = H * dh
    g = torch.Generator().manual_seed(B * 1000 + Tk)
    q, k, v = torch.randn(B * Tq, D, generator=g), torch.randn(B * Tk, D, generator=g), torch.randn(B * Tk, D, generator=g)
    w = torch.randn(B * Tq, D, generator=g)
    qd, kd, vd = (t.double().requires_grad_(True) for t in (q, k, v))
    qh, kh, vh = (t.view(B, -1, H, dh).transpose(1, 2) for t in (qd, kd, vd))
    ref = (torch.softmax(qh @ kh.transpose(-1, -2) * dh**-0.5, -1) @ vh).transpose(1, 2).reshape(B * Tq, D)
    (ref * w.double()).sum().backward()
    outs = []
    for _ in range(2):
        qg, kg, vg = (t.cuda().requires_grad_(True) for t in (q, k, v))
        o = C.CrossAttnFn.apply(qg, kg, vg, None, B, Tq, Tk, H, dh)
        (o * w.cuda()).sum().backward()
        outs.append((o.detach().clone(), qg.grad.clone(), kg.grad.clone(), vg.grad.clone()))
    assert all(torch.equal(a, b) for a, b in zip(*outs))
    o, dq, dk, dv = outs[0]
    for got, want in ((o, ref.detach()), (dq, qd.grad), (dk, kd.grad), (dv, vd.grad)):
        assert (got.cpu().double() - want).abs().max().item() <= 2e-5 * max(1.0, want.abs().max().item())
    # frozen keys / values (the prompt-tuning path): only dQ is computed
    qg = q.cuda().requires_grad_(True)
    o2 = C.CrossAttnFn.apply(qg, k.cuda(), v.cuda(), None, B, Tq, Tk, H, dh)
    (o2 * w.cuda()).sum().backward()
    assert torch.equal(o2.detach(), o) and torch.equal(qg.grad, dq)


def test_score_map_gradient_in_one_launch_equals_the_per_sample_gemms(monkeypatch):
    """ScoreMapFn: the one-GEMM + diagonal-gather forward and the few-query-kernel backward against the per-sample GEMMs they replace (B = 1 takes
    those; TVL_FQ_ATTN=0 forces them) and against float64 torch."""
    from tunevlseg_amd import hip
    from tunevlseg_amd.nets.denseclip import ScoreMapFn

    B, HW, K, C_ = 5, 36, 7, 128
    T = 1 + HW
    g = torch.Generator().manual_seed(3)
    v = torch.nn.functional.normalize(torch.randn(B * T, C_, generator=g), dim=1)
    t = torch.nn.functional.normalize(torch.randn(B, K, C_, generator=g), dim=2)
    w = torch.randn(B * HW, K, generator=g)
    td = t.double().requires_grad_(True)
    ref = torch.einsum("bic,bkc->bik", v.double().view(B, T, C_)[:, 1:], td).reshape(B * HW, K)
    (ref * w.double()).sum().backward()
    res = []
    for fq in (True, False):
        monkeypatch.setattr(hip, "FQ_ATTN", fq)
        tg = t.cuda().requires_grad_(True)
        s = ScoreMapFn.apply(v.cuda(), tg, B, HW, 1)
        (s * w.cuda()).sum().backward()
        res.append((s.detach().cpu(), tg.grad.cpu()))
        assert (res[-1][0].double() - ref.detach()).abs().max().item() <= 1e-5
        assert (res[-1][1].double() - td.grad).abs().max().item() <= 1e-4 * td.grad.abs().max().item()
    assert (res[0][1] - res[1][1]).abs().max().item() <= 1e-5 * res[1][1].abs().max().item()
