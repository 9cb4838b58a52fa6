"""Kernel-level parity: every C-ABI entry point against a float64 PyTorch restatement of the same op.

Runs on the GPU box only (-m gpu).  Tolerances are fp32 round-off of the exact-fp32 MFMA path:
rtol 2e-5 on GEMM/attention outputs relative to the operand scale, bit-exact for integer counts.
"""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from tunevlseg_amd import hip as H

    H.load()
    return H


def dev(t):
    return t.to("cuda").contiguous()


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def close(a, b, tol, what=""):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = (a - b).abs().max().item()
    scale = b.abs().max().item() + 1e-30
    assert err <= tol * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e} (tol {tol})"


def qgelu(x):
    return x * torch.sigmoid(1.702 * x)


# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K", [(300, 200, 100), (128, 128, 32), (1000, 768, 768), (4100, 2048, 96), (77, 64, 64), (65, 70, 36),
                                   (50, 25, 64)])
@pytest.mark.parametrize("layout", [0, 1, 2])
def test_gemm_layouts(hip, M, N, K, layout):
    A = rnd(M, K, seed=1)
    B = rnd(N, K, seed=2) if layout == 0 else rnd(K, N, seed=2)
    ref = A.double() @ (B.double().T if layout == 0 else B.double())
    Ad = dev(A.T) if layout == 2 else dev(A)  # TN stores A as [K, M]
    Bd = dev(B)
    Cd = torch.empty(M, N, device="cuda")
    hip.gemm(layout, M, N, K, Ad, Ad.shape[1], Bd, Bd.shape[1], Cd, N)
    close(Cd, ref, 3e-6 * math.sqrt(K), f"gemm layout {layout}")


@pytest.mark.parametrize("nsplit,tol", [(3, 2e-6), (2, 3e-5), (1, 2e-2)])
@pytest.mark.parametrize("M,N,K", [(1000, 768, 768), (300, 200, 100), (4100, 2048, 96), (130, 67, 70)])
def test_gemm_split_bf16(hip, nsplit, tol, M, N, K):
    """fp32 GEMM on the bf16 matrix cores; nsplit=3 must be fp32-equivalent (error relative to sum |a||b|)."""
    import ctypes as C

    A, W = rnd(M, K, seed=1), rnd(N, K, seed=2)
    bias, res = rnd(N, seed=3), rnd(M, N, seed=4)
    Cd = torch.empty(M, N, device="cuda")
    args = hip.GemmArgs(hip.NT, M, N, K, hip._p(dev(A)), K, hip._p(dev(W)), K, hip._p(Cd), N, None, None, 0, 0, None, None, 0, 0, 1.0,
                        hip._ident(), hip._ident())
    Ad, Wd, bd, rd = dev(A), dev(W), dev(bias), dev(res)
    args = hip.GemmArgs(hip.NT, M, N, K, hip._p(Ad), K, hip._p(Wd), K, hip._p(Cd), N, hip._p(bd), hip._p(rd), N, hip.ACT_RELU, None,
                        None, 0, 0, 1.0, hip._ident(), hip._ident())
    hip._call("tvl_gemm_bf16s", C.byref(args), nsplit)
    ref = torch.relu(A.double() @ W.double().T + bias.double()) + res.double()
    mag = (A.double().abs() @ W.double().abs().T).max().item()
    err = (Cd.cpu().double() - ref).abs().max().item()
    assert err <= tol * mag, f"nsplit {nsplit}: err {err:.3e} vs sum|a||b| {mag:.3e}"


def test_gemm_unaligned_k(hip):
    M, N, K = 130, 67, 70  # lda % 4 != 0 -> scalar-guarded loads
    A, W = rnd(M, K, seed=3), rnd(N, K, seed=4)
    Cd = torch.empty(M, N, device="cuda")
    hip.gemm(hip.NT, M, N, K, dev(A), K, dev(W), K, Cd, N)
    close(Cd, A.double() @ W.double().T, 2e-5, "gemm unaligned")


@pytest.mark.parametrize("act", ["quick_gelu", "relu", None])
def test_gemm_epilogue_forward(hip, act):
    M, N, K = 515, 192, 64
    x, W, b, res = rnd(M, K, seed=5), rnd(N, K, seed=6, scale=0.2), rnd(N, seed=7), rnd(M, N, seed=8)
    pre_ref = x.double() @ W.double().T + b.double()
    a_ref = {"quick_gelu": qgelu, "relu": F.relu, None: lambda v: v}[act](pre_ref)
    y, pre = hip.linear_fwd(dev(x), dev(W), dev(b), act=hip.ACT_IDS[act], residual=dev(res), want_pre=True)
    close(pre, pre_ref, 1e-5, "pre")
    close(y, a_ref + res.double(), 1e-5, "y")


@pytest.mark.parametrize("act", ["quick_gelu", "relu"])
def test_gemm_dgrad_with_dact(hip, act):
    M, N, K = 300, 96, 160  # dy [M,N], W [N,K]
    dy, W, z, res = rnd(M, N, seed=9), rnd(N, K, seed=10, scale=0.3), rnd(M, K, seed=11), rnd(M, K, seed=12)
    zz = z.double().requires_grad_(True)
    f = (qgelu(zz) if act == "quick_gelu" else F.relu(zz)).sum()
    (gz,) = torch.autograd.grad(f, zz)
    ref = (dy.double() @ W.double()) * gz + res.double()
    dx = hip.linear_dgrad(dev(dy), dev(W), dact=hip.ACT_IDS[act], dact_aux=dev(z), residual=dev(res))
    close(dx, ref, 1e-5, "dgrad")


def test_gemm_wgrad_and_rowmaps(hip):
    B_, T, P, Cc, N = 3, 12, 9, 16, 25
    tok = rnd(B_, T, Cc, seed=13)
    w = rnd(Cc, N, seed=14)
    strip = tok[:, 1:1 + P].reshape(B_ * P, Cc)
    # forward with a_map: rows 1..P of every T-row block
    out = torch.empty(B_ * P, N, device="cuda")
    amap = hip.RowMap(P, T, 1)
    hip.gemm(hip.NN, B_ * P, N, Cc, dev(tok.reshape(B_ * T, Cc)), Cc, dev(w), N, out, N, a_map=amap)
    close(out, strip.double() @ w.double(), 1e-5, "a_map")
    # wgrad with the K rows remapped
    dy = rnd(B_ * P, N, seed=15)
    dW = torch.empty(Cc, N, device="cuda")
    hip.gemm(hip.TN, Cc, N, B_ * P, dev(tok.reshape(B_ * T, Cc)), Cc, dev(dy), N, dW, N, a_map=amap)
    close(dW, strip.double().T @ dy.double(), 1e-5, "tn a_map")
    # dgrad scattered through c_map into a zeroed token buffer
    dtok = torch.zeros(B_ * T, Cc, device="cuda")
    hip.gemm(hip.NT, B_ * P, Cc, N, dev(dy), N, dev(w), N, dtok, Cc, c_map=amap)
    ref = torch.zeros(B_, T, Cc, dtype=torch.double)
    ref[:, 1:1 + P] = (dy.double() @ w.double().T).reshape(B_, P, Cc)
    close(dtok.reshape(B_, T, Cc), ref, 1e-5, "c_map")


# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("rows,cols", [(1001, 768), (37, 64), (5, 32), (9, 10), (130, 512)])
def test_layernorm_fwd_bwd(hip, rows, cols):
    x, g, b, dy, dres = rnd(rows, cols, seed=1) * 2 + 0.5, 1 + 0.1 * rnd(cols, seed=2), rnd(cols, seed=3), rnd(rows, cols, seed=4), rnd(rows, cols, seed=5)
    xd = x.double().requires_grad_(True)
    gd, bd = g.double().requires_grad_(True), b.double().requires_grad_(True)
    yref = F.layer_norm(xd, (cols,), gd, bd, 1e-5)
    yref.backward(dy.double())
    y, mean, rstd = hip.layernorm_fwd(dev(x), dev(g), dev(b), 1e-5)
    close(y, yref, 2e-6, "ln fwd")
    dgam = torch.zeros(cols, device="cuda")
    dbet = torch.zeros(cols, device="cuda")
    dx = hip.layernorm_bwd(dev(dy), dev(x), dev(g), mean, rstd, dres=dev(dres), dgamma=dgam, dbeta=dbet)
    close(dx, xd.grad + dres.double(), 5e-6, "ln dx")
    close(dgam, gd.grad, 2e-5, "ln dgamma")
    close(dbet, bd.grad, 2e-5, "ln dbeta")


# ----------------------------------------------------------------------------------------------
def attn_ref(qkv, B, T, H, dh, causal, key_mask):
    D = H * dh
    q, k, v = (qkv[..., i * D:(i + 1) * D].reshape(B, T, H, dh).transpose(1, 2) for i in range(3))
    w = (q @ k.transpose(-1, -2)) * dh ** -0.5
    allowed = torch.ones(T, T, dtype=torch.bool)
    if causal:
        allowed = allowed.tril()
    allowed = allowed.expand(B, 1, T, T)
    if key_mask is not None:
        allowed = allowed & key_mask.bool()[:, None, None, :]
    w = w.masked_fill(~allowed, float("-inf"))
    p = torch.softmax(w, -1)
    return (p @ v).transpose(1, 2).reshape(B, T, D), torch.logsumexp(w, -1)


@pytest.mark.parametrize("B,T,H,dh,causal,masked", [(2, 495, 2, 64, False, False), (3, 77, 2, 64, True, True), (2, 50, 3, 16, False, False),
                                                    (2, 17, 2, 8, True, True), (1, 130, 1, 32, True, False), (2, 21, 4, 16, False, True)])
def test_attention_fwd_bwd(hip, B, T, H, dh, causal, masked):
    D = H * dh
    qkv = rnd(B, T, 3 * D, seed=21)
    d_o = rnd(B, T, D, seed=22)
    km = None
    if masked:
        km = torch.ones(B, T, dtype=torch.int32)
        for b in range(B):
            km[b, T - 1 - 2 * b - 1:] = 0  # trailing padding; key 0 always visible
    qd = qkv.double().requires_grad_(True)
    oref, lseref = attn_ref(qd, B, T, H, dh, causal, km)
    oref.backward(d_o.double())
    qkv_d = dev(qkv.reshape(B * T, 3 * D))
    kmd = dev(km) if km is not None else None
    o, lse = hip.attn_fwd_packed(qkv_d, B, T, H, dh, dh ** -0.5, causal=causal, key_mask=kmd)
    close(o.reshape(B, T, D), oref, 1e-5, "attn o")
    close(lse, lseref, 1e-5, "attn lse")
    dqkv = hip.attn_bwd_packed(qkv_d, o, dev(d_o.reshape(B * T, D)), lse, B, T, H, dh, dh ** -0.5, causal=causal, key_mask=kmd)
    gref = qd.grad.reshape(B * T, 3 * D)
    close(dqkv[:, :D], gref[:, :D], 2e-5, "dq")
    close(dqkv[:, D:2 * D], gref[:, D:2 * D], 2e-5, "dk")
    close(dqkv[:, 2 * D:], gref[:, 2 * D:], 2e-5, "dv")


# ----------------------------------------------------------------------------------------------
def test_im2col_matches_conv(hip):
    B, Cc, H, ps, Dm = 2, 3, 64, 16, 20
    img, w = rnd(B, Cc, H, H, seed=31), rnd(Dm, Cc, ps, ps, seed=32, scale=0.05)
    cols = hip.im2col_patch(dev(img), ps)
    ref = F.conv2d(img.double(), w.double(), stride=ps).flatten(2).transpose(1, 2).reshape(-1, Dm)
    out = hip.linear_fwd(cols, dev(w.reshape(Dm, -1)))
    close(out, ref, 1e-5, "patch embed")


def test_vision_and_text_assemble(hip):
    B, P, n, D = 3, 16, 4, 32
    patch, cls, pos, ctx = rnd(B * P, D, seed=33), rnd(D, seed=34), rnd(1 + P, D, seed=35), rnd(n, D, seed=36)
    x0 = hip.vision_assemble(dev(patch), dev(cls), dev(pos), dev(ctx), 0, B, P, n, D)
    ref = torch.cat((torch.cat((cls.expand(B, 1, D), patch.reshape(B, P, D)), 1) + pos, ctx.expand(B, n, D)), 1)
    close(x0, ref, 1e-7, "vision assemble")
    # text: [BOS, ctx(2), tok1, tok2, last]
    L, V, Dt, T = 5, 50, 16, 6
    ids = torch.randint(0, V, (B, L), generator=torch.Generator().manual_seed(1))
    table, tpos, tctx = rnd(V, Dt, seed=37), rnd(T, Dt, seed=38), rnd(B, 2, Dt, seed=39)
    tmap = torch.tensor([0, -1, -2, 1, 2, L - 1], dtype=torch.int32)
    out = hip.text_assemble(dev(ids), dev(tmap), dev(table), dev(tctx), 2 * Dt, dev(tpos), B, T, Dt)
    emb = table[ids]
    ref = torch.cat((emb[:, :1], tctx, emb[:, 1:3], emb[:, -1:]), 1) + tpos
    close(out, ref, 1e-7, "text assemble")


def test_rows_overwrite_grad_gather_scatter(hip):
    B, T, D, n = 3, 10, 8, 3
    x, src = rnd(B, T, D, seed=40), rnd(n, D, seed=41)
    xd = dev(x)
    hip.rows_overwrite(xd, dev(src), 0, T - n, n)
    ref = x.clone()
    ref[:, -n:] = src
    close(xd, ref, 0, "overwrite bcast")
    src_b = rnd(B, n, D, seed=42)
    hip.rows_overwrite(xd, dev(src_b), n * D, 1, n)
    ref[:, 1:1 + n] = src_b
    close(xd, ref, 0, "overwrite per-sample")
    g = dev(rnd(B, T, D, seed=43))
    g0 = g.clone().cpu()
    dst = torch.empty(n, D, device="cuda")
    hip.rows_grad(g, dst, T - n, n, True, True)
    close(dst, g0[:, -n:].sum(0), 1e-6, "rows_grad reduce")
    assert g[:, -n:].abs().max().item() == 0
    dst_b = torch.ones(B, n, D, device="cuda")
    hip.rows_grad(g, dst_b, 1, n, False, False, accumulate=True)
    close(dst_b, 1 + g0[:, 1:1 + n], 1e-6, "rows_grad per-sample accumulate")
    idx = torch.tensor([2, 0, 9], dtype=torch.int32)
    out = hip.gather_rows(dev(x), dev(idx))
    close(out, x[torch.arange(B), idx.long()], 0, "gather")
    dx = torch.zeros(B, T, D, device="cuda")
    hip.scatter_rows_add(out, dev(idx), dx)
    ref = torch.zeros(B, T, D)
    ref[torch.arange(B), idx.long()] = x[torch.arange(B), idx.long()]
    close(dx, ref, 0, "scatter")


def test_film(hip):
    B, T, Cc = 3, 21, 16
    x, mul, add, dy = rnd(B, T, Cc, seed=44), rnd(B, Cc, seed=45), rnd(B, Cc, seed=46), rnd(B, T, Cc, seed=47)
    y = hip.film_fwd(dev(x), dev(mul), dev(add))
    close(y, mul[:, None] * x + add[:, None], 1e-6, "film fwd")
    dx, dmul, dadd = hip.film_bwd(dev(dy), dev(x), dev(mul), True)
    close(dx, mul[:, None] * dy, 1e-6, "film dx")
    close(dmul, (dy * x).sum(1), 1e-5, "film dmul")
    close(dadd, dy.sum(1), 1e-5, "film dadd")


def test_pixel_shuffle_is_conv_transpose(hip):
    B, Cc, G, ps = 2, 8, 4, 16
    feat, w, b = rnd(B, Cc, G, G, seed=48), rnd(Cc, 1, ps, ps, seed=49), rnd(1, seed=50)
    ref = F.conv_transpose2d(feat.double(), w.double(), b.double(), stride=ps)[:, 0]
    tok = feat.permute(0, 2, 3, 1).reshape(B * G * G, Cc)
    cols = torch.empty(B * G * G, ps * ps, device="cuda")
    hip.gemm(hip.NN, B * G * G, ps * ps, Cc, dev(tok), Cc, dev(w.reshape(Cc, ps * ps)), ps * ps, cols, ps * ps)
    logits = hip.pixel_shuffle_fwd(cols, dev(b), None, 1.0, 0.0, B, G, ps)
    close(logits, ref, 1e-5, "tconv")
    dl = rnd(B, G * ps, G * ps, seed=51)
    dcols = hip.pixel_unshuffle_bwd(dev(dl), 0.5, B, G, ps)
    ref_d = 0.5 * dl.reshape(B, G, ps, G, ps).permute(0, 1, 3, 2, 4).reshape(B * G * G, ps * ps)
    close(dcols, ref_d, 0, "unshuffle")


@pytest.mark.parametrize("k,G", [(5, 4), (3, 2), (5, 22)])
def test_upconv_taps(hip, k, G):
    B, Cc, ps = 2, 6, 16
    feat = rnd(B, Cc, G, G, seed=52).double().requires_grad_(True)
    w = rnd(1, Cc, k, k, seed=53, scale=0.3).double().requires_grad_(True)
    b = rnd(1, seed=54).double().requires_grad_(True)
    up = F.interpolate(feat, scale_factor=float(ps), mode="bilinear")
    up = F.pad(up, (k // 2,) * 4, mode="replicate")
    ref = F.conv2d(up, w, b)[:, 0]
    dout = rnd(B, G * ps, G * ps, seed=55)
    ref.backward(dout.double())
    tok = feat.detach().float().permute(0, 2, 3, 1).reshape(B * G * G, Cc)
    w2 = w.detach().float().reshape(Cc, k * k)
    taps = torch.empty(B * G * G, k * k, device="cuda")
    hip.gemm(hip.NN, B * G * G, k * k, Cc, dev(tok), Cc, dev(w2), k * k, taps, k * k)
    out = hip.upconv_taps_fwd(taps, dev(b.detach().float()), B, G, ps, k)
    close(out, ref, 1e-5, "upconv fwd")
    dtaps = hip.upconv_taps_bwd(dev(dout), B, G, ps, k)
    dtok = hip.linear_fwd(dtaps, dev(w2))  # [M, k*k] x [C, k*k]^T
    close(dtok.reshape(B, G, G, Cc).permute(0, 3, 1, 2), feat.grad, 2e-5, "upconv dfeat")
    dW = hip.linear_wgrad(dev(tok), dtaps)  # [C, k*k] = tok^T dtaps
    close(dW.reshape(1, Cc, k, k), w.grad, 2e-5, "upconv dw")
    close(hip.dot(dev(dout)), b.grad, 1e-5, "upconv dbias")


def test_dicece_stats_and_grad(hip):
    from oracle import clipseg_oracle as O

    B, S = 3, 48
    logits = rnd(B, 1, S, S, seed=56) * 2
    target = (torch.rand(B, 1, S, S, generator=torch.Generator().manual_seed(57)) > 0.7).float()
    ld = logits.double().requires_grad_(True)
    loss_ref = O.dice_ce_loss(ld, target.double())
    loss_ref.backward()
    fsum, isum, label = hip.dicece_stats(dev(logits), dev(target), 0.5, want_label=True)
    f = fsum.cpu()
    dice = (1 - (2 * f[:, 0] + 1e-5) / (f[:, 1] + f[:, 2] + 1e-5)).mean()
    loss = dice + 0.2 * f[:, 3].sum() / (B * S * S)
    assert abs(loss.item() - loss_ref.item()) < 1e-6
    tp, fp, fn, tn = O.confusion_counts(torch.sigmoid(logits), target.long())
    assert torch.equal(isum.cpu(), torch.stack((tp, fp, fn, tn), 1))
    assert torch.equal(label.cpu().bool(), torch.sigmoid(logits) > 0.5)
    loss_dev = hip.dicece_loss(fsum, S * S, 1.0, 0.2)   # the same arithmetic in one launch, fp32 scalar on the device
    assert loss_dev.dtype == torch.float32 and loss_dev.item() == loss.to(torch.float32).item()
    gs = torch.tensor([0.7], device="cuda")
    dl = hip.dicece_bwd(dev(logits), dev(target), fsum, 1.0, 0.2, 1e-5, 1e-5, gs)
    close(dl, 0.7 * ld.grad, 2e-5, "dicece grad")


def test_adamw_matches_torch(hip):
    n = 1000
    p0, g1, g2 = rnd(n, seed=58), rnd(n, seed=59), rnd(n, seed=60)
    pt = torch.nn.Parameter(p0.clone().double())
    opt = torch.optim.AdamW([pt], lr=2e-4, weight_decay=0.01)
    p, m, v = dev(p0), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    for t, g in enumerate((g1, g2), 1):
        pt.grad = g.double()
        opt.step()
        hip.adamw(p, dev(g), m, v, 2e-4, 0.9, 0.999, 1e-8, 0.01, t)
    close(p, pt.detach(), 1e-6, "adamw")


def test_small_reductions(hip):
    x = rnd(37, 20, seed=61)
    y, inv = hip.l2norm_fwd(dev(x))
    xd = x.double().requires_grad_(True)
    yr = xd / xd.norm(dim=-1, keepdim=True)
    dy = rnd(37, 20, seed=62)
    yr.backward(dy.double())
    close(y, yr, 1e-6, "l2norm")
    close(hip.l2norm_bwd(dev(dy), y, inv), xd.grad, 1e-5, "l2norm bwd")
    close(hip.colsum(dev(x)), x.double().sum(0), 1e-5, "colsum")
    big = rnd(5000, 3, seed=63)
    close(hip.colsum(dev(big)), big.double().sum(0), 1e-5, "colsum tall")
    close(hip.dot(dev(x), dev(dy)), (x.double() * dy.double()).sum().reshape(1), 1e-5, "dot")
    out = hip.bias_act(dev(x), dev(rnd(20, seed=64)), hip.ACT_QUICK_GELU)
    close(out, qgelu(x.double() + rnd(20, seed=64).double()), 1e-6, "bias_act")
    t = torch.empty(100, device="cuda")
    hip.fill(t, 3.0)
    assert (t == 3.0).all()
    hip.axpby(dev(torch.ones(100)), 2.0, t, 0.5)
    assert torch.allclose(t.cpu(), torch.full((100,), 3.5))


@pytest.mark.parametrize("rows,cols", [(32, 16), (100, 64), (15840, 768), (77, 512)])
def test_tp3_pack_is_exact(hip, rows, cols):
    """x == p0 + p1 + p2 in fp32 for every element (three bf16 pieces carry all 24 significand bits); padded rows are zero."""
    x = dev(rnd(rows, cols, seed=7) * torch.logspace(-3, 3, cols))
    img = hip.tp3_pack(x)
    assert img.buf.numel() == hip.load().tvl_tp3_bytes(rows, cols)
    assert torch.equal(img.float(), x)
    wide = dev(rnd(rows, cols + 24, seed=8))  # a column slice of a wider matrix (row stride != cols)
    assert torch.equal(hip.tp3_pack(wide[:, 8:8 + cols]).float(), wide[:, 8:8 + cols])


@pytest.mark.parametrize("tile", [0, 128, 192, 256])
@pytest.mark.parametrize("M,N,K", [(15840, 768, 3072), (1000, 768, 768), (15840, 2304, 768), (333, 80, 64), (4100, 2048, 96), (192, 256, 64)])
def test_gemm_tp3(hip, M, N, K, tile):
    """tvl_gemm_tp3 (LDS-DMA ring over pre-tiled bf16 pieces) against the fp64 product, full epilogue, ragged M / N, both
    outputs (fp32 and the tp3 image the next GEMM consumes), every tile height and both schedule variants."""
    A, B, bias, res = rnd(M, K, seed=1), rnd(N, K, seed=2), rnd(N, seed=3), rnd(M, N, seed=4)
    At, Bt = hip.tp3_pack(dev(A)), hip.tp3_pack(dev(B))
    pre_ref = A.double() @ B.double().T + bias.double()
    ref = qgelu(pre_ref) + res.double()
    old = hip.GEMM_TP3_TILE, hip.GEMM_TP3_VARIANT
    try:
        hip.GEMM_TP3_TILE = tile
        for variant in ((0, 1, 2) if tile in (0, 192) else (0,)):
            hip.GEMM_TP3_VARIANT = variant
            pre = torch.empty(M, N, device="cuda")
            Cf, Ct = hip.gemm_tp3(At, Bt, want_tp3=True, bias=dev(bias), residual=dev(res), act=hip.ACT_QUICK_GELU, pre_out=pre)
            close(Cf, ref, 3e-6 * math.sqrt(K), f"gemm_tp3 tile {tile} variant {variant}")
            close(pre, pre_ref, 3e-6 * math.sqrt(K), "gemm_tp3 pre_out")
            assert torch.equal(Ct.float(), Cf)  # the tp3 output is the exact split of the fp32 output
        # tp3-only output feeding a second GEMM; act' epilogue (the dz of the backward)
        z = dev(rnd(M, N, seed=5))
        _, dz = hip.gemm_tp3(At, Bt, want_f32=False, want_tp3=True, dact=hip.ACT_QUICK_GELU, dact_aux=z)
        s = torch.sigmoid(1.702 * z.double().cpu())
        dref = (A.double() @ B.double().T) * (s + 1.702 * z.double().cpu() * s * (1 - s))
        close(dz.float(), dref, 3e-6 * math.sqrt(K), "gemm_tp3 dact -> tp3")
        if N % 32 == 0 and N >= 64:
            W2 = rnd(48, N, seed=6)
            C2, _ = hip.gemm_tp3(dz, hip.tp3_pack(dev(W2)))
            close(C2, dref @ W2.double().T, 3e-6 * math.sqrt(K) * math.sqrt(N), "gemm_tp3 chained")
    finally:
        hip.GEMM_TP3_TILE, hip.GEMM_TP3_VARIANT = old


@pytest.mark.parametrize("rows,cols", [(15840, 768), (100, 64), (33, 1024), (489, 768)])
def test_layernorm_tp3(hip, rows, cols):
    """LayerNorm that writes its result only as a tp3 image / its backward that writes fp32 + tp3: identical to the fp32 kernels
    (the tp3 image is an exact split of the same fp32 values)."""
    x, g, b = dev(rnd(rows, cols, seed=1) * 3 + 0.5), dev(1 + 0.1 * rnd(cols, seed=2)), dev(0.1 * rnd(cols, seed=3))
    y, mean, rstd = hip.layernorm_fwd(x, g, b, 1e-5)
    yt, mean_t, rstd_t = hip.layernorm_fwd_tp3(x, g, b, 1e-5)
    assert torch.equal(yt.float(), y) and torch.equal(mean, mean_t) and torch.equal(rstd, rstd_t)
    dy, dres = dev(rnd(rows, cols, seed=4)), dev(rnd(rows, cols, seed=5))
    dx = hip.layernorm_bwd(dy, x, g, mean, rstd, dres=dres)
    dx2, dxt = hip.layernorm_bwd_tp3(dy, x, g, mean, rstd, dres=dres)
    assert torch.equal(dx2, dx) and torch.equal(dxt.float(), dx)


@pytest.mark.parametrize("B,T,H", [(2, 495, 12), (3, 100, 2), (1, 33, 1)])
def test_attention_tp3_outputs(hip, B, T, H):
    """d_h = 64 attention writing O / dQ|dK|dV as tp3 images: the same values as the fp32-output kernels (exact split)."""
    dh, D = 64, H * 64
    qkv = dev(rnd(B * T, 3 * D, seed=1))
    o, lse = hip.attn_fwd_packed(qkv, B, T, H, dh, dh**-0.5)
    ot, lse_t = hip.attn_fwd_packed_tp3(qkv, B, T, H, dh, dh**-0.5)
    assert torch.equal(ot.float(), o) and torch.equal(lse, lse_t)
    d_o = dev(rnd(B * T, D, seed=2))
    dqkv = hip.attn_bwd_packed(qkv, o, d_o, lse, B, T, H, dh, dh**-0.5)
    dqkv_t = hip.attn_bwd_packed_tp3(qkv, ot, d_o, lse, B, T, H, dh, dh**-0.5)
    close(dqkv_t.float(), dqkv, 1e-6, "attn bwd tp3")  # delta is summed in a different order


@pytest.mark.parametrize("M,N,K", [(64, 25, 21632), (64, 64, 5000), (130, 25, 15488), (32, 8, 100000)])
def test_gemm_tn_split_k(hip, M, N, K):
    """Weight gradients with a handful of output tiles run split-K (fp32 atomics); a row map on the K rows still applies."""
    A, B = rnd(K, M, seed=5), rnd(K, N, seed=6)
    ref = A.double().T @ B.double()
    Cd = torch.full((M, N), 7.0, device="cuda")  # must be overwritten, not accumulated into
    hip.gemm(hip.TN, M, N, K, dev(A), M, dev(B), N, Cd, N)
    close(Cd, ref, 3e-6 * math.sqrt(K), "gemm TN split-K")


@pytest.mark.parametrize("M,N,K", [(15840, 768, 3072), (15841, 3072, 100), (15840, 2304, 776), (40001, 256, 2304), (15840, 3072, 96)])
def test_gemm_wide_tiles_with_epilogue(hip, M, N, K):
    """Shapes that pick the 192x256 (8-wave) and 192x128 tiles, ragged in M and K, full epilogue incl. the post-residual order."""
    from tunevlseg_amd.hip import _bf16s_tile

    assert _bf16s_tile(M, N, K)[0] == 192
    A, B, bias, res = rnd(M, K, seed=1), rnd(N, K, seed=2), rnd(N, seed=3), rnd(M, N, seed=4)
    pre_ref = A.double() @ B.double().T + bias.double()
    for post in (False, True):
        ref = torch.relu(pre_ref + res.double()) if post else torch.relu(pre_ref) + res.double()
        Cd, pre = torch.empty(M, N, device="cuda"), torch.empty(M, N, device="cuda")
        hip.gemm(hip.NT, M, N, K, dev(A), K, dev(B), K, Cd, N, bias=dev(bias), act=hip.ACT_RELU | (hip.ACT_POST_RESIDUAL if post else 0),
                 residual=dev(res), ldr=N, pre_out=pre)
        close(Cd, ref, 3e-6 * math.sqrt(K), f"wide tile post={post}")
        close(pre, pre_ref, 3e-6 * math.sqrt(K), "wide tile pre_out")


@pytest.mark.parametrize("M,N,K", [(256, 512, 2048), (384, 512, 1536), (300, 130, 1100), (256, 2048, 1024)])
def test_gemm_split_k_skinny(hip, M, N, K):
    """Skinny, deep NT GEMMs take the deterministic split-K path (workspace partials + ordered reduction + epilogue)."""
    A, B, bias, res = rnd(M, K, seed=1), rnd(N, K, seed=2), rnd(N, seed=3), rnd(M, N, seed=4)
    ref_pre = A.double() @ B.double().T + bias.double()
    ref = qgelu(ref_pre) + res.double()
    outs = []
    for _ in range(2):
        Cd, pre = torch.empty(M, N, device="cuda"), torch.empty(M, N, device="cuda")
        hip.gemm_profile_start()
        hip.gemm(hip.NT, M, N, K, dev(A), K, dev(B), K, Cd, N, bias=dev(bias), act=hip.ACT_QUICK_GELU, residual=dev(res), ldr=N, pre_out=pre)
        prof = hip.gemm_profile_stop()
        assert any("splitk" in k for k in prof), prof.keys()
        close(Cd, ref, 3e-6 * math.sqrt(K), "split-K")
        close(pre, ref_pre, 3e-6 * math.sqrt(K), "split-K pre_out")
        outs.append(Cd.clone())
    assert torch.equal(outs[0], outs[1])  # fixed summation order: bitwise reproducible


def test_dicece_kernel_matches_hand_derived_vectors(hip):
    """Rows L1 / L2 on the HIP kernel: tests/golden/loss_metric_kav.json (derived by hand from the published definitions)."""
    import json
    from pathlib import Path

    from tunevlseg_amd import ops

    kav = json.loads((Path(__file__).resolve().parent / "golden" / "loss_metric_kav.json").read_text())
    for case in kav["cases"]:
        if not case["asserted"]:
            continue
        x = dev(torch.tensor(case["logits"], dtype=torch.float32)[:, None, :, None]).requires_grad_(True)
        t = dev(torch.tensor(case["mask"], dtype=torch.float32)[:, None, :, None])
        e = case["expect"]
        loss, isum = ops.DiceCELossFn.apply(x, t, kav["lambda_dice"], kav["lambda_ce"], kav["threshold"])
        assert abs(loss.item() - e["loss"]) < 2e-6, case["name"]
        assert isum.cpu().tolist() == e["counts_tp_fp_fn_tn"], case["name"]


def test_dicece_loss_is_bitwise_reproducible(hip):
    """The float sums are reduced in a fixed order (two stages, no float atomics): identical bits run after run."""
    from tunevlseg_amd import ops

    x, t = dev(rnd(8, 1, 352, 352, seed=1)), dev((rnd(8, 1, 352, 352, seed=2) > 0.5).float())
    first = ops.DiceCELossFn.apply(x, t, 1.0, 0.2, 0.5)[0]
    for _ in range(5):
        assert torch.equal(ops.DiceCELossFn.apply(x, t, 1.0, 0.2, 0.5)[0], first)


def test_input_side_normalize_and_mask(hip):
    """Row f2 on the device: uint8 HWC image -> normalised NCHW float, uint8 mask -> float / 255; the loss sees the float mask,
    the metrics its .long() (only grey level 255 counts as foreground)."""
    g = torch.Generator().manual_seed(0)
    img = torch.randint(0, 256, (3, 37, 41, 3), generator=g, dtype=torch.uint8)
    mean, std = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
    ref = ((img.double() / 255 - torch.tensor(mean, dtype=torch.float64)) / torch.tensor(std, dtype=torch.float64)).permute(0, 3, 1, 2)
    close(hip.normalize_u8(img.cuda(), mean, std), ref, 1e-6, "normalize_u8")
    m = torch.randint(0, 256, (3, 37, 41), generator=g, dtype=torch.uint8)
    m[0, :5] = 255
    out = hip.mask_u8(m.cuda())
    assert torch.equal(out.cpu(), (m.float() / 255)[:, None])
    assert torch.equal(out.long().cpu(), (m == 255).long()[:, None])


@pytest.mark.parametrize("B,T,H", [(2, 495, 12), (3, 100, 2), (1, 33, 1), (4, 64, 3), (2, 485, 12), (4, 8, 2), (5, 31, 1)])
def test_attention_on_tp3_qkv(hip, B, T, H):
    """Attention reading Q / K / V from the tp3 image of the packed QKV matrix (LDS-DMA key tiles aligned to the image's row
    blocks, neighbouring samples' keys masked) against the fp32-input kernel: same piece arithmetic, different key tiling."""
    dh, D = 64, H * 64
    qkv = dev(rnd(B * T, 3 * D, seed=1))
    o_ref, lse_ref = hip.attn_fwd_packed(qkv, B, T, H, dh, dh**-0.5)
    o, lse = hip.attn_tp3_fwd(hip.tp3_pack(qkv), B, T, H, dh**-0.5)
    close(o.float(), o_ref, 2e-6, "attn tp3 fwd O")
    close(lse, lse_ref, 1e-6, "attn tp3 fwd lse")


@pytest.mark.parametrize("B,T,H", [(2, 495, 12), (3, 100, 2), (1, 33, 1), (4, 64, 3), (2, 485, 12), (4, 8, 2), (5, 31, 1)])
def test_attention_backward_on_tp3_operands(hip, B, T, H):
    """dQ | dK | dV from tp3 images of QKV, O and dO (query / key tiles aligned to the images' row blocks, neighbouring samples'
    rows masked, delta formed from the pieces) against the fp32-input backward kernels: same piece arithmetic, different tiling."""
    dh, D = 64, H * 64
    qkv = dev(rnd(B * T, 3 * D, seed=2))
    d_o = dev(rnd(B * T, D, seed=3))
    o_t, lse = hip.attn_fwd_packed_tp3(qkv, B, T, H, dh, dh**-0.5)
    ref = hip.attn_bwd_packed_tp3(qkv, o_t, d_o, lse, B, T, H, dh, dh**-0.5).float()
    got = hip.attn_tp3_bwd(hip.tp3_pack(qkv), o_t, hip.tp3_pack(d_o), lse, B, T, H, dh**-0.5).float()
    for name, c0 in (("dQ", 0), ("dK", D), ("dV", 2 * D)):
        close(got[:, c0:c0 + D], ref[:, c0:c0 + D], 5e-6, f"attn tp3 bwd {name}")


def test_dropout_statistics_and_reproducibility(hip):
    """tvl_dropout (the SharedAttn learner's TransformerEncoderLayer, configs/model/shared_attn_clipseg.yaml:21 uses p = 0.25):
    the dropped fraction is binomial around p, survivors are scaled by 1 / (1 - p), the same (seed) reproduces the mask -- which is how
    the backward re-creates it -- and another seed gives an independent one."""
    n, p = 1 << 20, 0.25
    x = dev(torch.ones(n))
    a, b, c = hip.dropout(x, p, 1234), hip.dropout(x, p, 1234), hip.dropout(x, p, 1235)
    assert torch.equal(a, b)
    dropped = (a == 0).double().mean().item()
    sigma = (p * (1 - p) / n) ** 0.5
    assert abs(dropped - p) < 5 * sigma, (dropped, p, sigma)
    kept = a[a != 0]
    assert torch.allclose(kept, torch.full_like(kept, 1 / (1 - p)), rtol=1e-6, atol=0)
    # independence of two seeds: P(both dropped) = p^2 within 5 sigma
    both = ((a == 0) & (c == 0)).double().mean().item()
    assert abs(both - p * p) < 5 * (p * p * (1 - p * p) / n) ** 0.5, both
    # no visible structure along the index (runs test on 64-element blocks: block means spread like a binomial)
    blk = (a == 0).double().view(-1, 64).mean(1)
    assert abs(blk.var().item() - p * (1 - p) / 64) < 0.1 * p * (1 - p) / 64
    assert torch.equal(hip.dropout(x, 0.0, 7), x)


@pytest.mark.parametrize("rows,cols", [(100, 64), (32, 768), (495, 768)])
def test_h2_pack_round_trip_and_scales(hip, rows, cols):
    """Two fp16 pieces of the power-of-two scaled value: 22 significand bits per element relative to its ROW maximum (per-row mode)
    or TENSOR maximum, exact power-of-two scales that put the largest magnitude in [2^13, 2^14)."""
    x = rnd(rows, cols, seed=4) * torch.logspace(-6, 2, rows)[:, None]   # rows spanning eight decades, as gradients do
    for per_row in (True, False):
        t = hip.h2_pack(dev(x), per_row)
        inv = t.inv_scale.cpu()
        assert torch.equal(inv, torch.exp2(torch.round(torch.log2(inv))))   # exact powers of two
        top = (x.abs().amax(1) if per_row else x.abs().max().reshape(1)) / inv
        assert (top >= 2.0**13).all() and (top < 2.0**14).all()
        err = (t.float().cpu() - x).abs()
        bound = (x.abs().amax(1, keepdim=True) if per_row else x.abs().max()) * 2.0**-21
        assert (err <= bound).all(), (err / bound).max()


@pytest.mark.parametrize("rows,cols", [(70, 64), (2049, 512)])
def test_h2_pack_with_relu_gate(hip, rows, cols):
    """tvl_h2_pack_masked: x * (mask > 0) packed in one go; scales and row norms are those of the GATED matrix; a strided mask."""
    x = rnd(rows, cols, seed=8) * torch.logspace(-3, 1, rows)[:, None]
    wide = rnd(rows, cols + 16, seed=9)
    mask = dev(wide)[:, 8:8 + cols]   # row stride cols + 16
    gated = torch.where(wide[:, 8:8 + cols] > 0, x, torch.zeros_like(x))   # (x * False would leave -0.0 behind)
    for per_row in (True, False):
        t = hip.h2_pack(dev(x), per_row, want_norm=per_row, relu_mask=mask)
        ref = hip.h2_pack(dev(gated), per_row, want_norm=per_row)
        assert torch.equal(t.inv_scale, ref.inv_scale) and torch.equal(t.buf, ref.buf)   # bit-identical to packing the gated matrix
        if per_row:
            assert torch.allclose(t.row_norm, ref.row_norm, rtol=1e-6, atol=0)   # (the two instantiations contract their FMAs differently)


@pytest.mark.parametrize("tile", [0, 1926, 2566, 1920, 2560])
@pytest.mark.parametrize("M,N,K", [(1000, 768, 768), (495, 512, 3072), (33, 256, 64), (257, 512, 96), (700, 272, 128)])
def test_gemm_h2_matches_fp64(hip, M, N, K, tile):
    """tvl_gemm_h2: three fp16-piece products per k-step, row scales of A and the tensor scale of B undone in the epilogue.  Both kernel
    generations and both row tiles: 1926 / 2566 = the v_mfma_f32_16x16x32_f16 ring (csrc/gemm_h2m_kernel.h; K = 96 is its shortest k-loop,
    K = 64 falls back), 1920 / 2560 = the 32x32x16 one; 0 = the library's own choice."""
    _gemm = hip.gemm_h2
    hip_gemm_h2 = lambda *a, **k: _gemm(*a, tile_m=tile, **k)  # noqa: E731
    a = rnd(M, K, seed=5) * torch.logspace(-5, 1, M)[:, None]
    b = rnd(N, K, seed=6) * 0.03
    bias = rnd(N, seed=7)
    ref = a.double() @ b.double().T + bias.double()
    den = a.abs().double() @ b.abs().double().T + bias.abs().double()
    A, B = hip.h2_pack(dev(a), True), hip.h2_pack(dev(b), False)
    c, _ = hip_gemm_h2(A, B, bias=dev(bias))
    assert ((c.cpu().double() - ref).abs() / den).max().item() < 2e-6
    c3, _ = hip.gemm_tp3(hip.tp3_pack(dev(a)), hip.tp3_pack(dev(b)), bias=dev(bias))
    e2, e3 = ((c.cpu().double() - ref) / den).pow(2).mean().sqrt().item(), ((c3.cpu().double() - ref) / den).pow(2).mean().sqrt().item()
    assert e2 < 3 * e3 + 1e-9, (e2, e3)   # as accurate as the six-product bf16 scheme
    # tp3 output of the same call (what the attention / fc2 consume)
    _, ct = hip_gemm_h2(A, B, bias=dev(bias), want_f32=False, want_tp3=True)
    assert torch.equal(ct.float(), c)


def test_layernorm_h2_forward_and_backward(hip):
    rows, cols = 495, 768
    x, dy, dres = rnd(rows, cols, seed=8) * 3 + 0.5, rnd(rows, cols, seed=9) * torch.logspace(-7, -3, rows)[:, None], rnd(rows, cols, seed=10) * 1e-6
    g, b = 1 + 0.1 * rnd(cols, seed=11), 0.1 * rnd(cols, seed=12)
    y_ref = torch.nn.functional.layer_norm(x.double(), (cols,), g.double(), b.double(), 1e-5)
    y, mean, rstd = hip.layernorm_fwd_h2(dev(x), dev(g), dev(b), 1e-5)
    err = (y.float().cpu().double() - y_ref).abs()
    assert (err <= y_ref.abs().amax(1, keepdim=True) * 2.0**-20).all()
    y3, mean3, rstd3 = hip.layernorm_fwd_tp3(dev(x), dev(g), dev(b), 1e-5)
    assert torch.equal(mean, mean3) and torch.equal(rstd, rstd3)
    dx, dxt = hip.layernorm_bwd_h2(dev(dy), dev(x), dev(g), mean, rstd, dres=dev(dres))
    dx3, _ = hip.layernorm_bwd_tp3(dev(dy), dev(x), dev(g), mean, rstd, dres=dev(dres))
    assert torch.equal(dx, dx3)   # the fp32 result is the same kernel code
    e = (dxt.float() - dx).abs()
    assert (e <= dx.abs().amax(1, keepdim=True) * 2.0**-20).all()
    # the largest row norm arrives by tagged atomicMax in an un-cleared slot: equal to the reduction over row_norm, call after call
    assert torch.equal(y.norm_max, y.row_norm.max().reshape(1)) and torch.equal(dxt.norm_max, dxt.row_norm.max().reshape(1))
    from tunevlseg_amd import hip as H
    first = y.norm_max.item()
    H._max_slot_calls += 4096 - 2   # the next two calls land on the two slots just used, one turn of the pool later
    ys, _, _ = hip.layernorm_fwd_h2(dev(x), dev(g * 1e-3), dev(b * 1e-3), 1e-5)      # smaller norms than the slot's previous owner
    assert torch.equal(ys.norm_max, ys.row_norm.max().reshape(1)) and ys.norm_max.item() < 2e-3 * first
    assert ys.norm_max.data_ptr() == y.norm_max.data_ptr()


def test_gemm_h2_with_h2_output_feeds_the_next_gemm(hip):
    """fc1 -> QuickGELU -> fc2 on two-piece operands end to end: the first GEMM's epilogue writes its result as an h2 image whose row
    scales come from the Cauchy-Schwarz bound ||A row|| max ||B row|| + max |bias| (no value may overflow fp16, every row keeps ~20 bits)."""
    M, D, F = 700, 768, 3072
    x = rnd(M, D, seed=13) * torch.logspace(-3, 1, M)[:, None]
    w1, b1, w2 = rnd(F, D, seed=14) * 0.04, rnd(F, seed=15) * 0.1, rnd(D, F, seed=16) * 0.02
    X = hip.h2_pack(dev(x), True, want_norm=True)
    W1, W2 = hip.weight_h2(dev(w1)), hip.weight_h2(dev(w2))
    z = torch.empty(M, F, device="cuda")
    _, a = hip.gemm_h2(X, W1, want_f32=False, want_h2=True, out_add=float(b1.abs().max()), bias=dev(b1), act=hip.ACT_QUICK_GELU, pre_out=z)
    z_ref = x.double() @ w1.double().T + b1.double()
    a_ref = z_ref * torch.sigmoid(1.702 * z_ref)
    assert torch.isfinite(a.float()).all()
    top = a_ref.abs().amax(1) / a.inv_scale.cpu().double()
    assert (top < 2.0**14).all()                       # the bound held: nothing near fp16's 65504
    assert (a.float().cpu().double() - a_ref).abs().max().item() <= 1e-5 * a_ref.abs().max().item()
    err_rows = (a.float().cpu().double() - a_ref).abs().amax(1) / a_ref.abs().amax(1)
    assert err_rows.max().item() < 2e-5, err_rows.max()   # every row, however small, keeps its own precision
    y, _ = hip.gemm_h2(a, W2)
    y_ref = a_ref @ w2.double().T
    den = a_ref.abs() @ w2.abs().double().T
    assert ((y.cpu().double() - y_ref).abs() / den).max().item() < 5e-6


@pytest.mark.parametrize("B,T,H", [(2, 495, 12), (3, 100, 2), (1, 33, 1), (4, 64, 3), (4, 8, 2), (5, 31, 1)])
def test_attention_on_h2_operands(hip, B, T, H):
    """Forward and backward attention on two-piece fp16 images of QKV / dO (one power-of-two scale per tensor) against the tp3 kernels
    (six bf16-piece products): same tiles and mappings, half the MFMAs; dO rows spanning four decades as gradients do."""
    dh, D = 64, H * 64
    qkv = dev(rnd(B * T, 3 * D, seed=21))
    d_o = dev(rnd(B * T, D, seed=22) * torch.logspace(-6, -2, B * T)[:, None])
    q3 = hip.tp3_pack(qkv)
    o_ref, lse_ref = hip.attn_tp3_fwd(q3, B, T, H, dh**-0.5)
    g_ref = hip.attn_tp3_bwd(q3, o_ref, hip.tp3_pack(d_o), lse_ref, B, T, H, dh**-0.5).float()
    qh = hip.h2_pack(qkv, per_row=False)
    o, lse = hip.attn_h2_fwd(qh, B, T, H, dh**-0.5)
    close(o.float(), o_ref.float(), 3e-6, "attn h2 fwd O")
    close(lse, lse_ref, 2e-6, "attn h2 fwd lse")
    dh_ = hip.h2_pack(d_o, per_row=False)
    g = hip.attn_h2_bwd(qh, o, dh_, lse, B, T, H, dh**-0.5).float()
    for name, c0 in (("dQ", 0), ("dK", D), ("dV", 2 * D)):
        close(g[:, c0:c0 + D], g_ref[:, c0:c0 + D], 1e-5, f"attn h2 bwd {name}")
    # O as an h2 image that shares the QKV scale (the out-projection GEMM's A operand), and the backward reading it
    o2, lse2 = hip.attn_h2_fwd(qh, B, T, H, dh**-0.5, o_as_h2=True)
    assert torch.equal(lse2, lse)
    close(o2.float(), o_ref.float(), 3e-6, "attn h2 fwd O as h2")
    g2 = hip.attn_h2_bwd(qh, o2, dh_, lse, B, T, H, dh**-0.5).float()
    close(g2, g, 2e-6, "attn h2 bwd with the h2 O")
    # the gradient as an h2 image with one exact scale per (row, head, part), and the GEMM that rescales its accumulators per 64-column chunk
    gk = hip.attn_h2_bwd(qh, o2, dh_, lse, B, T, H, dh**-0.5, out_h2=True)
    inv = gk.kscale.cpu()
    assert torch.equal(inv, torch.exp2(torch.round(torch.log2(inv))))
    blocks = g2.cpu().view(B * T, 3 * H, 64)
    err = (gk.float().cpu().view(B * T, 3 * H, 64) - blocks).abs().amax(2)
    assert (err <= blocks.abs().amax(2) * 2.0**-19 + 1e-30).all()   # every (row, head) block keeps its own precision
    w = dev(rnd(D, 3 * D, seed=23) * 0.03)
    ref = g2.double().cpu() @ w.double().cpu().T
    den = g2.abs().double().cpu() @ w.abs().double().cpu().T
    got = hip.gemm_h2_ks(gk, hip.weight_h2(w)).cpu().double()
    assert ((got - ref).abs() / (den + 1e-300)).max().item() < 5e-6
    # one 128-row block only (the layer under the visual prompts): those rows are the full call's rows bit for bit, run after run
    full = gk.float()
    for blk in sorted({0, (T - 1) // 128}):
        row0 = blk * 128
        n = min(128, T - row0)
        for _ in range(3):
            part = hip.attn_h2_bwd(qh, o2, dh_, lse, B, T, H, dh**-0.5, out_h2=True, only_block=blk)
            rows = hip.h2k_gather_rows(part, B, T, row0, n)
            assert torch.equal(rows.view(B, n, 3 * D), full.view(B, T, 3 * D)[:, row0:row0 + n])
    with pytest.raises(RuntimeError):
        hip.attn_h2_bwd(qh, o2, dh_, lse, B, T, H, dh**-0.5, out_h2=True, only_block=(T + 127) // 128)


def test_gemm_h2_persistent_tile_walk_equals_one_workgroup_per_tile(hip):
    """fc1's shape at the headline batch (M = 15840, N = 3072, K = 768: 744 tiles of 256 x 256 -> the persistent walk with XCD-phased split
    first tiles, csrc/gemm_h2m_kernel.h) against the same kernel launched one workgroup per tile: the split tiles sum their two k-ranges in
    another order, everything else is the same arithmetic -> equal to fp32 rounding; a subsample of rows against float64."""
    if not hip.load().tvl_build_flags() & 1:
        pytest.skip("the persistent tile walk is a retired experiment: only in `make EXPERIMENTS=1` builds of the library")
    M, N, K = 15840, 3072, 768
    x = dev(rnd(M, K, seed=31))
    w = rnd(N, K, seed=32) * 0.05
    b1 = rnd(N, seed=33) * 0.1
    X, W = hip.h2_pack(x, True, want_norm=True), hip.weight_h2(dev(w))
    import os

    outs = []
    for persistent in (True, False):
        os.environ["TVL_GEMM_PERSIST"] = "1" if persistent else "0"   # the walk is an opt-in experiment (read per launch by the library)
        z = torch.empty(M, N, device="cuda")
        _, a = hip.gemm_h2(X, W, want_f32=False, want_h2=True, out_add=float(b1.abs().max()), bias=dev(b1), act=hip.ACT_QUICK_GELU, pre_out=z,
                           persistent=persistent)
        outs.append((z, a.float(), a.inv_scale.clone()))
    os.environ.pop("TVL_GEMM_PERSIST", None)
    (z1, a1, s1), (z0, a0, s0) = outs
    assert torch.equal(s1, s0)
    assert (z1 - z0).abs().max().item() <= 2e-6 * z0.abs().max().item() and not torch.equal(z1, z0)   # the split tiles DID take the other order
    assert (a1 - a0).abs().max().item() <= 4e-6 * a0.abs().max().item()
    rows = torch.arange(0, M, 97)
    ref = x[rows].double().cpu() @ w.double().T + b1.double()
    den = x[rows].abs().double().cpu() @ w.abs().double().T + b1.abs().double()
    assert ((z1[rows].cpu().double() - ref).abs() / den).max().item() < 2e-6


def test_gemm_h2_pre_activation_in_accumulator_order_round_trips(hip):
    """fc1 -> z (private accumulator-order buffer, hip.gemm_aux) -> the QuickGELU' epilogue of the data gradient: the same dz image as with
    a row-major z -- bit for bit when the buffer holds z in fp32 (TVL_GEMM_ZHALF=0), within the one fp16 rounding of QuickGELU'(z) (2^-11
    relative per element) in the default mode, where the buffer holds the derivative itself as one fp16 per element; the activation image is
    bit-identical either way; shapes that do not pick the 256-row tile of the 16x16x32 ring get no such buffer."""
    M, N, K = 15840, 3072, 768
    assert hip.gemm_aux(1000, 768, "cuda") is None and hip.gemm_aux(M, 2304, "cuda") is None
    x, g = dev(rnd(M, K, seed=41)), dev(rnd(M, K, seed=42) * 1e-3)
    w1, w2t, b1 = rnd(N, K, seed=43) * 0.05, rnd(N, K, seed=44) * 0.05, rnd(N, seed=45) * 0.1
    X, G = hip.h2_pack(x, True, want_norm=True), hip.h2_pack(g, True, want_norm=True)
    W1, W2T = hip.weight_h2(dev(w1)), hip.weight_h2(dev(w2t))
    res = []
    for blocked in (True, False):
        z = hip.gemm_aux(M, N, "cuda") if blocked else torch.empty(M, N, device="cuda")
        assert z is not None and (z.dim() == 1) == blocked
        _, a = hip.gemm_h2(X, W1, want_f32=False, want_h2=True, out_add=float(b1.abs().max()), bias=dev(b1), act=hip.ACT_QUICK_GELU, pre_out=z, aux_blocked=blocked)
        _, dz = hip.gemm_h2(G, W2T, want_f32=False, want_h2=True, out_mul=1.125 * W2T._bound, dact=hip.ACT_QUICK_GELU, dact_aux=z, aux_blocked=blocked)
        res.append((a.buf.clone(), dz.buf.clone(), dz.inv_scale.clone(), dz.float()))
    (a1, d1, s1, f1), (a0, d0, s0, f0) = res
    assert torch.equal(a1, a0) and torch.equal(s1, s0)
    if hip.GEMM_ZHALF:
        assert hip.gemm_aux(M, N, "cuda").numel() == (hip.load().tvl_gemm_aux_floats(M, N) + 1) // 2   # one fp16 per element
        # one fp16 rounding of the factor: 2^-11 relative where QuickGELU'(z) is a normal fp16, 2^-25 absolute (times the row's largest |dy W2|, bounded
        # here by the row's largest |dz| / min |factor| ~ a few row maxima) where it is subnormal (|factor| < 6e-5: z far in the negative tail)
        rowmax = f0.abs().amax(1, keepdim=True)
        assert ((f1 - f0).abs() <= 2.0**-10 * f0.abs() + 2.0**-20 * rowmax).all()
        assert float((f1 - f0).abs().max()) > 0.0                          # ... and it is the half-precision path that ran
    else:
        assert torch.equal(d1, d0)


@pytest.mark.parametrize("M,F", [(15520, 2048), (77, 128), (64, 256), (1000, 2048)])
def test_mlp64_block_forward_and_backward_match_float64(hip, M, F):
    """LayerNorm(x + W2 relu(W1 x + b1) + b2) and its data gradient in one launch each (csrc/mlp64.hip) against float64 torch;
    rows spread over five decades (every row carries its own power-of-two scale) and a few exactly-zero rows."""
    from tunevlseg_amd import hip as H

    g = torch.Generator().manual_seed(M + F)
    x = torch.randn(M, 64, generator=g) * torch.logspace(-3, 2, M)[torch.randperm(M, generator=g)][:, None]
    x[::97] = 0
    W1, b1 = torch.randn(F, 64, generator=g) * 0.2, torch.randn(F, generator=g) * 0.3
    W2, b2 = torch.randn(64, F, generator=g) * 0.05, torch.randn(64, generator=g) * 0.1
    gamma, beta = 1 + 0.2 * torch.randn(64, generator=g), 0.1 * torch.randn(64, generator=g)
    dout = torch.randn(M, 64, generator=g) * torch.logspace(-6, -2, M)[:, None]
    xd = x.double().requires_grad_(True)
    t2_ref = xd + torch.relu(xd @ W1.double().T + b1.double()) @ W2.double().T + b2.double()
    out_ref = torch.nn.functional.layer_norm(t2_ref, (64,), gamma.double(), beta.double(), 1e-5)
    out_ref.backward(dout.double())
    w = H.Mlp64Weights(dev(W1), dev(b1), dev(W2))
    out, t2, mean, rstd = H.mlp64_fwd(dev(x), w, dev(b1), dev(b2), dev(gamma), dev(beta), 1e-5)
    scale_t = t2_ref.detach().abs().amax(1, keepdim=True).clamp(min=1e-30)
    assert ((t2.cpu().double() - t2_ref.detach()).abs() / scale_t).max().item() < 2e-6
    assert (out.cpu().double() - out_ref.detach()).abs().max().item() < 2e-5
    # against the op-by-op fp32 path of the same library (what the fused kernel replaces)
    u = H.linear_fwd(dev(x), dev(W1), dev(b1), act=H.ACT_RELU)
    t2_ops = H.linear_fwd(u, dev(W2), dev(b2), residual=dev(x))
    assert ((t2 - t2_ops).abs().cpu().double() / scale_t).max().item() < 4e-6
    dx = H.mlp64_bwd(dev(dout), dev(x), t2, mean, rstd, w, dev(b1), dev(gamma))
    ref = xd.grad
    rel = ((dx.cpu().double() - ref).norm(dim=1) / ref.norm(dim=1).clamp(min=1e-300))
    assert rel.max().item() < 2e-5, rel.max()   # row by row: small-gradient rows keep their own precision
    out_nostat, t2n, _, _ = H.mlp64_fwd(dev(x), w, dev(b1), dev(b2), dev(gamma), dev(beta), 1e-5, want_stats=False)
    assert t2n is None and torch.equal(out_nostat, out)


def test_h2_zero_rows_cuts_the_image_like_the_matrix(hip):
    """tvl_h2_zero_rows: rows b*T + row0 .. + n - 1 of an h2 image become exact zeros, every other row keeps both pieces."""
    from tunevlseg_amd import hip as H

    B, T, D, row0, n = 3, 45, 128, 41, 4
    x = rnd(B * T, D, seed=77) * torch.logspace(-2, 2, B * T)[:, None]
    img = H.h2_pack(dev(x), per_row=True)
    before = img.float().clone()
    H.h2_zero_rows(img, B, T, row0, n)
    after = img.float()
    want = before.clone().view(B, T, D)
    want[:, row0:row0 + n] = 0
    assert torch.equal(after, want.view(B * T, D))
