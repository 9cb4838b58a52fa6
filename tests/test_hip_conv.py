"""Kernel-level parity of the CRIS conv-path entry points (csrc/conv.hip, cross-attention) against float64 PyTorch.

Runs on the GPU box only (-m gpu).  Layout reminder: maps are NHWC pixel matrices [B*H*W, C].
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


def nhwc(x):  # [B,C,H,W] -> [B*H*W, C]
    B, C, H, W = x.shape
    return x.permute(0, 2, 3, 1).reshape(B * H * W, C).contiguous()


def nchw(m, B, H, W):
    return m.reshape(B, H, W, -1).permute(0, 3, 1, 2)


@pytest.mark.parametrize("B,C,H,W,stride,Cout", [(2, 8, 6, 6, 1, 5), (1, 3, 10, 12, 2, 4), (3, 6, 7, 5, 1, 9), (2, 64, 13, 13, 1, 32),
                                                 (1, 3, 32, 32, 2, 16)])
def test_conv3x3_as_im2col_gemm(hip, B, C, H, W, stride, Cout):
    x, w = rnd(B, C, H, W, seed=1), rnd(Cout, C, 3, 3, seed=2)
    ref = F.conv2d(x.double(), w.double(), stride=stride, padding=1)
    Ho, Wo = ref.shape[2:]
    cols = hip.im2col3x3(dev(nhwc(x)), B, H, W, stride)
    K = cols.shape[1]
    wm = torch.zeros(Cout, K)
    wm[:, : 9 * C] = w.permute(0, 2, 3, 1).reshape(Cout, 9 * C)  # (ky, kx, c) column order
    y = torch.empty(B * Ho * Wo, Cout, device="cuda")
    hip.gemm(0, B * Ho * Wo, Cout, K, cols, K, dev(wm), K, y, Cout)
    close(nchw(y, B, Ho, Wo), ref, 1e-5 * math.sqrt(9 * C), "conv3x3")
    # the stem reads the NCHW image directly
    cols2 = hip.im2col3x3_nchw(dev(x), stride)
    assert torch.equal(cols2, cols)


@pytest.mark.parametrize("B,C,H,W,stride,Cout", [(2, 8, 12, 12, 1, 5), (1, 32, 26, 26, 1, 64), (3, 64, 13, 13, 1, 40), (2, 16, 20, 14, 2, 24),
                                                 (1, 512, 26, 26, 1, 512), (4, 36, 9, 11, 1, 130)])
@pytest.mark.parametrize("mode", ["bf16x6", "f32"])
def test_conv3x3_implicit_gemm(hip, B, C, H, W, stride, Cout, mode):
    """hip.conv3x3: implicit GEMM (split-bf16) / im2col + GEMM (fp32) with bias + ReLU epilogue, reading a channel slice."""
    x, w, b = rnd(B, C + 4, H, W, seed=21), rnd(Cout, C, 3, 3, seed=22) * (9 * C) ** -0.5, rnd(Cout, seed=23)
    ref = F.relu(F.conv2d(x[:, 4:].double(), w.double(), b.double(), stride=stride, padding=1))
    Ho, Wo = ref.shape[2:]
    wm = w.permute(0, 2, 3, 1).reshape(Cout, 9 * C)
    wide = dev(nhwc(x))
    old = hip.GEMM_MODE
    hip.set_gemm_mode(mode)
    try:
        y = hip.conv3x3(wide[:, 4:], B, H, W, dev(wm), dev(b), hip.ACT_RELU, stride)
    finally:
        hip.set_gemm_mode(old)
    close(nchw(y, B, Ho, Wo), ref, 2e-5, f"conv3x3 {mode}")


def test_im2col_reads_channel_slice(hip):
    B, C, H, W = 2, 8, 5, 5
    x = rnd(B, 2 * C, H, W, seed=3)
    wide = dev(nhwc(x))
    a = hip.im2col3x3(wide[:, C:], B, H, W, 1)
    b = hip.im2col3x3(dev(nhwc(x[:, C:])), B, H, W, 1)
    assert torch.equal(a, b)


@pytest.mark.parametrize("B,C,H,W,k", [(2, 5, 8, 6, 2), (1, 16, 4, 4, 2), (2, 3, 9, 6, 3)])
def test_avgpool(hip, B, C, H, W, k):
    x = rnd(B, C, H, W, seed=4).double().requires_grad_()
    ref = F.avg_pool2d(x, k)
    g = rnd(*ref.shape, seed=5)
    ref.backward(g.double())
    y = hip.avgpool_fwd(dev(nhwc(x.detach().float())), B, H, W, k)
    close(nchw(y, B, H // k, W // k), ref, 1e-6, "avgpool fwd")
    dx = hip.avgpool_bwd(dev(nhwc(g)), B, H, W, k)
    close(nchw(dx, B, H, W), x.grad, 1e-6, "avgpool bwd")
    # write into a column slice of a wider buffer (channel concat)
    wide = torch.zeros(B * (H // k) * (W // k), C + 3, device="cuda")
    hip.avgpool_fwd(dev(nhwc(x.detach().float())), B, H, W, k, out=wide[:, 3:])
    assert torch.equal(wide[:, 3:], y) and wide[:, :3].abs().max() == 0


@pytest.mark.parametrize("B,C,H,W,s", [(2, 5, 3, 4, 2), (1, 8, 13, 13, 2), (2, 3, 1, 5, 2), (1, 4, 6, 6, 4)])
def test_bilinear_up(hip, B, C, H, W, s):
    x = rnd(B, C, H, W, seed=6).double().requires_grad_()
    ref = F.interpolate(x, scale_factor=s, mode="bilinear")
    g = rnd(*ref.shape, seed=7)
    ref.backward(g.double())
    y = hip.bilinear_up_fwd(dev(nhwc(x.detach().float())), B, H, W, s)
    close(nchw(y, B, H * s, W * s), ref, 2e-6, "bilinear fwd")
    dx = hip.bilinear_up_bwd(dev(nhwc(g)), B, H, W, s)
    close(nchw(dx, B, H, W), x.grad, 2e-6, "bilinear bwd")


@pytest.mark.parametrize("B,Hi,Wi,Ho,Wo", [(2, 6, 6, 24, 24), (1, 24, 24, 96, 96), (2, 5, 7, 13, 30), (1, 104, 104, 416, 416), (1, 4, 4, 4, 4)])
def test_bicubic_align_corners(hip, B, Hi, Wi, Ho, Wo):
    x = rnd(B, 1, Hi, Wi, seed=8).double().requires_grad_()
    ref = F.interpolate(x, (Ho, Wo), mode="bicubic", align_corners=True)
    g = rnd(*ref.shape, seed=9)
    ref.backward(g.double())
    extra = rnd(B, Ho, Wo, seed=10)
    y = hip.bicubic_ac_fwd(dev(x.detach().float()[:, 0]), Ho, Wo)
    close(y, ref[:, 0], 2e-5, "bicubic fwd")
    y2 = hip.bicubic_ac_fwd(dev(x.detach().float()[:, 0]), Ho, Wo, extra=dev(extra), a=0.3, r=0.7)
    close(y2, 0.3 * ref[:, 0] + 0.7 * extra.double(), 2e-5, "bicubic fwd mix")
    dx = hip.bicubic_ac_bwd(dev(g[:, 0]), Hi, Wi, a=0.3)
    close(dx, 0.3 * x.grad[:, 0], 2e-5, "bicubic bwd")


@pytest.mark.parametrize("B,C,H,W", [(2, 16, 6, 6), (3, 5, 7, 9), (1, 256, 24, 24), (2, 70, 12, 11)])
def test_dynconv(hip, B, C, H, W):
    x = rnd(B, C, H, W, seed=11).double().requires_grad_()
    word = (rnd(B, 9 * C + 1, seed=12) * 0.2).double().requires_grad_()
    wgt, bias = word[:, :-1].reshape(B, C, 3, 3), word[:, -1]
    ref = F.conv2d(x.reshape(1, B * C, H, W), wgt, bias, padding=1, groups=B).transpose(0, 1)  # [B,1,H,W]
    g = rnd(*ref.shape, seed=13)
    ref.backward(g.double())
    xm, wd = dev(nhwc(x.detach().float())), dev(word.detach().float())
    out = hip.dynconv_fwd(xm, wd, B, H, W)
    close(out, ref[:, 0], 1e-5 * math.sqrt(9 * C), "dynconv fwd")
    dx, dword = hip.dynconv_bwd(dev(g[:, 0]), xm, wd, B, H, W)
    close(nchw(dx, B, H, W), x.grad, 1e-5, "dynconv dx")
    close(dword, word.grad, 1e-5 * math.sqrt(H * W), "dynconv dword")


@pytest.mark.parametrize("B,H,dh,T,Tk,masked", [(2, 2, 16, 36, 10, True), (1, 8, 64, 676, 17, True), (2, 4, 32, 50, 77, False),
                                                 (3, 2, 8, 5, 130, True)])
def test_cross_attention(hip, B, H, dh, T, Tk, masked):
    D = H * dh
    q, k, v = rnd(B, T, D, seed=14), rnd(B, Tk, D, seed=15), rnd(B, Tk, D, seed=16)
    km = torch.ones(B, Tk, dtype=torch.int32)
    if masked:
        for b in range(B):
            km[b, Tk - 1 - (b % 3) * 2:] = 0
    qd, kd, vd = (t.double().requires_grad_() for t in (q, k, v))
    qh = qd.view(B, T, H, dh).transpose(1, 2)
    kh = kd.view(B, Tk, H, dh).transpose(1, 2)
    vh = vd.view(B, Tk, H, dh).transpose(1, 2)
    s = qh @ kh.transpose(-1, -2) * dh**-0.5
    s = s.masked_fill(km[:, None, None, :] == 0, float("-inf"))
    ref = (s.softmax(-1) @ vh).transpose(1, 2).reshape(B, T, D)
    g = rnd(B, T, D, seed=17)
    ref.backward(g.double())
    q2, k2, v2 = dev(q.view(B * T, D)), dev(k.view(B * Tk, D)), dev(v.view(B * Tk, D))
    o, lse = hip.attn_fwd(q2, k2, v2, B, T, Tk, H, dh, dh**-0.5, key_mask=dev(km) if masked else None)
    close(o.view(B, T, D), ref, 2e-5, "xattn fwd")
    dq, dk, dv = torch.empty_like(q2), torch.empty_like(k2), torch.empty_like(v2)
    hip.attn_bwd(q2, k2, v2, o, dev(g.view(B * T, D)), lse, dq, dk, dv, B, T, Tk, H, dh, dh**-0.5, key_mask=dev(km) if masked else None)
    close(dq.view(B, T, D), qd.grad, 3e-5, "xattn dq")
    close(dk.view(B, Tk, D), kd.grad, 3e-5, "xattn dk")
    close(dv.view(B, Tk, D), vd.grad, 3e-5, "xattn dv")


def test_self_attention_still_matches_through_generic_entry(hip):
    """Tk == T through the separate-pointer entry: packed-slice views of one QKV buffer."""
    B, H, dh, T = 2, 2, 16, 40
    D = H * dh
    qkv = dev(rnd(B * T, 3 * D, seed=18))
    o1, lse1 = hip.attn_fwd_packed(qkv, B, T, H, dh, dh**-0.5)
    o2, lse2 = hip.attn_fwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], B, T, T, H, dh, dh**-0.5)
    assert torch.equal(o1, o2) and torch.equal(lse1, lse2)


@pytest.mark.parametrize("B,C,H,W,Cout", [(32, 512, 26, 26, 512), (8, 192, 104, 104, 256)])
def test_conv3x3_implicit_gemm_wide_tile(hip, B, C, H, W, Cout):
    """Conv shapes that take the 192x256 8-wave tile (N % 256 == 0, K = 9C >= 1536)."""
    from tunevlseg_amd.hip import _bf16s_tile

    assert _bf16s_tile(B * H * W, Cout, 9 * C)[:2] == (192, 256)
    x, w, b = rnd(B, C, H, W, seed=31), rnd(Cout, C, 3, 3, seed=32) * (9 * C) ** -0.5, rnd(Cout, seed=33)
    ref = F.relu(F.conv2d(x.double(), w.double(), b.double(), padding=1))
    y = hip.conv3x3(dev(nhwc(x)), B, H, W, dev(w.permute(0, 2, 3, 1).reshape(Cout, 9 * C)), dev(b), hip.ACT_RELU)
    close(nchw(y, B, H, W), ref, 2e-5, "conv3x3 wide tile")


@pytest.mark.parametrize("B,C,H,W,Cout,bias_relu", [(32, 512, 26, 26, 512, True), (16, 32, 64, 64, 256, True), (5, 96, 21, 23, 144, False),
                                                    (2, 128, 40, 33, 128, True), (1, 64, 50, 47, 384, False)])
def test_conv3x3_two_piece_fp16(hip, B, C, H, W, Cout, bias_relu):
    """hip.conv3x3 over a frozen weight with C % 32 == 0: implicit GEMM on two fp16 pieces (tvl_conv3x3_h2), taps gathered by LDS-DMA, the
    padding taps from the zero block; ragged M (not a multiple of 32), both row tiles, generic and plain epilogue."""
    from tunevlseg_amd.hip import conv_h2_kernel_name

    assert hip.CONV_H2 and hip.GEMM_MODE == "bf16x6"
    x, w, b = rnd(B, C, H, W, seed=41), rnd(Cout, C, 3, 3, seed=42) * (9 * C) ** -0.5, rnd(Cout, seed=43)
    x[:, : C // 2] *= 1e-3   # a wide range under the one tensor scale
    ref = F.conv2d(x.double(), w.double(), b.double() if bias_relu else None, padding=1)
    ref = F.relu(ref) if bias_relu else ref
    wm = hip.mark_frozen(dev(w.permute(0, 2, 3, 1).reshape(Cout, 9 * C)))
    prof = []
    hip._gemm_prof, old = prof, hip._gemm_prof
    try:
        y = hip.conv3x3(dev(nhwc(x)), B, H, W, wm, dev(b) if bias_relu else None, hip.ACT_RELU if bias_relu else hip.ACT_NONE)
        torch.cuda.synchronize()
    finally:
        hip._gemm_prof = old
    assert prof and prof[0][0] == conv_h2_kernel_name(B * H * W, Cout, bias_relu, hip.ACT_RELU if bias_relu else hip.ACT_NONE), prof   # the h2 kernel ran, not the bf16 one
    close(nchw(y, B, H, W), ref, 2e-5, "conv3x3 h2")
    # against the three-bf16-piece implicit GEMM on the same input: both are fp32-equivalent
    hip.CONV_H2 = False
    try:
        y3 = hip.conv3x3(dev(nhwc(x)), B, H, W, wm, dev(b) if bias_relu else None, hip.ACT_RELU if bias_relu else hip.ACT_NONE)
    finally:
        hip.CONV_H2 = True
    close(y, y3, 1e-5, "conv3x3 h2 vs bf16x3")


@pytest.mark.parametrize("B,C,H,W,s,Cout", [(2, 64, 13, 11, 2, 128), (3, 96, 9, 10, 2, 144)])
def test_bilinear_upsample_written_as_conv_operand(hip, B, C, H, W, s, Cout):
    """tvl_bilinear_up_h2: the upsample as the h2 image of its result (one scale = the input's; ragged row count), and the 3x3 conv reading it
    (``conv3x3(packed=...)``) against the conv of the fp32 upsample."""
    x = dev(rnd(B * H * W, C, seed=51))
    x[:, : C // 2] *= 1e-2
    up = hip.bilinear_up_fwd(x, B, H, W, s)
    img = hip.bilinear_up_h2(x, B, H, W, s)
    assert not img.per_row and img.rows == B * H * s * W * s
    amax = x.abs().max().item()
    assert 2.0**13 <= amax / img.inv_scale.item() < 2.0**14   # the input's maximum sets the scale
    close(img.float(), up, 2.0**-20, "upsample as h2")
    assert img.buf[-(C // 16) * 2048:].abs().max().item() == 0   # the zero block the padding taps read
    w, b = rnd(Cout, C, 3, 3, seed=52) * (9 * C) ** -0.5, rnd(Cout, seed=53)
    wm = hip.mark_frozen(dev(w.permute(0, 2, 3, 1).reshape(Cout, 9 * C)))
    Ho, Wo = H * s, W * s
    if hip.conv3x3_takes_h2(B * Ho * Wo, C, wm):
        y = hip.conv3x3(None, B, Ho, Wo, wm, dev(b), hip.ACT_RELU, packed=img)
        y_ref = hip.conv3x3(up, B, Ho, Wo, wm, dev(b), hip.ACT_RELU)
        close(y, y_ref, 1e-5, "conv over the packed upsample")
        ref = F.relu(F.conv2d(F.interpolate(nchw(x.cpu(), B, H, W).double(), scale_factor=s, mode="bilinear"), w.double(), b.double(), padding=1))
        close(nchw(y, B, Ho, Wo), ref, 2e-5, "upsample + conv vs float64")


def test_conv3x3_two_piece_full_size_properties(hip):
    """BASELINE configs[2]'s largest conv (projector vis3: B = 32, 104 x 104, 512 -> 256; M = 346 112 rows, K = 4608) through properties that
    need no reference: (1) scaling the map by a power of two scales the result EXACTLY (the image's pieces are identical, only its
    power-of-two scale moves); (2) permuting the samples permutes the result bit for bit (a row's k-loop does not depend on its tile);
    (3) an input that is zero outside one sample leaves the other samples' outputs exactly zero (no tap crosses a sample boundary)."""
    B, C, H, W, Cout = 32, 512, 104, 104, 256
    g = torch.Generator(device="cuda").manual_seed(7)
    x = torch.randn(B * H * W, C, device="cuda", generator=g)
    wm = hip.mark_frozen(torch.randn(Cout, 9 * C, device="cuda", generator=g) * (9 * C) ** -0.5)
    assert hip.conv3x3_takes_h2(B * H * W, C, wm)
    y = hip.conv3x3(x, B, H, W, wm)
    y4 = hip.conv3x3(x * 4.0, B, H, W, wm)
    assert torch.equal(y4, y * 4.0)
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(3)).cuda()
    xp = x.view(B, H * W, C)[perm].reshape(B * H * W, C).contiguous()
    yp = hip.conv3x3(xp, B, H, W, wm)
    assert torch.equal(yp.view(B, H * W, Cout), y.view(B, H * W, Cout)[perm])
    del y4, yp, xp
    xs = torch.zeros_like(x)
    xs.view(B, H * W, C)[5] = x.view(B, H * W, C)[5]
    ys = hip.conv3x3(xs, B, H, W, wm).view(B, H * W, Cout)
    assert ys[:5].abs().max().item() == 0 and ys[6:].abs().max().item() == 0
    # (the scale of xs is that of sample 5 alone, which may differ from the full map's: compare to rounding, not to the bit)
    ref5 = y.view(B, H * W, Cout)[5]
    assert (ys[5] - ref5).abs().max().item() <= 2e-6 * ref5.abs().max().item()


def _frozen_linear(hip, n, k, seed, bias=True):
    from tunevlseg_amd.cris_ops import FrozenLinear

    w = dev(rnd(n, k, seed=seed) * k ** -0.5)
    return FrozenLinear(hip.mark_frozen(w), dev(rnd(n, seed=seed + 1) * 0.1) if bias else None, hip.mark_frozen(w.t().contiguous()))


def test_cris_decoder_blocks_on_two_pieces_match_the_unfused_path(hip):
    """SelfAttnBlockFn / FFNBlockFn (QK | V projections -> one packed image -> attention -> out-projection; LayerNorm -> GEMM -> ReLU ->
    LayerNorm -> GEMM + residual, all operands as two fp16 pieces) against the op-by-op path on fp32 tensors: outputs and input gradients."""
    from tunevlseg_amd import cris_ops as CO
    from tunevlseg_amd import ops

    B, T, H, dh, F_ = 4, 676, 8, 64, 2048
    D = H * dh
    fqk, fv, fo = _frozen_linear(hip, 2 * D, D, 61), _frozen_linear(hip, D, D, 63), _frozen_linear(hip, D, D, 65)
    assert CO.SelfAttnBlockFn.takes(B * T, dh, fqk, fv, fo)
    xq0, xv0, dy = dev(rnd(B, T, D, seed=67)), dev(rnd(B, T, D, seed=68)), dev(rnd(B * T, D, seed=69))
    res = {}
    for fused in (True, False):
        CO.SELF_ATTN_H2 = fused
        try:
            xq, xv = xq0.clone().requires_grad_(True), xv0.clone().requires_grad_(True)
            out = CO.self_attn_block(xq, xv, fqk, fv, fo, B, T, H, dh)
            out.backward(dy)
            res[fused] = (out.detach(), xq.grad, xv.grad)
        finally:
            CO.SELF_ATTN_H2 = True
    for a, b, what in zip(res[True], res[False], ("out", "d xq", "d xv")):
        close(a.reshape(b.shape), b, 2e-5, f"self-attention block {what}")

    f0, f4 = _frozen_linear(hip, F_, D, 71), _frozen_linear(hip, D, F_, 73)
    g3, b3, gf, bf = dev(1 + 0.1 * rnd(D, seed=75)), dev(0.1 * rnd(D, seed=76)), dev(1 + 0.1 * rnd(F_, seed=77)), dev(0.1 * rnd(F_, seed=78))
    assert CO.FFNBlockFn.takes(B * T, D, F_, f0, f4)
    x0, dy3 = dev(rnd(B, T, D, seed=79)), dev(rnd(B, T, D, seed=80))
    x = x0.clone().requires_grad_(True)
    y = CO.FFNBlockFn.apply(x, g3, b3, f0, gf, bf, f4, 1e-5)
    y.backward(dy3)
    xr = x0.clone().requires_grad_(True)
    v2 = ops.layer_norm(xr, g3, b3, 1e-5)
    v2 = ops.layer_norm(CO.flinear_g(v2, f0, hip.ACT_RELU), gf, bf, 1e-5)
    yr = ops.add(xr, CO.flinear_g(v2, f4))
    yr.backward(dy3)
    close(y, yr, 2e-5, "ffn block out")
    close(x.grad, xr.grad, 2e-5, "ffn block d x")
