"""autograd nodes of the CRIS path (BASELINE configs[2]); every forward/backward body is a sequence of HIP launches.

Feature maps are NHWC pixel matrices ``[B*H*W, C]`` (csrc/conv.hip).  The whole CRIS model is frozen and in eval mode
(reference coop_cris.py:66-68), so BatchNorm is folded into the preceding conv/linear once at load time and every node
below carries a data-gradient-only backward; weight gradients exist only for the new last layer and the mix ratio.
"""
from __future__ import annotations

from dataclasses import dataclass

import torch

from . import hip
from .ops import Fn, _c


@dataclass
class FrozenLinear:
    """``y = act(x W^T + b)``: nn.Linear, or a 1x1 conv (+ folded eval BatchNorm) over a pixel matrix."""
    W: torch.Tensor            # [N, K]
    b: torch.Tensor | None     # [N]
    Wt: torch.Tensor           # [K, N]  (data gradient as an NT GEMM)


@dataclass
class FrozenConv3:
    """3x3 / pad 1 conv (+ folded eval BatchNorm) as a GEMM over the im2col matrix."""
    Wm: torch.Tensor           # [Cout, Kpad]  columns ordered (ky, kx, c), zero padded to a multiple of 4
    b: torch.Tensor | None     # [Cout]
    Wd: torch.Tensor | None    # [Cin, 9*Cout (padded)] tap-flipped transpose: dX = im2col(dY) Wd^T  (stride 1 only)
    cin: int
    cout: int


# ----------------------------------------------------------------------------------------------------------------------
# no-grad helpers (image tower, text-independent part of the neck)
# ----------------------------------------------------------------------------------------------------------------------
def flinear(x2d: torch.Tensor, fl: FrozenLinear, act: int = hip.ACT_NONE, residual=None, out=None) -> torch.Tensor:
    M, K = x2d.shape
    N = fl.W.shape[0]
    y = out if out is not None else torch.empty((M, N), device=x2d.device, dtype=torch.float32)
    hip.gemm(hip.NT, M, N, K, x2d, x2d.stride(0), fl.W, K, y, y.stride(0), bias=fl.b, act=act,
             residual=residual, ldr=0 if residual is None else residual.stride(0))
    return y


def fconv3(x2d: torch.Tensor, fc: FrozenConv3, B: int, H: int, W: int, act: int = hip.ACT_NONE, stride: int = 1, out=None) -> torch.Tensor:
    return hip.conv3x3(x2d, B, H, W, fc.Wm, fc.b, act, stride, out)


# ----------------------------------------------------------------------------------------------------------------------
# autograd nodes
# ----------------------------------------------------------------------------------------------------------------------
class FLinearFn(Fn):
    @staticmethod
    def forward(ctx, x, fl: FrozenLinear, act):
        shape = x.shape
        x2d = _c(x).view(-1, shape[-1])
        y = flinear(x2d, fl, act)
        ctx.fl, ctx.act, ctx.shape = fl, act, shape
        ctx.save_for_backward(y if act == hip.ACT_RELU else None)
        if act not in (hip.ACT_NONE, hip.ACT_RELU):
            raise NotImplementedError("FLinearFn: only ReLU / identity epilogues are differentiated")
        return y.view(*shape[:-1], y.shape[1])

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        fl = ctx.fl
        dy2d = _c(dy).view(-1, fl.W.shape[0])
        # relu'(pre) == (y > 0): the gate is applied while dy is packed for the GEMM (or in a pass of its own where the GEMM reads fp32)
        return hip.linear_dgrad(dy2d, fl.W, Wt=fl.Wt, relu_mask=y if ctx.act == hip.ACT_RELU else None).view(ctx.shape), None, None


def flinear_g(x, fl: FrozenLinear, act: int = hip.ACT_NONE):
    return FLinearFn.apply(x, fl, act)


class FConv3Fn(Fn):
    @staticmethod
    def forward(ctx, x2d, fc: FrozenConv3, B, H, W, act):
        x2d = _c(x2d)
        y = fconv3(x2d, fc, B, H, W, act)
        ctx.fc, ctx.act, ctx.geom = fc, act, (B, H, W)
        ctx.save_for_backward(y if act == hip.ACT_RELU else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        fc = ctx.fc
        B, H, W = ctx.geom
        return hip.conv3x3(_c(dy), B, H, W, fc.Wd, x_relu_mask=y if ctx.act == hip.ACT_RELU else None), None, None, None, None, None


def fconv3_g(x2d, fc: FrozenConv3, B: int, H: int, W: int, act: int = hip.ACT_RELU):
    return FConv3Fn.apply(x2d, fc, B, H, W, act)


class UpConv3Fn(Fn):
    """``conv3x3(F.interpolate(x, scale_factor=s, mode="bilinear"))`` + folded BN + ReLU (the projector, reference layers.py:100-107) as one
    node: where the conv runs on two fp16 pieces, the upsample writes the conv's operand image itself and the upsampled map never exists in
    fp32 (nothing needs it again: frozen weights have no weight gradient, the ReLU mask comes from the output)."""

    @staticmethod
    def forward(ctx, x2d, fc: FrozenConv3, B, H, W, s, act):
        x2d = _c(x2d)
        Ho, Wo = H * s, W * s
        if hip.conv3x3_takes_h2(B * Ho * Wo, x2d.shape[1], fc.Wm) and x2d.stride(0) % 4 == 0:
            y = hip.conv3x3(None, B, Ho, Wo, fc.Wm, fc.b, act, packed=hip.bilinear_up_h2(x2d, B, H, W, s))
        else:
            y = fconv3(hip.bilinear_up_fwd(x2d, B, H, W, s), fc, B, Ho, Wo, act)
        ctx.fc, ctx.act, ctx.geom = fc, act, (B, H, W, s)
        ctx.save_for_backward(y if act == hip.ACT_RELU else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        B, H, W, s = ctx.geom
        dup = hip.conv3x3(_c(dy), B, H * s, W * s, ctx.fc.Wd, x_relu_mask=y if ctx.act == hip.ACT_RELU else None)
        return hip.bilinear_up_bwd(dup, B, H, W, s), None, None, None, None, None, None


def up_conv3_g(x2d, fc: FrozenConv3, B: int, H: int, W: int, s: int = 2, act: int = hip.ACT_RELU):
    return UpConv3Fn.apply(x2d, fc, B, H, W, s, act)


class ReluFn(Fn):
    @staticmethod
    def forward(ctx, x):
        y = hip.bias_act(_c(x).view(-1, x.shape[-1]), None, hip.ACT_RELU).view(x.shape)
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        return hip.dact_mul(_c(dy).view(-1, y.shape[-1]), y.view(-1, y.shape[-1]), hip.ACT_RELU).view(y.shape)


class BilinearUpFn(Fn):
    @staticmethod
    def forward(ctx, x2d, B, H, W, s):
        ctx.geom = (B, H, W, s)
        return hip.bilinear_up_fwd(_c(x2d), B, H, W, s)

    @staticmethod
    def backward(ctx, dy):
        B, H, W, s = ctx.geom
        return hip.bilinear_up_bwd(_c(dy), B, H, W, s), None, None, None, None


class CatColsFn(Fn):
    """torch.cat(dim=1) of NCHW maps == column concat of pixel matrices (reference layers.py:424,429,440)."""

    @staticmethod
    def forward(ctx, *parts):
        rows = parts[0].shape[0]
        widths = [p.shape[1] for p in parts]
        out = torch.empty((rows, sum(widths)), device=parts[0].device, dtype=torch.float32)
        off = 0
        for p, w in zip(parts, widths):
            hip.copy2d(p, out[:, off:off + w])
            off += w
        ctx.widths = widths
        return out

    @staticmethod
    def backward(ctx, d):
        d = _c(d)
        grads, off = [], 0
        for i, w in enumerate(ctx.widths):
            if ctx.needs_input_grad[i]:
                g = torch.empty((d.shape[0], w), device=d.device, dtype=torch.float32)
                hip.copy2d(d[:, off:off + w], g)
                grads.append(g)
            else:
                grads.append(None)
            off += w
        return tuple(grads)


class BcastAddFn(Fn):
    """x[b, t, :] + pos[t, :]: fixed sin/cos position codes added to every sample (layers.py:318-320)."""

    @staticmethod
    def forward(ctx, x, pos, B):
        x2 = _c(x).view(B, -1)
        return hip.bias_act(x2, pos.view(-1), hip.ACT_NONE).view(x.shape)

    @staticmethod
    def backward(ctx, d):
        return d, None, None


class SelfAttnQKFn(Fn):
    """Self-attention whose q and k come from one packed projection ``qk`` [B*T, 2D] and v from another matrix."""

    @staticmethod
    def forward(ctx, qk, v, B, T, H, dh):
        qk, v = _c(qk), _c(v)
        D = H * dh
        o, lse = hip.attn_fwd(qk[:, :D], qk[:, D:], v, B, T, T, H, dh, dh**-0.5)
        ctx.save_for_backward(qk, v, o, lse)
        ctx.geom = (B, T, H, dh)
        return o

    @staticmethod
    def backward(ctx, d_o):
        qk, v, o, lse = ctx.saved_tensors
        B, T, H, dh = ctx.geom
        D = H * dh
        dqk, dv = torch.empty_like(qk), torch.empty_like(v)
        hip.attn_bwd(qk[:, :D], qk[:, D:], v, o, _c(d_o), lse, dqk[:, :D], dqk[:, D:], dv, B, T, T, H, dh, dh**-0.5)
        return dqk, dv, None, None, None, None


class SelfAttnBlockFn(Fn):
    """One decoder self-attention block, ``out_proj(attn(q = k = Wqk xq, v = Wv xv))`` (reference layers.py:322-328 with nn.MultiheadAttention's
    packed in-projection), on two fp16 pieces end to end: both projections write ONE [B*T, 3D] matrix, which is packed once with one scale;
    attention reads that image and writes O as an image sharing its scale; the out-projection consumes O's image directly.  The backward
    mirrors it: dO packed once, dQ | dK | dV from the attention kernels as one image, unpacked for the two projection data gradients."""

    @staticmethod
    def takes(M: int, dh: int, *fls: FrozenLinear) -> bool:
        return bool(hip.GEMM_MODE == "bf16x6" and hip.GEMM_H2 and hip.ATTN_H2 and SELF_ATTN_H2 and dh == 64 and M >= 2048
                    and all(getattr(f.W, "_tvl_frozen", False) and f.W.shape[1] % 32 == 0 for f in fls))

    @staticmethod
    def forward(ctx, xq, xv, fqk: FrozenLinear, fv: FrozenLinear, fo: FrozenLinear, B, T, H, dh):
        D = H * dh
        xq2, xv2 = _c(xq).view(B * T, -1), _c(xv).view(B * T, -1)
        buf = torch.empty((B * T, 3 * D), device=xq2.device, dtype=torch.float32)
        flinear(xq2, fqk, out=buf[:, :2 * D])
        flinear(xv2, fv, out=buf[:, 2 * D:])
        qkv_h = hip.h2_pack(buf, per_row=False)
        del buf
        o_h, lse = hip.attn_h2_fwd(qkv_h, B, T, H, dh**-0.5, o_as_h2=True)
        out, _ = hip.gemm_h2(o_h, hip.weight_h2_cached(fo.W), bias=fo.b)
        ctx.save_for_backward(qkv_h.buf, qkv_h.inv_scale, o_h.buf, lse)
        ctx.mods, ctx.geom, ctx.in_shapes = (fqk, fv, fo), (B, T, H, dh), (xq.shape, xv.shape)
        return out

    @staticmethod
    def backward(ctx, d_out):
        qkv_buf, qkv_inv, o_buf, lse = ctx.saved_tensors
        fqk, fv, fo = ctx.mods
        B, T, H, dh = ctx.geom
        D, M = H * dh, B * T
        qkv_h = hip.H2.wrap(M, 3 * D, qkv_buf, qkv_inv, per_row=False)
        o_h = hip.H2.wrap(M, D, o_buf, qkv_inv, per_row=False)
        do_h = hip.h2_pack(hip.linear_dgrad(_c(d_out).view(M, D), fo.W, Wt=fo.Wt), per_row=False)
        g = hip.attn_h2_bwd(qkv_h, o_h, do_h, lse, B, T, H, dh**-0.5).float()   # dQ | dK | dV [M, 3D]
        dxq = hip.linear_dgrad(g[:, :2 * D], fqk.W, Wt=fqk.Wt)
        dxv = hip.linear_dgrad(g[:, 2 * D:], fv.W, Wt=fv.Wt)
        return dxq.view(ctx.in_shapes[0]), dxv.view(ctx.in_shapes[1]), None, None, None, None, None, None, None


SELF_ATTN_H2 = __import__("os").environ.get("TVL_CRIS_SELF_ATTN_H2", "1") != "0"   # A/B switch of the fused decoder self-attention block


def self_attn_block(xq, xv, fqk: FrozenLinear, fv: FrozenLinear, fo: FrozenLinear, B: int, T: int, H: int, dh: int):
    """[B*T, D] output of the decoder's self-attention sub-block (before its LayerNorm)."""
    if SelfAttnBlockFn.takes(B * T, dh, fqk, fv, fo):
        return SelfAttnBlockFn.apply(xq, xv, fqk, fv, fo, B, T, H, dh)
    D = H * dh
    qk = flinear_g(xq, fqk).view(B * T, 2 * D)
    v = flinear_g(xv, fv).view(B * T, D)
    return flinear_g(SelfAttnQKFn.apply(qk, v, B, T, H, dh), fo)


class FFNBlockFn(Fn):
    """``x + W4 LN_f(relu(W0 LN_3(x) + b0)) + b4`` -- the decoder layer's feed-forward sub-block (reference layers.py:296-300,351-355, eval-mode
    dropout) with both LayerNorms writing the next GEMM's two-piece operand image directly (csrc/layernorm.hip) and the residual added in the
    second GEMM's epilogue: neither normalised tensor exists in fp32.  Backward: plain fp32 LayerNorm gradients, the ReLU gate applied while
    the [M, 2048] gradient is packed."""

    @staticmethod
    def takes(M: int, D: int, F_: int, f0: FrozenLinear, f4: FrozenLinear) -> bool:
        return bool(hip.GEMM_MODE == "bf16x6" and hip.GEMM_H2 and FFN_H2 and M >= 2048 and D % 32 == 0 and F_ % 32 == 0 and max(D, F_) <= 2048
                    and getattr(f0.W, "_tvl_frozen", False) and getattr(f4.W, "_tvl_frozen", False))

    @staticmethod
    def forward(ctx, x, g3, b3, f0: FrozenLinear, gf, bf, f4: FrozenLinear, eps):
        shape = x.shape
        x2d = _c(x).view(-1, shape[-1])
        xh, mean3, rstd3 = hip.layernorm_fwd_h2(x2d, g3, b3, eps)
        h, _ = hip.gemm_h2(xh, hip.weight_h2_cached(f0.W), bias=f0.b, act=hip.ACT_RELU)
        hh, meanf, rstdf = hip.layernorm_fwd_h2(h, gf, bf, eps)
        out, _ = hip.gemm_h2(hh, hip.weight_h2_cached(f4.W), bias=f4.b, residual=x2d)
        ctx.save_for_backward(x2d, g3, mean3, rstd3, h, gf, meanf, rstdf)
        ctx.mods, ctx.shape = (f0, f4), shape
        return out.view(shape)

    @staticmethod
    def backward(ctx, d_out):
        x2d, g3, mean3, rstd3, h, gf, meanf, rstdf = ctx.saved_tensors
        f0, f4 = ctx.mods
        d2d = _c(d_out).view(x2d.shape)
        dn = hip.linear_dgrad(d2d, f4.W, Wt=f4.Wt)                         # d LN_f output  [M, F]
        dh = hip.layernorm_bwd(dn, h, gf, meanf, rstdf)                    # d relu output
        dx3 = hip.linear_dgrad(dh, f0.W, Wt=f0.Wt, relu_mask=h)            # gate by (h > 0) while packing; d LN_3 output  [M, D]
        dx = hip.layernorm_bwd(dx3, x2d, g3, mean3, rstd3, dres=d2d)       # + the residual branch
        return dx.view(ctx.shape), None, None, None, None, None, None, None


FFN_H2 = __import__("os").environ.get("TVL_CRIS_FFN_H2", "1") != "0"   # A/B switch of the fused decoder feed-forward block


class CrossAttnFn(Fn):
    """T visual queries x Tk word keys with a key-padding mask (layers.py:341-349); also DenseCLIP's context decoder (K class queries x 1 + H*W
    visual keys, models.py:463-481), which takes the few-query kernels (csrc/attention_fq.hip) and computes dK / dV only when asked."""

    @staticmethod
    def forward(ctx, q, k, v, key_mask, B, T, Tk, H, dh):
        q, k, v = _c(q), _c(k), _c(v)
        ctx.geom = (B, T, Tk, H, dh)
        ctx.fq = hip.fq_attention_ok(T, Tk, dh, False, key_mask)
        if ctx.fq:
            o, P = hip.fq_attn_fwd(q, k, v, B, T, Tk, H, dh, dh**-0.5)
            ctx.save_for_backward(q, k, v, o, P, None)
            return o
        o, lse = hip.attn_fwd(q, k, v, B, T, Tk, H, dh, dh**-0.5, key_mask=key_mask)
        ctx.save_for_backward(q, k, v, o, lse, key_mask)
        return o

    @staticmethod
    def backward(ctx, d_o):
        q, k, v, o, lse, key_mask = ctx.saved_tensors
        B, T, Tk, H, dh = ctx.geom
        if ctx.fq:
            dq, dk, dv = hip.fq_attn_bwd(q, k, v, o, _c(d_o), lse, B, T, Tk, H, dh, dh**-0.5, need_dq=ctx.needs_input_grad[0],
                                         need_dkv=ctx.needs_input_grad[1] or ctx.needs_input_grad[2])
            return dq, dk, dv, None, None, None, None, None, None
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        hip.attn_bwd(q, k, v, o, _c(d_o), lse, dq, dk, dv, B, T, Tk, H, dh, dh**-0.5, key_mask=key_mask)
        return dq, dk, dv, None, None, None, None, None, None


class DynConvFn(Fn):
    """Projector tail: per-sample 3x3 kernel + bias from the text state (layers.py:106-118)."""

    @staticmethod
    def forward(ctx, x2d, word, B, H, W):
        x2d, word = _c(x2d), _c(word)
        ctx.save_for_backward(x2d, word)
        ctx.geom = (B, H, W)
        return hip.dynconv_fwd(x2d, word, B, H, W)

    @staticmethod
    def backward(ctx, dout):
        x2d, word = ctx.saved_tensors
        B, H, W = ctx.geom
        dx, dword = hip.dynconv_bwd(_c(dout), x2d, word, B, H, W, need_dx=ctx.needs_input_grad[0])
        return dx, dword, None, None, None


class BicubicFn(Fn):
    """F.interpolate(pred, img_size, mode="bicubic", align_corners=True) on [B,h,w] maps (coop_cris.py:235)."""

    @staticmethod
    def forward(ctx, pred, Ho, Wo):
        ctx.geom = pred.shape[1:]
        return hip.bicubic_ac_fwd(_c(pred), Ho, Wo)

    @staticmethod
    def backward(ctx, d):
        return hip.bicubic_ac_bwd(_c(d), *ctx.geom), None, None


class UpconvFn(Fn):
    """Upsample(x ps, bilinear) -> Conv2d(C->1, k, same, replicate) on a [B*G*G, C] map; the C x (G*ps)^2 tensor is never
    materialised (csrc/upconv.hip).  ``conv_w`` / ``conv_b`` are trainable (coop_cris.py:74-86)."""

    @staticmethod
    def forward(ctx, z, conv_w, conv_b, B, G, ps):
        z = _c(z)
        Cc, k = z.shape[1], conv_w.shape[-1]
        w2 = _c(conv_w).view(Cc, k * k)
        taps = torch.empty((B * G * G, k * k), device=z.device, dtype=torch.float32)
        hip.gemm(hip.NN, B * G * G, k * k, Cc, z, Cc, w2, k * k, taps, k * k)
        ctx.save_for_backward(z, w2)
        ctx.geom = (B, G, ps, k, Cc)
        return hip.upconv_taps_fwd(taps, conv_b, B, G, ps, k)

    @staticmethod
    def backward(ctx, d):
        z, w2 = ctx.saved_tensors
        B, G, ps, k, Cc = ctx.geom
        d = _c(d)
        M = B * G * G
        dtaps = hip.upconv_taps_bwd(d, B, G, ps, k)
        dz = dw = db = None
        if ctx.needs_input_grad[0]:
            dz = torch.empty((M, Cc), device=d.device, dtype=torch.float32)
            hip.gemm(hip.NT, M, Cc, k * k, dtaps, k * k, w2, k * k, dz, Cc)
        if ctx.needs_input_grad[1]:
            dw = torch.empty((Cc, k * k), device=d.device, dtype=torch.float32)
            hip.gemm(hip.TN, Cc, k * k, M, z, Cc, dtaps, k * k, dw, k * k)
            dw = dw.view(1, Cc, k, k)
        if ctx.needs_input_grad[2]:
            db = hip.dot(d)
        return dz, dw, db, None, None, None


class MixFn(Fn):
    """``(1 - r) * main + r * extra`` with a trainable scalar r (coop_cris.py:240-242); r is read on the device."""

    @staticmethod
    def forward(ctx, main, extra, ratio):
        main, extra = _c(main), _c(extra)
        ratio = ratio.detach() if isinstance(ratio, torch.Tensor) else hip.const_f32(float(ratio), main.device)
        ctx.save_for_backward(main, extra, ratio)
        return hip.mix(main, extra, ratio)

    @staticmethod
    def backward(ctx, d):
        main, extra, ratio = ctx.saved_tensors
        d = _c(d)
        dmain = hip.scale_dev(d, ratio, True) if ctx.needs_input_grad[0] else None
        dextra = hip.scale_dev(d, ratio, False) if ctx.needs_input_grad[1] else None
        dr = (hip.dot(d, extra) - hip.dot(d, main)).view(()) if ctx.needs_input_grad[2] else None
        return dmain, dextra, dr
