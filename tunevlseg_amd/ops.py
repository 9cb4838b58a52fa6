"""autograd nodes of the hot path; every forward/backward body is a sequence of HIP launches.

The frozen towers only ever need *data* gradients (SURVEY.md §8a "Backward requirements"),
so the two layer nodes below carry hand-written dgrad-only backwards and save exactly what
those need (layer input, LN statistics, packed QKV, attention output + LSE, MLP pre-activation).
Weight gradients exist only for the small trainable pieces (prompts, meta-nets, new last layer).
"""
from __future__ import annotations

from dataclasses import dataclass

import torch

from . import hip

Fn = torch.autograd.Function


@dataclass
class LayerWeights:
    """Frozen weights of one transformer layer, QKV packed as [3D, D] (built once per device)."""
    ln1_w: torch.Tensor
    ln1_b: torch.Tensor
    wqkv: torch.Tensor
    bqkv: torch.Tensor
    wo: torch.Tensor
    bo: torch.Tensor
    ln2_w: torch.Tensor
    ln2_b: torch.Tensor
    w1: torch.Tensor
    b1: torch.Tensor
    w2: torch.Tensor
    b2: torch.Tensor
    # W^T copies (frozen): data gradients become NT GEMMs for the split-bf16 kernel
    wqkv_t: torch.Tensor | None = None
    wo_t: torch.Tensor | None = None
    w1_t: torch.Tensor | None = None
    w2_t: torch.Tensor | None = None
    _tp3: dict | None = None
    _h2: dict | None = None
    _mlp64: object | None = None

    def mlp64(self):
        """Fragment-ordered images of (w1, w2) for the one-kernel feed-forward block of a 64-wide decoder layer (hip.Mlp64Weights), or
        None when the layer does not have that geometry."""
        if self._mlp64 is None:
            ok = hip.MLP64 and self.w1.shape[1] == 64 and tuple(self.w2.shape) == (64, self.w1.shape[0]) and self.w1.shape[0] % 128 == 0
            self._mlp64 = hip.Mlp64Weights(self.w1, self.b1, self.w2) if ok else False
        return self._mlp64 or None

    def tp3(self) -> dict:
        """The eight weight operands of the layer's tp3 GEMMs (forward: W as stored [N, K]; data gradients: W^T), packed once."""
        if self._tp3 is None:
            t = lambda w: w.detach().t().contiguous()  # noqa: E731
            self._tp3 = {k: hip.tp3_pack(w) for k, w in (
                ("wqkv", self.wqkv), ("wo", self.wo), ("w1", self.w1), ("w2", self.w2),
                ("wqkv_t", self.wqkv_t if self.wqkv_t is not None else t(self.wqkv)), ("wo_t", self.wo_t if self.wo_t is not None else t(self.wo)),
                ("w1_t", self.w1_t if self.w1_t is not None else t(self.w1)), ("w2_t", self.w2_t if self.w2_t is not None else t(self.w2)))}
            hip._built(self.wqkv)   # one-time images: complete before any other stream / thread (autograd's worker) can pick them up
        return self._tp3

    def h2(self) -> dict:
        """Two-piece fp16 images (per-tensor power-of-two scale) of the four weights whose GEMMs take a LayerNorm-produced A operand:
        QKV and fc1 forward, the out-projection and fc2 data gradients (W^T)."""
        if self._h2 is None:
            t = lambda w: w.detach().t().contiguous()  # noqa: E731
            self._h2 = {k: hip.weight_h2(w) for k, w in (
                ("wqkv", self.wqkv), ("wo", self.wo), ("w1", self.w1), ("w2", self.w2),
                ("wo_t", self.wo_t if self.wo_t is not None else t(self.wo)), ("w2_t", self.w2_t if self.w2_t is not None else t(self.w2)),
                ("w1_t", self.w1_t if self.w1_t is not None else t(self.w1)),
                ("wqkv_t", self.wqkv_t if self.wqkv_t is not None else t(self.wqkv)))}
            self._h2["b1_max"] = float(self.b1.detach().abs().max().item()) * 1.0001
            self._h2["bqkv_max"] = float(self.bqkv.detach().abs().max().item()) * 1.0001
        return self._h2


@dataclass
class AttnSpec:
    heads: int
    act: int
    eps: float
    causal: bool = False
    key_mask: torch.Tensor | None = None  # int32 [B, T]


def _c(t: torch.Tensor) -> torch.Tensor:
    return t if t.is_contiguous() else t.contiguous()


# ----------------------------------------------------------------------------------------------
# pre-LN encoder layer (HF CLIPSegEncoderLayer, modeling_clipseg.py:341-371)
# ----------------------------------------------------------------------------------------------
def _tp3_of(t2d: torch.Tensor):
    """The Tp3 image its producer attached to a gradient tensor (LayerNorm backward writes both forms), if still valid."""
    cached = getattr(t2d, "_tvl_tp3", None)
    if cached is not None and cached[0] == (t2d.data_ptr(), t2d._version, t2d.numel()):
        return cached[1]
    return None


QUICK_GELU_LIP = 1.125   # sup |d/dx x sigmoid(1.702 x)| = 1.0998 (at x = 1.51), rounded up: bound of the dz rows (tvl_gemm_h2_out)


class EncoderLayerTp3Fn(Fn):
    """The same layer on the tp3 kernels (hip.tp3_path_ok): every GEMM operand is handed over pre-split and pre-tiled by its
    producer -- LayerNorm and attention write Tp3 images directly, fc1's epilogue writes QuickGELU(z) as Tp3 (+ z in fp32 for
    the backward), the fc2 data gradient writes dz as Tp3 -- so no activation is split twice and no fp32 copy exists that
    only a GEMM would read."""

    @staticmethod
    def forward(ctx, h, lw: LayerWeights, spec: AttnSpec, grad_rows=None):
        h = _c(h)
        B, T, D = h.shape
        M = B * T
        H = spec.heads
        dh = D // H
        need = ctx.needs_input_grad[0]
        ctx.grad_rows = grad_rows
        W = lw.tp3()
        h2d = h.view(M, D)
        # the GEMMs whose A operand a LayerNorm kernel writes run on two fp16 pieces (3 MFMAs per product instead of 6): that producer has
        # the whole row, hence its exact power-of-two scale (hip.GEMM_H2, csrc/gemm_h2.hip)
        use_h2 = hip.GEMM_H2 and hip.ATTN_TP3
        ln_fwd = hip.layernorm_fwd_h2 if use_h2 else hip.layernorm_fwd_tp3
        gemm_ln = hip.gemm_h2 if use_h2 else hip.gemm_tp3
        WL = lw.h2() if use_h2 else W
        x1, mean1, rstd1 = ln_fwd(h2d, lw.ln1_w, lw.ln1_b, spec.eps, want_stats=need)
        attn_h2 = use_h2 and hip.ATTN_H2
        qkv_inv = None
        if attn_h2:   # Q | K | V as ONE-scale two-piece fp16 image: 3 MFMAs per attention product (csrc/attention_h2.hip)
            _, qkv_h = hip.gemm_h2(x1, WL["wqkv"], want_f32=False, want_h2=True, out_per_tensor=True, out_add=WL["bqkv_max"], bias=lw.bqkv)
            o, lse = hip.attn_h2_fwd(qkv_h, B, T, H, dh**-0.5, want_lse=need, o_as_h2=True)   # O shares the QKV scale: |O| <= max |V|
            qkv, qkv_inv = qkv_h.buf, qkv_h.inv_scale
        elif hip.ATTN_TP3:   # Q | K | V never exist in fp32: the attention kernels read the GEMM epilogue's tp3 image by LDS-DMA
            _, qkv_t = gemm_ln(x1, WL["wqkv"], want_f32=False, want_tp3=True, bias=lw.bqkv)
            o, lse = hip.attn_tp3_fwd(qkv_t, B, T, H, dh**-0.5, want_lse=need)
            qkv = qkv_t.buf
        else:
            qkv, _ = hip.gemm_tp3(x1, W["wqkv"], bias=lw.bqkv)
            o, lse = hip.attn_fwd_packed_tp3(qkv, B, T, H, dh, dh**-0.5, want_lse=need)
        del x1
        if isinstance(o, hip.H2):
            h2, _ = hip.gemm_h2(o, WL["wo"], bias=lw.bo, residual=h2d)
        else:
            h2, _ = hip.gemm_tp3(o, W["wo"], bias=lw.bo, residual=h2d)
        x2, mean2, rstd2 = ln_fwd(h2, lw.ln2_w, lw.ln2_b, spec.eps, want_stats=need)
        h2_mlp = use_h2 and spec.act == hip.ACT_QUICK_GELU
        # the MLP pre-activation is read by exactly one kernel, the QuickGELU' epilogue of the fc2 data gradient: on the h2 path it stays
        # in the GEMM's own accumulator order (hip.gemm_aux: both sides fully coalesced) instead of a row-major [M, F] matrix
        z = (hip.gemm_aux(M, lw.w1.shape[0], h.device) if h2_mlp else None) if need else None
        if need and z is None:
            z = torch.empty((M, lw.w1.shape[0]), device=h.device, dtype=torch.float32)
        out = torch.empty((B, T, D), device=h.device, dtype=torch.float32)  # a base tensor: deep prompts overwrite rows in place
        if h2_mlp:   # fc1's epilogue writes QuickGELU(z) as h2 too: |QuickGELU(z)| <= |z| <= ||x2 row|| max_n ||W1 row n|| + max |b1|
            _, a = hip.gemm_h2(x2, WL["w1"], want_f32=False, want_h2=True, out_add=WL["b1_max"], bias=lw.b1, act=spec.act, pre_out=z,
                               aux_blocked=z is not None and z.dim() == 1)
            hip.gemm_h2(a, WL["w2"], out=out.view(M, D), bias=lw.b2, residual=h2)
        else:
            _, a = gemm_ln(x2, WL["w1"], want_f32=False, want_tp3=True, bias=lw.b1, act=spec.act, pre_out=z)
            hip.gemm_tp3(a, W["w2"], out=out.view(M, D), bias=lw.b2, residual=h2)
        del x2
        if need:
            ctx.save_for_backward(h2d, mean1, rstd1, qkv, o.buf, lse, h2, mean2, rstd2, z, qkv_inv)
            ctx.lw, ctx.spec, ctx.shape = lw, spec, (B, T, D)
        return out

    @staticmethod
    def backward(ctx, dout):
        h2d, mean1, rstd1, qkv, o_buf, lse, h2, mean2, rstd2, z, qkv_inv = ctx.saved_tensors
        lw, spec = ctx.lw, ctx.spec
        B, T, D = ctx.shape
        M = B * T
        H = spec.heads
        dh = D // H
        W = lw.tp3()
        dout2d = _c(dout).view(M, D)
        use_h2 = hip.GEMM_H2 and hip.ATTN_TP3
        gemm_ln = hip.gemm_h2 if use_h2 else hip.gemm_tp3
        ln_bwd = hip.layernorm_bwd_h2 if use_h2 else hip.layernorm_bwd_tp3
        WL = lw.h2() if use_h2 else W
        dout_t = _tp3_of(dout)   # attached by the layer above's LayerNorm backward, unless autograd summed into it
        if dout_t is None or isinstance(dout_t, hip.H2) != use_h2:
            dout_t = hip.h2_pack(dout2d, per_row=True, want_norm=True) if use_h2 else hip.tp3_pack(dout2d)
        if use_h2 and spec.act == hip.ACT_QUICK_GELU:   # |dz| <= QUICK_GELU_LIP ||dout row|| max_n ||W2^T row n||
            _, dz = hip.gemm_h2(dout_t, WL["w2_t"], want_f32=False, want_h2=True, out_mul=QUICK_GELU_LIP * WL["w2_t"]._bound, dact=spec.act, dact_aux=z,
                                aux_blocked=z.dim() == 1)
            dx2, _ = hip.gemm_h2(dz, WL["w1_t"])
        else:
            _, dz = gemm_ln(dout_t, WL["w2_t"], want_f32=False, want_tp3=True, dact=spec.act, dact_aux=z)
            dx2, _ = hip.gemm_tp3(dz, W["w1_t"])
        del dz
        dh2, dh2_t = ln_bwd(dx2, h2, lw.ln2_w, mean2, rstd2, dres=dout2d)
        del dx2
        o_t = hip.H2.wrap(M, D, o_buf, qkv_inv, per_row=False) if qkv_inv is not None else hip.Tp3(M, D, o_buf.device, o_buf)
        rows = ctx.grad_rows
        if rows is not None and not (qkv_inv is not None and hip.DQKV_H2 and hip.GRAD_ROWS and rows[1] > 0 and rows[0] // 128 == (rows[0] + rows[1] - 1) // 128):
            rows = None
        if rows is not None:
            # Only rows [row0, row0 + n) of every sample of this layer's input gradient have a consumer (the first layer under the visual
            # prompts: patches, CLS and positions are frozen).  Those rows need dQ, dK, dV of the prompt tokens only -- one 128-row block of
            # the attention backward instead of four -- a [B*n, 3D] x [3D, D] data gradient instead of [M, 3D] x [3D, D], and n rows of
            # LayerNorm backward per sample.  Everything above (MLP, LayerNorm 2, dO) is still needed for all rows: every query's dO
            # reaches the prompt keys.  Same gradients, less dead work.
            row0, n = rows
            _, do = hip.gemm_h2(dh2_t, WL["wo_t"], want_f32=False, want_h2=True, out_per_tensor=True)
            dqkv = hip.attn_h2_bwd(hip.H2.wrap(M, 3 * D, qkv, qkv_inv, per_row=False), o_t, do, lse, B, T, H, dh**-0.5, out_h2=True, only_block=row0 // 128)
            del dh2_t, do
            g_rows = hip.h2k_gather_rows(dqkv, B, T, row0, n)                      # [B*n, 3D] fp32
            del dqkv
            dx1_rows = hip.linear_dgrad(g_rows, lw.wqkv, Wt=lw.wqkv_t)              # [B*n, D]
            ridx = hip.const_i64([b * T + row0 + j for b in range(B) for j in range(n)], h2d.device)
            dh_rows = hip.layernorm_bwd(dx1_rows, h2d.index_select(0, ridx), lw.ln1_w, mean1.index_select(0, ridx), rstd1.index_select(0, ridx),
                                        dres=dh2.index_select(0, ridx))
            dh_in = torch.zeros((M, D), device=h2d.device, dtype=torch.float32)
            dh_in.index_copy_(0, ridx, dh_rows)
            return dh_in.view(B, T, D), None, None, None
        if qkv_inv is not None:   # the forward ran the attention on two fp16 pieces: dO as a one-scale h2 image too
            _, do = hip.gemm_h2(dh2_t, WL["wo_t"], want_f32=False, want_h2=True, out_per_tensor=True)
            dqkv = hip.attn_h2_bwd(hip.H2.wrap(M, 3 * D, qkv, qkv_inv, per_row=False), o_t, do, lse, B, T, H, dh**-0.5, out_h2=hip.DQKV_H2)
        elif hip.ATTN_TP3:
            _, do = gemm_ln(dh2_t, WL["wo_t"], want_f32=False, want_tp3=True)
            dqkv = hip.attn_tp3_bwd(hip.Tp3(M, 3 * D, qkv.device, qkv), o_t, do, lse, B, T, H, dh**-0.5)
        else:
            do, _ = hip.gemm_tp3(dh2_t, W["wo_t"])
            dqkv = hip.attn_bwd_packed_tp3(qkv, o_t, do, lse, B, T, H, dh, dh**-0.5)
        del dh2_t, do
        if isinstance(dqkv, hip.H2K):   # exact power-of-two scale per (row, head, Q / K / V part): the GEMM rescales its accumulators at the chunk boundaries
            dx1 = hip.gemm_h2_ks(dqkv, WL["wqkv_t"])
        else:
            dx1, _ = hip.gemm_tp3(dqkv, W["wqkv_t"])
        del dqkv
        dh_in, dh_in_t = ln_bwd(dx1, h2d, lw.ln1_w, mean1, rstd1, dres=dh2)
        g = dh_in.view(B, T, D)
        g._tvl_tp3 = ((g.data_ptr(), g._version, g.numel()), dh_in_t)  # the layer below starts its backward with a tp3 GEMM on this
        g._tvl_owned = True   # a fresh tensor nobody else holds yet: its single consumer may edit it in place (RowsOverwriteFn.backward)
        return g, None, None, None


def encoder_layer(h, lw: LayerWeights, spec: AttnSpec, grad_rows=None):
    """One pre-LN encoder layer; picks the tp3 kernels when the shape qualifies.  ``grad_rows`` = (row0, n): the caller guarantees that
    only rows [row0, row0 + n) of every sample of ``h`` have a gradient consumer (see EncoderLayerTp3Fn.backward)."""
    B, T, D = h.shape
    if hip.tp3_path_ok(B * T, D, lw.w1.shape[0], D // spec.heads, spec.causal, spec.key_mask):
        return EncoderLayerTp3Fn.apply(h, lw, spec, grad_rows)
    return EncoderLayerFn.apply(h, lw, spec)


class EncoderLayerFn(Fn):
    @staticmethod
    def forward(ctx, h, lw: LayerWeights, spec: AttnSpec):
        h = _c(h)
        B, T, D = h.shape
        M = B * T
        H = spec.heads
        dh = D // H
        need = ctx.needs_input_grad[0]
        h2d = h.view(M, D)
        x1, mean1, rstd1 = hip.layernorm_fwd(h2d, lw.ln1_w, lw.ln1_b, spec.eps, want_stats=need)
        qkv = hip.linear_fwd(x1, lw.wqkv, lw.bqkv)
        del x1
        o, lse = hip.attn_fwd_packed(qkv, B, T, H, dh, dh**-0.5, spec.causal, spec.key_mask, want_lse=need)
        h2 = hip.linear_fwd(o, lw.wo, lw.bo, residual=h2d)
        x2, mean2, rstd2 = hip.layernorm_fwd(h2, lw.ln2_w, lw.ln2_b, spec.eps, want_stats=need)
        if need:
            a, z = hip.linear_fwd(x2, lw.w1, lw.b1, act=spec.act, want_pre=True)
        else:
            a, z = hip.linear_fwd(x2, lw.w1, lw.b1, act=spec.act), None
        del x2
        # the result must be a base tensor, not a view: deep prompts overwrite its rows in place afterwards
        out = torch.empty((B, T, D), device=h.device, dtype=torch.float32)
        hip.linear_fwd(a, lw.w2, lw.b2, residual=h2, out=out.view(M, D))
        if need:
            ctx.save_for_backward(h2d, mean1, rstd1, qkv, o, lse, h2, mean2, rstd2, z)
            ctx.lw, ctx.spec, ctx.shape = lw, spec, (B, T, D)
        return out

    @staticmethod
    def backward(ctx, dout):
        h2d, mean1, rstd1, qkv, o, lse, h2, mean2, rstd2, z = ctx.saved_tensors
        lw, spec = ctx.lw, ctx.spec
        B, T, D = ctx.shape
        H = spec.heads
        dh = D // H
        dout2d = _c(dout).view(B * T, D)
        dz = hip.linear_dgrad(dout2d, lw.w2, dact=spec.act, dact_aux=z, Wt=lw.w2_t)
        dx2 = hip.linear_dgrad(dz, lw.w1, Wt=lw.w1_t)
        del dz
        dh2 = hip.layernorm_bwd(dx2, h2, lw.ln2_w, mean2, rstd2, dres=dout2d)
        del dx2
        do = hip.linear_dgrad(dh2, lw.wo, Wt=lw.wo_t)
        dqkv = hip.attn_bwd_packed(qkv, o, do, lse, B, T, H, dh, dh**-0.5, spec.causal, spec.key_mask)
        del do
        dx1 = hip.linear_dgrad(dqkv, lw.wqkv, Wt=lw.wqkv_t)
        del dqkv
        dh_in = hip.layernorm_bwd(dx1, h2d, lw.ln1_w, mean1, rstd1, dres=dh2)
        return dh_in.view(B, T, D), None, None


# ----------------------------------------------------------------------------------------------
# post-LN decoder layer (HF CLIPSegDecoderLayer, modeling_clipseg.py:374-410)
# ----------------------------------------------------------------------------------------------
class DecoderLayerFn(Fn):
    @staticmethod
    def forward(ctx, x, lw: LayerWeights, spec: AttnSpec):
        x = _c(x)
        B, T, D = x.shape
        M = B * T
        H = spec.heads
        dh = D // H
        need = ctx.needs_input_grad[0]
        x2d = x.view(M, D)
        qkv = hip.linear_fwd(x2d, lw.wqkv, lw.bqkv)
        o, lse = hip.attn_fwd_packed(qkv, B, T, H, dh, dh**-0.5, False, None, want_lse=need)
        t1 = hip.linear_fwd(o, lw.wo, lw.bo, residual=x2d)
        x1, m1, r1 = hip.layernorm_fwd(t1, lw.ln1_w, lw.ln1_b, spec.eps, want_stats=need)
        fused = lw.mlp64() if (D == 64 and spec.act == hip.ACT_RELU) else None
        if fused is not None:   # the whole feed-forward block + LayerNorm2 in one launch; the backward recomputes the relu gate from x1
            out, t2, m2, r2 = hip.mlp64_fwd(x1, fused, lw.b1, lw.b2, lw.ln2_w, lw.ln2_b, spec.eps, want_stats=need)
            zu = x1
        else:
            if need:
                u, zu = hip.linear_fwd(x1, lw.w1, lw.b1, act=spec.act, want_pre=True)
            else:
                u, zu = hip.linear_fwd(x1, lw.w1, lw.b1, act=spec.act), None
            t2 = hip.linear_fwd(u, lw.w2, lw.b2, residual=x1)
            del u
            out, m2, r2 = hip.layernorm_fwd(t2, lw.ln2_w, lw.ln2_b, spec.eps, want_stats=need)
        if need:
            ctx.save_for_backward(qkv, o, lse, t1, m1, r1, zu, t2, m2, r2)
            ctx.lw, ctx.spec, ctx.shape, ctx.fused = lw, spec, (B, T, D), fused
        return out.view(B, T, D)

    @staticmethod
    def backward(ctx, dout):
        qkv, o, lse, t1, m1, r1, zu, t2, m2, r2 = ctx.saved_tensors
        lw, spec = ctx.lw, ctx.spec
        B, T, D = ctx.shape
        H = spec.heads
        dh = D // H
        dout2d = _c(dout).view(B * T, D)
        if ctx.fused is not None:   # zu holds x1, the block's input
            dx1 = hip.mlp64_bwd(dout2d, zu, t2, m2, r2, ctx.fused, lw.b1, lw.ln2_w)
        else:
            dt2 = hip.layernorm_bwd(dout2d, t2, lw.ln2_w, m2, r2)
            dzu = hip.linear_dgrad(dt2, lw.w2, dact=spec.act, dact_aux=zu, Wt=lw.w2_t)
            dx1 = hip.linear_dgrad(dzu, lw.w1, residual=dt2, Wt=lw.w1_t)
            del dzu, dt2
        dt1 = hip.layernorm_bwd(dx1, t1, lw.ln1_w, m1, r1)
        del dx1
        do = hip.linear_dgrad(dt1, lw.wo, Wt=lw.wo_t)
        dqkv = hip.attn_bwd_packed(qkv, o, do, lse, B, T, H, dh, dh**-0.5, False, None)
        dx = hip.linear_dgrad(dqkv, lw.wqkv, residual=dt1, Wt=lw.wqkv_t)
        return dx.view(B, T, D), None, None


# ----------------------------------------------------------------------------------------------
# generic small pieces
# ----------------------------------------------------------------------------------------------
class LinearFn(Fn):
    """y = act(x W^T + b) [+ residual]; data gradient always, weight/bias gradients when they are trainable."""

    @staticmethod
    def forward(ctx, x, W, b, act, residual, Wt=None):
        shape = x.shape
        x2d = _c(x).view(-1, shape[-1])
        W = _c(W)
        ctx.Wt = Wt   # W^T kept by the caller for a frozen W: the data gradient becomes an NT GEMM (split-bf16 kernel)
        need_any = any(ctx.needs_input_grad)
        if act != hip.ACT_NONE and need_any:
            y, pre = hip.linear_fwd(x2d, W, b, act=act, residual=None if residual is None else _c(residual).view(-1, W.shape[0]), want_pre=True)
        else:
            y, pre = hip.linear_fwd(x2d, W, b, act=act, residual=None if residual is None else _c(residual).view(-1, W.shape[0])), None
        ctx.act = act
        ctx.has_b = b is not None
        ctx.in_shape = shape
        ctx.save_for_backward(x2d if ctx.needs_input_grad[1] else None, W, pre)
        return y.view(*shape[:-1], W.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2d, W, pre = ctx.saved_tensors
        dy2d = _c(dy).view(-1, W.shape[0])
        if ctx.act != hip.ACT_NONE:
            dz = hip.dact_mul(dy2d, pre, ctx.act)  # tiny tensors only (meta-nets, FiLM inputs)
        else:
            dz = dy2d
        dx = dW = db = dres = None
        if ctx.needs_input_grad[0]:
            dx = hip.linear_dgrad(dz, W, Wt=ctx.Wt).view(ctx.in_shape)
        if ctx.needs_input_grad[1]:
            dW = hip.linear_wgrad(dz, x2d)
        if ctx.has_b and ctx.needs_input_grad[2]:
            db = hip.colsum(dz)
        if ctx.needs_input_grad[4]:
            dres = dy
        return dx, dW, db, None, dres, None


def linear(x, W, b=None, act=hip.ACT_NONE, residual=None, Wt=None):
    return LinearFn.apply(x, W, b, act, residual, Wt)


class LayerNormFn(Fn):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        shape = x.shape
        x2d = _c(x).view(-1, shape[-1])
        need = any(ctx.needs_input_grad)
        y, mean, rstd = hip.layernorm_fwd(x2d, gamma, beta, eps, want_stats=need)
        if need:
            ctx.save_for_backward(x2d, gamma, mean, rstd)
            ctx.has_beta = beta is not None
        return y.view(shape)

    @staticmethod
    def backward(ctx, dy):
        x2d, gamma, mean, rstd = ctx.saved_tensors
        dy2d = _c(dy).view(x2d.shape)
        dgamma = torch.zeros_like(gamma) if ctx.needs_input_grad[1] else None
        dbeta = torch.zeros_like(gamma) if (ctx.has_beta and ctx.needs_input_grad[2]) else None
        dx = hip.layernorm_bwd(dy2d, x2d, gamma, mean, rstd, dgamma=dgamma, dbeta=dbeta)
        return (dx.view(dy.shape) if ctx.needs_input_grad[0] else None), dgamma, dbeta, None


def layer_norm(x, gamma, beta, eps):
    return LayerNormFn.apply(x, gamma, beta, eps)


class L2NormFn(Fn):
    @staticmethod
    def forward(ctx, x):
        y, inv = hip.l2norm_fwd(_c(x))
        ctx.save_for_backward(y, inv)
        return y

    @staticmethod
    def backward(ctx, dy):
        y, inv = ctx.saved_tensors
        return hip.l2norm_bwd(_c(dy), y, inv)


class OuterAddFn(Fn):
    """out[b, j, :] = bias[b, :] + ctx[j, :]   (CoCoOp: cocoop_context_learner.py:50-58)."""

    @staticmethod
    def forward(ctx, bias, cvec):
        ctx.shapes = (bias.shape, cvec.shape)
        return hip.outer_add(_c(bias), _c(cvec))

    @staticmethod
    def backward(ctx, dout):
        dbias, dc = hip.outer_add_bwd(_c(dout))
        return dbias, dc


class VisionAssembleFn(Fn):
    """[CLS+pos0 | patches+pos | prompts] (HF:190-206 + vpt_context_learner.py:46-64); only the prompts carry grad."""

    @staticmethod
    def forward(ctx, patch, cls, pos, prompts):
        B = patch.shape[0]
        P, D = patch.shape[1], patch.shape[2]
        n = 0 if prompts is None else prompts.shape[-2]
        per_sample = prompts is not None and prompts.dim() == 3
        ctx.meta = (B, P, n, D, per_sample)
        return hip.vision_assemble(_c(patch).view(B * P, D), cls, pos, None if prompts is None else _c(prompts),
                                   n * D if per_sample else 0, B, P, n, D)

    @staticmethod
    def backward(ctx, dx0):
        B, P, n, D, per_sample = ctx.meta
        dprompts = None
        if n and ctx.needs_input_grad[3]:
            dprompts = torch.empty((B, n, D) if per_sample else (n, D), device=dx0.device, dtype=torch.float32)
            hip.rows_grad(_c(dx0), dprompts, 1 + P, n, not per_sample, False)
        return None, None, None, dprompts


class TextAssembleFn(Fn):
    """[BOS, ctx(n), words.., last] + pos  (coop_context_learner.py:136-181, coop_clipseg.py:40-73)."""

    @staticmethod
    def forward(ctx, ids, tmap, table, ctxv, pos, n_ctx):
        B = ids.shape[0]
        T = tmap.shape[0]
        D = table.shape[1]
        per_sample = ctxv is not None and ctxv.dim() == 3
        ctx.meta = (B, T, D, n_ctx, per_sample)
        return hip.text_assemble(ids, tmap, table, None if ctxv is None else _c(ctxv), n_ctx * D if per_sample else 0, pos, B, T, D)

    @staticmethod
    def backward(ctx, dx):
        B, T, D, n, per_sample = ctx.meta
        dctx = None
        if n and ctx.needs_input_grad[3]:
            dctx = torch.empty((B, n, D) if per_sample else (n, D), device=dx.device, dtype=torch.float32)
            hip.rows_grad(_c(dx), dctx, 1, n, not per_sample, False)
        return None, None, None, dctx, None, None


_EXCLUSIVE_RC: int | None = None


def _exclusive_refcount() -> int:
    """Reference count a gradient tensor shows inside a custom node's backward when the producing node's backward just returned it and nobody
    else holds it (measured once on a toy CPU graph of the same structure: it is a property of the autograd engine, not of the device).  A
    tensor hook that kept the gradient, or any other holder, shows up as a larger count in RowsOverwriteFn.backward."""
    global _EXCLUSIVE_RC
    if _EXCLUSIVE_RC is None:
        import sys

        seen = []

        class _Producer(Fn):
            @staticmethod
            def forward(ctx, x):
                return x * 2

            @staticmethod
            def backward(ctx, d):
                return d * 2

        class _Consumer(Fn):
            @staticmethod
            def forward(ctx, h):
                ctx.mark_dirty(h)
                return h

            @staticmethod
            def backward(ctx, dh):
                seen.append(sys.getrefcount(dh))
                return dh

        with torch.enable_grad():
            x = torch.ones(1, requires_grad=True)
            _Producer.apply(_Consumer.apply(x * 1)).sum().backward()
        _EXCLUSIVE_RC = seen[0]
    return _EXCLUSIVE_RC


class RowsOverwriteFn(Fn):
    """In-place ``h[:, row0:row0+n] = src`` (base_visual_learner.py:18-23, coop_context_learner.py:124-134).

    Backward hands the slot gradients to ``src`` and cuts the gradient of the values that were replaced.
    """

    @staticmethod
    def forward(ctx, h, src, row0):
        n = src.shape[-2]
        per_sample = src.dim() == 3
        hip.rows_overwrite(h, _c(src), n * h.shape[-1] if per_sample else 0, row0, n)
        ctx.meta = (row0, n, per_sample)
        ctx.mark_dirty(h)
        return h

    @staticmethod
    def backward(ctx, dh):
        row0, n, per_sample = ctx.meta
        B, T, D = dh.shape
        img = _tp3_of(dh) if dh.is_contiguous() else None
        # In place only when the producer marked the tensor as exclusively owned (EncoderLayerTp3Fn.backward) AND autograd hands this node the
        # very object the producer returned: a tensor hook on the layer output, gradient accumulation from a second consumer or retain_graph
        # replays give a different object (or one whose flag was consumed below), and those get the clone.  The flag is cleared on read.
        import sys

        owned = bool(getattr(dh, "_tvl_owned", False)) and not torch.is_grad_enabled() and sys.getrefcount(dh) <= _exclusive_refcount()
        if getattr(dh, "_tvl_owned", False):
            dh._tvl_owned = False
        if isinstance(img, hip.H2) and owned:
            # dh is the fresh tensor an encoder layer's backward just produced, with its operand image attached, and this node is its only
            # consumer: cut the rows in place -- in the fp32 gradient and in the image -- instead of cloning 49 MB and packing them again
            dh_in = dh
            hip.h2_zero_rows(img, B, T, row0, n)
        else:
            dh_in = dh.clone(memory_format=torch.contiguous_format)
        dsrc = torch.empty((B, n, D) if per_sample else (n, D), device=dh.device, dtype=torch.float32)
        hip.rows_grad(dh_in, dsrc, row0, n, not per_sample, True)
        return dh_in, (dsrc if ctx.needs_input_grad[1] else None), None


class GatherRowsFn(Fn):
    """EOS pooling ``x[arange(B), idx]`` (coop_clipseg.py:261-289)."""

    @staticmethod
    def forward(ctx, x, idx):
        ctx.save_for_backward(idx)
        ctx.shape = x.shape
        return hip.gather_rows(_c(x), idx)

    @staticmethod
    def backward(ctx, dout):
        (idx,) = ctx.saved_tensors
        dx = torch.zeros(ctx.shape, device=dout.device, dtype=torch.float32)
        hip.scatter_rows_add(_c(dout), idx, dx)
        return dx, None


class FilmFn(Fn):
    """``film_mul(c) * x + film_add(c)`` over tokens (base_clipseg.py:110-114)."""

    @staticmethod
    def forward(ctx, x, mul, add):
        x, mul, add = _c(x), _c(mul), _c(add)
        ctx.save_for_backward(x, mul)
        return hip.film_fwd(x, mul, add)

    @staticmethod
    def backward(ctx, dy):
        x, mul = ctx.saved_tensors
        want = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        dx, dmul, dadd = hip.film_bwd(_c(dy), x, mul, want)
        return (dx if ctx.needs_input_grad[0] else None), dmul, dadd


class SegHeadFn(Fn):
    """Strip CLS/prompt tokens -> ConvTranspose2d(C->1, k=s=ps) [+ new last layer] (base_clipseg.py:132-155,
    vpt_clipseg.py:287-302).  mix: 0 none, 1 ``logits += f(out)`` (VPT), 2 ``(1-r) logits + r f(out)`` (Base/MaPLe); the
    trainable ``r`` is read on the device (``tvl_mix`` / ``tvl_scale_dev``): no ``.item()``, no host synchronisation."""

    @staticmethod
    def forward(ctx, tokens, wt, bt, conv_w, conv_b, ratio, mix, G, ps):
        tokens = _c(tokens)
        B, T, Cc = tokens.shape
        Mrows = B * G * G
        amap = hip.RowMap(G * G, T, 1)
        tok2d = tokens.view(B * T, Cc)
        wt2d = _c(wt).view(Cc, ps * ps)
        cols = torch.empty((Mrows, ps * ps), device=tokens.device, dtype=torch.float32)
        hip.gemm(hip.NN, Mrows, ps * ps, Cc, tok2d, Cc, wt2d, ps * ps, cols, ps * ps, a_map=amap)
        extra = tconv = rdev = None
        k = 0
        if mix:
            k = conv_w.shape[-1]
            w2 = _c(conv_w).view(Cc, k * k)
            taps = torch.empty((Mrows, k * k), device=tokens.device, dtype=torch.float32)
            hip.gemm(hip.NN, Mrows, k * k, Cc, tok2d, Cc, w2, k * k, taps, k * k, a_map=amap)
            extra = hip.upconv_taps_fwd(taps, conv_b, B, G, ps, k)
        if mix == 2:
            rdev = ratio.detach().to(torch.float32) if isinstance(ratio, torch.Tensor) else hip.const_f32(float(ratio), tokens.device)
            tconv = hip.pixel_shuffle_fwd(cols, bt, None, 1.0, 0.0, B, G, ps)
            logits = hip.mix(tconv, extra, rdev)
        else:
            logits = hip.pixel_shuffle_fwd(cols, bt, extra, 1.0, 1.0 if mix else 0.0, B, G, ps)
        keep = mix == 2 and ctx.needs_input_grad[5]
        ctx.save_for_backward(tokens, wt2d, conv_w if mix else None, extra if keep else None, tconv if keep else None, rdev)
        ctx.meta = (B, T, Cc, G, ps, k, mix)
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        tokens, wt2d, conv_w, extra, tconv, rdev = ctx.saved_tensors
        B, T, Cc, G, ps, k, mix = ctx.meta
        dlogits = _c(dlogits)
        Mrows = B * G * G
        amap = hip.RowMap(G * G, T, 1)
        need = ctx.needs_input_grad
        dtok = dwt = dbt = dcw = dcb = dratio = None
        # gradients reaching the two branches: d_main for the transposed conv, d_extra for the new last layer
        d_main = hip.scale_dev(dlogits, rdev, True) if mix == 2 else dlogits
        d_extra = hip.scale_dev(dlogits, rdev, False) if mix == 2 else dlogits
        dcols = hip.pixel_unshuffle_bwd(d_main, 1.0, B, G, ps)
        if need[0]:
            dtok = torch.zeros((B * T, Cc), device=dlogits.device, dtype=torch.float32)
            hip.gemm(hip.NT, Mrows, Cc, ps * ps, dcols, ps * ps, wt2d, ps * ps, dtok, Cc, c_map=amap)
        if need[1]:
            dwt = torch.empty((Cc, ps * ps), device=dlogits.device, dtype=torch.float32)
            hip.gemm(hip.TN, Cc, ps * ps, Mrows, tokens.view(B * T, Cc), Cc, dcols, ps * ps, dwt, ps * ps, a_map=amap)
            dwt = dwt.view(Cc, 1, ps, ps)
        if need[2]:
            dbt = hip.dot(d_main)
        if mix:
            dtaps = hip.upconv_taps_bwd(d_extra, B, G, ps, k)
            w2 = _c(conv_w).view(Cc, k * k)
            if need[0]:
                hip.gemm(hip.NT, Mrows, Cc, k * k, dtaps, k * k, w2, k * k, dtok, Cc, residual=dtok, ldr=Cc, c_map=amap)
            if need[3]:
                dcw = torch.empty((Cc, k * k), device=dlogits.device, dtype=torch.float32)
                hip.gemm(hip.TN, Cc, k * k, Mrows, tokens.view(B * T, Cc), Cc, dtaps, k * k, dcw, k * k, a_map=amap)
                dcw = dcw.view(1, Cc, k, k)
            if need[4]:
                dcb = hip.dot(d_extra)
            if mix == 2 and need[5]:
                dratio = (hip.dot(dlogits, extra) - hip.dot(dlogits, tconv)).view(())
        if dtok is not None:
            dtok = dtok.view(B, T, Cc)
        return dtok, dwt, dbt, dcw, dcb, dratio, None, None, None


class DiceCELossFn(Fn):
    """monai DiceCELoss(sigmoid=True, lambda_dice, lambda_ce) for one channel; also yields the integer confusion counts."""

    @staticmethod
    def forward(ctx, logits, target, lambda_dice, lambda_ce, threshold):
        logits, target = _c(logits), _c(target)
        B = logits.shape[0]
        N = logits[0].numel()
        fsum, isum, _ = hip.dicece_stats(logits, target, threshold)
        ctx.save_for_backward(logits, target, fsum)
        ctx.lam = (lambda_dice, lambda_ce)
        loss = hip.dicece_loss(fsum, N, lambda_dice, lambda_ce, 1e-5, 1e-5)
        ctx.mark_non_differentiable(isum)
        return loss, isum

    @staticmethod
    def backward(ctx, dloss, _disum):
        logits, target, fsum = ctx.saved_tensors
        gs = _c(dloss.to(torch.float32)).view(1)
        dl = hip.dicece_bwd(logits, target, fsum, ctx.lam[0], ctx.lam[1], 1e-5, 1e-5, gs)
        return dl, None, None, None, None


class SpliceRowsFn(Fn):
    """out[b,t] = x[b, map[t]] or ctx[-map[t]-1]; the API-compatible ``learner.forward`` paths
    (vpt_context_learner.py:46-64, coop_context_learner.py:136-181).  The nets use the fused assemble kernels instead."""

    @staticmethod
    def forward(ctx, x, cvec, tmap_list):
        x, cvec = _c(x), _c(cvec)
        B, L, D = x.shape
        n = cvec.shape[-2]
        per_sample = cvec.dim() == 3
        tmap = hip.const_i32(tmap_list, x.device)
        ctx.meta = (B, L, D, n, per_sample, tuple(tmap_list))
        return hip.splice_rows(x, tmap, cvec, n * D if per_sample else 0)

    @staticmethod
    def backward(ctx, dout):
        B, L, D, n, per_sample, tmap_list = ctx.meta
        dout = _c(dout)
        dx = dc = None
        if ctx.needs_input_grad[0]:
            inv = [-1] * L
            for t, m in enumerate(tmap_list):
                if m >= 0:
                    inv[m] = t
            zero = torch.zeros((1, D), device=dout.device, dtype=torch.float32)
            dx = hip.splice_rows(dout, hip.const_i32(inv, dout.device), zero, 0)
        if ctx.needs_input_grad[1]:
            rows = [t for t, m in enumerate(tmap_list) if m < 0]
            if rows != list(range(rows[0], rows[0] + n)) or [tmap_list[t] for t in rows] != [-(j + 1) for j in range(n)]:
                raise NotImplementedError("context rows must form one contiguous ascending block")
            dc = torch.empty((B, n, D) if per_sample else (n, D), device=dout.device, dtype=torch.float32)
            hip.rows_grad(dout, dc, rows[0], n, not per_sample, False)
        return dx, dc, None


def splice_rows(x, cvec, tmap_list):
    return SpliceRowsFn.apply(x, cvec, list(tmap_list))


def concat_rows(x, cvec):
    L, n = x.shape[1], cvec.shape[-2]
    return SpliceRowsFn.apply(x, cvec, [*range(L), *[-(j + 1) for j in range(n)]])


_dropout_calls = 0


class DropoutFn(Fn):
    """nn.Dropout (training): counter-based mask, regenerated in backward from the same (seed, call index)."""

    @staticmethod
    def forward(ctx, x, p):
        global _dropout_calls
        _dropout_calls += 1
        ctx.p = p
        ctx.seed = (torch.initial_seed() * 0x9E3779B1 + _dropout_calls) & 0xFFFFFFFFFFFFFFFF
        return hip.dropout(_c(x), p, ctx.seed)

    @staticmethod
    def backward(ctx, dy):
        return hip.dropout(_c(dy), ctx.p, ctx.seed), None


def dropout(x, p: float, training: bool):
    if not training or p <= 0.0:
        return x
    return DropoutFn.apply(x, p)


class AddFn(Fn):
    """a + b (residual adds inside the SharedAttn learner's transformer layer)."""

    @staticmethod
    def forward(ctx, a, b):
        out = _c(b).clone()
        hip.axpby(_c(a), 1.0, out, 1.0)
        return out

    @staticmethod
    def backward(ctx, d):
        return d, d


def add(a, b):
    return AddFn.apply(a, b)


def split_cols(x, k: int):
    """Column split of a small [n, E] tensor -- pure data movement (autograd slice views), no arithmetic."""
    return x[:, :k], x[:, k:]
