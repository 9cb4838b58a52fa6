"""ctypes binding of ``libtvl_hip.so`` (the C ABI declared in ``include/tvl_hip.h``).

This is the only place the Python host touches the native library.  There is no
fallback: if the library is missing, or a tensor is not an fp32 contiguous device
tensor, the call raises.  PyTorch supplies device memory and the current HIP
stream; every wrapper passes raw device pointers + sizes + the stream handle.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from pathlib import Path

import torch

if os.environ.get("TVL_POISON") == "1":
    # Debug mode: every buffer this package asks torch for uninitialised comes back full of NaN patterns (fp32 NaN; bytes
    # 0x7E = fp16 / bf16 NaN pairs in the operand images), so a kernel that reads a location its producer did not write shows up in the
    # single-process parity suite instead of only when another process's leftovers happen to sit there.  `python -m pytest tests -m gpu` under it.
    _torch_empty, _torch_empty_like, _torch_empty_strided, _new_empty = torch.empty, torch.empty_like, torch.empty_strided, torch.Tensor.new_empty

    def _poisoned(t):
        if t.is_cuda:
            if t.dtype in (torch.float32, torch.float64, torch.float16, torch.bfloat16):
                t.fill_(float("nan"))
            elif t.dtype == torch.uint8:
                t.fill_(0x7E)
            elif t.dtype in (torch.int32, torch.int64, torch.int16, torch.int8):   # index / count / flag buffers: a huge negative value (a wild index faults, a count is absurd)
                t.fill_(torch.iinfo(t.dtype).min + 7)
        return t

    torch.empty = lambda *a, **k: _poisoned(_torch_empty(*a, **k))
    torch.empty_like = lambda *a, **k: _poisoned(_torch_empty_like(*a, **k))
    torch.empty_strided = lambda *a, **k: _poisoned(_torch_empty_strided(*a, **k))
    torch.Tensor.new_empty = lambda self, *a, **k: _poisoned(_new_empty(self, *a, **k))

_HERE = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ["TVL_HIP_LIB"]) if os.environ.get("TVL_HIP_LIB") else _HERE / "csrc" / "libtvl_hip.so"   # (override: a `make DIAG=1` build for the tools)

NT, NN, TN = 0, 1, 2
ACT_NONE, ACT_QUICK_GELU, ACT_RELU, ACT_SIGMOID, ACT_GELU = 0, 1, 2, 3, 4
ACT_POST_RESIDUAL = 0x100  # OR into act: activation after the residual add
ACT_IDS = {None: ACT_NONE, "none": ACT_NONE, "quick_gelu": ACT_QUICK_GELU, "relu": ACT_RELU, "sigmoid": ACT_SIGMOID, "gelu": ACT_GELU}


class RowMap(C.Structure):
    _fields_ = [("div", C.c_int32), ("mul", C.c_int32), ("off", C.c_int32)]


class GemmArgs(C.Structure):
    _fields_ = [
        ("layout", C.c_int32), ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("A", C.c_void_p), ("lda", C.c_int32),
        ("B", C.c_void_p), ("ldb", C.c_int32),
        ("C", C.c_void_p), ("ldc", C.c_int32),
        ("bias", C.c_void_p),
        ("residual", C.c_void_p), ("ldr", C.c_int32),
        ("act", C.c_int32),
        ("pre_out", C.c_void_p),
        ("dact_aux", C.c_void_p), ("ld_aux", C.c_int32), ("dact", C.c_int32),
        ("alpha", C.c_float),
        ("a_map", RowMap), ("c_map", RowMap),
    ]


class GemmTp3Args(C.Structure):
    _fields_ = [
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("A", C.c_void_p), ("a_rows", C.c_int64),
        ("B", C.c_void_p), ("b_rows", C.c_int64),
        ("C", C.c_void_p), ("ldc", C.c_int32),
        ("C_tp3", C.c_void_p),
        ("bias", C.c_void_p),
        ("residual", C.c_void_p), ("ldr", C.c_int32),
        ("act", C.c_int32),
        ("pre_out", C.c_void_p),
        ("dact_aux", C.c_void_p), ("ld_aux", C.c_int32), ("dact", C.c_int32),
        ("alpha", C.c_float),
        ("tile_m", C.c_int32), ("variant", C.c_int32),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64),
        ("aux_blocked", C.c_int32), ("a_scale_one", C.c_int32),
    ]


class ConvGeom(C.Structure):
    _fields_ = [("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("C", C.c_int32), ("stride", C.c_int32)]


class AttnFwdArgs(C.Structure):
    _fields_ = [
        ("q", C.c_void_p), ("k", C.c_void_p), ("v", C.c_void_p),
        ("q_bs", C.c_int64), ("k_bs", C.c_int64), ("v_bs", C.c_int64),
        ("q_ts", C.c_int32), ("k_ts", C.c_int32), ("v_ts", C.c_int32),
        ("o", C.c_void_p), ("ldo", C.c_int32),
        ("lse", C.c_void_p),
        ("key_mask", C.c_void_p),
        ("B", C.c_int32), ("H", C.c_int32), ("T", C.c_int32), ("dh", C.c_int32), ("causal", C.c_int32),
        ("scale", C.c_float), ("Tk", C.c_int32),
    ]


class AttnBwdArgs(C.Structure):
    _fields_ = [
        ("q", C.c_void_p), ("k", C.c_void_p), ("v", C.c_void_p),
        ("q_bs", C.c_int64), ("k_bs", C.c_int64), ("v_bs", C.c_int64),
        ("q_ts", C.c_int32), ("k_ts", C.c_int32), ("v_ts", C.c_int32),
        ("o", C.c_void_p), ("d_o", C.c_void_p), ("ldo", C.c_int32),
        ("lse", C.c_void_p),
        ("delta", C.c_void_p),
        ("dq", C.c_void_p), ("dk", C.c_void_p), ("dv", C.c_void_p),
        ("dq_bs", C.c_int64), ("dk_bs", C.c_int64), ("dv_bs", C.c_int64),
        ("dq_ts", C.c_int32), ("dk_ts", C.c_int32), ("dv_ts", C.c_int32),
        ("key_mask", C.c_void_p),
        ("B", C.c_int32), ("H", C.c_int32), ("T", C.c_int32), ("dh", C.c_int32), ("causal", C.c_int32),
        ("scale", C.c_float), ("Tk", C.c_int32),
    ]


_P, _I, _L, _F = C.c_void_p, C.c_int32, C.c_int64, C.c_float
# name -> argtypes (stream appended automatically); mirrors include/tvl_hip.h one to one
_SIGS = {
    "tvl_gemm_f32": [C.POINTER(GemmArgs)],
    "tvl_gemm_bf16s": [C.POINTER(GemmArgs), _I],
    "tvl_gemm_bf16s_splitk": [C.POINTER(GemmArgs), _I, _P, _L],
    "tvl_layernorm_fwd": [_P, _P, _P, _P, _P, _P, _L, _I, _F],
    "tvl_layernorm_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I],
    "tvl_attn_fwd": [C.POINTER(AttnFwdArgs)],
    "tvl_attn_bwd": [C.POINTER(AttnBwdArgs)],
    "tvl_im2col_patch": [_P, _P, _I, _I, _I, _I, _I],
    "tvl_vision_assemble": [_P, _P, _P, _P, _L, _P, _I, _I, _I, _I],
    "tvl_text_assemble": [_P, _I, _P, _P, _L, _P, _L, _P, _P, _I, _I, _I],
    "tvl_splice_rows": [_P, _I, _P, _P, _L, _P, _I, _I, _I],
    "tvl_rows_overwrite": [_P, _P, _L, _I, _I, _I, _I, _I],
    "tvl_rows_grad": [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I],
    "tvl_gather_rows": [_P, _P, _P, _I, _I, _I],
    "tvl_scatter_rows_add": [_P, _P, _P, _I, _I, _I],
    "tvl_film_fwd": [_P, _P, _P, _P, _I, _I, _I],
    "tvl_film_bwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I],
    "tvl_pixel_shuffle_fwd": [_P, _P, _P, _F, _F, _P, _I, _I, _I],
    "tvl_pixel_unshuffle_bwd": [_P, _F, _P, _I, _I, _I],
    "tvl_upconv_taps_fwd": [_P, _I, _P, _P, _P, _I, _I, _I, _I],
    "tvl_upconv_taps_bwd": [_P, _P, _I, _P, _I, _I, _I, _I],
    "tvl_dicece_stats": [_P, _P, _P, _P, _P, _P, _I, _L, _F],
    "tvl_dicece_bwd": [_P, _P, _P, _P, _I, _L, _F, _F, _F, _F, _P],
    "tvl_dicece_loss": [_P, _P, _I, _L, _F, _F, _F, _F, _P],
    "tvl_mlp64_pack": [_P, _P, _I, _F, _F, _P],
    "tvl_mlp64_fwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _F, _F, _F, _F, _F],
    "tvl_mlp64_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _F, _F, _F],
    "tvl_normalize_u8": [_P, _P, _I, _I, _I, C.POINTER(C.c_float), C.POINTER(C.c_float)],
    "tvl_mask_u8": [_P, _P, _L],
    "tvl_resize_u8": [_P, _P, _P, _I, _I, _I, _I, _I, _P],
    "tvl_augment_u8": [_P, _P, _P, _P, C.POINTER(C.c_float), C.POINTER(C.c_float), _P, _P, _I, _I, _I],
    "tvl_mix": [_P, _P, _P, _P, _L],
    "tvl_scale_dev": [_P, _P, _I, _P, _L],
    "tvl_adamw": [_P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _I, _F, _P],
    "tvl_fill": [_P, _F, _L],
    "tvl_axpby": [_P, _F, _P, _F, _L],
    "tvl_bias_act": [_P, _P, _P, _L, _I, _I],
    "tvl_dropout": [_P, _P, _L, _F, C.c_uint64],
    "tvl_dact_mul": [_P, _P, _P, _L, _I],
    "tvl_outer_add": [_P, _P, _P, _I, _I, _I],
    "tvl_outer_add_bwd": [_P, _P, _P, _I, _I, _I],
    "tvl_l2norm_fwd": [_P, _P, _P, _I, _I],
    "tvl_l2norm_bwd": [_P, _P, _P, _P, _I, _I],
    "tvl_dot": [_P, _P, _P, _L, _I],
    "tvl_colsum": [_P, _P, _L, _I, _I],
    "tvl_copy2d": [_P, _I, _P, _I, _L, _I],
    "tvl_conv3x3_bf16s": [C.POINTER(GemmArgs), C.POINTER(ConvGeom), _I],
    "tvl_layernorm_fwd_tp3": [_P, _P, _P, _P, _P, _P, _L, _I, _F],
    "tvl_layernorm_bwd_tp3": [_P, _P, _P, _P, _P, _P, _P, _P, _L, _I],
    "tvl_attn_fwd_tp3": [C.POINTER(AttnFwdArgs), _P],
    "tvl_attn_bwd_tp3": [C.POINTER(AttnBwdArgs), _P, _P],
    "tvl_attn_tp3_fwd": [_P, _P, _P, _I, _I, _I, _F],
    "tvl_attn_tp3_fwd_diag": [_P, _P, _P, _I, _I, _I, _F, _I, _P],
    "tvl_attn_tp3_bwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _F],
    "tvl_tp3_pack": [_P, _L, _L, _I, _P],
    "tvl_tp3_unpack": [_P, _L, _I, _P, _L],
    "tvl_gemm_tp3": [C.POINTER(GemmTp3Args)],
    "tvl_h2_pack": [_P, _L, _L, _I, _P, _P, _P, _I, _P],
    "tvl_h2_pack_masked": [_P, _L, _P, _L, _L, _I, _P, _P, _P, _I, _P],
    "tvl_gemm_h2_out": [C.POINTER(GemmTp3Args), _P, _P, _P, _F, _F, _P, _I],
    "tvl_attn_h2_fwd": [_P, _P, _P, _I, _P, _I, _I, _I, _F],
    "tvl_attn_h2_bwd": [_P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _I, _P, _I, _I, _I, _F, _I],
    "tvl_h2k_gather_rows": [_P, _P, _I, _I, _I, _I, _I, _P],
    "tvl_gemm_h2_ks": [C.POINTER(GemmTp3Args), _P],
    "tvl_conv3x3_h2": [C.POINTER(GemmTp3Args), C.POINTER(ConvGeom), _P],
    "tvl_gemm_h2": [C.POINTER(GemmTp3Args), _P],
    "tvl_layernorm_fwd_h2": [_P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _F, _P, C.c_uint32],
    "tvl_layernorm_bwd_h2": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _P, C.c_uint32],
    "tvl_im2col3x3": [_P, _L, _L, _L, _L, _P, _I, _I, _I, _I, _I, _I],
    "tvl_avgpool_fwd": [_P, _I, _P, _I, _I, _I, _I, _I, _I],
    "tvl_avgpool_bwd": [_P, _I, _P, _I, _I, _I, _I, _I, _I],
    "tvl_bilinear_up_fwd": [_P, _I, _P, _I, _I, _I, _I, _I, _I],
    "tvl_bilinear_up_h2": [_P, _I, _P, _P, _P, _I, _I, _I, _I, _I],
    "tvl_h2_absmax": [_P, _L, _L, _I, _P],
    "tvl_h2_zero_rows": [_P, _I, _I, _I, _I, _I],
    "tvl_bilinear_up_bwd": [_P, _I, _P, _I, _I, _I, _I, _I, _I],
    "tvl_bicubic_ac_fwd": [_P, _P, _P, _F, _F, _I, _I, _I, _I, _I],
    "tvl_bicubic_ac_bwd": [_P, _F, _P, _I, _I, _I, _I, _I],
    "tvl_bicubic_resize_u8": [_P, _P, _I, _I, _I, _I],
    "tvl_dynconv_fwd": [_P, _I, _P, _I, _P, _P, _I, _I, _I, _I],
    "tvl_dynconv_bwd": [_P, _P, _I, _P, _I, _P, _I, _P, _P, _I, _I, _I, _I],
    "tvl_groupnorm_stats": [_P, _L, _I, _I, _I, _I, _F, _P, _P],
    "tvl_groupnorm_apply": [_P, _L, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I],
    "tvl_tconv2x2_unshuffle": [_P, _P, _I, _I, _I, _I, _I, _I],
    "tvl_colscale_add": [_P, _P, _P, _P, _L, _I],
    "tvl_colscale_bwd": [_P, _P, _P, _P, _P, _L, _I],
    "tvl_blockdiag_gather": [_P, _I, _P, _I, _I, _I, _I, _I],
    "tvl_fq_qk": [_P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _F],
    "tvl_fq_softmax": [_P, _P, _L, _I],
    "tvl_fq_pk": [_P, _L, _L, _L, _L, _P, _L, _I, _P, _I, _I, _I, _I, _I, _I, _F],
    "tvl_fq_ds": [_P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _I],
    "tvl_fq_tk": [_P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _F],
}
EXPORTS = ["tvl_last_error", "tvl_abi_version", "tvl_build_flags", "tvl_dynconv_bwd_work_floats", "tvl_tp3_bytes", "tvl_h2_bytes", "tvl_dicece_work_doubles",
           "tvl_gemm_aux_floats", "tvl_mlp64_image_bytes", "tvl_groupnorm_work_doubles", *_SIGS]

_lib = None
ABI_VERSION = 6   # include/tvl_hip.h TVL_ABI_VERSION


def load():
    """dlopen the in-tree library (torch must be imported first so both share one HIP runtime)."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise RuntimeError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `make -C {LIB_PATH.parent}` "
            "or `python -c 'import __graft_entry__ as g; g.build()'`. There is no CPU fallback."
        )
    lib = C.CDLL(str(LIB_PATH))
    lib.tvl_last_error.restype = C.c_char_p
    lib.tvl_abi_version.restype = C.c_int
    lib.tvl_build_flags.restype = C.c_int
    if lib.tvl_abi_version() != ABI_VERSION:
        raise RuntimeError(f"{LIB_PATH} has C-ABI version {lib.tvl_abi_version()}, this package binds version {ABI_VERSION}: rebuild it "
                           f"(`make -C {LIB_PATH.parent}`)")
    lib.tvl_dynconv_bwd_work_floats.argtypes = [_I, _I, _I, _I]
    lib.tvl_dynconv_bwd_work_floats.restype = C.c_int64
    lib.tvl_mlp64_image_bytes.argtypes = [_I]
    lib.tvl_mlp64_image_bytes.restype = C.c_int64
    lib.tvl_dicece_work_doubles.argtypes = [_I, _L]
    lib.tvl_dicece_work_doubles.restype = C.c_int64
    lib.tvl_gemm_aux_floats.argtypes = [_L, _L]
    lib.tvl_gemm_aux_floats.restype = C.c_int64
    lib.tvl_groupnorm_work_doubles.argtypes = [_I, _I]
    lib.tvl_groupnorm_work_doubles.restype = C.c_int64
    lib.tvl_tp3_bytes.argtypes = [_L, _I]
    lib.tvl_tp3_bytes.restype = C.c_int64
    for name, sig in _SIGS.items():
        fn = getattr(lib, name)
        fn.argtypes = [*sig, C.c_void_p]
        fn.restype = C.c_int
    _lib = lib
    return lib


_CONST_I32: dict = {}


def _built(t: torch.Tensor | None = None, device=None) -> None:
    """End of a ONE-TIME construction on the device (constant tables, packed weight images, slot pools, ``prepared()`` trees): wait until the
    stream that built it has finished.  Such objects are built lazily, on whatever stream first asks -- and consumed, from the next line on,
    by any stream or thread of the process (the text tower's side stream, autograd's worker, a capture stream): only the builder's stream is
    ordered behind the construction, so the construction is made complete before the object is handed out.  Costs one stream
    synchronisation per object per process, all in the first step; steady-state steps never get here."""
    dev = t.device if t is not None else device
    if dev is not None and torch.device(dev).type == "cuda" and not torch.cuda.is_current_stream_capturing():
        torch.cuda.current_stream(dev).synchronize()


def const_i32(values, device) -> torch.Tensor:
    """Device int32 tensor of a small host-side index list (token maps, row maps), built once per (values, device): the per-step
    ``torch.tensor(list, device=...)`` it replaces is a pageable host-to-device copy, which also forbids HIP-graph capture."""
    key = (tuple(int(v) for v in values), str(device))
    t = _CONST_I32.get(key)
    if t is None:
        if len(_CONST_I32) > 4096:
            _CONST_I32.clear()
        t = _CONST_I32[key] = torch.tensor(key[0], dtype=torch.int32, device=device)
        _built(t)
    return t


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)   # the current stream's handle without building a Stream object


_CONST_F32: dict = {}


def const_f32(value: float, device) -> torch.Tensor:
    """One-element device fp32 tensor of a host scalar, built once per (value, device) -- see const_i32."""
    key = (float(value), str(device))
    t = _CONST_F32.get(key)
    if t is None:
        if len(_CONST_F32) > 4096:
            _CONST_F32.clear()
        t = _CONST_F32[key] = torch.tensor(key[0], dtype=torch.float32, device=device)
        _built(t)
    return t


_NONFINITE: dict = {}


def nonfinite_flags(device) -> torch.Tensor:
    """int32 [2] on the device: [0] counts the losses that came out NaN / Inf (``tvl_dicece_loss``), [1] is set when a NaN / Inf gradient reached
    the optimiser update (``tvl_adamw``).  Written by the kernels themselves -- nothing in a step waits for them; :func:`check_finite` reads
    them where the host synchronises anyway (end of an epoch, end of the benchmark's timed region)."""
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    t = _NONFINITE.get(key)
    if t is None:
        t = _NONFINITE[key] = torch.zeros(2, device=torch.device("cuda", key), dtype=torch.int32)
        torch.cuda.current_stream(t.device).synchronize()
    return t


def check_finite(device=None, what: str = "training") -> None:
    """Raise ``FloatingPointError`` if any step since the last call produced a non-finite loss or gradient (reads and clears the device flags)."""
    for key, t in list(_NONFINITE.items()):
        if device is not None and torch.device(device).index not in (None, key):
            continue
        bad_loss, bad_grad = (int(v) for v in t.tolist())
        if bad_loss or bad_grad:
            t.zero_()
            raise FloatingPointError(f"{what}: {bad_loss} step(s) with a non-finite loss" + (", non-finite gradients reached the optimiser" if bad_grad else "")
                                     + f" on cuda:{key}")


def _stream():
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _call(name: str, *args):
    lib = _lib if _lib is not None else load()
    rc = getattr(lib, name)(*args, _stream())
    if rc != 0:
        raise RuntimeError(f"{name} failed (rc={rc}): {lib.tvl_last_error().decode()}")


def _p(t: torch.Tensor | None, dtype=torch.float32):
    if t is None:
        return None
    if not t.is_cuda or t.dtype != dtype or not t.is_contiguous():
        raise RuntimeError(f"HIP op needs a contiguous {dtype} device tensor, got {t.dtype} {t.device} contiguous={t.is_contiguous()}")
    return t.data_ptr()


def _ps(t: torch.Tensor | None):
    """Pointer of a 2-D fp32 device matrix whose rows may be strided (a column slice of a wider matrix: channel concat)."""
    if t is None:
        return None
    if not t.is_cuda or t.dtype != torch.float32 or t.dim() != 2 or (t.shape[1] > 1 and t.stride(1) != 1):
        raise RuntimeError(f"HIP op needs a 2-D fp32 device matrix with unit column stride, got {t.dtype} {t.device} {tuple(t.shape)} {t.stride()}")
    return t.data_ptr()


def _ident():
    return RowMap(0, 0, 0)


# --------------------------------------------------------------------------------------
# GEMM
# --------------------------------------------------------------------------------------
# Arithmetic of the large NT GEMMs (fp32 in, fp32 out either way):
#   "bf16x6" (default) bf16 MFMA, every fp32 operand split exactly into 3 bf16 pieces inside the kernel, 6 MFMAs per
#            k-step, fp32 accumulate: fp32-equivalent results (measured 3e-6 on full-size logits) at 1.4x the f32-MFMA speed
#   "f32"    v_mfma_f32_32x32x2_f32, exact fp32
#   "bf16x3" 2 pieces / 3 MFMAs (~2^-16 per product; fails the 1e-3 gradient gate on the full-size fixtures)
#   "bf16"   plain bf16 operands (fails the 1e-3 logit gate) -- both kept only as measured, documented reduced-precision modes
GEMM_MODE = os.environ.get("TVL_GEMM_MODE", "bf16x6")
SPLITK = os.environ.get("TVL_GEMM_SPLITK", "1") != "0"  # deterministic split-K for skinny, deep GEMMs
_NSPLIT = {"bf16x6": 3, "bf16x3": 2, "bf16": 1}


def set_gemm_mode(mode: str) -> None:
    global GEMM_MODE
    if mode != "f32" and mode not in _NSPLIT:
        raise ValueError(f"unknown GEMM mode {mode!r}")
    GEMM_MODE = mode


_gemm_prof: list | None = None  # bench.py: (kernel key, flops, start event, stop event) per launch
_aux_prof: list | None = None   # the same for the non-GEMM kernels of a vision layer: (key, flops, algorithmic bytes, e0, e1)


class _aux_span:
    """HIP-event bracket (torch's current stream = the launch stream) around one non-GEMM entry point while bench.py profiles."""

    __slots__ = ("key", "flops", "nbytes", "e0")

    def __init__(self, key: str, flops: float, nbytes: float):
        self.key, self.flops, self.nbytes, self.e0 = key, flops, nbytes, None

    def __enter__(self):
        if _aux_prof is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()

    def __exit__(self, *exc):
        if self.e0 is not None and _aux_prof is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            _aux_prof.append((self.key, self.flops, self.nbytes, self.e0, e1))
        return False


def csrc_sha() -> str:
    """sha256 over the kernel sources (csrc/*.hip, *.h, Makefile, include/tvl_hip.h): the identity of the code a recorded profile belongs to.
    (The GPU box gets a snapshot without .git, so a commit hash cannot be checked there; identical sources can.)"""
    import hashlib

    h = hashlib.sha256()
    files = sorted([*(_HERE / "csrc").glob("*.hip"), *(_HERE / "csrc").glob("*.h"), _HERE / "csrc" / "Makefile", _HERE.parent / "include" / "tvl_hip.h"])
    for f in files:
        h.update(f.name.encode())
        h.update(f.read_bytes())
    return h.hexdigest()[:16]


def gemm_profile_start():
    global _gemm_prof, _aux_prof
    _gemm_prof, _aux_prof = [], []


def gemm_profile_stop() -> dict:
    """Per kernel instantiation: launches, algorithmic FLOPs (2*M*N*K) and summed device time (ms, HIP events)."""
    global _gemm_prof, _aux_prof, last_aux_profile
    rec, _gemm_prof = _gemm_prof or [], None
    aux, _aux_prof = _aux_prof or [], None
    # what an event pair reads around a kernel that does (almost) nothing -- the markers' own cost, ~20-30 us on this stack, which would
    # otherwise be booked on every launch and turn a 27 us kernel into a 60 us one: median over 32 one-element fills, taken off each launch
    pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(32)] if rec else []
    dummy = torch.zeros(1, device="cuda") if rec else None
    for e0, e1 in pairs:
        e0.record()
        dummy.fill_(1.0)
        e1.record()
    torch.cuda.synchronize()
    empty = max(sorted(e0.elapsed_time(e1) for e0, e1 in pairs)[len(pairs) // 2] - 0.003, 0.0) if pairs else 0.0
    out: dict[str, dict] = {}
    for key, flops, e0, e1, *more in rec:
        d = out.setdefault(key, {"launches": 0, "flops": 0.0, "ms": 0.0, "bytes": 0.0})
        d["launches"] += 1
        d["flops"] += flops
        d["bytes"] += more[0] if more else 0.0   # algorithmic HBM bytes of the launch (operands once + every output once), where the caller knows them
        t = e0.elapsed_time(e1)
        d["ms"] += max(t - empty, 0.25 * t)
        d["raw_ms"] = d.get("raw_ms", 0.0) + t
    last_aux_profile = {}
    for key, flops, nbytes, e0, e1 in aux:
        d = last_aux_profile.setdefault(key, {"launches": 0, "flops": 0.0, "bytes": 0.0, "ms": 0.0, "raw_ms": 0.0})
        t = e0.elapsed_time(e1)
        d["launches"] += 1
        d["flops"] += flops
        d["bytes"] += nbytes
        d["ms"] += max(t - empty, 0.25 * t)
        d["raw_ms"] += t
    global last_empty_pair_ms
    last_empty_pair_ms = empty
    return out


last_empty_pair_ms = 0.0
last_aux_profile: dict = {}   # filled by gemm_profile_stop(): attention / LayerNorm spans of the profiled steps


def _bf16s_tile(M: int, N: int, K: int) -> tuple[int, int, int]:
    """Tile the split-bf16 GEMM picks (same rule as csrc/gemm_bf16s_kernel.h choose_bm)."""
    use192 = int(os.environ.get("TVL_GEMM_TILE192", "1"))
    use256 = int(os.environ.get("TVL_GEMM_TILE256", "1"))
    cands = ((192, 256, 1, 2, 0.90), (192, 128, 2, 2, 0.93), (128, 128, 2, 2, 1.0), (96, 128, 2, 1, 1.0), (64, 64, 4, 2, 1.12))
    best, tile, raw = None, (64, 64, 2), {}
    for bm, bn, per_cu, wgm, w in cands:
        if bn == 256 and (not use256 or not use192 or N % 256 != 0 or (K < 1536 and N < 3072)):
            continue
        if (bm == 192 and not use192) or (bm == 96 and use192):
            continue
        tiles = ((M + bm - 1) // bm) * ((N + bn - 1) // bn)
        raw[(bm, bn)] = ((tiles + 256 * per_cu - 1) // (256 * per_cu)) * per_cu * bm * bn
        if best is None or raw[(bm, bn)] * w < best:
            best, tile = raw[(bm, bn)] * w, (bm, bn, wgm)
    if tile[:2] == (192, 128) and K < 1536 and raw[(128, 128)] <= raw[(192, 128)]:
        tile = (128, 128, 2)
    if tile[:2] == (128, 128) and (192, 256) in raw and N >= 3072 and raw[(192, 256)] <= raw[(128, 128)]:
        tile = (192, 256, 2)
    return tile


def gemm_kernel_key(layout: int, M: int, N: int, vec: bool = True, split: int = 0, K: int = 1 << 20) -> str:
    """Name of the template instantiation the GEMM entry points pick (same rules as csrc/gemm.hip / gemm_bf16s.hip)."""
    if split:
        tile = _bf16s_tile(M, N, K)
        bk = 64 if (tile[0] == 64 and split == 3 and os.environ.get("TVL_GEMM_SMALL_BK", "64") == "64") else 32
        return f"gemm_bf16s_kernel<{tile[0]}, {tile[1]}, {tile[2]}, {split}, {'true' if vec else 'false'}, {bk}, {3 if bk == 64 else 1}, false>"
    best, tile = None, (64, 64, 2)
    for bm, bn, per_cu, wgm in ((128, 128, 2, 2), (96, 128, 2, 1), (64, 64, 4, 2)):
        if bm == 96 and layout == TN:
            continue
        tiles = ((M + bm - 1) // bm) * ((N + bn - 1) // bn)
        slots = 256 * per_cu
        cost = ((tiles + slots - 1) // slots) * per_cu * bm * bn * (1.12 if bm == 64 else 1.0)
        if best is None or cost < best:
            best, tile = cost, (bm, bn, wgm)
    ak, bk = {NT: ("true", "true"), NN: ("true", "false"), TN: ("false", "false")}[layout]
    return f"gemm_f32_kernel<{tile[0]}, {tile[1]}, {tile[2]}, {ak}, {bk}, {'true' if vec else 'false'}>"


class Tp3:
    """An fp32 matrix [rows, cols] held as three bf16 pieces per element in MFMA-fragment order (include/tvl_hip.h, "tp3"):
    the operand format of ``tvl_gemm_tp3``.  ``buf`` is the raw byte image (rows padded to a multiple of 32)."""

    __slots__ = ("buf", "rows", "cols")

    def __init__(self, rows: int, cols: int, device, buf: torch.Tensor | None = None):
        if cols % 16:
            raise RuntimeError(f"tp3 needs cols % 16 == 0, got {cols}")
        self.rows, self.cols = rows, cols
        n = (rows + 31) // 32 * (cols // 16) * 3072
        # rows beyond `rows` in the last 32-row block are read by whole-block consumers (attention key tiles: p = 0 times a stale
        # NaN would poison the sum): an image with padded rows starts zeroed
        # (only the last block row needs it: zeroing the whole image cost 4 % of the DenseCLIP step, whose M = 16 * 1601 is not a multiple of 32)
        if buf is None:
            buf = torch.empty(n, device=device, dtype=torch.uint8)
            if rows % 32:
                buf[(rows // 32) * (cols // 16) * 3072:].zero_()
        self.buf = buf

    @property
    def shape(self):
        return (self.rows, self.cols)

    def float(self) -> torch.Tensor:
        """Back to fp32 (p0 + p1 + p2): tests and debugging."""
        y = torch.empty((self.rows, self.cols), device=self.buf.device, dtype=torch.float32)
        _call("tvl_tp3_unpack", self.buf.data_ptr(), self.rows, self.cols, _p(y), self.cols)
        return y


def tp3_pack(x2d: torch.Tensor) -> Tp3:
    """fp32 [rows, cols] (unit column stride) -> Tp3.  Frozen weights: once at load; activations come from their producers."""
    rows, cols = x2d.shape
    out = Tp3(rows, cols, x2d.device)
    _call("tvl_tp3_pack", _ps(x2d), x2d.stride(0), rows, cols, out.buf.data_ptr())
    return out


def weight_tp3(W: torch.Tensor) -> Tp3:
    """Tp3 image of a frozen weight, cached on the tensor object (prepared weights are persistent, see mark_frozen)."""
    cached = getattr(W, "_tvl_tp3", None)
    if cached is not None and cached[0] == (W.data_ptr(), W._version):
        return cached[1]
    img = tp3_pack(W)
    if not W.requires_grad and W._base is None and W.grad_fn is None and getattr(W, "_tvl_frozen", False):
        W._tvl_tp3 = ((W.data_ptr(), W._version), img)
        _built(img.buf)
    return img


def weight_h2_cached(W: torch.Tensor) -> "H2":
    """H2 image of a frozen weight, cached on the tensor object (valid while its storage and version stay the same)."""
    cached = getattr(W, "_tvl_h2", None)
    if cached is not None and cached[0] == (W.data_ptr(), W._version):
        return cached[1]
    img = weight_h2(W)
    W._tvl_h2 = ((W.data_ptr(), W._version), img)
    return img


def conv_weight_h2_cached(Wm: torch.Tensor, C_in: int) -> "H2":
    """H2 image of a frozen 3x3 conv weight ``Wm`` [N, 9*C_in] (columns (ky, kx, c)) in tvl_conv3x3_h2's column order
    (c / 16, ky, kx, c % 16), cached on the tensor like ``weight_h2_cached``."""
    cached = getattr(Wm, "_tvl_h2_conv", None)
    if cached is not None and cached[0] == (Wm.data_ptr(), Wm._version):
        return cached[1]
    N = Wm.shape[0]
    img = weight_h2(Wm.detach().view(N, 9, C_in // 16, 16).permute(0, 2, 1, 3).reshape(N, 9 * C_in))
    Wm._tvl_h2_conv = ((Wm.data_ptr(), Wm._version), img)
    return img


def mark_frozen(t: torch.Tensor) -> torch.Tensor:
    """Declare a prepared weight tensor immutable for its lifetime (enables the tp3 cache)."""
    t._tvl_frozen = True
    return t


GEMM_TP3_TILE = int(os.environ.get("TVL_TP3_TILE", "0"))      # 0 = automatic; 128 / 192 / 256 rows per workgroup
# 0 = production.  Diagnostic variants (ablations with wrong results, stamps) exist only in a `make DIAG=1` library and are selected
# by the tools through set_tp3_variant(); the training path never reads an environment variable for this.
GEMM_TP3_VARIANT = 0


def set_tp3_variant(v: int) -> None:
    global GEMM_TP3_VARIANT
    GEMM_TP3_VARIANT = int(v)


def gemm_tp3(A: Tp3, B: Tp3, *, M: int | None = None, out: torch.Tensor | None = None, out_tp3: Tp3 | None = None, want_f32=True,
             want_tp3=False, bias=None, residual=None, act=ACT_NONE, pre_out=None, dact_aux=None, dact=ACT_NONE, alpha=1.0):
    """epilogue(alpha * A . B^T) over tp3 operands; returns (C fp32 or None, C as Tp3 or None)."""
    M = A.rows if M is None else M
    N, K = B.rows, A.cols
    if B.cols != K:
        raise RuntimeError(f"gemm_tp3: K mismatch {A.shape} x {B.shape}")
    dev = A.buf.device
    Cf = out if out is not None else (torch.empty((M, N), device=dev, dtype=torch.float32) if want_f32 else None)
    Ct = out_tp3 if out_tp3 is not None else (Tp3(M, N, dev) if want_tp3 else None)
    ldc = Cf.stride(0) if Cf is not None else (pre_out.stride(0) if pre_out is not None else N)
    args = GemmTp3Args(M, N, K, A.buf.data_ptr(), A.rows, B.buf.data_ptr(), B.rows, _ps(Cf), ldc, None if Ct is None else Ct.buf.data_ptr(),
                       _p(bias), _ps(residual), 0 if residual is None else residual.stride(0), act, _ps(pre_out), _ps(dact_aux),
                       0 if dact_aux is None else dact_aux.stride(0), dact, alpha, GEMM_TP3_TILE, GEMM_TP3_VARIANT)
    if _gemm_prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _call("tvl_gemm_tp3", C.byref(args))
    if _gemm_prof is not None:
        e1.record()
        _gemm_prof.append((tp3_kernel_name(M, N, bias is not None, residual is not None, act, dact, pre_out is not None, Cf is not None,
                                           Ct is not None, alpha), 2.0 * M * N * K, e0, e1))
    return Cf, Ct


_TP3_EPI_BUILT = {32, 33, 35, 85, 69, 72, 64, 65}   # launch_epi's compile-time epilogues (csrc/gemm_tp3_kernel.h); anything else runs the generic -1


def tp3_kernel_name(M, N, bias, residual, act, dact, pre_out, c_f32, c_tp3, alpha=1.0) -> str:
    """Instantiation tvl_gemm_tp3 launches for a call, spelled as rocprofv3 prints it (bench.py matches profiles/ by this name)."""
    tile = tp3_tile(M, N)
    variant = GEMM_TP3_VARIANT or (2 if tile == 256 else 3)
    epi = (1 if bias else 0) | (2 if residual else 0) | (4 if (act & 0xFF) else 0) | (8 if dact else 0) | (16 if pre_out else 0) | \
          (32 if c_f32 else 0) | (64 if c_tp3 else 0)
    if alpha != 1.0 or (act & ~0xFF) or epi not in _TP3_EPI_BUILT:
        epi = -1
    return f"gemm_tp3_kernel<{tile}, 256, {variant}, {epi}, 3, false, false>"


GEMM_WORKSPACE_BYTES = 64 << 20
_GEMM_WORK: dict = {}


def gemm_aux(M: int, N: int, device) -> torch.Tensor | None:
    """A buffer for fc1's pre-activation in the GEMM's own accumulator order (``aux_blocked``), or None when the shape does not qualify:
    written by fc1's epilogue, read by the QuickGELU' epilogue of its data gradient, never anything else -- fully coalesced both ways."""
    n = load().tvl_gemm_aux_floats(M, N)
    if n <= 0 or not GEMM_M16:
        return None
    # with TVL_GEMM_ZHALF (default) the buffer holds QuickGELU'(z) as one fp16 per element: half the floats
    return torch.empty((n + 1) // 2 if GEMM_ZHALF else n, device=device, dtype=torch.float32)


def gemm_workspace() -> torch.Tensor:
    """64 MiB of device memory per (device, stream) for the persistent tile walk of the image-writing GEMMs (tvlGemmTp3Args.workspace):
    a workgroup parks one partial tile there between the two passes over its split first tile."""
    key = (torch.cuda.current_device(), torch.cuda.current_stream().cuda_stream)
    w = _GEMM_WORK.get(key)
    if w is None:
        w = _GEMM_WORK[key] = torch.empty(GEMM_WORKSPACE_BYTES, device="cuda", dtype=torch.uint8)
    return w


class H2:
    """An fp32 matrix [rows, cols] as two fp16 pieces per element of the row- (activations) or tensor- (weights) scaled values, in
    MFMA-fragment order (include/tvl_hip.h, "h2"): the operand format of ``tvl_gemm_h2`` -- 3 MFMAs per product instead of tp3's 6.
    ``inv_scale``: fp32 [rows] (per row) or [1] (per tensor) exact powers of two that the GEMM's epilogue multiplies back in."""

    __slots__ = ("buf", "rows", "cols", "inv_scale", "per_row", "_alpha", "row_norm", "_bound", "norm_max")

    def __init__(self, rows: int, cols: int, device, per_row: bool, zero_tail: bool = False):
        if cols % 16:
            raise RuntimeError(f"h2 needs cols % 16 == 0, got {cols}")
        self.rows, self.cols, self.per_row = rows, cols, per_row
        n = (rows + 31) // 32 * (cols // 16) * 2048
        if zero_tail:   # one more 32-row block of zeros: where the padding taps of conv3x3_h2 read
            self.buf = torch.empty(n + (cols // 16) * 2048, device=device, dtype=torch.uint8)
            self.buf[n:].zero_()
        else:
            self.buf = torch.empty(n, device=device, dtype=torch.uint8)
            if rows % 32:   # padded rows of the last block row are read by whole-block consumers: they start as zeros (the rest is written by the producer)
                self.buf[(rows // 32) * (cols // 16) * 2048:].zero_()
        self.inv_scale = torch.empty(rows if per_row else 1, device=device, dtype=torch.float32)
        self._alpha = None   # host copy of a per-tensor inverse scale, read once (frozen weights)
        self.row_norm = None   # [rows] L2 norms of the rows (per-row operands whose consumer GEMM writes an h2 output)
        self.norm_max = None   # [1] the largest of them, left behind by the producer (max_slot): a one-scale output needs no reduction launch
        self._bound = None

    @classmethod
    def wrap(cls, rows: int, cols: int, buf: torch.Tensor, inv_scale: torch.Tensor, per_row: bool) -> "H2":
        """An H2 over existing storage (tensors saved for the backward)."""
        t = cls.__new__(cls)
        t.rows, t.cols, t.per_row, t.buf, t.inv_scale, t._alpha, t.row_norm, t._bound, t.norm_max = rows, cols, per_row, buf, inv_scale, None, None, None, None
        return t

    @property
    def shape(self):
        return (self.rows, self.cols)

    def float(self) -> torch.Tensor:
        """Back to fp32: (h0 + h1) * inverse scale.  Tests and debugging (plain tensor ops on the byte image)."""
        RB, KB = (self.rows + 31) // 32, self.cols // 16
        h = self.buf[: RB * KB * 2048].view(torch.float16).view(RB, KB, 2, 2, 32, 8).float().sum(2)   # [rb, kb, k-half, row, 8]
        x = h.permute(0, 3, 1, 2, 4).reshape(RB * 32, self.cols)[: self.rows]
        return x * (self.inv_scale[:, None] if self.per_row else self.inv_scale)

    def alpha(self) -> float:
        """Per-tensor inverse scale as a host float (one device read per weight image, then cached)."""
        if self._alpha is None:
            self._alpha = float(self.inv_scale[0].item())
        return self._alpha


def h2_pack(x2d: torch.Tensor, per_row: bool, want_norm: bool = False, zero_tail: bool = False, relu_mask: torch.Tensor | None = None) -> H2:
    """fp32 [rows, cols] -> H2 (per_row: activations, the A operand; per tensor: frozen weights, the B operand, and conv inputs).
    ``relu_mask``: pack ``x * (relu_mask > 0)`` -- a ReLU layer's data gradient gated on the way into the image."""
    rows, cols = x2d.shape
    out = H2(rows, cols, x2d.device, per_row, zero_tail)
    work = None if per_row else torch.empty(1, device=x2d.device, dtype=torch.int32)
    if per_row and want_norm:
        out.row_norm = torch.empty(rows, device=x2d.device, dtype=torch.float32)
    if relu_mask is not None:
        if relu_mask.shape != x2d.shape:
            raise RuntimeError(f"h2_pack: mask {tuple(relu_mask.shape)} vs matrix {tuple(x2d.shape)}")
        _call("tvl_h2_pack_masked", _ps(x2d), x2d.stride(0), _ps(relu_mask), relu_mask.stride(0), rows, cols, out.buf.data_ptr(), _p(out.inv_scale),
              _p(out.row_norm), 1 if per_row else 0, None if work is None else work.data_ptr())
    else:
        _call("tvl_h2_pack", _ps(x2d), x2d.stride(0), rows, cols, out.buf.data_ptr(), _p(out.inv_scale), _p(out.row_norm), 1 if per_row else 0,
              None if work is None else work.data_ptr())
    return out


def weight_h2(W: torch.Tensor) -> H2:
    """H2 image of a frozen weight [N, K] + the largest L2 norm of its rows (the factor of the output bound, tvl_gemm_h2_out)."""
    img = h2_pack(W.detach().contiguous(), per_row=False)
    img._bound = float(W.detach().float().norm(dim=1).max().item()) * 1.0001
    img.alpha()
    return img


def gemm_h2(A: H2, B: H2, *, M: int | None = None, out: torch.Tensor | None = None, out_tp3: Tp3 | None = None, want_f32=True,
            want_tp3=False, bias=None, residual=None, act=ACT_NONE, pre_out=None, dact_aux=None, dact=ACT_NONE, tile_m: int = 0,
            want_h2=False, out_mul: float | None = None, out_add: float = 0.0, out_per_tensor: bool = False, persistent: bool = True,
            aux_blocked: bool = False):
    """epilogue(A . B^T) over h2 operands (A row-scaled, B tensor-scaled); returns (C fp32 or None, C as Tp3 / H2 or None).
    ``want_h2``: the result as an H2 image (next GEMM's A operand); its row scales come from the bound
    ``A.row_norm[m] * out_mul + out_add`` (out_mul defaults to B's largest row norm)."""
    M = A.rows if M is None else M
    N, K = B.rows, A.cols
    if B.cols != K or B.per_row:
        raise RuntimeError(f"gemm_h2: need B {B.shape} per-tensor scaled and K = {K}")
    a_scale = A.inv_scale   # [rows], or [1] for a one-scale activation image (attention's O): a_scale_one
    dev = A.buf.device
    Cf = out if out is not None else (torch.empty((M, N), device=dev, dtype=torch.float32) if want_f32 else None)
    Ct = out_tp3 if out_tp3 is not None else (Tp3(M, N, dev) if want_tp3 else None)
    Ch = None
    if want_h2:
        if A.row_norm is None or (out_mul is None and B._bound is None):
            raise RuntimeError("gemm_h2(want_h2=True) needs A.row_norm (from A's producer) and a bound factor (weight_h2 / out_mul)")
        Ch = H2(M, N, dev, per_row=not out_per_tensor)
    if aux_blocked:   # pre_out / dact_aux: flat private buffers of gemm_aux(M, N)
        ldc, ld_aux, pre_p, aux_p = N, N, _p(pre_out), _p(dact_aux)
    else:
        ldc = Cf.stride(0) if Cf is not None else (pre_out.stride(0) if pre_out is not None else N)
        ld_aux, pre_p, aux_p = (0 if dact_aux is None else dact_aux.stride(0)), _ps(pre_out), _ps(dact_aux)
    args = GemmTp3Args(M, N, K, A.buf.data_ptr(), A.rows, B.buf.data_ptr(), B.rows, _ps(Cf), ldc, None if Ct is None else Ct.buf.data_ptr(),
                       _p(bias), _ps(residual), 0 if residual is None else residual.stride(0), act, pre_p, aux_p,
                       ld_aux, dact, B.alpha(), tile_m, GEMM_TP3_VARIANT,
                       gemm_workspace().data_ptr() if (persistent and (Ch is not None or pre_out is not None)) else None, GEMM_WORKSPACE_BYTES,
                       (2 if GEMM_ZHALF else 1) if aux_blocked else 0, 0 if A.per_row else 1)
    if _gemm_prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    if Ch is not None:
        # per-tensor mode: ONE bound from the largest row norm of A (a one-element device tensor), one scale for the whole image
        norm = (A.norm_max if A.norm_max is not None else A.row_norm.max().reshape(1)) if out_per_tensor else A.row_norm
        _call("tvl_gemm_h2_out", C.byref(args), _p(a_scale), Ch.buf.data_ptr(), _p(norm), float(B._bound if out_mul is None else out_mul),
              float(out_add), _p(Ch.inv_scale), 1 if out_per_tensor else 0)
    else:
        _call("tvl_gemm_h2", C.byref(args), _p(a_scale))
    if _gemm_prof is not None:
        e1.record()
        aux_b = 2.0 if (aux_blocked and GEMM_ZHALF) else 4.0
        nbytes = 4.0 * (M + N) * K + M * N * (4.0 * (Cf is not None) + 6.0 * (Ct is not None) + 4.0 * (Ch is not None) + aux_b * (pre_out is not None)
                                               + 4.0 * (residual is not None) + aux_b * (dact_aux is not None))
        _gemm_prof.append((h2_kernel_name(M, N, bias is not None, residual is not None, act, dact, pre_out is not None, Cf is not None,
                                          Ct is not None, tile_m, Ch is not None, K=K), 2.0 * M * N * K, e0, e1, nbytes))
    return Cf, (Ch if Ch is not None else Ct)


_H2_EPI_BUILT = {192: {193, 161, 160, 192, 163, 385, 384, 673}, 256: {213, 197, 200, 405, 389, 392, 161, 673, 160}}   # launch_h2's compile-time epilogues per row tile (csrc/gemm_h2.hip)


# fc1 leaves QuickGELU'(z) -- all its data gradient needs of z -- as ONE fp16 per element in the accumulator-order buffer instead of z as fp32: 2 B
# instead of 4 written by fc1 and read by dz (-195 MB per layer and step at the headline shape).  Its rounding (2^-12 relative, zero-mean) enters dz
# once and is averaged by the K = 3072 contraction of dx2; every fixture incl. the *_tails ones passes its unchanged gates (profiles/r4_gemm_experiments.md).
# 0 = fp32 z (A/B switch).
GEMM_ZHALF = os.environ.get("TVL_GEMM_ZHALF", "1") != "0"
GEMM_M16 = os.environ.get("TVL_GEMM_M16", "1") != "0"   # the h2 ring GEMMs on v_mfma_f32_16x16x32_f16 (csrc/gemm_h2m_kernel.h); 0 = the 32x32x16 generation
_H2M_EPI_BUILT = {385, 384, 163, 160, 405, 389, 392, 161, 673}   # launch_m_layer_epi's compile-time epilogues


def h2_kernel_name(M, N, bias, residual, act, dact, pre_out, c_f32, c_tp3, tile_m=0, c_h2=False, K=None) -> str:
    """Instantiation tvl_gemm_h2 launches, spelled as rocprofv3 prints it (NP = 2 as the last template argument)."""
    if GEMM_M16 and (K is None or K >= 96) and tile_m in (0, 192, 256, 1926, 2566):
        tile = {1926: 192, 2566: 256}.get(tile_m, tile_m)
        if tile not in (192, 256):
            t256, t192 = -(-M // 256) * -(-N // 256), -(-M // 192) * -(-N // 256)
            tile = 256 if -(-t256 // 256) * 256 <= -(-t192 // 256) * 192 else 192
        epi = (1 if bias else 0) | (2 if residual else 0) | (4 if (act & 0xFF) == ACT_QUICK_GELU else 0) | (512 if (act & 0xFF) == ACT_RELU else 0) | \
              (8 if dact else 0) | (16 if pre_out else 0) | (32 if c_f32 else 0) | (64 if c_tp3 else 0) | 128 | (256 if c_h2 else 0)
        if (act & ~0xFF) or (act & 0xFF) not in (ACT_NONE, ACT_QUICK_GELU, ACT_RELU) or epi not in _H2M_EPI_BUILT:
            epi = -1
        return f"gemm_h2m_kernel<{tile}, {epi}, false, false, false>"
    tile = tile_m
    if tile not in (192, 256):
        t256, t192 = -(-M // 256) * -(-N // 256), -(-M // 192) * -(-N // 256)
        tile = 256 if -(-t256 // 256) * 256 <= -(-t192 // 256) * 192 else 192
    epi = (1 if bias else 0) | (2 if residual else 0) | (4 if (act & 0xFF) == ACT_QUICK_GELU else 0) | (512 if (act & 0xFF) == ACT_RELU else 0) | \
          (8 if dact else 0) | (16 if pre_out else 0) | (32 if c_f32 else 0) | (64 if c_tp3 else 0) | 128 | (256 if c_h2 else 0)
    if (act & 0xFF) not in (ACT_NONE, ACT_QUICK_GELU, ACT_RELU):
        epi = -1
    if (act & ~0xFF) or epi not in _H2_EPI_BUILT[tile]:
        epi = -1
    return f"gemm_tp3_kernel<{tile}, 256, {2 if tile == 256 else 3}, {epi}, 2, false, false>"


def conv_h2_tile(M: int, N: int) -> int:
    """Rows per workgroup of tvl_conv3x3_h2 (the rule of csrc/gemm_h2.hip): fewest rounds x rows, ties to the larger tile.  (A 128-row
    tile for the 26 x 26 / 13 x 13 maps measured no gain on the CRIS step and was dropped.)"""
    if CONV_TILE in (192, 256):
        return CONV_TILE
    t256, t192 = -(-M // 256) * -(-N // 256), -(-M // 192) * -(-N // 256)
    return 256 if -(-t256 // 256) * 256 <= -(-t192 // 256) * 192 else 192


def conv_h2_kernel_name(M, N, bias, act) -> str:
    """Instantiation tvl_conv3x3_h2 launches (CONV = true as the last template argument)."""
    tile = conv_h2_tile(M, N)
    epi = 673 if (bias and act == ACT_RELU) else (160 if not (bias or act) else -1)
    if GEMM_M16:
        return f"gemm_h2m_kernel<{tile}, {epi}, false, true, false>"
    return f"gemm_tp3_kernel<{tile}, 256, {2 if tile == 256 else 3}, {epi}, 2, false, true>"


def tp3_tile(M: int, N: int) -> int:
    """Rows per workgroup tvl_gemm_tp3 picks (same rule as csrc/gemm_tp3.hip)."""
    if GEMM_TP3_TILE:
        return GEMM_TP3_TILE
    best, bm = None, 0
    for c in (256, 192, 128):
        tiles = ((M + c - 1) // c) * ((N + 255) // 256)
        cost = ((tiles + 255) // 256) * c * (1.08 if c == 128 else 1.0)
        if best is None or cost < best:
            best, bm = cost, c
    return bm


def _gemm_takes_h2(layout, M, N, K, A, lda, B, ldb, Cout, ldc, residual, ldr, pre_out, dact_aux, ld_aux, alpha, a_map, c_map) -> bool:
    """Large NT problems over a FROZEN weight (mark_frozen): both operands as two-piece fp16 images -- the weight's image is cached on the
    tensor, the activation is packed on the way in (one pass for its exact row scales: 8 B/element of extra traffic, repaid by 3 MFMAs
    per product on the DMA-ring kernel instead of 6 on the in-kernel-split one wherever the GEMM is not HBM-bound: N, K >= 256)."""
    return bool(layout == NT and GEMM_MODE == "bf16x6" and GEMM_H2 and PACK_H2 and getattr(B, "_tvl_frozen", False) and a_map is None and c_map is None
                and alpha == 1.0 and M >= 2048 and N >= PACK_H2_MIN_N and N % 16 == 0 and K % 32 == 0 and K >= PACK_H2_MIN_K and 2.0 * M * N * K >= 3e9
                and A.dim() == 2 and A.stride(1) == 1 and A.stride(0) == lda and lda % 4 == 0 and B.dim() == 2 and B.is_contiguous() and ldb == K
                and Cout.dim() == 2 and Cout.stride(1) == 1 and Cout.stride(0) == ldc and ldc % 4 == 0
                and (residual is None or (residual.dim() == 2 and residual.stride(1) == 1 and residual.stride(0) == ldr))
                and (pre_out is None or (pre_out.dim() == 2 and pre_out.stride(0) == ldc))
                and (dact_aux is None or (dact_aux.dim() == 2 and dact_aux.stride(0) == ld_aux)))


SPLITK_MIN_K = int(os.environ.get("TVL_GEMM_SPLITK_MIN_K", "768"))   # (1024 until round 3: the decoder's reduce Linears have K = 768)


def gemm(layout: int, M: int, N: int, K: int, A, lda, B, ldb, Cout, ldc, *, bias=None, residual=None, ldr=0, act=ACT_NONE,
         pre_out=None, dact_aux=None, ld_aux=0, dact=ACT_NONE, alpha=1.0, a_map=None, c_map=None, a_relu_mask=None):
    takes_h2 = _gemm_takes_h2(layout, M, N, K, A, lda, B, ldb, Cout, ldc, residual, ldr, pre_out, dact_aux, ld_aux, alpha, a_map, c_map)
    if a_relu_mask is not None and not takes_h2:   # A := A * (mask > 0): fused into the packing where the GEMM packs, a pass of its own elsewhere
        A = dact_mul(A, a_relu_mask, ACT_RELU)
    args = GemmArgs(layout, M, N, K, _ps(A) if A.dim() == 2 else _p(A), lda, _p(B), ldb, _ps(Cout) if Cout.dim() == 2 else _p(Cout), ldc,
                    _p(bias), _ps(residual) if (residual is not None and residual.dim() == 2) else _p(residual), ldr, act, _p(pre_out),
                    _p(dact_aux), ld_aux, dact, alpha, a_map or _ident(), c_map or _ident())
    if takes_h2:
        gemm_h2(h2_pack(A[:M], per_row=True, relu_mask=None if a_relu_mask is None else a_relu_mask[:M]), weight_h2_cached(B), out=Cout[:M], bias=bias,
                residual=None if residual is None else residual[:M], act=act, pre_out=None if pre_out is None else pre_out[:M], dact_aux=None if dact_aux is None else dact_aux[:M], dact=dact)
        return Cout
    split = _NSPLIT.get(GEMM_MODE, 0) if (layout == NT and M >= 256) else 0
    # skinny and deep (text-tower GEMMs: a few dozen 64x64 tiles, 48-64 k-slabs each): split K over more workgroups
    tiles64 = ((M + 63) // 64) * ((N + 63) // 64)
    if split == 3 and SPLITK and tiles64 <= 256 and K >= SPLITK_MIN_K:
        splits = min(8, K // 256)
        ws = torch.empty(splits * M * N, device=Cout.device, dtype=torch.float32)
        if _gemm_prof is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        _call("tvl_gemm_bf16s_splitk", C.byref(args), splits, _p(ws), ws.numel())
        if _gemm_prof is not None:
            e1.record()
            _gemm_prof.append(("gemm_bf16s_splitk_kernel<3, true>", 2.0 * M * N * K, e0, e1))
        return Cout
    if _gemm_prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    if split:
        _call("tvl_gemm_bf16s", C.byref(args), split)
    else:
        _call("tvl_gemm_f32", C.byref(args))
    if _gemm_prof is not None:
        e1.record()
        vec = lda % 4 == 0 and ldb % 4 == 0
        _gemm_prof.append((gemm_kernel_key(layout, M, N, vec, split, K), 2.0 * M * N * K, e0, e1))
    return Cout


def linear_fwd(x2d: torch.Tensor, W: torch.Tensor, b=None, *, act=ACT_NONE, residual=None, want_pre=False, out=None,
               a_map=None, c_map=None, M=None, out_rows=None):
    """y = act(x W^T + b) + residual ; x2d [M,K] (or more rows with a_map), W [N,K]."""
    Mx = x2d.shape[0] if M is None else M
    K = x2d.shape[1]
    N = W.shape[0]
    rows = Mx if out_rows is None else out_rows
    y = out if out is not None else torch.empty((rows, N), device=x2d.device, dtype=torch.float32)
    pre = torch.empty_like(y) if want_pre else None
    gemm(NT, Mx, N, K, x2d, K, W, K, y, N, bias=b, residual=residual, ldr=N, act=act, pre_out=pre, a_map=a_map, c_map=c_map)
    return (y, pre) if want_pre else y


def linear_dgrad(dy2d: torch.Tensor, W: torch.Tensor, *, dact=ACT_NONE, dact_aux=None, residual=None, out=None, c_map=None,
                 out_rows=None, M=None, Wt: torch.Tensor | None = None, relu_mask: torch.Tensor | None = None):
    """dx = (dy W) * act'(aux) + residual ; dy2d [M,N], W [N,K] -> [M,K].  ``Wt`` = W^T [K,N] (kept for frozen weights)
    turns the data gradient into an NT GEMM, which is what the split-bf16 kernel runs."""
    Mx = dy2d.shape[0] if M is None else M
    N, K = W.shape
    rows = Mx if out_rows is None else out_rows
    dx = out if out is not None else torch.empty((rows, K), device=dy2d.device, dtype=torch.float32)
    lda = dy2d.stride(0)   # dy may be a column range of a wider matrix (a packed gradient)
    if Wt is not None and GEMM_MODE != "f32":
        gemm(NT, Mx, K, N, dy2d, lda, Wt, N, dx, K, residual=residual, ldr=K, dact_aux=dact_aux, ld_aux=K, dact=dact, c_map=c_map, a_relu_mask=relu_mask)
    else:
        gemm(NN, Mx, K, N, dy2d, lda, W, K, dx, K, residual=residual, ldr=K, dact_aux=dact_aux, ld_aux=K, dact=dact, c_map=c_map, a_relu_mask=relu_mask)
    return dx


def linear_wgrad(dy2d: torch.Tensor, x2d: torch.Tensor, *, a_map=None, M=None) -> torch.Tensor:
    """dW[N,K] = dy^T x ; dy2d [M,N], x2d [M,K]."""
    Mx = dy2d.shape[0] if M is None else M
    N, K = dy2d.shape[1], x2d.shape[1]
    dW = torch.empty((N, K), device=dy2d.device, dtype=torch.float32)
    gemm(TN, N, K, Mx, dy2d, N, x2d, K, dW, K, a_map=a_map)
    return dW


# --------------------------------------------------------------------------------------
# LayerNorm
# --------------------------------------------------------------------------------------
def layernorm_fwd(x2d, gamma, beta, eps: float, want_stats=True):
    rows, cols = x2d.shape
    y = torch.empty_like(x2d)
    mean = torch.empty(rows, device=x2d.device, dtype=torch.float32) if want_stats else None
    rstd = torch.empty(rows, device=x2d.device, dtype=torch.float32) if want_stats else None
    _call("tvl_layernorm_fwd", _p(x2d), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), rows, cols, float(eps))
    return y, mean, rstd


def layernorm_bwd(dy2d, x2d, gamma, mean, rstd, dres=None, dgamma=None, dbeta=None):
    rows, cols = x2d.shape
    dx = torch.empty_like(x2d)
    _call("tvl_layernorm_bwd", _p(dy2d), _p(x2d), _p(gamma), _p(mean), _p(rstd), _p(dres), _p(dx), _p(dgamma), _p(dbeta), rows, cols)
    return dx


def layernorm_fwd_tp3(x2d, gamma, beta, eps: float, want_stats=True):
    """LayerNorm whose only consumer is a tp3 GEMM: returns (Tp3 image of y, mean, rstd); y is never written in fp32."""
    rows, cols = x2d.shape
    y = Tp3(rows, cols, x2d.device)
    mean = torch.empty(rows, device=x2d.device, dtype=torch.float32) if want_stats else None
    rstd = torch.empty(rows, device=x2d.device, dtype=torch.float32) if want_stats else None
    _call("tvl_layernorm_fwd_tp3", _p(x2d), _p(gamma), _p(beta), y.buf.data_ptr(), _p(mean), _p(rstd), rows, cols, float(eps))
    return y, mean, rstd


_MAX_SLOTS: dict = {}   # device index -> (int64 zeros [4096], the same memory as fp32 [8192])
_max_slot_calls = 0


def _max_slot(device):
    """(address of a 64-bit slot, tag, one-element fp32 view of the slot's low word) for a producer that leaves the largest of its row
    norms behind by tagged atomicMax (tvl_layernorm_fwd_h2 / _bwd_h2 ``max_slot``): slots rotate, the tag grows with every turn of the
    pool, so a slot never has to be cleared -- no fill and no reduction launch per use."""
    global _max_slot_calls
    key = device.index if device.index is not None else torch.cuda.current_device()
    pool = _MAX_SLOTS.get(key)
    if pool is None:
        z = torch.zeros(4096, device=device, dtype=torch.int64)
        torch.cuda.current_stream(device).synchronize()   # once: other streams may use the pool next
        pool = _MAX_SLOTS[key] = (z, z.view(torch.float32))
    n = _max_slot_calls
    _max_slot_calls += 1
    i, tag = n % 4096, n // 4096 + 1
    return pool[0].data_ptr() + 8 * i, tag, pool[1][2 * i: 2 * i + 1]


def layernorm_fwd_h2(x2d, gamma, beta, eps: float, want_stats=True):
    """LayerNorm whose only consumer is an h2 GEMM: returns (H2 image of y with per-row scales, mean, rstd)."""
    rows, cols = x2d.shape
    y = H2(rows, cols, x2d.device, per_row=True)
    y.row_norm = torch.empty(rows, device=x2d.device, dtype=torch.float32)
    mean = torch.empty(rows, device=x2d.device, dtype=torch.float32) if want_stats else None
    rstd = torch.empty(rows, device=x2d.device, dtype=torch.float32) if want_stats else None
    slot, tag, y.norm_max = _max_slot(x2d.device)
    with _aux_span("ln_fwd_h2", 0.0, 8.0 * rows * cols):   # read x fp32, write the two fp16 pieces
        _call("tvl_layernorm_fwd_h2", _p(x2d), _p(gamma), _p(beta), y.buf.data_ptr(), _p(y.inv_scale), _p(y.row_norm), _p(mean), _p(rstd), rows, cols, float(eps),
              slot, tag)
    return y, mean, rstd


def layernorm_bwd_h2(dy2d, x2d, gamma, mean, rstd, dres=None):
    """dx = [dres +] LN'(dy) as fp32 (the residual stream's gradient) AND as the H2 operand of the next data-gradient GEMM."""
    rows, cols = x2d.shape
    dx = torch.empty_like(x2d)
    dxt = H2(rows, cols, x2d.device, per_row=True)
    dxt.row_norm = torch.empty(rows, device=x2d.device, dtype=torch.float32)
    slot, tag, dxt.norm_max = _max_slot(x2d.device)
    with _aux_span("ln_bwd_h2", 0.0, (20.0 if dres is not None else 16.0) * rows * cols):   # read dy, x [, dres]; write dx fp32 + image
        _call("tvl_layernorm_bwd_h2", _p(dy2d), _p(x2d), _p(gamma), _p(mean), _p(rstd), _p(dres), _p(dx), dxt.buf.data_ptr(), _p(dxt.inv_scale),
              _p(dxt.row_norm), rows, cols, slot, tag)
    return dx, dxt


def layernorm_bwd_tp3(dy2d, x2d, gamma, mean, rstd, dres=None):
    """dx = [dres +] LN'(dy) as fp32 (the residual stream's gradient) AND as the Tp3 operand of the next data-gradient GEMM."""
    rows, cols = x2d.shape
    dx = torch.empty_like(x2d)
    dxt = Tp3(rows, cols, x2d.device)
    _call("tvl_layernorm_bwd_tp3", _p(dy2d), _p(x2d), _p(gamma), _p(mean), _p(rstd), _p(dres), _p(dx), dxt.buf.data_ptr(), rows, cols)
    return dx, dxt


CONV_TILE = int(os.environ.get("TVL_CONV_TILE", "0"))   # rows per workgroup of tvl_conv3x3_h2: 0 = its own rule, 192 / 256 = forced (A/B switch, tools/bench_conv.py)
CONV_H2 = os.environ.get("TVL_CONV_H2", "1") != "0"   # 3x3 convs over frozen weights (C % 32 == 0, N >= 128) as an implicit GEMM on two fp16 pieces
PACK_H2_MIN_N = int(os.environ.get("TVL_PACK_H2_MIN_N", "256"))
PACK_H2_MIN_K = int(os.environ.get("TVL_PACK_H2_MIN_K", "256"))
PACK_H2 = os.environ.get("TVL_PACK_H2", "1") != "0"   # generic large Linears over frozen weights: pack the activation to h2 on the way in (hip.gemm)
GEMM_H2 = os.environ.get("TVL_GEMM_H2", "1") != "0"   # the four LayerNorm-fed GEMMs of a tp3 layer on two fp16 pieces (3 MFMAs per product)
DQKV_H2 = os.environ.get("TVL_DQKV_H2", "1") != "0"   # dQ | dK | dV as an h2 image with per-(row, head) scales; QKV data gradient on tvl_gemm_h2_ks
ATTN_H2 = os.environ.get("TVL_ATTN_H2", "1") != "0"   # attention of the tp3 layers on two fp16 pieces (QKV / dO as tensor-scaled h2 images)
ATTN_TP3 = os.environ.get("TVL_ATTN_TP3", "1") != "0"   # 0: attention of the tp3 layers on the fp32-operand kernels (A/B switch)
TP3_MIN_ROWS = int(os.environ.get("TVL_TP3_MIN_ROWS", "1024"))  # below this the layer is launch-latency bound either way


def tp3_path_ok(M: int, D: int, F: int, dh: int, causal: bool, key_mask) -> bool:
    """Can an encoder layer run on the tp3 kernels (GEMM ring + LayerNorm / attention that write tp3)?  d_h = 64 without masks,
    widths that are multiples of 32, enough rows to fill the chip, fp32-equivalent arithmetic selected."""
    return (GEMM_MODE == "bf16x6" and os.environ.get("TVL_ATTN_MODE", "")[:1] != "f" and os.environ.get("TVL_TP3", "1") != "0"
            and dh == 64 and not causal and key_mask is None and D % 32 == 0 and F % 32 == 0 and 64 <= D <= 2048 and M >= TP3_MIN_ROWS)   # tvl_layernorm_*_tp3 / _h2: cols <= 2048


# --------------------------------------------------------------------------------------
# attention over a packed [B, T, 3*H*dh] QKV buffer
# --------------------------------------------------------------------------------------
def attn_fwd_packed_tp3(qkv: torch.Tensor, B: int, T: int, H: int, dh: int, scale: float, want_lse=True):
    """Unmasked d_h = 64 attention whose output feeds the out_proj tp3 GEMM: returns (Tp3 image of O [B*T, H*dh], lse)."""
    D = H * dh
    o = Tp3(B * T, D, qkv.device)
    lse = torch.empty((B, H, T), device=qkv.device, dtype=torch.float32) if want_lse else None
    base = _p(qkv)
    a = AttnFwdArgs(base, base + 4 * D, base + 8 * D, 3 * D * T, 3 * D * T, 3 * D * T, 3 * D, 3 * D, 3 * D, None, D, _p(lse),
                    None, B, H, T, dh, 0, float(scale))
    _call("tvl_attn_fwd_tp3", C.byref(a), o.buf.data_ptr())
    return o, lse


def attn_tp3_fwd(qkv_t: Tp3, B: int, T: int, H: int, scale: float, want_lse=True):
    """Attention over the Tp3 image of the packed QKV matrix [B*T, 3*H*64] (d_h = 64): returns (Tp3 image of O, lse)."""
    D = H * 64
    if qkv_t.rows != B * T or qkv_t.cols != 3 * D:
        raise RuntimeError(f"attn_tp3_fwd: QKV image is {qkv_t.shape}, expected {(B * T, 3 * D)}")
    o = Tp3(B * T, D, qkv_t.buf.device)
    lse = torch.empty((B, H, T), device=qkv_t.buf.device, dtype=torch.float32) if want_lse else None
    _call("tvl_attn_tp3_fwd", qkv_t.buf.data_ptr(), o.buf.data_ptr(), _p(lse), B, H, T, float(scale))
    return o, lse


def attn_tp3_bwd(qkv_t: Tp3, o_t: Tp3, do_t: Tp3, lse, B: int, T: int, H: int, scale: float) -> Tp3:
    """Backward of attn_tp3_fwd, all operands tp3 images: returns dQ | dK | dV as the Tp3 image of the packed gradient [B*T, 3*H*64]."""
    D = H * 64
    for name, t, cols in (("QKV", qkv_t, 3 * D), ("O", o_t, D), ("dO", do_t, D)):
        if t.rows != B * T or t.cols != cols:
            raise RuntimeError(f"attn_tp3_bwd: {name} image is {t.shape}, expected {(B * T, cols)}")
    g = Tp3(B * T, 3 * D, qkv_t.buf.device)
    delta = torch.empty((B, H, T), device=qkv_t.buf.device, dtype=torch.float32)
    _call("tvl_attn_tp3_bwd", qkv_t.buf.data_ptr(), o_t.buf.data_ptr(), do_t.buf.data_ptr(), _p(lse), _p(delta), g.buf.data_ptr(), B, H, T, float(scale))
    return g


def attn_h2_fwd(qkv_h: H2, B: int, T: int, H: int, scale: float, want_lse=True, o_as_h2=False):
    """Attention over the H2 image (ONE tensor scale) of the packed QKV matrix [B*T, 3*H*64]: returns (image of O, lse); O as Tp3, or
    (o_as_h2) as an H2 image that shares the QKV image's scale (|O| <= max |V|)."""
    D = H * 64
    if qkv_h.rows != B * T or qkv_h.cols != 3 * D or qkv_h.per_row:
        raise RuntimeError(f"attn_h2_fwd: QKV image is {qkv_h.shape} per_row={qkv_h.per_row}, expected {(B * T, 3 * D)} with one tensor scale")
    if o_as_h2:
        o = H2(B * T, D, qkv_h.buf.device, per_row=False)
        o.inv_scale = qkv_h.inv_scale
    else:
        o = Tp3(B * T, D, qkv_h.buf.device)
    lse = torch.empty((B, H, T), device=qkv_h.buf.device, dtype=torch.float32) if want_lse else None
    with _aux_span("attn_h2_fwd", 4.0 * B * H * T * T * 64, 4.0 * B * T * D * (3 + (1 if o_as_h2 else 1.5))):   # QKV image in, O image out
        _call("tvl_attn_h2_fwd", qkv_h.buf.data_ptr(), _p(qkv_h.inv_scale), o.buf.data_ptr(), 1 if o_as_h2 else 0, _p(lse), B, H, T, float(scale))
    return o, lse


class H2K:
    """dQ | dK | dV as two fp16 pieces with one exact power-of-two scale per (row, 64-column block): ``kscale`` [rows, cols / 64] inverse
    scales.  The A operand of ``gemm_h2_ks`` (the QKV data gradient)."""

    __slots__ = ("buf", "rows", "cols", "kscale")

    def __init__(self, rows: int, cols: int, device):
        self.rows, self.cols = rows, cols
        n = (rows + 31) // 32 * (cols // 16) * 2048
        self.buf = (torch.zeros if rows % 32 else torch.empty)(n, device=device, dtype=torch.uint8)
        self.kscale = torch.empty((rows, cols // 64), device=device, dtype=torch.float32)

    def float(self) -> torch.Tensor:
        RB, KB = (self.rows + 31) // 32, self.cols // 16
        h = self.buf.view(torch.float16).view(RB, KB, 2, 2, 32, 8).float().sum(2)
        x = h.permute(0, 3, 1, 2, 4).reshape(RB * 32, self.cols)[: self.rows]
        return x * self.kscale.repeat_interleave(64, dim=1)


def gemm_h2_ks(A: H2K, B: H2, out: torch.Tensor | None = None) -> torch.Tensor:
    """A . B^T with A scaled per (row, 64-column block of K) and B a tensor-scaled H2 weight: fp32 result (the QKV data gradient)."""
    M, N, K = A.rows, B.rows, A.cols
    if B.cols != K or B.per_row:
        raise RuntimeError(f"gemm_h2_ks: need B {B.shape} per-tensor scaled and K = {K}")
    Cf = out if out is not None else torch.empty((M, N), device=A.buf.device, dtype=torch.float32)
    args = GemmTp3Args(M, N, K, A.buf.data_ptr(), A.rows, B.buf.data_ptr(), B.rows, _ps(Cf), Cf.stride(0), None, None, None, 0, ACT_NONE, None, None, 0,
                       ACT_NONE, B.alpha(), 0, 0)
    if _gemm_prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _call("tvl_gemm_h2_ks", C.byref(args), _p(A.kscale))
    if _gemm_prof is not None:
        e1.record()
        _gemm_prof.append(("gemm_h2m_kernel<192, 160, true, false, false>" if (GEMM_M16 and K >= 96) else "gemm_tp3_kernel<192, 256, 3, 160, 2, true, false>", 2.0 * M * N * K, e0, e1))
    return Cf


def attn_h2_bwd(qkv_h: H2, o_t, do_h: H2, lse, B: int, T: int, H: int, scale: float, out_h2: bool = False, only_block: int = -1):
    """Backward of attn_h2_fwd: QKV and dO as tensor-scaled H2 images, O as Tp3 or as the H2 image the forward wrote; returns
    dQ | dK | dV as a Tp3 image, or (out_h2) as an H2K image with exact per-(row, head) scales.  ``only_block`` >= 0: only rows
    128 * only_block .. + 127 of every sample are computed (the rest of the image stays unwritten)."""
    D = H * 64
    if qkv_h.per_row or do_h.per_row or do_h.rows != B * T or do_h.cols != D:
        raise RuntimeError("attn_h2_bwd: QKV / dO must be tensor-scaled H2 images of [B*T, 3D] / [B*T, D]")
    dev = qkv_h.buf.device
    g = H2K(B * T, 3 * D, dev) if out_h2 else Tp3(B * T, 3 * D, dev)
    delta = torch.empty((B, H, T), device=dev, dtype=torch.float32)
    dn = torch.empty(B * H, device=dev, dtype=torch.int32)
    # 10 T^2 d_h FLOP per (b, h) (five products; the recomputed S of the second kernel is not counted); bytes: both kernels read QKV + dO,
    # the dQ one also O, together they write dQ | dK | dV
    part = min(128, T) / T if only_block >= 0 else 1.0   # one 128-row block of queries (dQ) and of keys (dK | dV) against all T of the other side
    with _aux_span("attn_h2_bwd", 10.0 * B * H * T * T * 64 * part, 4.0 * B * T * D * (2 * 4 + 1 + 3)):
        _call("tvl_attn_h2_bwd", qkv_h.buf.data_ptr(), _p(qkv_h.inv_scale), o_t.buf.data_ptr(), 1 if isinstance(o_t, H2) else 0, do_h.buf.data_ptr(),
              _p(do_h.inv_scale), _p(lse),
              _p(delta), dn.data_ptr(), g.buf.data_ptr(), 1 if out_h2 else 0, _p(g.kscale) if out_h2 else None, B, H, T, float(scale), int(only_block))
    return g


def h2k_gather_rows(g: "H2K", B: int, T: int, row0: int, n: int) -> torch.Tensor:
    """Rows ``b*T + row0 .. + n - 1`` of an H2K gradient image as an fp32 matrix [B*n, cols]."""
    if g.rows != B * T:
        raise RuntimeError(f"h2k_gather_rows: image has {g.rows} rows, expected {B * T}")
    out = torch.empty((B * n, g.cols), device=g.buf.device, dtype=torch.float32)
    _call("tvl_h2k_gather_rows", g.buf.data_ptr(), _p(g.kscale), g.cols, B, T, row0, n, _p(out))
    return out


_CONST_I64: dict = {}


def const_i64(values, device) -> torch.Tensor:
    """int64 twin of const_i32 (torch index ops want int64)."""
    key = (tuple(int(v) for v in values), str(device))
    t = _CONST_I64.get(key)
    if t is None:
        if len(_CONST_I64) > 1024:
            _CONST_I64.clear()
        t = _CONST_I64[key] = torch.tensor(key[0], dtype=torch.int64, device=device)
        _built(t)
    return t


def attn_bwd_packed_tp3(qkv, o_tp3: Tp3, d_o, lse, B: int, T: int, H: int, dh: int, scale: float) -> Tp3:
    """Backward of the above: delta from the Tp3 O, dQ | dK | dV as the Tp3 image of the packed gradient [B*T, 3*H*dh]."""
    D = H * dh
    dqkv = Tp3(B * T, 3 * D, qkv.device)
    delta = torch.empty((B, H, T), device=qkv.device, dtype=torch.float32)
    base = _p(qkv)
    s3 = 3 * D
    a = AttnBwdArgs(base, base + 4 * D, base + 8 * D, s3 * T, s3 * T, s3 * T, s3, s3, s3, None, _p(d_o), D, _p(lse), _p(delta),
                    None, None, None, 0, 0, 0, 0, 0, 0, None, B, H, T, dh, 0, float(scale))
    _call("tvl_attn_bwd_tp3", C.byref(a), o_tp3.buf.data_ptr(), dqkv.buf.data_ptr())
    return dqkv



def attn_fwd_packed(qkv: torch.Tensor, B: int, T: int, H: int, dh: int, scale: float, causal=False, key_mask=None, want_lse=True):
    D = H * dh
    o = torch.empty((B * T, D), device=qkv.device, dtype=torch.float32)
    lse = torch.empty((B, H, T), device=qkv.device, dtype=torch.float32) if want_lse else None
    base = _p(qkv)
    a = AttnFwdArgs(base, base + 4 * D, base + 8 * D, 3 * D * T, 3 * D * T, 3 * D * T, 3 * D, 3 * D, 3 * D, _p(o), D, _p(lse),
                    _p(key_mask, torch.int32), B, H, T, dh, int(bool(causal)), float(scale))
    _call("tvl_attn_fwd", C.byref(a))
    return o, lse


def attn_bwd_packed(qkv, o, d_o, lse, B: int, T: int, H: int, dh: int, scale: float, causal=False, key_mask=None):
    D = H * dh
    dqkv = torch.empty_like(qkv)
    delta = torch.empty((B, H, T), device=qkv.device, dtype=torch.float32)
    base, dbase = _p(qkv), _p(dqkv)
    s3 = 3 * D
    a = AttnBwdArgs(base, base + 4 * D, base + 8 * D, s3 * T, s3 * T, s3 * T, s3, s3, s3, _p(o), _p(d_o), D, _p(lse), _p(delta),
                    dbase, dbase + 4 * D, dbase + 8 * D, s3 * T, s3 * T, s3 * T, s3, s3, s3, _p(key_mask, torch.int32),
                    B, H, T, dh, int(bool(causal)), float(scale))
    _call("tvl_attn_bwd", C.byref(a))
    return dqkv


# --------------------------------------------------------------------------------------
# token plumbing
# --------------------------------------------------------------------------------------
def im2col_patch(img: torch.Tensor, ps: int) -> torch.Tensor:
    B, Cc, H, W = img.shape
    cols = torch.empty((B * (H // ps) * (W // ps), Cc * ps * ps), device=img.device, dtype=torch.float32)
    _call("tvl_im2col_patch", _p(img), _p(cols), B, Cc, H, W, ps)
    return cols


def vision_assemble(patch, cls, pos, ctx, ctx_bs: int, B: int, P: int, n: int, D: int) -> torch.Tensor:
    x0 = torch.empty((B, 1 + P + n, D), device=patch.device, dtype=torch.float32)
    _call("tvl_vision_assemble", _p(patch), _p(cls), _p(pos), _p(ctx), ctx_bs, _p(x0), B, P, n, D)
    return x0


def text_assemble(ids, tmap, table, ctx, ctx_bs: int, pos, B: int, T: int, D: int) -> torch.Tensor:
    out = torch.empty((B, T, D), device=table.device, dtype=torch.float32)
    _call("tvl_text_assemble", _p(ids, torch.int64), ids.shape[1], _p(tmap, torch.int32), _p(table), table.shape[0], _p(ctx), ctx_bs, _p(pos),
          _p(out), B, T, D)
    return out


def rows_overwrite(x, src, src_bs: int, row0: int, n: int):
    B, T, D = x.shape
    _call("tvl_rows_overwrite", _p(x), _p(src), src_bs, B, T, D, row0, n)


def rows_grad(g, dst, row0: int, n: int, reduce_batch: bool, zero_src: bool, accumulate: bool = False):
    B, T, D = g.shape
    _call("tvl_rows_grad", _p(g), _p(dst), B, T, D, row0, n, int(reduce_batch), int(zero_src), int(accumulate))


def h2_zero_rows(img: "H2", B: int, T: int, row0: int, n: int) -> None:
    """Rows ``b*T + row0 .. + n - 1`` of the image := 0 (the image of a gradient whose rows were cut in place)."""
    if img.rows != B * T:
        raise RuntimeError(f"h2_zero_rows: image has {img.rows} rows, expected {B * T}")
    _call("tvl_h2_zero_rows", img.buf.data_ptr(), img.cols, B, T, row0, n)


def gather_rows(x, idx):
    B, T, D = x.shape
    out = torch.empty((B, D), device=x.device, dtype=torch.float32)
    _call("tvl_gather_rows", _p(x), _p(idx, torch.int32), _p(out), B, T, D)
    return out


def scatter_rows_add(dout, idx, dx):
    B, T, D = dx.shape
    _call("tvl_scatter_rows_add", _p(dout), _p(idx, torch.int32), _p(dx), B, T, D)


# --------------------------------------------------------------------------------------
# decoder pieces
# --------------------------------------------------------------------------------------
def film_fwd(x, mul, add):
    B, T, Cc = x.shape
    y = torch.empty_like(x)
    _call("tvl_film_fwd", _p(x), _p(mul), _p(add), _p(y), B, T, Cc)
    return y


def film_bwd(dy, x, mul, want_cond_grads: bool):
    B, T, Cc = x.shape
    dx = torch.empty_like(x)
    dmul = torch.empty((B, Cc), device=x.device, dtype=torch.float32) if want_cond_grads else None
    dadd = torch.empty((B, Cc), device=x.device, dtype=torch.float32) if want_cond_grads else None
    _call("tvl_film_bwd", _p(dy), _p(x), _p(mul), _p(dx), _p(dmul), _p(dadd), B, T, Cc)
    return dx, dmul, dadd


def pixel_shuffle_fwd(cols, bias, extra, a: float, r: float, B: int, G: int, ps: int):
    logits = torch.empty((B, G * ps, G * ps), device=cols.device, dtype=torch.float32)
    _call("tvl_pixel_shuffle_fwd", _p(cols), _p(bias), _p(extra), float(a), float(r), _p(logits), B, G, ps)
    return logits


def pixel_unshuffle_bwd(dlogits, a: float, B: int, G: int, ps: int):
    dcols = torch.empty((B * G * G, ps * ps), device=dlogits.device, dtype=torch.float32)
    _call("tvl_pixel_unshuffle_bwd", _p(dlogits), float(a), _p(dcols), B, G, ps)
    return dcols


def upconv_taps_fwd(taps, bias, B: int, G: int, ps: int, k: int):
    out = torch.empty((B, G * ps, G * ps), device=taps.device, dtype=torch.float32)
    work = torch.empty(B * G * k * G * ps, device=taps.device, dtype=torch.float32)   # x pass -> y pass
    _call("tvl_upconv_taps_fwd", _p(taps), taps.shape[1], _p(bias), _p(out), _p(work), B, G, ps, k)
    return out


def upconv_taps_bwd(dout, B: int, G: int, ps: int, k: int):
    dtaps = torch.empty((B * G * G, k * k), device=dout.device, dtype=torch.float32)
    work = torch.empty(B * k * G * ps * G, device=dout.device, dtype=torch.float32)
    _call("tvl_upconv_taps_bwd", _p(dout), _p(dtaps), k * k, _p(work), B, G, ps, k)
    return dtaps


# --------------------------------------------------------------------------------------
# loss / metrics / optimiser / misc
# --------------------------------------------------------------------------------------
GRAD_ROWS = os.environ.get("TVL_GRAD_ROWS", "1") != "0"   # first vision layer: input gradient for the prompt rows only (ops.EncoderLayerTp3Fn); 0 = all rows (A/B switch)
MLP64 = os.environ.get("TVL_MLP64", "1") != "0"   # the decoder's feed-forward block as one kernel (csrc/mlp64.hip); 0 = op by op (A/B switch)


class Mlp64Weights:
    """The four fragment-ordered two-piece fp16 images of a frozen (W1 [F, 64], W2 [64, F]) pair + the scalars the kernels need."""

    __slots__ = ("img", "F", "inv_w1", "inv_w2", "w1_rownorm", "b1_max", "w2_colnorm")

    def __init__(self, W1: torch.Tensor, b1: torch.Tensor, W2: torch.Tensor):
        F = W1.shape[0]
        if W1.shape[1] != 64 or tuple(W2.shape) != (64, F) or F % 128:
            raise RuntimeError(f"mlp64 wants W1 [F, 64], W2 [64, F], F % 128 == 0; got {tuple(W1.shape)}, {tuple(W2.shape)}")
        W1, W2 = W1.detach().float().contiguous(), W2.detach().float().contiguous()

        def inv_scale(amax: float) -> float:   # h2::inv_scale_of on the host
            if not (1.0e-30 < amax < 3.0e38):
                return 1.0
            return math.ldexp(1.0, math.frexp(amax)[1] - 14)

        self.F = F
        self.inv_w1, self.inv_w2 = inv_scale(float(W1.abs().max().item())), inv_scale(float(W2.abs().max().item()))
        self.w1_rownorm = float(W1.norm(dim=1).max().item()) * 1.0001
        self.w2_colnorm = float(W2.norm(dim=0).max().item()) * 1.0001
        self.b1_max = float(b1.detach().abs().max().item()) * 1.0001
        self.img = torch.empty(load().tvl_mlp64_image_bytes(F), device=W1.device, dtype=torch.uint8)
        _call("tvl_mlp64_pack", _p(W1), _p(W2), F, 1.0 / self.inv_w1, 1.0 / self.inv_w2, self.img.data_ptr())


def mlp64_fwd(x2d, w: Mlp64Weights, b1, b2, gamma, beta, eps: float, want_stats=True):
    """LayerNorm(x + W2 relu(W1 x + b1) + b2) for x [M, 64] in one launch: returns (out, t2 = the LayerNorm's input, mean, rstd)."""
    M = x2d.shape[0]
    out = torch.empty_like(x2d)
    t2 = torch.empty_like(x2d) if want_stats else None
    mean = torch.empty(M, device=x2d.device, dtype=torch.float32) if want_stats else None
    rstd = torch.empty(M, device=x2d.device, dtype=torch.float32) if want_stats else None
    with _aux_span("mlp64_fwd", 4.0 * M * w.F * 64, 4.0 * M * 64 * (3 if want_stats else 2)):
        _call("tvl_mlp64_fwd", _p(x2d), w.img.data_ptr(), _p(b1), _p(b2), _p(gamma), _p(beta), _p(out), _p(t2), _p(mean), _p(rstd), M, w.F,
              w.inv_w1, w.inv_w2, w.w1_rownorm, w.b1_max, float(eps))
    return out, t2, mean, rstd


def mlp64_bwd(dout2d, x2d, t2, mean, rstd, w: Mlp64Weights, b1, gamma):
    """Gradient of :func:`mlp64_fwd` w.r.t. x (frozen weights: data gradient only), one launch."""
    M = x2d.shape[0]
    dx = torch.empty_like(x2d)
    with _aux_span("mlp64_bwd", 6.0 * M * w.F * 64, 4.0 * M * 64 * 4):
        _call("tvl_mlp64_bwd", _p(dout2d), _p(x2d), _p(t2), _p(mean), _p(rstd), w.img.data_ptr(), _p(b1), _p(gamma), _p(dx), M, w.F,
              w.inv_w1, w.inv_w2, w.w2_colnorm)
    return dx


def dicece_stats(logits, target, thr: float, want_label=False):
    B = logits.shape[0]
    N = logits[0].numel()
    fsum = torch.empty((B, 4), device=logits.device, dtype=torch.float64)
    isum = torch.empty((B, 4), device=logits.device, dtype=torch.int64)
    label = torch.empty(logits.shape, device=logits.device, dtype=torch.uint8) if want_label else None
    work = torch.empty(load().tvl_dicece_work_doubles(B, N), device=logits.device, dtype=torch.float64)
    _call("tvl_dicece_stats", _p(logits), _p(target), _p(fsum, torch.float64), _p(isum, torch.int64), _p(label, torch.uint8),
          _p(work, torch.float64), B, N, float(thr))
    return fsum, isum, label


def dicece_loss(fsum, N: int, lambda_dice, lambda_ce, smooth_nr=1e-5, smooth_dr=1e-5):
    """fp32 scalar DiceCE loss from the per-sample sums of :func:`dicece_stats` (one launch, float64 inside)."""
    loss = torch.empty((), device=fsum.device, dtype=torch.float32)
    _call("tvl_dicece_loss", _p(fsum, torch.float64), _p(loss), fsum.shape[0], int(N), float(lambda_dice), float(lambda_ce),
          float(smooth_nr), float(smooth_dr), nonfinite_flags(fsum.device)[0:1].data_ptr())
    return loss


def dicece_bwd(logits, target, fsum, lambda_dice, lambda_ce, smooth_nr, smooth_dr, gscale):
    B = logits.shape[0]
    N = logits[0].numel()
    dl = torch.empty_like(logits)
    _call("tvl_dicece_bwd", _p(logits), _p(target), _p(fsum, torch.float64), _p(dl), B, N, float(lambda_dice), float(lambda_ce),
          float(smooth_nr), float(smooth_dr), _p(gscale))
    return dl


INTER_NEAREST, INTER_CUBIC = 0, 2   # cv2's values (the reference's YAML resolves ${import_eval:cv2.INTER_CUBIC} to 2)


def resize_u8(packed: torch.Tensor, offs: torch.Tensor, hw: torch.Tensor, C_: int, H: int, W: int, mode: int) -> torch.Tensor:
    """Ragged batch -> [B, H, W, C] uint8 (albumentations.Resize with cv2.INTER_CUBIC / INTER_NEAREST).  ``packed`` uint8 [bytes] holds the
    B images back to back, ``offs`` int64 [B] their byte offsets, ``hw`` int32 [B, 2] their (height, width)."""
    B = hw.shape[0]
    out = torch.empty((B, H, W, C_), device=packed.device, dtype=torch.uint8)
    _call("tvl_resize_u8", _p(packed, torch.uint8), _p(offs, torch.int64), _p(hw, torch.int32), B, C_, H, W, int(mode), _p(out, torch.uint8))
    return out


def augment_u8(img_u8: torch.Tensor, mask_u8_: torch.Tensor | None, params: torch.Tensor, flags: torch.Tensor, mean, std):
    """[B,H,W,3] uint8 (+ [B,H,W] uint8 mask) -> (normalised fp32 [B,3,H,W], fp32 mask / 255 [B,1,H,W] or None) with the per-sample
    affine warp / brightness-contrast of ``params`` [B, 8] / ``flags`` [B] (tvl_augment_u8)."""
    B, H, W, Cc = img_u8.shape
    if Cc != 3:
        raise RuntimeError(f"augment_u8 wants [B,H,W,3] uint8, got {tuple(img_u8.shape)}")
    out = torch.empty((B, 3, H, W), device=img_u8.device, dtype=torch.float32)
    om = torch.empty((B, 1, H, W), device=img_u8.device, dtype=torch.float32) if mask_u8_ is not None else None
    m3, s3 = (C.c_float * 3)(*[float(v) for v in mean]), (C.c_float * 3)(*[float(v) for v in std])
    _call("tvl_augment_u8", _p(img_u8, torch.uint8), _p(mask_u8_, torch.uint8), _p(params), _p(flags, torch.int32), m3, s3, _p(out), _p(om), B, H, W)
    return out, om


def normalize_u8(img_u8: torch.Tensor, mean, std) -> torch.Tensor:
    """Decoded images [B,H,W,3] uint8 -> normalised network input [B,3,H,W] float: ((x / 255) - mean) / std per channel."""
    B, H, W, Cc = img_u8.shape
    if Cc != 3 or img_u8.dtype != torch.uint8:
        raise RuntimeError(f"normalize_u8 wants [B,H,W,3] uint8, got {tuple(img_u8.shape)} {img_u8.dtype}")
    out = torch.empty((B, 3, H, W), device=img_u8.device, dtype=torch.float32)
    m3, s3 = (C.c_float * 3)(*[float(v) for v in mean]), (C.c_float * 3)(*[float(v) for v in std])
    _call("tvl_normalize_u8", _p(img_u8, torch.uint8), _p(out), B, H, W, m3, s3)
    return out


def mask_u8(mask_u8_: torch.Tensor) -> torch.Tensor:
    """Decoded grey-level masks [B,H,W] uint8 -> [B,1,H,W] float in [0, 1] (value / 255)."""
    B, H, W = mask_u8_.shape
    out = torch.empty((B, 1, H, W), device=mask_u8_.device, dtype=torch.float32)
    _call("tvl_mask_u8", _p(mask_u8_, torch.uint8), _p(out), mask_u8_.numel())
    return out


def mix(main, extra, ratio):
    """(1 - ratio) * main + ratio * extra, ``ratio`` a 0-d / 1-element DEVICE tensor (no host sync)."""
    out = torch.empty_like(main)
    _call("tvl_mix", _p(main), _p(extra), _p(ratio.reshape(1)), _p(out), main.numel())
    return out


def scale_dev(x, ratio, one_minus: bool):
    y = torch.empty_like(x)
    _call("tvl_scale_dev", _p(x), _p(ratio.reshape(1)), int(one_minus), _p(y), x.numel())
    return y


def adamw(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step_t: int, grad_scale: float = 1.0):
    _call("tvl_adamw", _p(p), _p(g), _p(m), _p(v), p.numel(), float(lr), float(beta1), float(beta2), float(eps), float(weight_decay),
          int(step_t), float(grad_scale), nonfinite_flags(p.device)[1:2].data_ptr())


def fill(t, val: float):
    _call("tvl_fill", _p(t), float(val), t.numel())


def axpby(x, a: float, y, b: float):
    _call("tvl_axpby", _p(x), float(a), _p(y), float(b), x.numel())


def bias_act(x2d, bias, act: int):
    y = torch.empty_like(x2d)
    _call("tvl_bias_act", _p(x2d), _p(bias), _p(y), x2d.shape[0], x2d.shape[1], act)
    return y


def l2norm_fwd(x2d):
    y = torch.empty_like(x2d)
    inv = torch.empty(x2d.shape[0], device=x2d.device, dtype=torch.float32)
    _call("tvl_l2norm_fwd", _p(x2d), _p(y), _p(inv), x2d.shape[0], x2d.shape[1])
    return y, inv


def l2norm_bwd(dy, y, inv):
    dx = torch.empty_like(y)
    _call("tvl_l2norm_bwd", _p(dy), _p(y), _p(inv), _p(dx), y.shape[0], y.shape[1])
    return dx


def dot(x, y=None, out=None, accumulate=False):
    if out is None:
        out = torch.empty(1, device=x.device, dtype=torch.float32)
    _call("tvl_dot", _p(x), _p(y), _p(out), x.numel(), int(accumulate))
    return out


def colsum(x2d, out=None, accumulate=False):
    if out is None:
        out = torch.empty(x2d.shape[1], device=x2d.device, dtype=torch.float32)
    _call("tvl_colsum", _p(x2d), _p(out), x2d.shape[0], x2d.shape[1], int(accumulate))
    return out


def dact_mul(dy2d, pre, act: int):
    if dy2d.shape != pre.shape or not dy2d.is_contiguous() or not pre.is_contiguous():
        raise RuntimeError(f"dact_mul: needs two contiguous tensors of one shape, got {tuple(dy2d.shape)} / {tuple(pre.shape)}")
    out = torch.empty_like(dy2d)
    _call("tvl_dact_mul", _p(dy2d), _p(pre), _p(out), dy2d.numel(), act)
    return out


def outer_add(bias, cvec):
    B, D = bias.shape
    n = cvec.shape[0]
    out = torch.empty((B, n, D), device=bias.device, dtype=torch.float32)
    _call("tvl_outer_add", _p(bias), _p(cvec), _p(out), B, n, D)
    return out


def outer_add_bwd(dout):
    B, n, D = dout.shape
    dbias = torch.empty((B, D), device=dout.device, dtype=torch.float32)
    dc = torch.empty((n, D), device=dout.device, dtype=torch.float32)
    _call("tvl_outer_add_bwd", _p(dout), _p(dbias), _p(dc), B, n, D)
    return dbias, dc


def splice_rows(x, tmap, ctx, ctx_bs: int):
    B, L, D = x.shape
    T = tmap.shape[0]
    out = torch.empty((B, T, D), device=x.device, dtype=torch.float32)
    _call("tvl_splice_rows", _p(x), L, _p(tmap, torch.int32), _p(ctx), ctx_bs, _p(out), B, T, D)
    return out


def dropout(x, p: float, seed: int):
    y = torch.empty_like(x)
    _call("tvl_dropout", _p(x), _p(y), x.numel(), float(p), int(seed) & 0xFFFFFFFFFFFFFFFF)
    return y


# --------------------------------------------------------------------------------------
# attention with separate q / k / v matrices (cross-attention: T queries x Tk keys)
# --------------------------------------------------------------------------------------
def _qkv_view(t: torch.Tensor, rows_per_batch: int):
    """(ptr, batch stride, row stride) of a [B*rows, >=H*dh] matrix that may be a column slice of a packed buffer."""
    return _ps(t), t.stride(0) * rows_per_batch, t.stride(0)


def attn_fwd(q, k, v, B: int, T: int, Tk: int, H: int, dh: int, scale: float, causal=False, key_mask=None, want_lse=True):
    D = H * dh
    o = torch.empty((B * T, D), device=q.device, dtype=torch.float32)
    lse = torch.empty((B, H, T), device=q.device, dtype=torch.float32) if want_lse else None
    (qp, qb, qt), (kp, kb, kt), (vp, vb, vt) = _qkv_view(q, T), _qkv_view(k, Tk), _qkv_view(v, Tk)
    a = AttnFwdArgs(qp, kp, vp, qb, kb, vb, qt, kt, vt, _p(o), D, _p(lse), _p(key_mask, torch.int32), B, H, T, dh,
                    int(bool(causal)), float(scale), Tk)
    _call("tvl_attn_fwd", C.byref(a))
    return o, lse


def attn_bwd(q, k, v, o, d_o, lse, dq, dk, dv, B: int, T: int, Tk: int, H: int, dh: int, scale: float, causal=False, key_mask=None):
    """dq/dk/dv are caller-provided matrices (possibly column slices of one packed gradient buffer)."""
    D = H * dh
    delta = torch.empty((B, H, T), device=q.device, dtype=torch.float32)
    (qp, qb, qt), (kp, kb, kt), (vp, vb, vt) = _qkv_view(q, T), _qkv_view(k, Tk), _qkv_view(v, Tk)
    (dqp, dqb, dqt), (dkp, dkb, dkt), (dvp, dvb, dvt) = _qkv_view(dq, T), _qkv_view(dk, Tk), _qkv_view(dv, Tk)
    a = AttnBwdArgs(qp, kp, vp, qb, kb, vb, qt, kt, vt, _p(o), _p(d_o), D, _p(lse), _p(delta), dqp, dkp, dvp, dqb, dkb, dvb,
                    dqt, dkt, dvt, _p(key_mask, torch.int32), B, H, T, dh, int(bool(causal)), float(scale), Tk)
    _call("tvl_attn_bwd", C.byref(a))


# --------------------------------------------------------------------------------------
# CRIS conv path: NHWC pixel matrices [B*H*W, C] (rows may be strided: channel slices of a concat buffer)
# --------------------------------------------------------------------------------------
def _out2d(out, rows: int, cols: int, like: torch.Tensor) -> torch.Tensor:
    if out is None:
        return torch.empty((rows, cols), device=like.device, dtype=torch.float32)
    if tuple(out.shape) != (rows, cols):
        raise RuntimeError(f"output matrix has shape {tuple(out.shape)}, expected {(rows, cols)}")
    return out


def im2col3x3(x2d: torch.Tensor, B: int, H: int, W: int, stride: int = 1) -> torch.Tensor:
    """x2d [B*H*W, C] NHWC -> cols [B*Ho*Wo, 9*C (padded to a multiple of 4)], column = (ky*3+kx)*C + c."""
    Cc = x2d.shape[1]
    ld = x2d.stride(0)
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    ldc = (9 * Cc + 3) // 4 * 4
    cols = torch.empty((B * Ho * Wo, ldc), device=x2d.device, dtype=torch.float32)
    _call("tvl_im2col3x3", _ps(x2d), H * W * ld, W * ld, ld, 1, _p(cols), ldc, B, H, W, Cc, stride)
    return cols


def im2col3x3_nchw(img: torch.Tensor, stride: int = 1) -> torch.Tensor:
    """NCHW image [B,C,H,W] -> cols, same column order (the stem conv reads the image directly)."""
    B, Cc, H, W = img.shape
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    ldc = (9 * Cc + 3) // 4 * 4
    cols = torch.empty((B * Ho * Wo, ldc), device=img.device, dtype=torch.float32)
    _call("tvl_im2col3x3", _p(img), Cc * H * W, W, 1, H * W, _p(cols), ldc, B, H, W, Cc, stride)
    return cols


def avgpool_fwd(x2d, B: int, H: int, W: int, k: int, out=None):
    Cc = x2d.shape[1]
    y = _out2d(out, B * (H // k) * (W // k), Cc, x2d)
    _call("tvl_avgpool_fwd", _ps(x2d), x2d.stride(0), _ps(y), y.stride(0), B, H, W, Cc, k)
    return y


def avgpool_bwd(dy2d, B: int, H: int, W: int, k: int, out=None):
    """H, W are the INPUT sizes of the pooled map."""
    Cc = dy2d.shape[1]
    dx = _out2d(out, B * H * W, Cc, dy2d)
    _call("tvl_avgpool_bwd", _ps(dy2d), dy2d.stride(0), _ps(dx), dx.stride(0), B, H, W, Cc, k)
    return dx


def bilinear_up_fwd(x2d, B: int, H: int, W: int, s: int, out=None):
    Cc = x2d.shape[1]
    y = _out2d(out, B * H * s * W * s, Cc, x2d)
    _call("tvl_bilinear_up_fwd", _ps(x2d), x2d.stride(0), _ps(y), y.stride(0), B, H, W, Cc, s)
    return y


def bilinear_up_bwd(dy2d, B: int, H: int, W: int, s: int, out=None):
    """H, W are the INPUT sizes (dy is [B*H*s*W*s, C])."""
    Cc = dy2d.shape[1]
    dx = _out2d(out, B * H * W, Cc, dy2d)
    _call("tvl_bilinear_up_bwd", _ps(dy2d), dy2d.stride(0), _ps(dx), dx.stride(0), B, H, W, Cc, s)
    return dx


def bilinear_up_h2(x2d: torch.Tensor, B: int, H: int, W: int, s: int) -> "H2":
    """bilinear_up_fwd written directly as the tensor-scaled H2 image (+ zero block) that ``conv3x3`` reads (``packed=``): the upsampled
    map never exists in fp32.  Its scale is the input's (a bilinear tap is a convex combination)."""
    Cc = x2d.shape[1]
    out = H2(B * H * s * W * s, Cc, x2d.device, per_row=False, zero_tail=True)
    bits = torch.empty(1, device=x2d.device, dtype=torch.int32)
    _call("tvl_h2_absmax", _ps(x2d), x2d.stride(0), x2d.shape[0], Cc, bits.data_ptr())
    _call("tvl_bilinear_up_h2", _ps(x2d), x2d.stride(0), bits.data_ptr(), out.buf.data_ptr(), _p(out.inv_scale), B, H, W, Cc, s)
    return out


def bicubic_ac_fwd(x, Ho: int, Wo: int, extra=None, a: float = 1.0, r: float = 0.0):
    """x [B,Hi,Wi] -> a * bicubic(x; align_corners=True) [+ r * extra] as [B,Ho,Wo]."""
    B, Hi, Wi = x.shape
    y = torch.empty((B, Ho, Wo), device=x.device, dtype=torch.float32)
    _call("tvl_bicubic_ac_fwd", _p(x), _p(y), _p(extra), float(a), float(r), B, Hi, Wi, Ho, Wo)
    return y


def bicubic_ac_bwd(dy, Hi: int, Wi: int, a: float = 1.0):
    B, Ho, Wo = dy.shape
    dx = torch.empty((B, Hi, Wi), device=dy.device, dtype=torch.float32)
    _call("tvl_bicubic_ac_bwd", _p(dy), float(a), _p(dx), B, Hi, Wi, Ho, Wo)
    return dx


def dynconv_fwd(x2d, word, B: int, H: int, W: int):
    """Per-sample 3x3 conv C->1 with kernel/bias from ``word`` [B, 9C+1] (reference cris_model/layers.py:106-118)."""
    Cc = x2d.shape[1]
    taps = torch.empty((B * H * W, 9), device=x2d.device, dtype=torch.float32)
    out = torch.empty((B, H, W), device=x2d.device, dtype=torch.float32)
    _call("tvl_dynconv_fwd", _ps(x2d), x2d.stride(0), _p(word), word.shape[1], _p(taps), _p(out), B, H, W, Cc)
    return out


def dynconv_bwd(dout, x2d, word, B: int, H: int, W: int, need_dx: bool = True):
    Cc = x2d.shape[1]
    dx = torch.empty((B * H * W, Cc), device=x2d.device, dtype=torch.float32) if need_dx else None
    dword = torch.empty_like(word)
    n = load().tvl_dynconv_bwd_work_floats(B, H, W, Cc)
    work = torch.empty(n, device=x2d.device, dtype=torch.float32)
    _call("tvl_dynconv_bwd", _p(dout), _ps(x2d), x2d.stride(0), _p(word), word.shape[1], _p(dx), Cc, _p(dword), _p(work), B, H, W, Cc)
    return dx, dword


def copy2d(src: torch.Tensor, dst: torch.Tensor) -> torch.Tensor:
    """dst[:, :] = src for 2-D matrices whose rows may be strided (channel concat / split)."""
    if src.shape != dst.shape:
        raise RuntimeError(f"copy2d shape mismatch {tuple(src.shape)} vs {tuple(dst.shape)}")
    _call("tvl_copy2d", _ps(src), src.stride(0), _ps(dst), dst.stride(0), src.shape[0], src.shape[1])
    return dst


def conv3x3_takes_h2(M: int, Cc: int, Wm: torch.Tensor, stride: int = 1) -> bool:
    """Will ``conv3x3`` run this problem on two fp16 pieces (tvl_conv3x3_h2)?  (Producers that can write the h2 image themselves ask.)"""
    N = Wm.shape[0]
    return bool(GEMM_MODE == "bf16x6" and GEMM_H2 and CONV_H2 and stride == 1 and Cc % 32 == 0 and N >= 128 and N % 16 == 0 and M >= 2048
                and getattr(Wm, "_tvl_frozen", False) and Wm.shape[1] == 9 * Cc)


def conv3x3(x2d: torch.Tensor | None, B: int, H: int, W: int, Wm: torch.Tensor, bias=None, act: int = ACT_NONE, stride: int = 1, out=None,
            packed: "H2 | None" = None, x_relu_mask: torch.Tensor | None = None):
    """3x3 / pad 1 conv of an NHWC pixel matrix with GEMM-ordered weights ``Wm`` [Cout, >=9C] (+ bias, activation).

    Split-bf16 modes run it as an implicit GEMM (no im2col matrix); the exact-fp32 mode and shapes the implicit kernel does
    not take (C % 4 != 0, tiny maps) build the im2col matrix and call the plain GEMM."""
    Cc = packed.cols if packed is not None else x2d.shape[1]
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    M, N = B * Ho * Wo, Wm.shape[0]
    if packed is not None and (packed.per_row or packed.rows != M or not conv3x3_takes_h2(M, Cc, Wm, stride)):
        raise RuntimeError("conv3x3(packed=...): needs a tensor-scaled H2 image of the [B*H*W, C] map and a problem conv3x3_takes_h2 accepts")
    y = _out2d(out, M, N, x2d if x2d is not None else packed.buf)
    split = _NSPLIT.get(GEMM_MODE, 0)
    if packed is not None or (conv3x3_takes_h2(M, Cc, Wm, stride) and x2d.stride(0) % 4 == 0 and y.stride(0) % 4 == 0):
        # two fp16 pieces: the map packed with one scale (+ a zero block for the padding taps), the taps gathered by the GEMM's LDS-DMA
        xa = packed if packed is not None else h2_pack(x2d, per_row=False, zero_tail=True, relu_mask=x_relu_mask)
        wb = conv_weight_h2_cached(Wm, Cc)
        args = GemmTp3Args(M, N, 9 * Cc, xa.buf.data_ptr(), M, wb.buf.data_ptr(), wb.rows, _ps(y), y.stride(0), None, _p(bias), None, 0, act, None, None, 0,
                           ACT_NONE, wb.alpha(), CONV_TILE, 0)
        geom = ConvGeom(B, H, W, Cc, 1)
        if _gemm_prof is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        _call("tvl_conv3x3_h2", C.byref(args), C.byref(geom), _p(xa.inv_scale))
        if _gemm_prof is not None:
            e1.record()
            _gemm_prof.append((conv_h2_kernel_name(M, N, bias is not None, act), 2.0 * M * N * 9 * Cc, e0, e1))
        return y
    if x_relu_mask is not None:   # the paths below read fp32 rows: gate them in a pass of their own
        x2d = dact_mul(x2d, x_relu_mask, ACT_RELU)
    if split == 3 and Cc % 4 == 0 and M >= 256 and x2d.stride(0) % 4 == 0:
        args = GemmArgs(NT, M, N, 9 * Cc, _ps(x2d), x2d.stride(0), _p(Wm), Wm.shape[1], _ps(y), y.stride(0), _p(bias), None, 0, act,
                        None, None, 0, ACT_NONE, 1.0, _ident(), _ident())
        geom = ConvGeom(B, H, W, Cc, stride)
        if _gemm_prof is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        _call("tvl_conv3x3_bf16s", C.byref(args), C.byref(geom), split)
        if _gemm_prof is not None:
            e1.record()
            _gemm_prof.append((gemm_kernel_key(NT, M, N, True, split, 9 * Cc).replace("32, 1, false>", "32, 1, true>"), 2.0 * M * N * 9 * Cc, e0, e1))
        return y
    cols = im2col3x3(x2d, B, H, W, stride)
    K = cols.shape[1]
    gemm(NT, M, N, K, cols, K, Wm, Wm.shape[1], y, y.stride(0), bias=bias, act=act)
    return y


def bicubic_resize_u8(pred: torch.Tensor, Ho: int, Wo: int) -> torch.Tensor:
    """One probability map [H, W] -> uint8 grey levels [Ho, Wo] (bicubic, align_corners=False, no antialias; x*255+0.5 clamped)."""
    H, W = pred.shape
    out = torch.empty((Ho, Wo), device=pred.device, dtype=torch.uint8)
    _call("tvl_bicubic_resize_u8", _p(pred), out.data_ptr(), H, W, Ho, Wo)
    return out


# --------------------------------------------------------------------------------------
# DenseCLIP (csrc/denseclip.hip): FPN taps of the ViT backbone over NHWC pixel matrices
# --------------------------------------------------------------------------------------
def groupnorm_nhwc(tokens: torch.Tensor, skip_rows: int, H: int, W: int, gamma, beta, eps: float, *, pool: int = 1, out: torch.Tensor | None = None) -> torch.Tensor:
    """``nn.GroupNorm(1, C)`` [+ ``nn.MaxPool2d(2, 2)``] over the map held by rows ``skip_rows ..`` of every sample of ``tokens`` [B, T, C]
    (reference models.py:697-701: the tap drops the CLS row) -> pixel matrix [B*(H/pool)*(W/pool), C], or into ``out`` (a column range of a
    wider matrix: the score map is concatenated behind fpn3, denseclip.py:166-168)."""
    B, T, Cc = tokens.shape
    if T != skip_rows + H * W or not tokens.is_contiguous():
        raise RuntimeError(f"groupnorm_nhwc: tokens {tuple(tokens.shape)} do not hold a {H}x{W} map behind {skip_rows} rows")
    lib = _lib if _lib is not None else load()
    stats = torch.empty(2 * B, device=tokens.device, dtype=torch.float32)
    work = torch.empty(int(lib.tvl_groupnorm_work_doubles(B, H * W)), device=tokens.device, dtype=torch.float64)
    src = tokens.view(B * T, Cc)[skip_rows:]   # first pixel of sample 0; sample stride T*C
    _call("tvl_groupnorm_stats", src.data_ptr(), T * Cc, Cc, B, H * W, Cc, float(eps), _p(stats), work.data_ptr())
    rows = B * (H // pool) * (W // pool)
    y = out if out is not None else torch.empty((rows, Cc), device=tokens.device, dtype=torch.float32)
    if y.shape != (rows, Cc):
        raise RuntimeError(f"groupnorm_nhwc: out {tuple(y.shape)} != {(rows, Cc)}")
    _call("tvl_groupnorm_apply", src.data_ptr(), T * Cc, Cc, _p(stats), _p(gamma), _p(beta), _ps(y), y.stride(0), B, H, W, Cc, pool)
    return y


def tconv2x2_unshuffle(blocked: torch.Tensor, B: int, H: int, W: int, Cc: int, levels: int, out: torch.Tensor | None = None) -> torch.Tensor:
    """[B*H*W*4^(levels-1), 4C] (the output of ``levels`` ConvTranspose2d(k=2, s=2)-as-GEMM steps) -> raster NHWC [B*(H*2^l)*(W*2^l), C]."""
    f = 1 << levels
    if blocked.numel() != B * H * W * f * f * Cc or not blocked.is_contiguous():
        raise RuntimeError(f"tconv2x2_unshuffle: {tuple(blocked.shape)} is not B*H*W*4^levels x C = {B * H * W * f * f} x {Cc}")
    y = out if out is not None else torch.empty((B * H * f * W * f, Cc), device=blocked.device, dtype=torch.float32)
    _call("tvl_tconv2x2_unshuffle", _p(blocked), _ps(y), y.stride(0), B, H, W, Cc, levels)
    return y


def colscale_add(a: torch.Tensor, b: torch.Tensor, g: torch.Tensor) -> torch.Tensor:
    out = torch.empty_like(a)
    _call("tvl_colscale_add", _p(a), _p(b), _p(g), _p(out), a.numel() // a.shape[-1], a.shape[-1])
    return out


def colscale_bwd(d: torch.Tensor, b: torch.Tensor, g: torch.Tensor, want_db: bool, want_dg: bool):
    db = torch.empty_like(d) if want_db else None
    dg = torch.empty_like(g) if want_dg else None
    _call("tvl_colscale_bwd", _p(d), _p(b), _p(g), _p(db), _p(dg), d.numel() // d.shape[-1], d.shape[-1])
    return db, dg


def blockdiag_gather(full: torch.Tensor, B: int, T: int, skip: int, rows: int, K: int) -> torch.Tensor:
    """out[b*rows + i, k] = full[b*T + skip + i, b*K + k] (full: [B*T, B*K])."""
    out = torch.empty((B * rows, K), device=full.device, dtype=torch.float32)
    _call("tvl_blockdiag_gather", _p(full), full.stride(0), _p(out), B, T, skip, rows, K)
    return out


# --------------------------------------------------------------------------------------
# few-query cross-attention (csrc/attention_fq.hip): Tq <= 32 queries, many keys, no masks
# --------------------------------------------------------------------------------------
FQ_ATTN = os.environ.get("TVL_FQ_ATTN", "1") != "0"   # 0 = the flash kernels also for few queries (A/B switch)


def fq_attention_ok(Tq: int, Tk: int, dh: int, causal: bool, key_mask) -> bool:
    return bool(FQ_ATTN and Tq <= 32 and Tk >= 256 and dh in (16, 32, 64) and not causal and key_mask is None)


def fq_attn_fwd(q, k, v, B: int, Tq: int, Tk: int, H: int, dh: int, scale: float):
    """-> (o [B*Tq, H*dh], P [B, H, Tq, Tk])."""
    P = torch.empty((B, H, Tq, Tk), device=q.device, dtype=torch.float32)
    _call("tvl_fq_qk", _ps(q), q.stride(0), _ps(k), k.stride(0), _p(P), B, H, Tq, Tk, dh, float(scale))
    _call("tvl_fq_softmax", _p(P), None, B * H * Tq, Tk)
    o = torch.empty((B * Tq, H * dh), device=q.device, dtype=torch.float32)
    _call("tvl_fq_pk", _p(P), H * Tq * Tk, Tq * Tk, Tk, 1, _ps(v), Tk * v.stride(0), v.stride(0), _p(o), o.stride(0), B, H, Tq, Tk, dh, 1.0)
    return o, P


def fq_attn_bwd(q, k, v, o, d_o, P, B: int, Tq: int, Tk: int, H: int, dh: int, scale: float, need_dq=True, need_dkv=True):
    """dq [B*Tq, D], dk, dv [B*Tk, D] (each None when not asked for)."""
    D = H * dh
    dS = torch.empty_like(P)
    _call("tvl_fq_qk", _ps(d_o), d_o.stride(0), _ps(v), v.stride(0), _p(dS), B, H, Tq, Tk, dh, 1.0)          # dP = dO V^T
    _call("tvl_fq_ds", _p(P), _p(dS), _ps(d_o), d_o.stride(0), _ps(o), o.stride(0), B, H, Tq, Tk, dh)      # dS = P (dP - delta)
    dq = dk = dv = None
    if need_dq:
        dq = torch.empty((B * Tq, D), device=q.device, dtype=torch.float32)
        _call("tvl_fq_pk", _p(dS), H * Tq * Tk, Tq * Tk, Tk, 1, _ps(k), Tk * k.stride(0), k.stride(0), _p(dq), D, B, H, Tq, Tk, dh, float(scale))
    if need_dkv:
        dk = torch.empty((B * Tk, D), device=q.device, dtype=torch.float32)
        dv = torch.empty((B * Tk, D), device=q.device, dtype=torch.float32)
        _call("tvl_fq_tk", _p(dS), _ps(q), q.stride(0), _p(dk), D, B, H, Tq, Tk, dh, float(scale))
        _call("tvl_fq_tk", _p(P), _ps(d_o), d_o.stride(0), _p(dv), D, B, H, Tq, Tk, dh, 1.0)
    return dq, dk, dv


def score_map_text_grad(dS: torch.Tensor, v_hat: torch.Tensor, B: int, T: int, skip: int, HW: int, K: int) -> torch.Tensor | None:
    """dT[b] = dS[b]^T V[b] (dS [B*HW, K] contiguous, V = rows skip.. of every sample of v_hat [B*T, C]) on the few-query kernel: the K classes are
    its "queries", the pixels its "keys", the C columns of V its heads of 64.  None when the shape does not fit (the caller then runs B TN GEMMs)."""
    Cc = v_hat.shape[1]
    if not (FQ_ATTN and K <= 32 and Cc % 64 == 0 and v_hat.stride(0) == Cc and dS.is_contiguous() and tuple(dS.shape) == (B * HW, K)):
        return None
    dt = torch.empty((B * K, Cc), device=dS.device, dtype=torch.float32)
    _call("tvl_fq_pk", _p(dS), HW * K, 0, 1, K, v_hat[skip:].data_ptr(), T * Cc, Cc, _p(dt), Cc, B, Cc // 64, K, HW, 64, 1.0)
    return dt.view(B, K, Cc)
