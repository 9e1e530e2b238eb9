"""ctypes binding of libldiff_hip.so (include/ldiff.h).  No fallback: if the shared object is
missing or cannot be loaded the import of any compute entry point raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libldiff_hip.so")
MAX_BLOCKS = 8

F32, F16, BF16 = 0, 1, 2


class UNetCfg(C.Structure):
    _fields_ = [("in_channels", C.c_int), ("out_channels", C.c_int), ("n_blocks", C.c_int),
                ("block_out_channels", C.c_int * MAX_BLOCKS), ("down_has_attn", C.c_int * MAX_BLOCKS),
                ("up_has_attn", C.c_int * MAX_BLOCKS), ("layers_per_block", C.c_int), ("heads", C.c_int),
                ("cross_attention_dim", C.c_int), ("norm_num_groups", C.c_int), ("norm_eps", C.c_float),
                ("flip_sin_to_cos", C.c_int), ("freq_shift", C.c_float)]


class VaeCfg(C.Structure):
    _fields_ = [("in_channels", C.c_int), ("out_channels", C.c_int), ("latent_channels", C.c_int), ("n_blocks", C.c_int),
                ("block_out_channels", C.c_int * MAX_BLOCKS), ("layers_per_block", C.c_int), ("norm_num_groups", C.c_int),
                ("scaling_factor", C.c_float)]


class ConvArgs(C.Structure):
    _fields_ = [("x", C.c_void_p), ("x2", C.c_void_p), ("C1", C.c_int), ("C2", C.c_int),
                ("B", C.c_int), ("Hin", C.c_int), ("Win", C.c_int), ("Hout", C.c_int), ("Wout", C.c_int),
                ("ks", C.c_int), ("stride", C.c_int), ("pad_t", C.c_int), ("pad_l", C.c_int), ("ups", C.c_int),
                ("w", C.c_void_p), ("N", C.c_int), ("Nrows", C.c_int),
                ("gn_scale", C.c_void_p), ("gn_shift", C.c_void_p), ("silu_in", C.c_int),
                ("bias", C.c_void_p), ("temb", C.c_void_p), ("ld_temb", C.c_int),
                ("res", C.c_void_p), ("ld_res", C.c_int), ("y", C.c_void_p), ("ldy", C.c_int), ("out_f32", C.c_int), ("stats", C.c_void_p),
                ("geglu", C.c_int), ("ld1", C.c_int), ("ld2", C.c_int), ("res_lo", C.c_int), ("y_lo", C.c_int), ("short_runs", C.c_int),
                ("lo8_slab0", C.c_int), ("lo8_scale", C.c_void_p), ("gemm_df", C.c_int),
                ("sc_x", C.c_void_p), ("sc_C", C.c_int), ("sc_ld", C.c_int), ("sc_w", C.c_void_p), ("sc_bias", C.c_void_p), ("c3d_ups", C.c_int), ("n_real", C.c_int), ("splitk", C.c_int)]


# name -> (restype, argtypes); every symbol include/ldiff.h declares
P, I, F, I64, U64 = C.c_void_p, C.c_int, C.c_float, C.c_int64, C.c_uint64
SIGNATURES = {
    "ldiff_version": (I, []),
    "ldiff_last_error": (C.c_char_p, []),
    "ldiff_unet_create": (I, [C.POINTER(P), C.POINTER(UNetCfg), I]),
    "ldiff_unet_load": (I, [P, C.c_char_p, P, I, C.POINTER(I64), I]),
    "ldiff_unet_set_precision": (I, [P, I]),
    "ldiff_unet_set_graph": (I, [P, I]),
    "ldiff_unet_graph_replays": (I64, [P]),
    "ldiff_unet_graph_nodes": (I64, [P]),
    "ldiff_unet_missing": (I, [P]),
    "ldiff_unet_missing_name": (C.c_char_p, [P, I]),
    "ldiff_unet_set_context": (I, [P, P, I, I, P]),
    "ldiff_unet_forward": (I, [P, P, I, I, I, F, P, P]),
    "ldiff_unet_set_additional_residuals": (I, [P, C.POINTER(P), I, P]),
    "ldiff_unet_check_finite": (I, [P, P]),
    "ldiff_unet_destroy": (None, [P]),
    "ldiff_vae_create": (I, [C.POINTER(P), C.POINTER(VaeCfg), I]),
    "ldiff_vae_load": (I, [P, C.c_char_p, P, I, C.POINTER(I64), I]),
    "ldiff_vae_set_precision": (I, [P, I, I]),
    "ldiff_vae_missing": (I, [P]),
    "ldiff_vae_missing_name": (C.c_char_p, [P, I]),
    "ldiff_vae_encode": (I, [P, P, I, I, I, P, P]),
    "ldiff_vae_decode": (I, [P, P, I, I, I, F, P, P, P, P, I, I, P]),
    "ldiff_vae_check_finite": (I, [P, P]),
    "ldiff_vae_destroy": (None, [P]),
    "ldiff_pndm_step": (I, [C.POINTER(F), C.POINTER(P), I, P, I64, P]),
    "ldiff_pndm_alphas_cumprod": (I, [C.POINTER(F), I]),
    "ldiff_pndm_coeffs": (I, [F, F, C.POINTER(F), C.POINTER(F)]),
    "ldiff_laplace_add": (I, [P, F, P, U64, U64, P, I64, P]),
    "ldiff_argmax_u8": (I, [P, I, I, I, I, P, P]),
    "ldiff_probe_argmax_u8": (I, [P, I, I, I, I, P, P, F, I, P, P]),
    "ldiff_window_accumulate": (I, [P, P, P, P, I, I, I, I, I, I, I, I, P]),
    "ldiff_luma_float": (I, [P, P, I, I, I, P]),
    "ldiff_bilinear_resize": (I, [P, P, I, I, I, I, I, I, P]),
    "ldiff_pipeline_create": (I, [C.POINTER(P), P, P]),
    "ldiff_pipeline_set_alphas_cumprod": (I, [P, C.POINTER(F), I]),
    "ldiff_pipeline_set_overlap": (I, [P, I]),
    "ldiff_pipeline_join": (I, [P, P]),
    "ldiff_sample": (I, [P, P, I, I, I, I, P, P, P, P]),
    "ldiff_pipeline_check_finite": (I, [P, P]),
    "ldiff_plms_timesteps": (I, [I, C.POINTER(I64), I]),
    "ldiff_pipeline_destroy": (None, [P]),
    "ldiff_op_conv": (I, [C.POINTER(ConvArgs), P]),
    "ldiff_op_conv_stats_blocks": (I, [C.POINTER(ConvArgs)]),
    "ldiff_op_gn_finalize": (I, [P, I, I, P, I, I, I, I, I, F, P, P, P, P, P]),
    "ldiff_op_attention": (I, [P, I, P, I, P, I, P, I, I, I, I, I, I, I64, I64, I64, F, P]),
    "ldiff_op_attention_prescaled": (I, [P, I, P, I, P, I, P, I, I, I, I, I, I, I64, I64, I64, P]),
    "ldiff_op_gn_stats": (I, [P, I, I, I, P, I, I, I, I, I, I, F, P, P, P, P, P]),
    "ldiff_op_layernorm": (I, [P, I, I, P, I, I, P, P, F, P]),
    "ldiff_op_ln_linear": (I, [P, I, I, I, I, P, P, F, P, I, I, P, I, P, I, I, F, P]),
    "ldiff_op_norm_apply": (I, [P, I, I, I, P, I, I, I, I, I, P, P, I, P, I, I, P]),
    "ldiff_op_dup_weights": (I, [P, P, I, I, I, I, I, I, P]),
    "ldiff_op_norm_apply_lo8": (I, [P, I, I, I, I, I, P, P, I, P, P]),
    "ldiff_op_lo8_weights": (I, [P, P, P, I, I, I, P]),
    "ldiff_op_geglu": (I, [P, P, I64, I, P]),
    "ldiff_op_nchw_to_nhwc": (I, [P, P, I, I, I, I, I, I, P]),
    "ldiff_op_im2col_t": (I, [P, P, I, I, I, I, I, I, I, I, I, I, I, P]),
    "ldiff_op_transpose": (I, [P, P, I, I, I, I, P]),
    "ldiff_op_colsum": (I, [P, P, I, I, I, P]),
    "ldiff_op_gn_train_fwd": (I, [P, P, P, P, P, P, I, I, I, I, F, I, P]),
    "ldiff_op_gn_train_bwd": (I, [P, P, P, P, P, P, P, P, P, I, I, I, I, I, P]),
    "ldiff_op_ln_bwd": (I, [P, P, P, P, P, P, I, I, F, P]),
    "ldiff_op_geglu_bwd": (I, [P, P, P, I64, I, P]),
    "ldiff_op_silu": (I, [P, P, I64, P]),
    "ldiff_op_silu_bwd": (I, [P, P, P, I64, P]),
    "ldiff_op_attention_bwd": (I, [P, I, P, I, P, I, P, I, P, P, P, I, I, I, I, I, I64, I64, I64, F, P]),
    "ldiff_op_adamw": (I, [P, P, P, P, I64, F, F, F, F, F, I, P]),
    "ldiff_op_pack_weight": (I, [P, P, I, I, I, I, I, I, P]),
    "ldiff_op_unpack_wgrad": (I, [P, P, I, I, I, I, I, P]),
    "ldiff_op_pack_weight_multi": (I, [P, P, I, I, P]),
    "ldiff_op_adamw_multi": (I, [P, P, P, I64, F, F, F, F, F, I, P]),
    "ldiff_op_infonce": (I, [P, I, I, I64, P, P, P, P, I, P, I, F, P, P, P]),
    "ldiff_stream_create_cu_share": (I, [I, I, P]),
    "ldiff_stream_destroy": (I, [P]),
    "ldiff_vae_set_side_cu_share": (I, [P, I, I]),
    "ldiff_prof_enable": (I, [I]),
    "ldiff_prof_set_filter": (I, [C.c_char_p]),
    "ldiff_prof_collect": (I, [P, I]),
}


class ProfRow(C.Structure):
    _fields_ = [("name", C.c_char * 64), ("launches", C.c_int64), ("ms", C.c_double), ("flops", C.c_double), ("bytes", C.c_double)]


def prof_collect():
    """Rows of the live HIP-event profile: list of dict(name, launches, ms, flops, bytes)."""
    lib = load()
    rows = (ProfRow * 128)()
    n = lib.ldiff_prof_collect(C.cast(rows, C.c_void_p), 128)
    check(n if n < 0 else 0)
    return [dict(name=r.name.decode(), launches=r.launches, ms=r.ms, flops=r.flops, bytes=r.bytes) for r in rows[:min(n, 128)]]

_lib = None


def load() -> C.CDLL:
    """Load the HIP library (once).  Raises RuntimeError with a build hint when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    # torch first: it ships its own HIP runtime, and the library must bind to THAT copy -- loaded before torch, libldiff_hip.so pulls in
    # /opt/rocm's libamdhip64 and the process ends up with two runtimes, of which ours sees no device (seen as "no ROCm-capable device is
    # detected" from ldiff_vae_create when build() and smoke() ran in one process)
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the gfx950 HIP extension is not built. Run `python -c \"import __graft_entry__ as g; g.build()\"` "
            "at the repo root (needs hipcc). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing: fail loudly
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class NonFiniteError(RuntimeError):
    """LDIFF_ERR_NONFINITE: an activation left fp16's range (or turned NaN) somewhere in a graph; the call's results are invalid
    (include/ldiff.h, "Non-finite detection")."""


def check(rc: int) -> None:
    """Map ldiff_status to the exceptions the reference raises: bad shapes -> ValueError, else RuntimeError."""
    if rc == 0:
        return
    msg = load().ldiff_last_error().decode("utf-8", "replace")
    if rc == -1:
        raise ValueError(msg)
    if rc == -4:
        raise NonFiniteError(f"ldiff error {rc}: {msg}")
    raise RuntimeError(f"ldiff error {rc}: {msg}")


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("ldiffusion_amd needs a ROCm GPU (gfx950); torch.cuda.is_available() is False and there is no CPU fallback")


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
