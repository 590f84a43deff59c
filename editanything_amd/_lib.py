"""ctypes binding of the C ABI declared in include/editanything_hip.h.

The product library is ``editanything_amd/csrc/libeditanything_hip.so`` (hipcc,
gfx950).  There is NO fallback: if it is missing or does not load, importing the
compute path raises ``RuntimeError`` -- a silent CPU/eager path would void every
parity and performance claim.  (``bind()`` is also used by the CPU test-suite to
bind tests/emu/libeditanything_emu.so, the host emulation of the same kernels.)
"""
import ctypes as C
import os

EA_OK = 0
ERRORS = {-1: "EA_ERR_BAD_SHAPE", -2: "EA_ERR_BAD_ARG", -3: "EA_ERR_UNSUPPORTED",
          -4: "EA_ERR_WORKSPACE", -5: "EA_ERR_LAUNCH"}
ACT_NONE, ACT_SILU, ACT_GELU, ACT_GEGLU = 0, 1, 2, 3

_vp = C.c_void_p
_i = C.c_int
_ll = C.c_longlong
_f = C.c_float
_sz = C.c_size_t


class Epilogue(C.Structure):
    """Mirror of ``struct ea_epilogue``."""
    _fields_ = [
        ("bias", _vp), ("bias_per_row", C.c_int32),
        ("rowvec", _vp), ("rowvec_ld", C.c_int32), ("rows_per_group", C.c_int32),
        ("act", C.c_int32), ("scale", C.c_float), ("row_scale", _vp),
        ("residual", _vp), ("residual32", _vp), ("ldr", C.c_int32),
        ("out", _vp), ("ldc", C.c_int32), ("out_f32", C.c_int32), ("geglu_block", C.c_int32),
        ("ln_stats", _vp), ("ln_parts", C.c_int32), ("ln_colsum", _vp), ("ln_eps", C.c_float), ("row_stats_out", _vp),
        ("gn_stats_out", _vp), ("gn_rows_per_sample", C.c_int32), ("gn_cpg", C.c_int32),
        ("gn_next_out", _vp), ("gn_next_gamma", _vp), ("gn_next_beta", _vp), ("gn_next_eps", C.c_float), ("gn_next_silu", C.c_int32),
        ("acc_scale_k", C.c_int32), ("acc_scale", C.c_float),
    ]


class Tuning(C.Structure):
    """Mirror of ``struct ea_tuning`` (tools / tests only)."""
    _fields_ = [("force_generic", C.c_int32), ("variant", C.c_int32), ("splits", C.c_int32), ("bn", C.c_int32),
                ("no_register_direct", C.c_int32), ("debug", C.c_int32)]


class ConvSrc(C.Structure):
    """Mirror of ``struct ea_conv_src``."""
    _fields_ = [
        ("x1", _vp), ("c1", C.c_int32), ("x2", _vp), ("c2", C.c_int32), ("x2_add", _vp),
        ("B", C.c_int32), ("Hin", C.c_int32), ("Win", C.c_int32),
        ("ksize", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32), ("ups", C.c_int32),
        ("Hout", C.c_int32), ("Wout", C.c_int32),
    ]


# name -> (restype, argtypes); must list every symbol include/editanything_hip.h declares
SIGNATURES = {
    "ea_version": (_i, []),
    "ea_device_info": (_i, [C.POINTER(_i), C.POINTER(_i), C.c_char_p, _i]),
    "ea_set_tuning": (_i, [C.POINTER(Tuning)]),
    "ea_tools_build": (_i, []),
    "ea_sam_vo_perm": (_i, [_i]),
    "ea_sam_i2t_f16": (_i, [_vp, _ll, _vp, _ll, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _i, _i, _i, _vp]),
    "ea_sam_t2i_f16": (_i, [_vp, _ll, _vp, _vp, _f, _vp, _i, _i, _i, _vp]),
    "ea_sam_upscale_tail_f16": (_i, [_vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "ea_gemm_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "ea_row_stats_parts": (_i, [_i]),
    "ea_gemm_ln_fold_ok": (_i, [_i, _i, _i]),
    "ea_gemm_gn_stats_chunk_rows": (_i, [_i, _i, _i, _i, _i, _i]),
    "ea_gemm_gn_next_ok": (_i, [_i, _i, _i, _i, _i, _i]),
    "ea_groupnorm_apply_f16": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _f, _i, _vp, _i, _vp]),
    "ea_gemm_f16": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _ll, _ll, _ll, _ll, C.POINTER(Epilogue), _vp, _sz, _vp]),
    "ea_conv2d_f16": (_i, [C.POINTER(ConvSrc), _vp, _i, C.POINTER(Epilogue), _vp, _sz, _vp]),
    "ea_gemm_f16_pair": (_i, [_vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, C.POINTER(Epilogue), C.POINTER(Epilogue), _vp, _sz, _vp]),
    "ea_conv2d_f16_pair": (_i, [C.POINTER(ConvSrc), C.POINTER(ConvSrc), _vp, _vp, _i, C.POINTER(Epilogue), C.POINTER(Epilogue),
                                _vp, _sz, _vp]),
    "ea_split3_f32": (_i, [_vp, _vp, _ll, _i, _i, _vp]),
    "ea_layernorm_split3_f32": (_i, [_vp, _vp, _vp, _f, _vp, _i, _i, _vp, _vp]),
    "ea_attention_exact_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _ll, _ll, _ll, _ll, _f, _vp, _vp, _i, _vp]),
    "ea_groupnorm_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "ea_groupnorm_f16": (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _i, _vp, _sz, _vp]),
    "ea_groupnorm_silu_conv3x3": (_i, [C.POINTER(ConvSrc), _vp, _vp, _i, _f, _vp, _vp, _i, C.POINTER(Epilogue),
                                       _vp, _sz, _vp]),
    "ea_layernorm_f16": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _f, _vp]),
    "ea_layernorm_rows_f16": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _f, _vp, _vp]),
    "ea_gather_add_rows_f32": (_i, [_vp, _vp, _vp, _i, _i, _vp]),
    "ea_ln_gemm_f16": (_i, [_vp, _i, _vp, _vp, _f, _vp, _vp, _i, _i, _i, _i, C.POINTER(Epilogue), _vp, _sz, _vp]),
    "ea_attention_f16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _ll, _ll, _ll, _ll, _ll, _ll, _ll, _ll, _f,
                              _vp, _vp, _i, _vp]),
    "ea_sam_window_attn_f16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _ll, _ll, _ll, _ll, _ll, _ll, _ll, _ll, _f, _vp, _vp, _vp]),
    "ea_relpos_tables_f16": (_i, [_vp, _i, _i, _i, _i, _ll, _ll, _vp, _vp, _vp, _vp, _vp]),
    "ea_sam_mask_postprocess": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _i, _f, _f, _vp, _vp, _vp]),
    "ea_sam_mask_postprocess_indexed": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _f, _f, _vp, _vp, _vp]),
    "ea_sam_mask_postprocess_ex": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _f, _f, _vp, _vp, _i, _vp]),
    "ea_sam_upscale_f16": (_i, [_vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "ea_sam_token_self_attn_f16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp]),
    "ea_sam_fold_heads_f16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "ea_sam_unfold_heads_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "ea_sam_id_map": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _f, _i, _vp, _vp]),
    "ea_softmax_rows_f32_f16": (_i, [_vp, _vp, _i, _i, _f, _vp]),
    "ea_cfg_ddim_step": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _ll, _vp]),
    "ea_nchw_f32_to_nhwc_f16": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _f, _f, _vp]),
    "ea_nhwc_f16_to_nchw_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _f, _f, _vp]),
    "ea_silu_f32": (_i, [_vp, _vp, _ll, _vp]),
    "ea_add_f16": (_i, [_vp, _vp, _vp, _ll, _vp]),
    "ea_lincomb_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _ll, _vp]),
    "ea_gather_rows": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _i, _vp]),
}

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libeditanything_hip.so")


class EaError(RuntimeError):
    pass


def check(status, what):
    if status != EA_OK:
        raise EaError(f"{what} failed: {ERRORS.get(status, status)}")


def bind(path):
    """dlopen `path` and attach prototypes for every declared entry point."""
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    return lib


_lib = None


def lib():
    """The product library; raises loudly when it has not been built / cannot be loaded."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). editanything_amd has no CPU / eager fallback.")
        # PyTorch-ROCm ships its own libamdhip64; load it FIRST so that our library's DT_NEEDED resolves to the same HIP
        # runtime instance (two runtimes in one process = stream handles from one are invalid in the other: every
        # launch fails).  torch owns the device memory and the streams, so it is always present on the product path.
        import torch  # noqa: F401
        _lib = bind(LIB_PATH)
    return _lib


def set_tuning(library=None, **fields):
    """Tools / tests: set the calling thread's contraction tuning (`ea_set_tuning`); no fields = reset."""
    library = library or lib()
    if not fields:
        return library.ea_set_tuning(None)
    t = Tuning()
    for k, v in fields.items():
        setattr(t, k, int(v))
    return library.ea_set_tuning(C.byref(t))


def apply_env_tuning(library=None):
    """Tools only: translate the historical EA_GEMM* environment switches into one explicit ea_set_tuning() call."""
    import os
    g = os.environ.get
    return set_tuning(library, force_generic=int(g("EA_GEMM_FORCE", "") == "generic"), variant=int(g("EA_GEMM2_VARIANT", "0") or 0),
                      splits=int(g("EA_GEMM2_SPLITS", "0") or 0), bn=int(g("EA_GEMM2_BN", "0") or 0),
                      no_register_direct=int(g("EA_GEMM2_TR", "1") == "0"), debug=int(g("EA_GEMM2_DEBUG", "0") or 0))
