"""ctypes binding of the C ABI in include/dlc.h (libdlc_hip.so, built in-tree).

There is NO CPU fallback: if the shared library is missing or no MI355X is
visible the product path raises.  The CPU oracle under ``oracle/`` is test
infrastructure and is never imported from here.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdlc_hip.so")

DLC_OK = 0
DLC_ERR_BAD_ARG, DLC_ERR_BAD_SHAPE, DLC_ERR_UNSUPPORTED, DLC_ERR_HIP, DLC_ERR_WORKSPACE = -1, -2, -3, -4, -5
DLC_BF16, DLC_F16, DLC_F32, DLC_F64, DLC_I8 = 0, 1, 2, 3, 4
DLC_ACT_NONE, DLC_ACT_SIGMOID, DLC_ACT_RELU = 0, 1, 2
DLC_B_KN, DLC_B_NK = 0, 1
DLC_MAX_K = 128
DLC_ABI_VERSION = 9          # include/dlc.h; load() refuses a library built from another header
DLC_SELECT_COOP = 1
DLC_SIM_FORCE_F64, DLC_SIM_NO_HOST_SYNC = 1, 2

_vp, _i64, _int, _sz, _dbl, _flt = C.c_void_p, C.c_int64, C.c_int, C.c_size_t, C.c_double, C.c_float

# name -> (restype, argtypes): every function include/dlc.h declares.
SIGNATURES = {
    "dlc_abi_version": (_int, []),
    "dlc_create": (_int, [_int, C.POINTER(_vp)]),
    "dlc_destroy": (_int, [_vp]),
    "dlc_last_error": (C.c_char_p, [_vp]),
    "dlc_status_string": (C.c_char_p, [_int]),
    "dlc_gemm_bias_act": (_int, [_vp, _int, _int, _int, _i64, _i64, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _i64, _vp]),
    "dlc_bias_act": (_int, [_vp, _int, _int, _i64, _i64, _vp, _i64, _vp, _vp, _i64, _vp]),
    "dlc_sdav_encode_workspace_bytes": (_sz, [_i64, C.POINTER(_i64), _int, _int]),
    "dlc_sdav_encode": (_int, [_vp, _int, _i64, _int, C.POINTER(_i64), _vp, C.POINTER(_vp), C.POINTER(_vp), _vp, _vp,
                               _sz, _vp]),
    "dlc_sdav_split_panels_bytes": (_sz, [_int, C.POINTER(_i64)]),
    "dlc_sdav_split_prepare": (_int, [_vp, _int, C.POINTER(_i64), C.POINTER(_vp), _vp, _sz, _vp]),
    "dlc_sdav_encode_split_workspace_bytes": (_sz, [_i64, C.POINTER(_i64), _int]),
    "dlc_sdav_encode_split": (_int, [_vp, _i64, _int, C.POINTER(_i64), _vp, _vp, C.POINTER(_vp), _vp, _vp, _sz, _vp]),
    "dlc_sdav_train_workspace_bytes": (_sz, [_i64, _i64, C.POINTER(_i64), _int, _int]),
    "dlc_sdav_train_step": (_int, [_vp, _int, _i64, _i64, _int, C.POINTER(_i64), _vp, C.POINTER(_vp), C.POINTER(_vp),
                                  C.POINTER(_vp), _vp, _dbl, _dbl, _dbl, _dbl, _vp, _vp, _sz, _vp]),
    "dlc_random_mask_f64": (_int, [_vp, _vp, _i64, _i64, C.c_uint64, C.c_uint64, _vp]),
    "dlc_rgb_to_gray_u8": (_int, [_vp, _vp, _i64, _vp, _vp]),
    "dlc_harris_keypoints_workspace_bytes": (_sz, [_i64, _int, _int]),
    "dlc_harris_keypoints_u8": (_int, [_vp, _vp, _i64, _int, _int, _int, _vp, _vp, _vp, _vp, _sz, _vp]),
    "dlc_extract_patches": (_int, [_vp, _vp, _i64, _int, _int, _vp, _int, _int, _int, _vp, _vp]),
    "dlc_im2col_nhwc_f64": (_int, [_vp, _vp, _i64, _int, _int, _int, _int, _int, _int, _int, _int, _int, _int, _vp, _vp]),
    "dlc_conv2d_nhwc_f64": (_int, [_vp, _vp, _i64, _int, _int, _int, _vp, _vp, _int, _int, _int, _int, _int, _int, _int,
                                  _int, _int, _vp, _vp]),
    "dlc_space_to_depth_nhwc_f64": (_int, [_vp, _vp, _i64, _int, _int, _int, _int, _vp, _vp]),
    "dlc_maxpool3x3s2_nhwc_f64": (_int, [_vp, _vp, _i64, _int, _int, _int, _vp, _vp]),
    "dlc_minmax_quant_gather_i8": (_int, [_vp, C.POINTER(_vp), C.POINTER(_i64), _int, _i64, _vp, _i64, _vp, _vp, _vp]),
    "dlc_cnnvtl_frame_minmax_init": (_int, [_vp, _vp, _i64, _vp]),
    "dlc_conv2d_nhwc_f64_stats": (_int, [_vp, _vp, _i64, _int, _int, _int, _vp, _vp, _int, _int, _int, _int, _int, _int, _int,
                                         _int, _int, _vp, _vp, _vp]),
    "dlc_quant_gather_i8": (_int, [_vp, C.POINTER(_vp), C.POINTER(_i64), _int, _i64, _vp, _i64, _vp, _vp, _vp, _vp]),
    "dlc_sdav_range_words": (_sz, [_i64]),
    "dlc_sdav_distinctive_score": (_int, [_vp, _vp, _i64, _i64, _dbl, _dbl, _vp, _vp, _vp]),
    "dlc_sdav_similarity_workspace_bytes": (_sz, [_i64, _i64, _i64, _int, _i64]),
    "dlc_sdav_similarity_matrix": (_int, [_vp, _vp, _i64, _i64, _i64, _vp, _dbl, _dbl, _vp, _vp, _int, _i64, _vp, _vp, _vp,
                                         _vp, _sz, _vp]),
    "dlc_topk_rows_f64": (_int, [_vp, _vp, _i64, _i64, _i64, _i64, _int, _vp, _vp, _vp, _vp]),
    "dlc_sdav_stream_state_bytes": (_sz, [_i64, _i64, _i64]),
    "dlc_sdav_stream_init": (_int, [_vp, _vp, _sz, _i64, _i64, _i64, _dbl, _dbl, _vp, _vp]),
    "dlc_sdav_stream_append": (_int, [_vp, _vp, _sz, _i64, _i64, _i64, _vp, _i64, _i64, _vp, _vp]),
    "dlc_sdav_stream_query": (_int, [_vp, _vp, _sz, _i64, _i64, _i64, _vp, _i64, _vp, _dbl, _dbl, _vp, _vp, _vp]),
    "dlc_sdav_stream_query_batch_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "dlc_sdav_stream_query_batch": (_int, [_vp, _vp, _sz, _i64, _i64, _i64, _vp, _i64, _i64, _vp, _dbl, _dbl, _vp, _i64, _vp,
                                           _vp, _sz, _vp]),
    "dlc_sdav_stream_query_batch_staged": (_int, [_vp, _vp, _sz, _i64, _i64, _i64, _vp, _i64, _i64, _vp, _dbl, _dbl, _vp, _i64,
                                                  _vp, _vp, _sz, _int, _vp]),
    "dlc_cnnvtl_distance_matrix": (_int, [_vp, _vp, _i64, _i64, _i64, _vp, _vp]),
    "dlc_l2_normalize_rows": (_int, [_vp, _int, _vp, _i64, _i64, _i64, _int, _int, _vp, _i64, _vp]),
    "dlc_cosine_topk_workspace_bytes": (_sz, [_i64, _i64, _i64, _int]),
    "dlc_cosine_topk": (_int, [_vp, _int, _vp, _i64, _i64, _vp, _i64, _i64, _i64, _int, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _sz,
                              _vp]),
    "dlc_max_row_norm": (_int, [_vp, _int, _vp, _i64, _i64, _i64, _vp, _vp]),
    "dlc_cosine_tau_scale": (_int, [_vp, _int, _vp, _i64, _i64, _i64, _vp, _vp, _vp]),
    "dlc_cosine_topk_older": (_int, [_vp, _int, _vp, _i64, _i64, _vp, _i64, _i64, _i64, _int, _i64, _i64, _vp, _vp, _vp, _vp, _vp,
                                    _vp, _sz, _vp]),
    "dlc_cosine_score_error_bound": (_dbl, [_i64, _i64, _i64, _int]),
    "dlc_cosine_score_error_bound_any_plan": (_dbl, [_i64]),
    "dlc_cosine_score_groups": (_int, [_vp, _int, _vp, _i64, _i64, _vp, _i64, _i64, _i64, _int, _vp, _sz, _vp]),
    "dlc_cosine_select_topk": (_int, [_vp, _int, _vp, _i64, _i64, _vp, _i64, _i64, _i64, _int, _i64, _vp, _vp, _vp, _vp, _vp, _vp,
                                     _sz, _int, _vp]),
    "dlc_cosine_groups_per_query": (_int, [_int]),
    "dlc_cosine_select_groups": (_int, [_vp, _int, _vp, _i64, _i64, _vp, _i64, _i64, _i64, _int, _vp, _sz, _vp, _vp, _int,
                                       _vp]),
    "dlc_cosine_rescore_topk": (_int, [_vp, _int, _vp, _i64, _i64, _vp, _i64, _i64, _i64, _int, _i64, _vp, _vp, _vp, _int,
                                      _vp, _vp, _vp, _vp, _int, _vp]),
    "dlc_cosine_exhaustive_topk": (_int, [_vp, _int, _vp, _i64, _i64, _vp, _i64, _i64, _i64, _int, _i64, _vp, _i64, _dbl,
                                         _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "dlc_topk_merge_strided": (_int, [_vp, _vp, _i64, _vp, _i64, _int, _i64, _int, _vp, _dbl, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dlc_topk_merge": (_int, [_vp, _vp, _vp, _int, _i64, _int, _vp, _vp, _vp, _vp]),
    "dlc_topk_keep_older": (_int, [_vp, _vp, _vp, _i64, _int, _i64, _int, _vp, _vp, _vp]),
    "dlc_cosine_scores_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "dlc_cosine_scores": (_int, [_vp, _int, _vp, _i64, _i64, _vp, _i64, _i64, _i64, _vp, _i64, _vp, _sz, _vp]),
    "dlc_host_to_device": (_int, [_vp, _vp, _vp, _sz, _vp]),
    "dlc_device_to_host": (_int, [_vp, _vp, _vp, _sz, _vp]),
    "dlc_set_host_threads": (_int, [_vp, _int]),
    "dlc_set_scratch": (_int, [_vp, _vp, _sz]),
    "dlc_set_profiling": (_int, [_vp, _int]),
    "dlc_profile_gemm_ms": (_int, [_vp, C.POINTER(_flt), _int]),
}

_lib = None


class DlcError(RuntimeError):
    """A HIP-side failure (DLC_ERR_HIP / DLC_ERR_WORKSPACE) reported by the C ABI."""


def load():
    """dlopen libdlc_hip.so and declare every prototype.  Raises ImportError
    (loudly) when the extension has not been built -- never falls back."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "deeploopcloser_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C deeploopcloser_amd/csrc`.  There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.dlc_abi_version() != DLC_ABI_VERSION:
        raise ImportError("deeploopcloser_amd: %s has ABI %d, these bindings are for ABI %d -- rebuild it"
                          % (LIB_PATH, lib.dlc_abi_version(), DLC_ABI_VERSION))
    _lib = lib
    return lib


def raise_for_status(lib, ctx, rc):
    if rc == DLC_OK:
        return
    msg = lib.dlc_last_error(ctx).decode() if ctx else ""
    text = "%s: %s" % (lib.dlc_status_string(rc).decode(), msg)
    if rc in (DLC_ERR_BAD_ARG, DLC_ERR_BAD_SHAPE, DLC_ERR_UNSUPPORTED):
        raise ValueError(text)
    raise DlcError(text)
