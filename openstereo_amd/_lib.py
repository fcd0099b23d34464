"""ctypes binding of the C-ABI library (include/openstereo_amd.h).

The product path has NO fallback: if the gfx950 library is missing or a call fails, a
RuntimeError is raised.  Nothing in here imports the CPU oracle.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OSA_LIB_PATH") or os.path.join(_HERE, "lib", "libopenstereo_amd.so")   # override: A/B experiments only

_lock = threading.Lock()
_lib = None

c_fp = C.c_void_p      # device float*
c_i = C.c_int
c_f = C.c_float
c_st = C.c_void_p      # hipStream_t
c_ll = C.c_longlong


class F16x3Ranges(C.Structure):
    """osa_f16x3_ranges (include/openstereo_amd.h): device pointers of the operands' range blocks."""
    _fields_ = [("x_meta", C.c_void_p), ("residual_meta", C.c_void_p), ("redir_meta", C.c_void_p),
                ("y_meta", C.c_void_p), ("bound_coef", C.c_void_p), ("redir_bound_coef", C.c_void_p), ("weight_scale", C.c_void_p)]


c_rng = C.POINTER(F16x3Ranges)


class NhwcRef(C.Structure):
    """osa_nhwc_ref: an NHWC tensor operand with its own channel stride (elements) and element type (fp32 / fp16)."""
    _fields_ = [("ptr", C.c_void_p), ("cs", C.c_int), ("f16", C.c_int)]


c_ref = C.POINTER(NhwcRef)

# name -> (restype, argtypes).  Mirrors include/openstereo_amd.h one to one; tests check that
# every symbol declared in the header is listed here and exported by the .so.
SIGNATURES = {
    "osa_abi_version": (c_i, []),
    "osa_last_error": (C.c_char_p, []),
    "osa_target_arch": (C.c_char_p, []),
    "osa_build_volume_f32": (c_i, [c_fp, c_fp, c_i, c_i, c_fp, c_fp, c_i, c_fp, c_i, c_i, c_i,
                                   c_i, c_i, c_i, c_i, c_i, c_fp, c_st]),
    "osa_build_volume_nhwc_f32": (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_fp, c_fp, c_i, c_i, c_fp, c_i, c_i,
                                        c_i, c_i, c_i, c_i, c_i, c_fp, c_st]),
    "osa_corr_volume_f32": (c_i, [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_st]),
    "osa_ncdhw_to_ndhwc_f32": (c_i, [c_fp, c_fp, c_i, c_i, c_ll, c_i, c_i, c_st]),
    "osa_ndhwc_to_ncdhw_f32": (c_i, [c_fp, c_fp, c_i, c_i, c_ll, c_i, c_i, c_st]),
    "osa_conv3d_packed_floats": (C.c_size_t, [c_i, c_i, c_i, c_i, c_i]),
    "osa_conv3d_pack_f32": (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_st]),
    "osa_deconv3d_packed_floats": (C.c_size_t, [c_i, c_i, c_i]),
    "osa_deconv3d_pack_f32": (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_i, c_st]),
    "osa_conv3d_ndhwc_f32": (c_i, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp,
                                   c_i, c_i, c_i, c_i, c_i, c_i,
                                   c_i, c_i, c_i,
                                   c_i, c_i, c_i, c_i,
                                   c_i, c_i, c_i,
                                   c_i, c_i, c_i,
                                   c_fp, c_i,
                                   c_i, c_f, c_st]),
    "osa_deconv3d_ndhwc_f32": (c_i, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp,
                                     c_i, c_i, c_i, c_i, c_i, c_i,
                                     c_i, c_i, c_i,
                                     c_i, c_i, c_i,
                                     c_fp, c_i,
                                     c_i, c_f, c_st]),
    "osa_conv3d_pack_f16x3": (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_f, c_st]),
    "osa_deconv3d_pack_f16x3": (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_i, c_f, c_st]),
    "osa_conv3d_ndhwc_f16x3": (c_i, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp,
                                     c_i, c_i, c_i, c_i, c_i, c_i,
                                     c_i, c_i, c_i,
                                     c_i, c_i, c_i, c_i,
                                     c_i, c_i, c_i,
                                     c_i, c_i, c_i,
                                     c_fp, c_i,
                                     c_i, c_f, c_f, c_rng, c_st]),
    "osa_deconv3d_ndhwc_f16x3": (c_i, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp,
                                       c_i, c_i, c_i, c_i, c_i, c_i,
                                       c_i, c_i, c_i,
                                       c_i, c_i, c_i,
                                       c_fp, c_i,
                                       c_i, c_f, c_f, c_rng, c_st]),
    "osa_conv3d_pack_ex": (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_st]),
    "osa_pair_volume_f32": (c_i, [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_st]),
    "osa_conv3d_wgrad_f32": (c_i, [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i,
                                   c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_st]),
    "osa_conv3d_wgrad_workspace_bytes": (C.c_size_t, [c_i] * 20),
    "osa_conv3d_wgrad_ws_f32": (c_i, [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i,
                                      c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_fp, C.c_size_t, c_st]),
    "osa_conv3d_wgrad_f16x3_workspace_bytes": (C.c_size_t, [c_i] * 20),
    "osa_conv3d_wgrad_ws_f16x3": (c_i, [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i,
                                        c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_fp, c_fp, c_fp, C.c_size_t, c_st]),
    "osa_conv3d_wgrad_ws_f16": (c_i, [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i,
                                      c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_fp, c_fp, c_i, c_i, c_fp, C.c_size_t, c_st]),
    "osa_conv3d_small_co_ndhwc_f32": (c_i, [c_fp, c_fp, c_fp, c_fp, c_fp,
                                            c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i,
                                            c_i, c_i, c_i, c_i, c_i, c_i, c_st]),
    "osa_conv3d_small_co_packed_floats": (C.c_size_t, [c_i, c_i, c_i, c_i, c_i]),
    "osa_conv3d_small_co_pack_f32": (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_st]),
    "osa_conv3d_small_co_packed_ndhwc_f32": (c_i, [c_fp, c_fp, c_fp, c_fp, c_fp,
                                            c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i,
                                            c_i, c_i, c_i, c_i, c_i, c_i, c_st]),
    "osa_build_volume_bwd_f32": (c_i, [c_fp, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_st]),
    "osa_softargmin_bwd_f32": (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_i, c_st]),
    "osa_softmax_softargmin_bwd_f32": (c_i, [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_st]),
    "osa_upsample_softargmin_bwd_f32": (c_i, [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_st]),
    "osa_amax_f32": (c_i, [c_fp, C.c_longlong, c_fp, c_st]),
    "osa_upsample_softargmin_bwd_workspace_bytes": (C.c_size_t, [c_i] * 4),
    "osa_upsample_softargmin_bwd_ws_f32": (c_i, [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_fp, C.c_size_t, c_st]),
    "osa_deconv3d_redir_ndhwc_f32": (c_i, [c_fp, c_fp, c_fp, c_fp, c_fp,
                                           c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i,
                                           c_i, c_i, c_i,
                                           c_fp, c_i, c_i, c_fp, c_fp, c_fp,
                                           c_i, c_f, c_st]),
    "osa_deconv3d_redir_ndhwc_f16x3": (c_i, [c_fp, c_fp, c_fp, c_fp, c_fp,
                                             c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i,
                                             c_i, c_i, c_i,
                                             c_fp, c_i, c_i, c_fp, c_fp, c_fp, c_f,
                                             c_i, c_f, c_f, c_rng, c_st]),
    "osa_deconv2d_packed_floats": (C.c_size_t, [c_i, c_i, c_i]),
    "osa_deconv2d_pack_f32": (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_i, c_st]),
    "osa_deconv2d_pack_f16x3": (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_i, c_f, c_st]),
    "osa_deconv2d_nhwc_f32": (c_i, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp,
                                    c_i, c_i, c_i, c_i, c_i,
                                    c_i, c_i, c_i,
                                    c_i, c_i, c_i,
                                    c_fp, c_i, c_i, c_f, c_st]),
    "osa_deconv2d_nhwc_f16x3": (c_i, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp,
                                      c_i, c_i, c_i, c_i, c_i,
                                      c_i, c_i, c_i,
                                      c_i, c_i, c_i,
                                      c_fp, c_i, c_i, c_f, c_f, c_rng, c_st]),
    "osa_dwconv2d_pack_f32": (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_st]),
    "osa_dwconv2d_nhwc_f32": (c_i, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp,
                                    c_i, c_i, c_i, c_i, c_i, c_i, c_i,
                                    c_i, c_i, c_i, c_i, c_i, c_i, c_i,
                                    c_i, c_fp, c_st]),
    "osa_dwconv2d_nhwc_f16io": (c_i, [c_fp, c_i, c_fp, c_fp, c_fp, c_fp, c_i] + [c_i] * 11 + [c_i, c_fp, c_st]),
    "osa_gru_combine_f32": (c_i, [c_fp, c_fp, c_fp, c_fp, c_ll, c_i, c_i, c_i, c_i, c_i, c_fp, c_st]),
    "osa_gru_gates_rz_fwd": (c_i, [c_ref, c_fp, c_fp, c_ref, c_ref, c_ref, c_ref, c_ref, c_ll, c_i, c_st]),
    "osa_gru_gates_rz_bwd": (c_i, [c_ref, c_fp, c_fp, c_ref, c_ref, c_ref, c_ref, c_ref, c_ref, c_ref, c_ll, c_i, c_st]),
    "osa_gru_gates_q_fwd": (c_i, [c_ref, c_ref, c_fp, c_ref, c_ref, c_ref, c_ll, c_i, c_st]),
    "osa_gru_gates_q_bwd": (c_i, [c_ref, c_ref, c_fp, c_ref, c_ref, c_ref, c_ref, c_ref, c_ref, c_ll, c_i, c_st]),
    "osa_disp_update_f32": (c_i, [c_fp, c_fp, c_i, c_fp, c_fp, c_i, c_ll, c_fp, c_fp, c_st]),
    "osa_conv3d_pack_ex_auto": (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_fp, c_fp, c_st]),
    "osa_deconv3d_pack_f16x3_auto": (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_i, c_fp, c_fp, c_st]),
    "osa_deconv2d_pack_f16x3_auto": (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_i, c_fp, c_fp, c_st]),
    "osa_cat_fms_f32": (c_i, [c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_st]),
    "osa_pool2x_nhwc_f32": (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_fp, c_fp, c_st]),
    "osa_resize_bilinear_nhwc_f32": (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_fp, c_fp, c_st]),
    "osa_context_upsample_logits_f32": (c_i, [c_fp, c_fp, c_i, C.POINTER(c_ll), c_fp, c_i, c_i, c_i, c_i, c_f, c_st]),
    "osa_context_upsample_logits_bwd_f32": (c_i, [c_fp, c_fp, c_i, C.POINTER(c_ll), c_fp, c_fp, c_fp, C.POINTER(c_ll), c_fp, c_i, c_i, c_i, c_i, c_f, c_st]),
    "osa_context_upsample_f32": (c_i, [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_f, c_st]),
    "osa_allpairs_corr_f32": (c_i, [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_st]),
    "osa_geo_rows_f32": (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_st]),
    "osa_avgpool_rows_f32": (c_i, [c_fp, c_fp, c_ll, c_i, c_st]),
    "osa_geo_lookup_f32": (c_i, [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(c_i), C.POINTER(c_i), c_i,
                                 c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_st]),
    "osa_geo_lookup_nhwc_f32": (c_i, [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(c_i), C.POINTER(c_i), c_i,
                                      c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_st]),
    "osa_geo_lookup_bwd_acc_f32": (c_i, [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(c_i), C.POINTER(c_i), c_i,
                                         c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_st]),
    "osa_geo_lookup_bwd_f32": (c_i, [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(c_i), C.POINTER(c_i), c_i,
                                     c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_st]),
    "osa_preprocess_pair_f32": (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, C.POINTER(c_f), C.POINTER(c_f), c_fp, c_i, c_st]),
    "osa_conv3d_pack_f16": (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_st]),
    "osa_deconv3d_pack_f16": (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_i, c_st]),
    "osa_deconv2d_pack_f16": (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_i, c_st]),
    "osa_conv3d_ndhwc_f16": (c_i, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp,
                                   c_i, c_i, c_i, c_i, c_i, c_i,
                                   c_i, c_i, c_i,
                                   c_i, c_i, c_i, c_i,
                                   c_i, c_i, c_i,
                                   c_i, c_i, c_i,
                                   c_fp, c_i,
                                   c_i, c_f, c_st]),
    "osa_deconv3d_ndhwc_f16": (c_i, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp,
                                     c_i, c_i, c_i, c_i, c_i, c_i,
                                     c_i, c_i, c_i,
                                     c_i, c_i, c_i,
                                     c_fp, c_i,
                                     c_i, c_f, c_st]),
    "osa_deconv2d_nhwc_f16": (c_i, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp,
                                    c_i, c_i, c_i, c_i, c_i,
                                    c_i, c_i, c_i,
                                    c_i, c_i, c_i,
                                    c_fp, c_i, c_i, c_f, c_st]),
    "osa_instnorm_workspace_floats": (C.c_size_t, [c_i, c_ll, c_i]),
    "osa_instnorm_nhwc_f32": (c_i, [c_fp, c_fp, c_i, c_ll, c_i, c_i, c_i, c_f, c_i, c_f, c_fp, c_fp, c_st]),
    "osa_conv3d_wgrad_ws_multi": (c_i, [c_i, c_fp, c_fp, c_i, c_fp] + [c_i] * 22 + [c_fp, c_fp, c_i, c_i, c_fp, C.c_size_t, c_st]),
    "osa_channel_sums_workspace_bytes": (C.c_size_t, [c_ll, c_i]),
    "osa_channel_affine": (c_i, [c_fp, c_i, c_i, c_fp, c_i, c_i, c_fp, c_fp, c_fp, c_fp, c_i, c_ll, c_i, c_i, c_st]),
    "osa_channel_sums_multi": (c_i, [c_fp, c_i, c_i, c_i, c_ll, c_i, c_fp, c_fp, C.c_size_t, c_st]),
    "osa_channel_sums": (c_i, [c_fp, c_i, c_i, c_fp, c_i, c_i, c_fp, c_fp, c_fp, c_i, c_ll, c_i, c_fp, c_fp, C.c_size_t, c_st]),
    "osa_conv3d_march_launches": (c_ll, []),
    "osa_conv3d_march_s2_launches": (c_ll, []),
    "osa_conv_b_ring_mask": (c_i, [c_i]),
    "osa_conv_b_ring_launches": (c_ll, []),
    "osa_volume_walk_step": (c_i, [c_i]),
    "osa_volume_walk_launches": (c_ll, []),
    "osa_build_volume_nhwc_split_eligible": (c_i, [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i]),
    "osa_build_volume_nhwc_split_f16x3": (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_fp, c_fp, c_i, c_i, c_fp, c_i, c_i,
                                                c_i, c_i, c_i, c_i, c_i, c_fp, c_fp, c_fp, c_st]),
    "osa_softargmin_f32": (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_i, c_st]),
    "osa_softmax_softargmin_f32": (c_i, [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_st]),
    "osa_upsample_softargmin_f32": (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_st]),
}


class EngineError(RuntimeError):
    pass


def load():
    """dlopen the library (once) and declare every prototype. Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise EngineError(
                f"{LIB_PATH} is missing: build it with `python -m openstereo_amd.build` "
                "(hipcc, gfx950). There is no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(lib, name)   # AttributeError if the .so does not export it
            except AttributeError:
                if os.environ.get("OSA_LIB_PATH"):      # A/B experiment builds may predate an entry point; the shipped library may not
                    continue
                raise
            fn.restype = res
            fn.argtypes = args
        if lib.osa_abi_version() != abi_version(lib):
            raise EngineError(f"ABI version mismatch: library reports {lib.osa_abi_version()}, include/openstereo_amd.h declares {abi_version(lib)}")
        if os.environ.get("OSA_B_RING_MASK") and hasattr(lib, "osa_conv_b_ring_mask"):
            lib.osa_conv_b_ring_mask(int(os.environ["OSA_B_RING_MASK"], 0))     # A/B runs: which tiles take their weights through the LDS ring
        if os.environ.get("OSA_VOL_WALK") and hasattr(lib, "osa_volume_walk_step"):
            lib.osa_volume_walk_step(int(os.environ["OSA_VOL_WALK"]))               # A/B runs: 0 = chunked volume builder, 4 / 8 = d-walking form
        _lib = lib
    return _lib


_abi = None


def abi_version(lib=None) -> int:
    """OSA_ABI_VERSION of include/openstereo_amd.h -- the ONE place the number lives (the loaders and the tests read it from here)."""
    global _abi
    if _abi is None:
        import re
        hdr = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "include", "openstereo_amd.h")
        try:
            with open(hdr) as f:
                _abi = int(re.search(r"#define\s+OSA_ABI_VERSION\s+(\d+)", f.read()).group(1))
        except (OSError, AttributeError):
            # a copy of the package without the repository's include/ directory (wheel / site-packages): the library was compiled against the
            # header, so what it reports IS the version
            h = lib if lib is not None else _lib
            if h is None:
                return -1
            _abi = int(h.osa_abi_version())
    return _abi


CALLS = {}          # name -> number of calls that went through ctypes (the C++ extension's dispatcher calls do not pass here): tests / DESIGN.md count them


def call(name: str, *args):
    """Invoke an int-returning entry point; turn a non-zero status into EngineError."""
    lib = load()
    CALLS[name] = CALLS.get(name, 0) + 1
    rc = getattr(lib, name)(*args)
    if rc != 0:
        msg = lib.osa_last_error()
        raise EngineError(f"{name} failed ({rc}): {msg.decode() if msg else '?'}")
    return rc
