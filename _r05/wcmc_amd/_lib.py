"""ctypes binding of ``libwcmc_hip.so`` (C ABI: ``include/wcmc_hip.h``).

The library is the product; there is no CPU fallback.  ``lib()`` raises if the
shared object is missing (run ``python -c 'import __graft_entry__ as g; g.build()'``
or ``make -C wcmc_amd/csrc``).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libwcmc_hip.so")
if os.environ.get("WCMC_DEBUG_LIB") == "1":      # `make -C wcmc_amd/csrc debug`: + timing-only ablation / stamp instances
    LIB_PATH = os.path.join(_HERE, "libwcmc_hip_debug.so")
if os.environ.get("WCMC_LIB_AB"):                # A/B timing of two builds (scripts/): a file name next to the product library
    LIB_PATH = os.path.join(_HERE, os.path.basename(os.environ["WCMC_LIB_AB"]))

_c = ctypes
P, I, L, F, D, Z = _c.c_void_p, _c.c_int, _c.c_int64, _c.c_float, _c.c_double, _c.c_size_t

# name -> (restype, [argtypes]); mirrors include/wcmc_hip.h one to one
SIGNATURES = {
    "wcmc_abi_version": (I, []),
    "wcmc_last_error": (_c.c_char_p, []),
    "wcmc_to_nhwc": (I, [P, L, L, L, L, P, L, L, L, I, I, I, I, P]),
    "wcmc_from_nhwc": (I, [P, L, L, L, P, L, L, L, L, I, I, I, I, P]),
    "wcmc_conv2d_packed_elems": (Z, [I, I, I]),
    "wcmc_conv2d_pack_weight": (I, [P, P, I, I, I, I, P]),
    "wcmc_conv2d_igemm": (I, [P, L, L, L, I, I, I, I, P, P, P, L, L, L, I, I, I, I, F,
                              P, L, L, L, I, F, P]),
    "wcmc_conv2d_wgrad_workspace_bytes": (Z, [I, I, I, I, I, I]),
    "wcmc_conv2d_wgrad": (I, [P, L, L, L, I, I, I, I, P, L, L, L, I, I, I, P, P, P, Z, P]),
    "wcmc_split_elems": (Z, [I, I, I, I]),
    "wcmc_split_bf16": (I, [P, L, L, L, P, I, I, I, I, P]),
    "wcmc_split_from_nchw": (I, [P, L, L, L, L, P, I, I, I, I, P]),
    "wcmc_split_gated_bf16": (I, [P, L, L, L, P, L, L, L, I, F, P, I, I, I, I, P]),
    "wcmc_cat_broadcast_split": (I, [P, L, L, L, P, L, L, L, P, I, I, I, I, I, I, P]),
    "wcmc_cat_upsample_split": (I, [P, L, L, L, P, L, L, L, P, I, I, I, I, I, P]),
    "wcmc_add_broadcast_split": (I, [P, L, L, L, P, L, L, L, F, P, I, I, I, I, I, P]),
    "wcmc_split_dy_colsum_bf16": (I, [P, L, L, L, P, L, L, L, I, F, P, L, L, L, I, F, P, P, I, I, I, I, P]),
    "wcmc_conv2d_packed_elems_bf16x3": (Z, [I, I, I, I]),
    "wcmc_conv2d_pack_weight_bf16x3": (I, [P, P, I, I, I, I, P]),
    "wcmc_conv2d_pack_chain_bf16x3": (I, [I, P, P, P, P, P, I, P]),
    "wcmc_conv2d_igemm_bf16x3": (I, [P, I, I, I, I, P, P, P, L, L, L, P, I, I, I, I, F, P, I, F, P, P, P, I, P]),
    "wcmc_conv2d_out_f16_supported": (I, [I, I, I]),
    "wcmc_split_to_f16_elems": (Z, [I, I, I, I]),
    "wcmc_split_to_f16": (I, [P, I, I, I, I, P, P]),
    "wcmc_conv2d_out_f16": (I, [P, I, I, I, I, P, P, P, L, L, L, I, I, I, P]),
    "wcmc_conv1x1_pair_supported": (I, [I, I, I]),
    "wcmc_conv1x1_pair_bf16x3": (I, [P, I, I, I, I, P, P, I, I, F, P, P, P, I, F, P, P, P, I, I, F, P, L, L, L, P]),
    "wcmc_conv2d_igemm_colsum_elems": (Z, [I, I, I, I]),
    "wcmc_colsum_finish": (I, [P, I, I, I, I, P, P]),
    "wcmc_conv2d_wgrad_bf16x3_workspace_bytes": (Z, [I, I, I, I, I, I]),
    "wcmc_conv2d_wgrad_reduce_multi": (I, [I, P, P, P, P, P, P, P, P, P, P, I, P]),
    "wcmc_conv2d_wgrad_bf16x3": (I, [P, I, I, I, I, P, I, I, I, P, P, P, Z, I, P, I, P]),
    "wcmc_act_backward": (I, [P, L, L, L, P, L, L, L, P, L, L, L, I, I, I, I, I, F, P]),
    "wcmc_kernel_apply_fwd": (I, [P, L, L, L, P, L, L, L, L, P, L, L, L, L, P, I, I, I, I, I, P]),
    "wcmc_kernel_apply_bwd": (I, [P, L, L, L, P, L, L, L, L, P, L, L, L, L, P, L, L, L, L, P,
                                  P, L, L, L, P, I, I, I, I, I, P]),
    "wcmc_recombine_fwd": (I, [P, L, L, L, L, P, L, L, L, L, P, L, L, L, L, P, I, I, I, I, P]),
    "wcmc_image_loss_workspace_bytes": (Z, []),
    "wcmc_image_loss_fwd": (I, [P, L, L, L, L, P, L, L, L, L, F, P, P, P, Z, I, I, I, I, P]),
    "wcmc_l1_mean_bwd": (I, [P, L, L, L, L, P, L, L, L, L, P, P, I, I, I, I, P]),
    "wcmc_recombine_bwd": (I, [P, P, L, L, L, L, P, L, L, L, L, P, P, I, I, I, I, P]),
    "wcmc_maxpool2_fwd": (I, [P, L, L, L, P, L, L, L, I, I, I, I, P]),
    "wcmc_maxpool2_bwd": (I, [P, L, L, L, P, L, L, L, P, L, L, L, I, I, I, I, P]),
    "wcmc_maxpool2_bwd_add": (I, [P, L, L, L, P, L, L, L, P, L, L, L, P, L, L, L, I, I, I, I, P]),
    "wcmc_upsample2_fwd": (I, [P, L, L, L, P, L, L, L, I, I, I, I, P]),
    "wcmc_upsample2_bwd": (I, [P, L, L, L, P, L, L, L, I, I, I, I, P]),
    "wcmc_sample_cat_fwd": (I, [P, L, L, L, L, L, P, L, L, L, L, L, P, I, I, I, I, I, I, P]),
    "wcmc_spp_reduce": (I, [P, L, L, L, P, L, L, L, I, I, I, I, I, F, P]),
    "wcmc_spp_broadcast": (I, [P, L, L, L, P, L, L, L, I, I, I, I, I, F, I, P]),
    "wcmc_pbuffer_cat_fwd": (I, [P, L, L, L, L, P, L, L, L, L, L, P, L, L, L, I, I, I, I, I, I, P]),
    "wcmc_pbuffer_cat_bwd": (I, [P, L, L, L, P, L, L, L, L, L, I, I, I, I, I, I, P]),
    "wcmc_feature_mse_workspace_bytes": (Z, [I, I, I, I, I]),
    "wcmc_feature_mse_fwd": (I, [P, L, L, L, L, L, P, L, L, L, L, P, P, P, P, Z, I, I, I, I, I, P]),
    "wcmc_feature_mse_bwd": (I, [P, L, L, L, L, L, P, P, P, P, P, Z, I, I, I, I, I, P]),
    "wcmc_grs_fwd": (I, [P, L, L, L, L, L, P, L, L, L, L, P, P, F, P, P, Z, I, I, I, I, I, P]),
    "wcmc_grs_bwd": (I, [P, L, L, L, L, L, P, P, P, P, P, Z, I, I, I, I, I, P]),
    "wcmc_random_permutation": (I, [P, L, ctypes.c_uint64, P]),
    "wcmc_permutation_key": (ctypes.c_uint64, [ctypes.c_uint64, ctypes.c_uint64, I]),
    "wcmc_random_permutation_dev": (I, [P, L, P, I, P]),
    "wcmc_step_counter_advance": (I, [P, P]),
    "wcmc_final2_supported": (I, [I, I, I, I, L]),
    "wcmc_final2_bwd_workspace_bytes": (Z, []),
    "wcmc_final2_fwd": (I, [P, I, P, I, I, I, L, P, P, P, P, I, P, P]),
    "wcmc_final2_bwd": (I, [P, I, P, I, I, I, L, P, P, P, P, I, P, P, P, P, P, P, P, P, P, P, Z, P]),
    "wcmc_embed3_supported": (I, [I, I, I, I]),
    "wcmc_embed3_bwd_workspace_bytes": (Z, []),
    "wcmc_embed3_fwd": (I, [P, L, I, P, P, P, P, P, P, P, P]),
    "wcmc_embed3_mean_supported": (I, [I, L]),
    "wcmc_embed3_mean_fwd": (I, [P, L, I, P, P, P, P, P, P, P, P, I, L, P]),
    "wcmc_embed3_bwd": (I, [P, L, I, P, P, P, P, P, P, P, I, P, I, I, L, F, P, P, P, P, P, P, P, Z, P]),
    "wcmc_image_loss2_fwd": (I, [I, P, L, L, L, L, P, L, L, L, L, F, P, P, Z, I, I, I, I, P]),
    "wcmc_image_loss2_bwd": (I, [I, P, L, L, L, L, P, L, L, L, L, F, P, P, I, I, I, I, P]),
    "wcmc_grad_norm_clip_workspace_bytes": (Z, [I, P]),
    "wcmc_grad_norm_clip": (I, [I, P, P, F, P, P, Z, P]),
    "wcmc_weight_norm_fwd": (I, [I, P, P, P, P, P, P, P]),
    "wcmc_weight_norm_bwd": (I, [I, P, P, P, P, P, P, P, P, P]),
    "wcmc_clip_adam": (I, [P, P, P, P, L, F, D, D, D, D, I, F, P, P]),
    "wcmc_clip_adam_hyper": (None, [D, D, D, D, I, P]),
    "wcmc_clip_adam_dev": (I, [P, P, P, P, L, F, F, P, P, P]),
    "wcmc_step_guard": (I, [P, I, P, P, P, P]),
    "wcmc_step_guard_local": (I, [P, I, P, P, P, P]),
    "wcmc_step_guard_global": (I, [P, I, P, P, P, P, P]),
    "wcmc_preprocess_llpm": (I, [P, L, I, I, P, P]),
    "wcmc_preprocess_kpcn_workspace_bytes": (Z, [I, I]),
    "wcmc_preprocess_kpcn": (I, [P, I, I, I, I, I, P, P, Z, P]),
    "wcmc_gradients": (I, [P, I, I, I, P, P]),
    "wcmc_assemble_kpcn_patches": (I, [P, P, P, P, I, I, I, I, I, P, P, P, P, P, P, P, P, P, P]),
}

_lib = None


def lib():
    """Load the shared library once; fail loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise RuntimeError(
                "wcmc_amd: %s is missing -- the HIP hot path has not been built "
                "(make -C wcmc_amd/csrc). There is no CPU fallback." % LIB_PATH)
        h = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(h, name)
            fn.restype, fn.argtypes = res, args
        if h.wcmc_abi_version() != 2:
            raise RuntimeError("wcmc_amd: ABI version mismatch in %s" % LIB_PATH)
        _lib = h
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().wcmc_last_error().decode("utf-8", "replace")
        raise RuntimeError("wcmc_hip %s failed (%d): %s" % (what, rc, msg))
