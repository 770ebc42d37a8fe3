"""ctypes binding of libdram_hip.so (C ABI: include/dram_hip.h).

The library is the product: there is no CPU or PyTorch fallback.  Importing this
module without the built library raises ImportError with the build command.
"""
import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_size_t, c_void_p

# torch FIRST: it bundles its own libamdhip64.so (soname libamdhip64.so.7).  Loaded after torch, libdram_hip.so binds
# to that already-mapped runtime, i.e. to the one that owns the device buffers and streams it is handed.  Loaded
# before torch, the dynamic loader resolves the soname to /opt/rocm's copy instead and torch then maps its own on
# top: two HIP runtimes in one process, and every launch from this library fails ("no ROCm-capable device").
import torch  # noqa: F401  (import order matters, see above)

_HERE = os.path.dirname(os.path.abspath(__file__))
# (DRAM_HIP_LIB: another build of the same library -- diagnostics / A-B experiments only)
LIB_PATH = os.environ.get("DRAM_HIP_LIB") or os.path.join(os.path.dirname(_HERE), "libdram_hip.so")

P, I, L, F, Z = c_void_p, c_int, c_int64, c_float, c_size_t

# name -> (restype, argtypes); mirrors include/dram_hip.h one to one
SIGNATURES = {
    "dram_last_error": (c_char_p, []),
    "dram_abi_version": (I, []),
    "dram_conv3d_k3_packed_floats": (Z, [I, I]),
    "dram_conv3d_k3_pack_weights": (I, [P, P, I, I, I, P]),
    "dram_conv3d_k3_fwd": (I, [P, P, P, P, I, I, I, I, I, I, P]),
    "dram_conv3d_k3_fwd_cat": (I, [P, I, P, I, I, I, I, I, I, I, P, P, P, I, I, I, I, I, P]),
    "dram_conv3d_k3_fwd_ex": (I, [P, I, P, I, I, I, I, I, I, I, P, P, P, I, P, I, I, I, I, I, I, I, I, I, I, I, P]),
    "dram_conv3d_k3_wgrad_ws_bytes": (Z, [I, I, I, I, I, I]),
    "dram_conv3d_k3_wgrad": (I, [P, P, P, P, Z, I, I, I, I, I, I, P]),
    "dram_conv3d_k3_wgrad_ex": (I, [P, I, P, I, I, I, I, I, I, I, P, P, P, Z, I, I, I, I, I, P]),
    "dram_channel_sum_ws_bytes": (Z, [I, I, L]),
    "dram_channel_sum": (I, [P, P, P, Z, I, I, L, P]),
    "dram_norm_ws_bytes": (Z, [I, I, L]),
    "dram_norm_fwd_train": (I, [P, P, P, P, P, P, P, P, P, F, F, I, I, I, I, I, L, P, Z, P]),
    "dram_bn_fwd_eval": (I, [P, P, P, P, P, P, P, P, P, F, I, I, I, L, P]),
    "dram_norm_bwd": (I, [P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, L, P, Z, P]),
    "dram_bn_stats": (I, [P, P, I, I, L, P, Z, P]),
    "dram_bn_bwd_sums": (I, [P, P, P, P, P, P, I, I, I, L, P, Z, P]),
    "dram_bn_bwd_apply_sums": (I, [P, P, P, P, P, P, P, ctypes.c_double, P, I, I, I, L, P, Z, P]),
    "dram_relu_fwd": (I, [P, P, L, P]),
    "dram_relu_bwd": (I, [P, P, P, L, P]),
    "dram_maxpool3d_2_fwd": (I, [P, P, P, I, I, I, I, I, P]),
    "dram_maxpool3d_2_bwd": (I, [P, P, P, I, I, I, I, I, P]),
    "dram_upsample_trilinear_ac_fwd": (I, [P, P, I, I, I, I, I, I, I, I, P]),
    "dram_upsample_trilinear_ac_bwd": (I, [P, P, I, I, I, I, I, I, I, I, P]),
    "dram_upsample_trilinear_ac_bwd_ws_bytes": (Z, [I, I, I, I, I, I, I, I]),
    "dram_upsample_trilinear_ac_bwd_ws": (I, [P, P, P, Z, I, I, I, I, I, I, I, I, P]),
    "dram_crop_concat_fwd": (I, [P, P, P, I, I, I, I, I, I, I, I, I, I, I, I, P]),
    "dram_crop_concat_bwd": (I, [P, P, P, I, I, I, I, I, I, I, I, I, I, I, I, P]),
    "dram_conv3d_k1_fwd": (I, [P, P, P, P, I, I, I, L, P]),
    "dram_conv3d_k1_bwd_ws_bytes": (Z, [I, I, I, L]),
    "dram_conv3d_k1_bwd": (I, [P, P, P, P, P, P, P, Z, I, I, I, L, P]),
    "dram_masked_mean_ws_bytes": (Z, [I, I, L]),
    "dram_masked_mean_fwd": (I, [P, P, P, P, P, Z, I, I, L, P]),
    "dram_masked_mean_bwd": (I, [P, P, P, P, I, I, L, P]),
    "dram_label_bboxes": (I, [P, P, I, I, I, I, P]),
    "dram_lobe_chunks": (I, [P, P, P, P, I, I, I, I, I, F, F, P]),
    "dram_lobe_paste": (I, [P, P, P, P, I, I, I, I, I, P]),
    "dram_lung_hist256": (I, [P, P, P, P, L, P]),
    "dram_threshold_mask": (I, [P, P, F, L, P]),
    "dram_scan_hist256": (I, [P, P, P, I, I, L, P]),
    "dram_lesion_post": (I, [P, P, P, P, P, F, I, I, ctypes.c_double, L, P]),
    "dram_mask_overlap": (I, [P, P, P, L, P]),
    "dram_resample_volume": (I, [P, P, I, I, I, I, I, I, I, I, P, P, P]),
    "dram_intreg_loss_ws_bytes": (Z, [I, L]),
    "dram_intreg_loss_state_floats": (I, [I]),
    "dram_intreg_loss_fwd": (I, [P, P, P, P, P, P, P, F, P, P, P, Z, I, L, P]),
    "dram_intreg_loss_bwd": (I, [P, P, P, P, P, P, P, P, P, F, P, P, I, L, P]),
    "dram_prelu_fwd": (I, [P, P, P, I, I, I, L, P]),
    "dram_prelu_bwd_ws_bytes": (Z, [I, I, L]),
    "dram_prelu_bwd": (I, [P, P, P, P, P, P, Z, I, I, I, L, P]),
    "dram_global_max_fwd": (I, [P, P, P, I, L, P]),
    "dram_global_max_bwd": (I, [P, P, P, I, L, P]),
    "dram_resize_trilinear_fwd": (I, [P, P, I, I, I, I, I, I, I, I, F, F, F, P]),
    "dram_resize_trilinear_bwd": (I, [P, P, I, I, I, I, I, I, I, I, F, F, F, P]),
    "dram_resize_nearest": (I, [P, P, I, I, I, I, I, I, I, I, F, F, F, P]),
    "dram_spatial_permute_flip": (I, [P, P, I, I, I, I, I, P, P, P]),
    "dram_pcm_attention_fwd": (I, [P, P, P, I, I, I, P, I, I, I, I, I, P]),
    "dram_pcm_attention_bwd": (I, [P, P, P, P, P, I, I, I, P, P, P, I, I, I, I, I, P]),
    "dram_pcm_attention_split_fwd": (I, [P, P, P, I, I, I, I, P, I, I, I, I, I, P]),
    "dram_pcm_attention_split_bwd": (I, [P, P, P, P, P, I, I, I, I, P, P, P, P, I, I, I, I, I, P]),
    "dram_pcm_attention_sum_fwd": (I, [P, P, P, I, I, P, I, I, I, I, I, P]),
    "dram_pcm_attention_sum_bwd": (I, [P, P, P, P, P, I, I, P, P, P, I, I, I, I, I, P]),
    "dram_pcm_aggregate_fwd": (I, [P, P, P, I, P, I, I, I, I, I, P]),
    "dram_pcm_aggregate_bwd": (I, [P, P, P, P, I, P, P, I, I, I, I, I, P]),
    # fused conv -> norm -> ReLU -> conv chains ("lazy" tensors)
    "dram_conv3d_k3_stats_parts": (I, [I, I, I, I, I]),
    "dram_conv3d_k3_fwd_fused": (I, [P, I, P, I, P, I, P, I, I, I, I, I, I, I, P, P, P, P, I, I, I, I, I, I, P]),
    "dram_conv3d_k3_wgrad_lazy_ok": (I, [I, I, I, I, I, I, I]),
    "dram_conv3d_k3_wgrad_fused": (I, [P, I, P, I, P, I, P, I, I, I, I, I, I, I, P, P, P, Z, I, I, I, I, I, P]),
    "dram_conv3d_k3_fwd_choice": (I, [I, I, I, I, I, I, I, I, I, I, I, c_char_p, Z]),
    "dram_conv3d_k3_fwd_choice_src": (I, [I, I, I, I, I, I, I, I, I, I, I, I, I, I, I, I, I, c_char_p, Z]),
    "dram_conv3d_k3_wgrad_choice": (I, [I, I, I, I, I, I, I, I, c_char_p, Z]),
    "dram_conv3d_k3_launch_counts": (I, [P, I]),
    "dram_norm_parts_ws_bytes": (Z, [I, I, I]),
    "dram_norm_finalize_parts": (I, [P, I, P, P, P, P, P, P, P, F, F, I, I, I, I, L, P, Z, P]),
    "dram_bn_parts_stats": (I, [P, I, P, I, I, L, P, Z, P]),
    "dram_bn_eval_coef": (I, [P, P, P, P, P, P, P, F, I, I, P]),
    "dram_row_affine_act": (I, [P, P, P, I, L, L, P]),
    "dram_maxpool3d_2_fwd_lazy": (I, [P, P, I, P, P, I, I, I, I, I, P]),
    "dram_maxpool3d_2_bwd_acc": (I, [P, P, P, I, I, I, I, I, P]),
    "dram_upsample_trilinear_ac_fwd_lazy": (I, [P, P, I, P, I, I, I, I, I, I, I, I, P]),
    "dram_conv3d_k1_fwd_lazy": (I, [P, P, I, P, P, P, I, I, I, L, P]),
    "dram_conv3d_k1_bwd_lazy": (I, [P, P, P, I, P, P, P, P, P, Z, I, I, I, L, P]),
    # affine-consistency losses
    "dram_sigmoid_fwd": (I, [P, P, L, P]),
    "dram_sigmoid_bwd": (I, [P, P, P, L, P]),
    "dram_masked_smooth_l1_ws_bytes": (Z, [I, I, L]),
    "dram_masked_smooth_l1_fwd": (I, [P, P, P, P, P, Z, I, I, L, P]),
    "dram_masked_smooth_l1_bwd": (I, [P, P, P, P, P, P, P, I, I, L, P]),
    "dram_affine_sample_fwd": (I, [P, P, P, I, I, I, I, I, P]),
    "dram_affine_sample_bwd": (I, [P, P, P, I, I, I, I, I, P]),
    # measured ceilings (bench.py)
    "dram_calibrate_hbm_copy": (I, [P, P, Z, P]),
    "dram_calibrate_mfma_f32": (I, [P, I, I, P, P]),
}


class DramHipError(RuntimeError):
    """A libdram_hip.so entry point returned a DRAM_E* status."""


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP library is the only implementation of this path "
            f"(no CPU fallback). Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"or `make -C bodyct-dram_amd/csrc`.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


def call(name, *args):
    """Call an int-returning entry point; raise DramHipError on a non-zero status."""
    rc = getattr(lib, name)(*args)
    if rc != 0:
        msg = lib.dram_last_error()
        raise DramHipError(f"{name} failed ({rc}): {msg.decode() if msg else '?'}")
