"""ctypes binding of libgank.so (include/gank.h).  There is NO fallback: if the HIP library is
missing or a symbol is absent this module raises, and every compute entry point of the package
goes through it."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# GANK_DTYPE = bf16 (default) | fp16: the element type of activations / MFMA operands for this PROCESS.  The two are separate
# builds of the same kernels (libgank.so, libgank_f16.so: gank_act_dtype()); everything above the C ABI takes the matching
# torch dtype from ACT_DTYPE_NAME (kernels.BF16).
DTYPE = os.environ.get("GANK_DTYPE", "bf16").lower()
if DTYPE not in ("bf16", "fp16"):
    raise RuntimeError(f"GANK_DTYPE={DTYPE!r}: bf16 or fp16")
ACT_DTYPE_NAME = "bfloat16" if DTYPE == "bf16" else "float16"
LIB_PATH = os.path.join(_HERE, os.environ.get("GANK_LIB_NAME", "libgank.so" if DTYPE == "bf16" else "libgank_f16.so"))   # GANK_LIB_NAME: experiment builds

P, I, L, F = C.c_void_p, C.c_int, C.c_long, C.c_float

# flags (include/gank.h)
IN_UPSAMPLE2X, IN_RELU, OUT_TANH, DY_UPSAMPLE2X, RES_UPSAMPLE2X, STATS_PREZEROED, OUT_POOLSUM2X = 1, 2, 4, 8, 64, 256, 512
STAT_SLOTS = 16   # GANK_STAT_SLOTS


class SnDesc(C.Structure):
    """gank_sn_desc"""
    _fields_ = [(n, P) for n in ("W", "u_in", "u_out", "v", "W_bar", "scal", "a", "b", "bpart",
                                 "dW_bar", "dW", "rowdot", "ga", "u_snap")] + \
               [("K", I), ("C", I), ("row_offset", I), ("chunk_offset", I)]


class WgradItem(C.Structure):
    """gank_wgrad_item"""
    _fields_ = [("x", P), ("dy", P), ("dw", P), ("dbias", P)]


class SlabJob(C.Structure):
    """gank_slab_job"""
    _fields_ = [("slabs", P), ("out", P), ("n", L), ("stride", L), ("nslabs", I), ("scale", F), ("fold", I), ("out_run", L), ("out_pitch", L)]


class PrepDesc(C.Structure):
    """gank_prep_desc"""
    _fields_ = [("w", P), ("wf", P), ("wd", P), ("ksize", I), ("Cin", I), ("Cout", I), ("kind", I), ("cin_pitch", I)]


class Res8Head(C.Structure):
    """gank_res8_head"""
    _fields_ = [("logits", P), ("head_w", P), ("pooled", P), ("loss", P), ("w_grad", P), ("b_grad", P), ("n_real", I), ("mode", I),
                ("loss_scale", F)]


class LabelDenseDesc(C.Structure):
    """gank_label_dense_desc"""
    _fields_ = [("table", P), ("bias", P), ("out", P), ("V", I), ("D", I), ("weight", I)]


class CriticFeedDesc(C.Structure):
    """gank_critic_feed_desc"""
    _fields_ = [("real_all", P), ("labels_all", P), ("fake_all", P), ("both", P), ("labels2", P), ("slot", P), ("rng_state", P),
                ("done_counter", P), ("B", I), ("n_slots", I)]


# name -> argument ctypes (all return int unless listed in _RET)
PROTOTYPES = {
    "gank_version": [],
    "gank_act_dtype": [],
    "gank_last_error": [],
    "gank_conv2d_prep_weights": [P, P, P, I, I, I, P],
    "gank_conv2d_prep_weights_batched": [C.POINTER(PrepDesc), I, P],
    "gank_conv2d_fprop": [P, P, P, P, P, P, I, I, I, I, I, I, I, F, P],
    "gank_conv2d_fprop_stats": [P, P, P, P, P, P, I, I, I, I, I, I, I, F, P, I, P, P],
    "gank_conv2d_dgrad": [P, P, P, P, P, I, I, I, I, I, I, I, F, P],
    "gank_meanpool_conv1x1_fprop": [P, P, P, P, P, I, I, I, I, I, P],
    "gank_image_conv_pair_fprop": [P, P, P, P, P, P, P, P, I, I, I, I, I, P],
    "gank_conv2d_wgrad_batched": [C.POINTER(WgradItem), I, I, I, I, I, I, I, I, F, P],
    "gank_conv2d_wgrad_narrow_pair": [P, P, P, P, I, I, I, I, I, P, P, P, P, I, I, I, I, I, F, P],
    "gank_conv2d_wgrad_ws_elems": [I, I, I, I, I, I, I],
    "gank_conv1x1_wgrad_dgrad": [P, P, P, P, P, I, P, I, I, I, I, I, P],
    "gank_conv2d_wgrad": [P, P, P, P, P, L, I, I, I, I, I, I, I, F, P],
    "gank_upconv3x3_prep_weights": [P, P, P, I, I, P],
    "gank_upconv3x3_fprop": [P, P, P, P, P, I, I, I, I, I, I, P],
    "gank_upconv3x3_fprop_stats": [P, P, P, P, P, I, I, I, I, I, I, P, I, P, P],
    "gank_upconv3x3_dgrad": [P, P, P, P, I, I, I, I, I, P],
    "gank_convpool3x3_prep_weights": [P, P, P, I, I, P],
    "gank_convpool3x3_fprop": [P, P, P, P, P, I, I, I, I, I, I, P],
    "gank_convpool3x3_dgrad": [P, P, P, P, I, I, I, I, I, P],
    "gank_convpool3x3_wgrad": [P, P, P, P, P, L, I, I, I, I, I, I, P],
    "gank_convpool3x3_wgrad_ws_elems": [I, I, I, I, I],
    "gank_upconv3x3_wgrad_ws_elems": [I, I, I, I, I],
    "gank_upconv3x3_wgrad": [P, P, P, P, L, I, I, I, I, I, P],
    "gank_conv2d_general_fprop": [P, P, P, P, I, I, I, I, I, I, I, I, I, I, I, P],
    "gank_conv2d_general_dgrad": [P, P, P, P, I, I, I, I, I, I, I, I, I, P],
    "gank_conv2d_general_wgrad": [P, P, P, P, I, I, I, I, I, I, I, I, I, I, I, P],
    "gank_res8_chain_fwd": [P, P, P, P, P, P, I, I, I, P],
    "gank_res8_chain_bwd": [P, P, P, P, P, P, P, P, P, I, I, I, P],
    "gank_phase_stack4": [P, P, I, I, I, P],
    "gank_pad_rows": [P, P, C.c_long, I, I, I, I, P],
    "gank_tile_rows": [P, P, I, I, I, P],
    "gank_fewout_pack": [P, P, I, I, I, I, I, P],
    "gank_zero_f32": [P, C.c_long, P],
    "gank_img16_conv3x3": [P, P, P, P, P, P, I, I, I, I, P],
    "gank_cbn_relu_img16_conv3x3": [P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, P, I, P],
    "gank_img16_conv3x3_stats": [P, P, P, P, P, P, I, I, I, I, P, I, P],
    "gank_res8_chain_fwd_head": [P, P, P, P, P, P, P, P, P, I, I, I, P],
    "gank_res8_chain_bwd_head": [C.POINTER(Res8Head), P, P, P, P, P, P, P, I, I, I, P],
    "gank_res8_conv3x3": [P, P, P, P, P, I, I, I, I, P, I, P],
    "gank_im2col_narrow": [P, P, I, I, I, I, I, I, I, I, I, I, P],
    "gank_tap_gather_up2": [P, P, P, I, I, I, I, I, I, I, I, P],
    "gank_tap_scatter_up2": [P, P, I, I, I, I, I, I, I, P],
    "gank_depth_to_space2": [P, P, I, I, I, I, P],
    "gank_space_to_depth2": [P, P, I, I, I, I, P],
    "gank_cpool_res_fprop": [P, P, P, P, P, I, I, I, I, I, I, P],
    "gank_cpool_res_dgrad": [P, P, P, P, I, I, I, I, I, P],
    "gank_cpool_res_dgrad_image_wgrad": [P, P, P, P, P, P, P, P, P, I, I, I, I, I, P, P],
    "gank_sum_slabs": [C.POINTER(SlabJob), I, P],
    "gank_convpool3x3_wgrad_job": [P, P, P, P, P, L, I, I, I, I, I, I, C.POINTER(SlabJob), P],
    "gank_conv2d_wgrad_slab_elems": [I, I, I, I, I, I, I],
    "gank_conv2d_wgrad_slabs": [P, P, P, P, I, I, I, I, I, I, I, F, P, L, C.POINTER(SlabJob), P],
    "gank_conv2d_wgrad_slab_splits": [I, I, I, I, I, I, I],
    "gank_conv2d_wgrad_slabs_rows": [P, P, P, P, I, I, I, I, I, I, I, I, F, P, L, C.POINTER(SlabJob), P],
    "gank_conv2d_wgrad_slabs_rows_tap_sums": [P, P, P, P, I, I, I, I, I, I, I, I, F, P, L, C.POINTER(SlabJob), P, I, P, P],
    "gank_conv2d_wgrad_batched_ws_elems": [I, I, I, I, I, I, I, I],
    "gank_conv2d_wgrad_batched_slabs": [C.POINTER(WgradItem), I, I, I, I, I, I, I, I, F, P, L, C.POINTER(SlabJob), P],
    "gank_deconv2d_prep_phases": [P, P, I, I, I, P],
    "gank_deconv2d_fprop": [P, P, P, P, I, I, I, I, I, I, P],
    "gank_deconv2d_dgrad": [P, P, P, I, I, I, I, I, I, P],
    "gank_deconv2d_wgrad": [P, P, P, I, I, I, I, I, I, P],
    "gank_colsum_bf16": [P, P, L, I, F, P],
    "gank_sn_ws_floats": [I, I],
    "gank_sn_power_iter_fwd": [C.POINTER(SnDesc), I, P],
    "gank_sn_power_iter_bwd": [C.POINTER(SnDesc), I, P],
    "gank_sn_power_iter_fwd_prep": [C.POINTER(SnDesc), I, C.POINTER(PrepDesc), C.POINTER(C.c_int), I, C.POINTER(LabelDenseDesc), P],
    "gank_sn_power_iter_fwd_a": [C.POINTER(SnDesc), I, P],
    "gank_sn_power_iter_fwd_b_prep": [C.POINTER(SnDesc), I, C.POINTER(PrepDesc), C.POINTER(C.c_int), I, C.POINTER(LabelDenseDesc), P, P, P, I, P],
    "gank_sn_power_iter_fwd_b_prep_feed": [C.POINTER(SnDesc), I, C.POINTER(PrepDesc), C.POINTER(C.c_int), I, C.POINTER(LabelDenseDesc), P, P, P, I,
                                           C.POINTER(CriticFeedDesc), P],
    "gank_sn_power_iter_bwd_gw": [C.POINTER(SnDesc), I, P],
    "gank_sn_adam_fwd_a": [C.POINTER(SnDesc), I, C.POINTER(C.c_void_p), P, P, P, P, L, P, P, P, P, I, P, P, P],
    "gank_label_dense_table": [P, P, P, P, P, I, I, I, P],
    "gank_concat_label_fwd": [P, P, P, P, I, I, I, I, I, P],
    "gank_concat_label_bwd": [P, P, P, I, I, I, I, P],
    "gank_label_dense_bwd": [P, P, P, P, P, P, P, I, I, I, I, P],
    "gank_label_conv3x3_bwd_pooled": [P, P, P, I, P, I, I, I, I, I, P, P, P, P, I, I, I, P],
    "gank_label_dense_bwd_parts": [P, I, P, P, P, P, P, I, I, I, P],
    "gank_sum_slabs_label_bwd": [P, I, P, P, P, I, P, I, I, I, I, I, P, P, P, I, I, I, P],
    "gank_concat_label_pool_fwd": [P, P, P, P, P, I, I, I, I, I, I, P],
    "gank_concat_label_unpool_bwd": [P, P, P, P, I, I, I, I, I, P],
    "gank_concat_label_unpool_bwd_factored": [P, P, P, P, P, I, P, P, I, I, I, I, I, I, P],
    "gank_label_conv3x3_table": [P, I, I, I, I, P, I, P, P, P, I, P, P],
    "gank_label_conv3x3_table_pooled": [P, I, I, I, I, P, I, P, P, P, I, P, P, P, I, I, I, P],
    "gank_label_conv3x3_table_pooled_shortcut": [P, I, I, I, I, P, I, P, P, P, I, P, P, P, I, I, I, P, I, P, I, P, P],
    "gank_img16_conv3x3_label_bias": [P, P, P, P, I, P, I, I, I, I, P],
    "gank_img16_conv3x3_dgrad_unpool": [P, P, P, P, I, F, P, I, I, I, P],
    "gank_img16_conv3x3_label_bwd": [P, P, P, P, I, I, I, I, P, P, I, P, I, I, I, I, P, P, P],
    "gank_label_conv3x3_bwd_ws_floats": [I, I],
    "gank_label_conv3x3_bwd": [P, P, P, I, P, I, I, I, I, I, I, I, P, P, P, P, P],
    "gank_cbn_parts": [L],
    "gank_cbn_fwd": [P, P, P, P, P, P, P, I, I, I, I, I, I, P],
    "gank_cbn_fwd_from_sums": [P, P, P, P, P, P, P, P, I, I, I, I, I, I, F, P],
    "gank_cbn_fwd_eps": [P, P, P, P, P, P, P, I, I, I, I, I, I, F, P],
    "gank_layer_norm_fwd": [P, P, P, P, P, I, I, I, F, P],
    "gank_layer_norm_bwd": [P, P, P, P, P, P, P, I, I, I, P],
    "gank_pixel_norm_fwd": [P, P, L, I, F, P],
    "gank_pixel_norm_bwd": [P, P, P, L, I, F, P],
    "gank_cbn_bwd": [P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, P],
    "gank_cbn_stats": [P, P, P, I, I, I, I, F, P],
    "gank_cbn_stats_from_sums": [P, P, P, I, I, L, F, P],
    "gank_cbn_relu_conv3x3_fprop": [P, P, P, P, P, P, P, P, I, I, I, I, I, I, I, I, P],
    "gank_cbn_bwd_remask": [P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, P],
    "gank_cbn_bwd_ws_floats": [I, I, I, I],
    "gank_cbn_bwd_ws": [P, P, P, P, P, P, P, P, P, P, P, L, I, I, I, I, I, I, P],
    "gank_pool2x2": [P, P, P, I, I, I, I, F, P],
    "gank_unpool2x2_add": [P, P, P, I, I, I, I, F, P],
    "gank_add_bf16": [P, P, P, L, P],
    "gank_relu_fwd": [P, P, L, F, P],
    "gank_relu_bwd": [P, P, P, L, F, P],
    "gank_tanh_bwd": [P, P, P, L, P],
    "gank_scale_f32": [P, P, P, L, P],
    "gank_weighted_sum4_f32": [P, P, P, P, F, F, F, F, P, L, P],
    "gank_linear_fwd": [P, P, P, P, I, I, I, P],
    "gank_linear_fwd_f32out": [P, P, P, P, I, I, I, P],
    "gank_linear_bwd": [P, P, P, P, P, P, I, I, I, P],
    "gank_copy_bytes": [P, P, L, P],
    "gank_copy_bytes_gather": [P, P, I, L, P],
    "gank_copy_bytes_gather2": [P, P, I, L, P, P, I, L, P],
    "gank_cast_f32_bf16": [P, P, L, P],
    "gank_cast_bf16_f32": [P, P, L, P],
    "gank_relu_meanpool_hw_fwd": [P, P, I, I, I, P],
    "gank_relu_meanpool_hw_bwd": [P, P, P, I, I, I, P],
    "gank_concat_tile_fwd": [P, P, P, I, I, I, I, P],
    "gank_concat_tile_bwd": [P, P, P, I, I, I, I, P],
    "gank_embedding_fwd": [P, P, P, I, I, I, P],
    "gank_embedding_bwd": [P, P, P, I, I, I, P],
    "gank_critic_head_hinge": [P, P, P, P, P, P, P, P, I, I, I, I, P],
    "gank_critic_head_hinge_scaled": [P, P, P, P, P, P, P, P, I, I, I, I, F, P],
    "gank_hinge_d_loss": [P, P, P, P, I, I, P],
    "gank_gan_pointwise_loss": [P, P, P, P, I, I, I, P],
    "gank_hinge_g_loss": [P, P, P, P, I, P],
    "gank_wgan_d_loss": [P, P, P, P, I, I, P],
    "gank_softmax_xent": [P, P, P, P, P, I, I, P],
    "gank_loss_grad_scale": [P, P, P, L, P],
    "gank_bn_bwd_bwd": [P, P, P, P, P, P, P, P, P, L, I, P],
    "gank_bn_moving_update": [P, P, P, P, P, I, I, L, F, F, P],
    "gank_gp_loss": [P, P, P, P, I, L, F, P],
    "gank_lerp_rows": [P, P, P, P, I, L, P],
    "gank_sum_hw": [P, P, I, I, I, F, P],
    "gank_bcast_hw": [P, P, I, I, I, F, P],
    "gank_rng_uniform_f32": [P, L, P, P],
    "gank_axpby_bf16": [P, P, F, F, P, L, P],
    "gank_blend_dev": [P, P, P, P, L, I, P],
    "gank_minibatch_std_fwd": [P, P, P, I, I, I, P],
    "gank_minibatch_std_bwd": [P, P, P, P, I, I, I, P],
    "gank_resize_bilinear": [P, P, I, I, I, I, I, I, P],
    "gank_concat_channels": [P, P, P, L, I, I, P],
    "gank_pool2d": [P, P, I, I, I, I, I, I, I, I, I, I, I, I, P],
    "gank_relu_to_channels": [P, P, L, I, I, I, P],
    "gank_split_channels": [P, P, P, L, I, I, P],
    "gank_l1_loss": [P, P, P, P, P, L, P],
    "gank_dropout_fwd": [P, P, P, L, F, P, P],
    "gank_dropout_bwd": [P, P, P, L, F, P],
    "gank_adam_tf": [P, P, P, P, P, P, P, L, L, P],
    "gank_adam_tf_health": [P, P, P, P, P, P, P, L, L, P, P],
    "gank_counter_add": [P, C.c_int64, P],
    "gank_preprocess_real": [P, P, P, I, P],
    "gank_rng_normal_bf16": [P, L, P, P],
    "gank_generator_feed": [P, L, I, P, L, P, L, P, P],
    "gank_rng_labels": [P, L, I, P, P],
    "gank_prof_enable": [I],
    "gank_prof_reset": [],
    "gank_prof_collect": [I, C.POINTER(C.c_double), C.POINTER(C.c_double)],
    "gank_prof_calibrate": [I, P],
    "gank_prof_bytes": [I],
    "gank_prof_kernel_stats": [I, I, C.c_char_p, I, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)],
    "gank_critic_feed": [P, P, P, P, P, P, P, P, I, I, P],
    "gank_debug_tr_probe": [P, P],
}
_RET = {"gank_last_error": C.c_char_p, "gank_sn_ws_floats": C.c_long, "gank_cbn_bwd_ws_floats": C.c_long, "gank_label_conv3x3_bwd_ws_floats": C.c_long, "gank_conv2d_wgrad_ws_elems": C.c_long, "gank_convpool3x3_wgrad_ws_elems": C.c_long, "gank_upconv3x3_wgrad_ws_elems": C.c_long, "gank_conv2d_wgrad_batched_ws_elems": C.c_long, "gank_conv2d_wgrad_slab_elems": C.c_long, "gank_prof_calibrate": C.c_double, "gank_prof_bytes": C.c_double}

_lib = None


def load():
    """Load libgank.so and bind every symbol of include/gank.h.  Raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run "
            "`python -m gan_lib_tensorflow_amd.build` (needs hipcc). There is no CPU fallback.")
    # libgank.so must bind to the HIP runtime that torch ships (torch/lib/libamdhip64.so), not to a second copy from
    # /opt/rocm: with two runtimes in one process the second one sees no device ("no ROCm-capable device is detected" on
    # the first launch).  Importing torch first makes its runtime the one the loader reuses for our DT_NEEDED entry.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, args in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise RuntimeError(f"libgank.so does not export {name}; rebuild it") from e
        fn.argtypes = args
        fn.restype = _RET.get(name, C.c_int)
    if lib.gank_act_dtype() != (1 if DTYPE == "fp16" else 0):
        raise RuntimeError(f"{LIB_PATH} was built for {'fp16' if lib.gank_act_dtype() else 'bf16'} buffers, this process runs GANK_DTYPE={DTYPE}")
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().gank_last_error()
        raise RuntimeError(f"gank: {what}: {msg.decode() if msg else 'error'}")
