"""ctypes binding of libnerf_mi355x.so (include/nerf_mi355x.h).

Loading is explicit and loud: a missing library or a box without a gfx950 GPU raises;
there is no CPU or PyTorch fallback anywhere in this package.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NERF_MI355X_LIB") or os.path.join(HERE, "libnerf_mi355x.so")

NERF_MAX_SKIPS = 8
NERF_NUM_SLOTS = 16

EXPORTS = (
    "nerf_last_error", "nerf_version", "nerf_device_count", "nerf_ctx_create", "nerf_ctx_destroy",
    "nerf_load_weights", "nerf_num_weight_tensors", "nerf_embed", "nerf_mlp_forward", "nerf_run_network",
    "nerf_raw2outputs", "nerf_sample_pdf", "nerf_render_rays", "nerf_profile_enable", "nerf_profile_read",
    "nerf_workspace_bytes", "nerf_generate_rays", "nerf_image_metrics", "nerf_train_step", "nerf_get_weights",
    "nerf_get_gradients", "nerf_stratified_z", "nerf_resample", "nerf_render_frame", "nerf_set_precision",
    "nerf_get_precision", "nerf_precision_status", "nerf_get_adam_state", "nerf_set_adam_state",
    "nerf_shard_bounds", "nerf_render_shard", "nerf_precision_peek", "nerf_precision_check",
    "nerf_precision_detail", "nerf_profile_read_train", "nerf_set_render_precision",
    "nerf_pack_rays",
)
NERF_W_PRECISION, NERF_W_PRECISION_FALLBACK = 1, 2
NERF_GUARD_OFF, NERF_GUARD_REPORT, NERF_GUARD_FALLBACK = 0, 1, 2


class NerfArch(C.Structure):
    _fields_ = [("D", C.c_int32), ("W", C.c_int32), ("input_ch", C.c_int32), ("input_ch_views", C.c_int32),
                ("output_ch", C.c_int32), ("n_skips", C.c_int32), ("skips", C.c_int32 * NERF_MAX_SKIPS),
                ("use_viewdirs", C.c_int32)]


_FP = C.c_void_p  # device / host float pointers travel as integers


class RenderArgs(C.Structure):
    _fields_ = [("rays", _FP), ("n_rays", C.c_int64), ("ray_stride", C.c_int32), ("N_samples", C.c_int32),
                ("N_importance", C.c_int32), ("slot_coarse", C.c_int32), ("slot_fine", C.c_int32),
                ("lindisp", C.c_int32), ("white_bkgd", C.c_int32), ("perturb", C.c_int32),
                ("t_rand", _FP), ("u_rand", _FP), ("noise0", _FP), ("noise", _FP),
                ("rgb_map", _FP), ("disp_map", _FP), ("acc_map", _FP), ("raw", _FP), ("rgb0", _FP),
                ("disp0", _FP), ("acc0", _FP), ("z_std", _FP), ("z_vals_coarse", _FP),
                ("weights_coarse", _FP), ("z_samples", _FP), ("z_vals_fine", _FP), ("weights_fine", _FP),
                ("depth_map", _FP), ("z_vals_fine_in", _FP), ("stream", C.c_void_p)]


class TrainArgs(C.Structure):
    _fields_ = [("rays", _FP), ("target", _FP), ("n_rays", C.c_int64), ("ray_stride", C.c_int32),
                ("N_samples", C.c_int32), ("N_importance", C.c_int32), ("slot_coarse", C.c_int32),
                ("slot_fine", C.c_int32), ("lindisp", C.c_int32), ("white_bkgd", C.c_int32), ("perturb", C.c_int32),
                ("t_rand", _FP), ("u_rand", _FP), ("noise0", _FP), ("noise", _FP), ("lr", C.c_float),
                ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float), ("step", C.c_int32),
                ("apply_update", C.c_int32), ("loss", _FP), ("rgb_map", _FP), ("rgb0", _FP), ("stream", C.c_void_p),
                ("z_vals_fine_in", _FP), ("stats", _FP)]


class Camera(C.Structure):
    _fields_ = [("H", C.c_int32), ("W", C.c_int32), ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float),
                ("cy", C.c_float), ("c2w", C.c_float * 12), ("c2w_static", C.c_float * 12),
                ("has_static", C.c_int32), ("ndc", C.c_int32), ("ndc_focal", C.c_double), ("near", C.c_float),
                ("far", C.c_float), ("use_viewdirs", C.c_int32)]


class FrameArgs(C.Structure):
    _fields_ = [("cam", Camera), ("first_pixel", C.c_int64), ("n_pixels", C.c_int64), ("chunk", C.c_int64),
                ("N_samples", C.c_int32), ("N_importance", C.c_int32), ("slot_coarse", C.c_int32),
                ("slot_fine", C.c_int32), ("lindisp", C.c_int32), ("white_bkgd", C.c_int32), ("rgb_map", _FP),
                ("disp_map", _FP), ("acc_map", _FP), ("rgb0", _FP), ("disp0", _FP), ("acc0", _FP), ("z_std", _FP),
                ("stream", C.c_void_p), ("precision_guard", C.c_int32)]


_lib = None


def library_path():
    return LIB_PATH


def load():
    """dlopen the library once and declare every prototype."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python nerf-projects_amd/build.py` "
            "(or __graft_entry__.build()). This package has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    i32, i64, vp = C.c_int, C.c_int64, C.c_void_p
    lib.nerf_last_error.restype = C.c_char_p
    lib.nerf_last_error.argtypes = []
    lib.nerf_version.restype = C.c_char_p
    lib.nerf_version.argtypes = []
    lib.nerf_device_count.restype = i32
    lib.nerf_device_count.argtypes = []
    lib.nerf_ctx_create.restype = i32
    lib.nerf_ctx_create.argtypes = [i32, C.POINTER(vp)]
    lib.nerf_ctx_destroy.restype = None
    lib.nerf_ctx_destroy.argtypes = [vp]
    lib.nerf_load_weights.restype = i32
    lib.nerf_load_weights.argtypes = [vp, i32, C.POINTER(NerfArch), C.POINTER(vp), i32]
    lib.nerf_num_weight_tensors.restype = i32
    lib.nerf_num_weight_tensors.argtypes = [C.POINTER(NerfArch)]
    lib.nerf_embed.restype = i32
    lib.nerf_embed.argtypes = [vp, vp, i64, i32, vp, vp]
    lib.nerf_mlp_forward.restype = i32
    lib.nerf_mlp_forward.argtypes = [vp, i32, vp, i64, vp, vp]
    lib.nerf_run_network.restype = i32
    lib.nerf_run_network.argtypes = [vp, i32, vp, vp, i64, i64, vp, vp]
    lib.nerf_raw2outputs.restype = i32
    lib.nerf_raw2outputs.argtypes = [vp, vp, i32, vp, vp, vp, i32, i64, i32, vp, vp, vp, vp, vp, vp]
    lib.nerf_sample_pdf.restype = i32
    lib.nerf_sample_pdf.argtypes = [vp, vp, vp, vp, i64, i32, i32, vp, vp]
    lib.nerf_render_rays.restype = i32
    lib.nerf_render_rays.argtypes = [vp, C.POINTER(RenderArgs)]
    lib.nerf_profile_enable.restype = i32
    lib.nerf_profile_enable.argtypes = [vp, i32]
    lib.nerf_profile_read.restype = i32
    lib.nerf_profile_read.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(i64), C.POINTER(i64), i32]
    lib.nerf_profile_read_train.restype = i32
    lib.nerf_profile_read_train.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(i64), C.POINTER(i64), i32]
    lib.nerf_generate_rays.restype = i32
    lib.nerf_generate_rays.argtypes = [vp, C.POINTER(Camera), i64, i64, vp, vp]
    lib.nerf_image_metrics.restype = i32
    lib.nerf_image_metrics.argtypes = [vp, vp, vp, i32, i32, C.c_float, vp, vp]
    lib.nerf_train_step.restype = i32
    lib.nerf_train_step.argtypes = [vp, C.POINTER(TrainArgs)]
    lib.nerf_get_weights.restype = i32
    lib.nerf_get_weights.argtypes = [vp, i32, C.POINTER(vp), i32]
    lib.nerf_get_gradients.restype = i32
    lib.nerf_get_gradients.argtypes = [vp, i32, C.POINTER(vp), i32]
    lib.nerf_stratified_z.restype = i32
    lib.nerf_stratified_z.argtypes = [vp, vp, i32, i64, i32, i32, vp, vp, vp]
    lib.nerf_resample.restype = i32
    lib.nerf_resample.argtypes = [vp, vp, vp, vp, i64, i32, i32, vp, vp, vp, vp]
    lib.nerf_render_frame.restype = i32
    lib.nerf_render_frame.argtypes = [vp, C.POINTER(FrameArgs)]
    lib.nerf_shard_bounds.restype = i32
    lib.nerf_shard_bounds.argtypes = [i64, i32, i32, C.POINTER(i64), C.POINTER(i64)]
    lib.nerf_render_shard.restype = i32
    lib.nerf_render_shard.argtypes = [vp, C.POINTER(FrameArgs), i32, i32, C.POINTER(i64), C.POINTER(i64)]
    lib.nerf_workspace_bytes.restype = i64
    lib.nerf_workspace_bytes.argtypes = [vp]
    lib.nerf_set_precision.restype = i32
    lib.nerf_set_precision.argtypes = [vp, i32]
    lib.nerf_pack_rays.restype = i32
    lib.nerf_pack_rays.argtypes = [vp, C.POINTER(Camera), vp, i32, vp, i32, i64, vp, vp]
    lib.nerf_set_render_precision.restype = i32
    lib.nerf_set_render_precision.argtypes = [vp, i32]
    lib.nerf_get_precision.restype = i32
    lib.nerf_get_precision.argtypes = [vp]
    lib.nerf_get_adam_state.restype = i32
    lib.nerf_get_adam_state.argtypes = [vp, i32, C.POINTER(vp), C.POINTER(vp), i32]
    lib.nerf_set_adam_state.restype = i32
    lib.nerf_set_adam_state.argtypes = [vp, i32, C.POINTER(vp), C.POINTER(vp), i32]
    lib.nerf_precision_status.restype = i32
    lib.nerf_precision_status.argtypes = [vp, C.POINTER(i64), i32]
    lib.nerf_precision_detail.restype = i32
    lib.nerf_precision_detail.argtypes = [vp, C.POINTER(i64), i32]
    lib.nerf_precision_peek.restype = i32
    lib.nerf_precision_peek.argtypes = [vp, C.POINTER(i64)]
    lib.nerf_precision_check.restype = i32
    lib.nerf_precision_check.argtypes = [vp, vp, C.POINTER(i64)]
    _lib = lib
    return lib


def check(rc):
    """Preserve the reference's exception convention: errors (negative codes) are Python exceptions. Positive codes are
    the library's warnings (NERF_W_*: the work was done); they become RuntimeWarnings and are returned."""
    if rc < 0:
        msg = load().nerf_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"nerf_mi355x error {rc}: {msg}")
    if rc > 0:
        import warnings
        warnings.warn(load().nerf_last_error().decode("utf-8", "replace"), RuntimeWarning, stacklevel=3)
    return rc
