"""ctypes binding of libsings_hip.so (the C ABI declared in include/sings_hip.h).

The product path has no CPU fallback: if the HIP library is missing or cannot be loaded this
module raises, loudly.  Build it with ``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C sings_amd/csrc``.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SINGS_HIP_LIB: another build of the SAME library (kernel experiments: tools/build_variants.sh); never a fallback
LIB_PATH = os.environ.get("SINGS_HIP_LIB") or os.path.join(_HERE, "libsings_hip.so")
_lib = None


class SgRasterSettings(C.Structure):
    _fields_ = [
        ("image_height", C.c_int32), ("image_width", C.c_int32),
        ("tanfovx", C.c_float), ("tanfovy", C.c_float), ("scale_modifier", C.c_float),
        ("sh_degree", C.c_int32), ("sh_coeffs", C.c_int32),
        ("prefiltered", C.c_int32), ("debug", C.c_int32), ("flags", C.c_int32),
        ("bg", C.c_void_p), ("viewmatrix", C.c_void_p), ("projmatrix", C.c_void_p), ("campos", C.c_void_p),
        ("count_signal", C.c_void_p), ("count_signal_host", C.c_void_p),
    ]


class SgLayout(C.Structure):
    _fields_ = [(n, C.c_size_t) for n in (
        "geom_recA", "geom_recB", "geom_recC", "geom_depth", "geom_flags", "geom_slot", "geom_bytes",
        "bin_header", "bin_tile_count", "bin_ranges", "bin_cursor", "bin_pair_keys", "bin_point_list",
        "bin_point_keys", "bin_pair_gid", "bin_pair_tile", "bin_pair_local", "bin_sort_items", "bin_rank_items", "bin_items", "bin_ck_start", "bin_plan", "bin_pair_mask", "bin_item_w", "bin_item_perm", "bin_rec_valid", "bin_tile_keys", "bin_bytes", "img_final_T", "img_n_contrib", "img_ckpt", "img_bytes", "bwd_bytes")]


class SgSkinInputs(C.Structure):
    _fields_ = [("J", C.c_int32), ("rot_format", C.c_int32)] + [(n, C.c_void_p) for n in (
        "xyz_canon", "rot_canon", "lbs_weights", "A", "smpl_scale", "transl", "ext_trans", "ext_rot", "ext_scale")]


class SgFrameBatch(C.Structure):
    _fields_ = [("K", C.c_int32), ("camera_stride", C.c_int32), ("transl_stride", C.c_int32), ("reserved", C.c_int32)]


MAX_FRAMES = 16                      # SG_MAX_FRAMES


class SgLinearSide(C.Structure):
    _fields_ = [("cout", C.c_int), ("act", C.c_int), ("aux", C.c_void_p), ("dh", C.c_void_p), ("W", C.c_void_p),
                ("dz_out", C.c_void_p)]


class SgTriplane(C.Structure):
    _fields_ = [("n_scales", C.c_int), ("feat", C.c_int), ("res", (C.c_int * 3) * 4), ("planes", (C.c_void_p * 3) * 4),
                ("aabb", (C.c_float * 3) * 2), ("feature_minor", C.c_int), ("reserved", C.c_int)]


# every symbol include/sings_hip.h declares
EXPORTS = ("sg_abi_version", "sg_version", "sg_last_error", "sg_layout", "sg_rasterize_forward", "sg_rasterize_backward",
           "sg_rasterize_backward_records", "sg_rasterize_backward_gaussians", "sg_skinned_backward_gaussians",
           "sg_mark_visible", "sg_read_num_rendered", "sg_signal_alloc", "sg_signal_free", "sg_profile_enable", "sg_profile_collect",
           "sg_kernel_name", "sg_skin_ws_floats", "sg_skinned_forward", "sg_skinned_backward",
           "sg_photo_loss_ws_bytes", "sg_photo_loss", "sg_photo_loss_backward", "sg_reg_ws_bytes", "sg_region_laplacian", "sg_rows_laplacian", "sg_mesh_edge_loss",
           "sg_l2norm_reg", "sg_knn_ws_bytes", "sg_gaussian_edge_loss", "sg_gaussian_edge_prepare", "sg_gaussian_edge_finish", "sg_joint_transforms", "sg_joint_transforms_backward", "sg_lbs_forward", "sg_lbs_backward", "sg_matrix_to_quaternion", "sg_matrix_to_quaternion_backward", "sg_rotation_convert", "sg_rotation_convert_backward", "sg_quaternion_multiply", "sg_quaternion_multiply_backward", "sg_triplane_ws_bytes", "sg_triplane_bwd_ws_bytes", "sg_triplane_forward",
           "sg_triplane_backward", "sg_triplane_backward_prepare", "sg_triplane_backward_prepared", "sg_bias_act_ws_bytes", "sg_bias_act_forward", "sg_bias_act_backward", "sg_scales_head_forward", "sg_scales_head_backward",
           "sg_weight_grad_ws_bytes", "sg_weight_grad", "sg_linear_forward", "sg_linear_backward", "sg_linear_backward_accumulate", "sg_linear_backward_fan", "sg_copy_probe",
           "sg_frames_layout", "sg_rasterize_forward_frames", "sg_skinned_forward_frames", "sg_read_num_rendered_frames", "sg_photo_loss_backward_frames",
           "sg_rasterize_backward_records_frames", "sg_rasterize_backward_gaussians_frames", "sg_skin_ws_floats_frames",
           "sg_skinned_backward_gaussians_frames", "sg_photo_loss_frames")
NUM_KERNELS = 8
ABI_VERSION = 7                      # SG_ABI_VERSION
FLAG_SHORT_LISTS = 1                 # SG_FLAG_SHORT_LISTS
COUNT_FLAG_HALF_ROWS = 4             # SG_COUNT_FLAG_HALF_ROWS
COUNT_FLAG_HALF_LONG_ROWS = 8        # SG_COUNT_FLAG_HALF_LONG_ROWS
FLAG_WS_CLEAN = 2                    # SG_FLAG_WS_CLEAN
FLAG_THROUGHPUT = 4                  # SG_FLAG_THROUGHPUT
FLAG_FORWARD_BINNING = 8             # SG_FLAG_FORWARD_BINNING
FLAG_FORWARD_COMPOSITE = 16          # SG_FLAG_FORWARD_COMPOSITE
FLAG_SH_PLANAR = 32                  # SG_FLAG_SH_PLANAR
FLAG_LONG_ROWS = 64                  # SG_FLAG_LONG_ROWS
NUM_RENDERED_LONG_LIST = -2          # SG_NUM_RENDERED_LONG_LIST


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"sings_amd: HIP library not built ({LIB_PATH} missing). There is no CPU fallback; "
            "run `make -C sings_amd/csrc` (needs hipcc, --offload-arch=gfx950).")
    # The process must hold ONE HIP runtime: import torch first so that libamdhip64.so.7 resolves to the
    # copy PyTorch-ROCm already loaded (loading ours first makes the two runtimes disagree about devices).
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    vp, i32, sz = C.c_void_p, C.c_int, C.c_size_t
    lib.sg_version.restype = C.c_char_p
    lib.sg_abi_version.restype = C.c_int
    if lib.sg_abi_version() != ABI_VERSION:
        raise RuntimeError(f"sings_amd: {LIB_PATH} speaks ABI {lib.sg_abi_version()}, this host code ABI {ABI_VERSION} "
                           "(include/sings_hip.h SG_ABI_VERSION): rebuild with `make -C sings_amd/csrc`")
    lib.sg_last_error.restype = C.c_char_p
    lib.sg_layout.argtypes = [i32, i32, i32, sz, C.POINTER(SgLayout)]
    lib.sg_rasterize_forward.argtypes = ([C.POINTER(SgRasterSettings), i32] + [vp] * 7 +
                                         [vp, vp, sz, vp, vp, vp, i32, C.POINTER(C.c_int64), vp])
    lib.sg_rasterize_backward.argtypes = ([C.POINTER(SgRasterSettings), i32] + [vp] * 7 +
                                          [vp, vp, vp, sz, vp, vp, vp] + [vp] * 8 + [vp])
    lib.sg_rasterize_backward_records.argtypes = [C.POINTER(SgRasterSettings), i32, vp, vp, sz, vp, vp, vp, vp]
    lib.sg_rasterize_backward_gaussians.argtypes = ([C.POINTER(SgRasterSettings), i32] + [vp] * 7 + [vp, vp, vp, sz, vp, i32] +
                                                    [vp] * 8 + [vp])
    lib.sg_skinned_backward_gaussians.argtypes = ([C.POINTER(SgRasterSettings), i32, C.POINTER(SgSkinInputs)] + [vp] * 3 +
                                                  [vp, vp, vp, sz, vp, vp, i32] + [vp] * 2 + [vp] * 8 + [vp])
    for f in ("sg_rasterize_backward_records", "sg_rasterize_backward_gaussians", "sg_skinned_backward_gaussians"):
        getattr(lib, f).restype = C.c_int
    lib.sg_mark_visible.argtypes = [i32, vp, vp, vp, vp, vp]
    lib.sg_read_num_rendered.argtypes = [vp, C.POINTER(C.c_int64), vp]
    lib.sg_signal_alloc.argtypes = [i32, C.POINTER(vp), C.POINTER(vp)]; lib.sg_signal_alloc.restype = C.c_int
    lib.sg_signal_free.argtypes = [vp]; lib.sg_signal_free.restype = C.c_int
    lib.sg_profile_enable.argtypes = [i32]
    lib.sg_profile_collect.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_int64), i32]
    lib.sg_copy_probe.argtypes = [vp, vp, sz, i32, vp]; lib.sg_copy_probe.restype = C.c_int
    # K frames per call (include/sings_hip.h, "K frames of the SAME Gaussians per call")
    FB = C.POINTER(SgFrameBatch)
    lib.sg_frames_layout.argtypes = [i32, i32, i32, sz, i32, C.POINTER(SgLayout)] + [C.POINTER(sz)] * 4
    lib.sg_rasterize_forward_frames.argtypes = ([C.POINTER(SgRasterSettings), FB, i32] + [vp] * 7 +
                                                [vp, vp, sz, vp, vp, vp, C.POINTER(C.c_int64), vp])
    lib.sg_skinned_forward_frames.argtypes = ([C.POINTER(SgRasterSettings), FB, i32, C.POINTER(SgSkinInputs)] + [vp] * 3 +
                                              [vp, vp, sz, vp, vp, vp, vp, vp, vp, C.POINTER(C.c_int64), vp])
    lib.sg_read_num_rendered_frames.argtypes = [vp, i32, i32, i32, sz, i32, C.POINTER(C.c_int64), vp]
    lib.sg_rasterize_backward_records_frames.argtypes = [C.POINTER(SgRasterSettings), FB, i32, vp, vp, sz, vp, vp, vp, vp]
    lib.sg_rasterize_backward_gaussians_frames.argtypes = ([C.POINTER(SgRasterSettings), FB, i32] + [vp] * 7 +
                                                           [vp, vp, vp, sz, vp, i32] + [vp] * 8 + [vp])
    lib.sg_skin_ws_floats_frames.argtypes = [i32, i32]; lib.sg_skin_ws_floats_frames.restype = sz
    lib.sg_skinned_backward_gaussians_frames.argtypes = ([C.POINTER(SgRasterSettings), FB, i32, C.POINTER(SgSkinInputs)] + [vp] * 3 +
                                                         [vp, vp, vp, sz, vp, vp, i32] + [vp] * 2 + [vp] * 8 + [vp])
    lib.sg_photo_loss_frames.argtypes = [i32, i32, i32, C.c_float, C.c_float, vp, vp, sz, vp, sz] + [vp] * 8
    lib.sg_photo_loss_backward_frames.argtypes = [i32, i32, i32, C.c_float, C.c_float, vp, vp, sz, vp, sz, vp, vp, vp, i32, vp, vp]
    for f in ("sg_frames_layout", "sg_rasterize_forward_frames", "sg_skinned_forward_frames", "sg_read_num_rendered_frames",
              "sg_rasterize_backward_records_frames", "sg_rasterize_backward_gaussians_frames",
              "sg_skinned_backward_gaussians_frames", "sg_photo_loss_frames"):
        getattr(lib, f).restype = C.c_int
    lib.sg_kernel_name.argtypes = [i32]
    lib.sg_kernel_name.restype = C.c_char_p
    lib.sg_skin_ws_floats.argtypes = [i32]
    lib.sg_skin_ws_floats.restype = sz
    lib.sg_skinned_forward.argtypes = ([C.POINTER(SgRasterSettings), i32, C.POINTER(SgSkinInputs)] + [vp] * 3 +
                                       [vp, vp, sz, vp, vp, vp, vp, vp, vp, C.POINTER(C.c_int64), vp])
    lib.sg_skinned_backward.argtypes = ([C.POINTER(SgRasterSettings), i32, C.POINTER(SgSkinInputs)] + [vp] * 3 +
                                        [vp, vp, vp, sz, vp, vp, vp] + [vp] * 3 + [vp] * 8 + [vp])
    lib.sg_photo_loss_ws_bytes.argtypes = [i32, i32]
    lib.sg_photo_loss_ws_bytes.restype = sz
    lib.sg_photo_loss.argtypes = [i32, i32, C.c_float, C.c_float] + [vp] * 11
    lib.sg_photo_loss.restype = C.c_int
    lib.sg_photo_loss_backward.argtypes = [i32, i32, C.c_float, C.c_float] + [vp] * 8
    lib.sg_photo_loss_backward.restype = C.c_int
    lib.sg_matrix_to_quaternion.argtypes = [i32, vp, vp, vp]; lib.sg_matrix_to_quaternion.restype = C.c_int
    lib.sg_matrix_to_quaternion_backward.argtypes = [i32, vp, vp, vp, vp]; lib.sg_matrix_to_quaternion_backward.restype = C.c_int
    lib.sg_joint_transforms.argtypes = [i32, i32] + [vp] * 6; lib.sg_joint_transforms.restype = C.c_int
    lib.sg_joint_transforms_backward.argtypes = [i32, i32] + [vp] * 8; lib.sg_joint_transforms_backward.restype = C.c_int
    lib.sg_lbs_forward.argtypes = [i32, i32] + [vp] * 6; lib.sg_lbs_forward.restype = C.c_int
    lib.sg_lbs_backward.argtypes = [i32, i32] + [vp] * 9; lib.sg_lbs_backward.restype = C.c_int
    lib.sg_triplane_ws_bytes.argtypes = [C.POINTER(SgTriplane)]; lib.sg_triplane_ws_bytes.restype = sz
    lib.sg_triplane_bwd_ws_bytes.argtypes = [C.POINTER(SgTriplane), i32]; lib.sg_triplane_bwd_ws_bytes.restype = sz
    lib.sg_triplane_forward.argtypes = [C.POINTER(SgTriplane), i32, vp, vp, vp, vp]
    lib.sg_triplane_backward.argtypes = [C.POINTER(SgTriplane), i32, vp, vp, vp, C.POINTER(C.c_void_p * 3 * 4), vp, vp]
    lib.sg_triplane_backward_prepared.argtypes = lib.sg_triplane_backward.argtypes
    lib.sg_triplane_backward_prepare.argtypes = [C.POINTER(SgTriplane), i32, vp, vp, vp]
    lib.sg_bias_act_ws_bytes.argtypes = [i32, i32]; lib.sg_bias_act_ws_bytes.restype = sz
    lib.sg_bias_act_forward.argtypes = [i32, i32, i32, vp, vp, vp, vp, vp, vp]
    lib.sg_bias_act_backward.argtypes = [i32, i32, i32, vp, vp, vp, vp, vp, vp, vp]
    lib.sg_scales_head_forward.argtypes = [i32, vp, vp, vp, vp]; lib.sg_scales_head_forward.restype = C.c_int
    lib.sg_scales_head_backward.argtypes = [i32, vp, vp, vp, vp, vp]; lib.sg_scales_head_backward.restype = C.c_int
    for f in ("sg_triplane_forward", "sg_triplane_backward", "sg_triplane_backward_prepare", "sg_triplane_backward_prepared",
              "sg_bias_act_forward", "sg_bias_act_backward"):
        getattr(lib, f).restype = C.c_int
    lib.sg_weight_grad_ws_bytes.argtypes = [i32, i32, i32]; lib.sg_weight_grad_ws_bytes.restype = sz
    lib.sg_weight_grad.argtypes = [i32, i32, i32, vp, vp, vp, vp, vp, vp]; lib.sg_weight_grad.restype = C.c_int
    lib.sg_linear_forward.argtypes = [i32, i32, i32, i32] + [vp] * 7; lib.sg_linear_forward.restype = C.c_int
    lib.sg_linear_backward.argtypes = [i32, i32, i32, i32] + [vp] * 7; lib.sg_linear_backward.restype = C.c_int
    lib.sg_linear_backward_accumulate.argtypes = [i32, i32, i32, i32] + [vp] * 7; lib.sg_linear_backward_accumulate.restype = C.c_int
    lib.sg_linear_backward_fan.argtypes = [i32, i32, i32, i32] + [vp] * 5 + [C.POINTER(SgLinearSide), vp]
    lib.sg_linear_backward_fan.restype = C.c_int
    lib.sg_reg_ws_bytes.argtypes = [i32]; lib.sg_reg_ws_bytes.restype = sz
    lib.sg_knn_ws_bytes.argtypes = [i32]; lib.sg_knn_ws_bytes.restype = sz
    lib.sg_region_laplacian.argtypes = [i32, i32] + [vp] * 11
    lib.sg_rows_laplacian.argtypes = [i32, i32, i32] + [vp] * 14
    lib.sg_mesh_edge_loss.argtypes = [i32, i32] + [vp] * 8
    lib.sg_l2norm_reg.argtypes = [i32] + [vp] * 11
    lib.sg_gaussian_edge_loss.argtypes = [i32, i32] + [vp] * 8
    lib.sg_gaussian_edge_prepare.argtypes = [i32, vp, vp, vp]
    lib.sg_gaussian_edge_finish.argtypes = [i32, i32] + [vp] * 7
    for f in ("sg_region_laplacian", "sg_rows_laplacian", "sg_mesh_edge_loss", "sg_l2norm_reg", "sg_gaussian_edge_loss", "sg_gaussian_edge_prepare",
              "sg_gaussian_edge_finish"):
        getattr(lib, f).restype = C.c_int
    for f in ("sg_layout", "sg_rasterize_forward", "sg_rasterize_backward", "sg_mark_visible",
              "sg_read_num_rendered", "sg_skinned_forward", "sg_skinned_backward"):
        getattr(lib, f).restype = C.c_int
    _lib = lib
    return lib


def check(status, what):
    if status != 0:
        raise RuntimeError(f"sings_hip {what} failed: {load().sg_last_error().decode()}")


def layout(P, W, H, cap):
    L = SgLayout()
    check(load().sg_layout(P, W, H, cap, C.byref(L)), "sg_layout")
    return L
