"""ctypes binding of the C-ABI in include/splatco_raster.h (libsplatco_raster.so, gfx950).

This is the binding a maintainer of the reference would write in place of the pybind module
`diff_gaussian_rasterization._C`.  There is NO fallback: if the HIP library
is missing or was built against another ABI version, importing this module raises.
"""
import ctypes as C
import os

# PyTorch-ROCm carries its own libamdhip64: load it BEFORE this library, so that both use ONE HIP runtime (the library's
# device pointers and streams are torch's).  Loaded the other way round -- e.g. build() and smoke() in one process --
# this library binds to /opt/rocm's copy and its first HIP call fails with "no ROCm-capable device is detected".
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
# developer override for A/B experiments with variant builds of the same library
LIB_PATH = os.environ.get("SPLATCO_RASTER_LIB", os.path.join(_HERE, "csrc", "libsplatco_raster.so"))

SYMBOLS = [
    "scr_abi_version", "scr_last_error", "scr_geom_bytes", "scr_binning_bytes", "scr_image_bytes",
    "scr_backward_scratch_bytes", "scr_visible_filter", "scr_mark_visible", "scr_forward_plan",
    "scr_forward_run", "scr_backward", "scr_debug_get", "scr_profile_enable", "scr_profile_read",
    "scr_profile_kernel_name", "scr_expand_scratch_bytes", "scr_expand_plan", "scr_expand_run",
    "scr_expand_backward", "scr_mask_index_plan", "scr_mask_index_run", "scr_plane_sample_scratch_bytes", "scr_plane_sample_backward", "scr_triplane_backward_multi_scratch_bytes", "scr_triplane_backward_multi", "scr_plane_row_pairs", "scr_triplane_forward", "scr_triplane_backward_scratch_bytes", "scr_triplane_backward",
    "scr_l1_ssim_scratch_bytes", "scr_l1_ssim_forward", "scr_l1_ssim_backward",
    "scr_scaling_reg_scratch_bytes", "scr_scaling_reg_forward", "scr_scaling_reg_backward",
    "scr_pair_l1_scratch_bytes", "scr_pair_l1_forward", "scr_pair_l1_backward",
    "scr_tpa_scratch_bytes", "scr_tpa_stats", "scr_tpa_forward", "scr_tpa_backward", "scr_tpa_backward_stats",
    "scr_statis_compute", "scr_statis_apply", "scr_copy_probe",
    "scr_knn", "scr_knn_curvature", "scr_anchor_gather_stat_rows", "scr_anchor_gather_stat_buffer_rows", "scr_anchor_gather", "scr_anchor_gather_backward", "scr_mlp_heads_hidden_bytes", "scr_mlp_heads_partial_bytes", "scr_mlp_heads_forward", "scr_mlp_heads_backward",
    "scr_norm_linear_scratch_bytes", "scr_norm_linear_forward", "scr_norm_linear_backward",
    "scr_norm_fold", "scr_norm_fold_backward", "scr_norm_running_stats", "scr_box_coords", "scr_forward_plan_run",
    "scr_profile_stride", "scr_debug_force_deep_lists", "scr_adam_step", "scr_tv_add_grad",
    "scr_markers_enable", "scr_marker_push", "scr_marker_pop", "scr_norm_linear_dx",
]
PLAN_NONFINITE_COLOUR, PLAN_LARGE_RECTS = 1, 2      # SCR_PLAN_*
PROF_COUNT = 19
ABI_VERSION = 27

(DBG_TILES_TOUCHED, DBG_POINT_OFFSETS, DBG_RANGES, DBG_POINT_LIST, DBG_N_CONTRIB, DBG_FINAL_T, DBG_SPLAT_RECORDS, DBG_QMASK,
 DBG_GM_INDEX) = range(9)


class AdamTensor(C.Structure):
    """scr_adam_tensor (include/splatco_raster.h)."""
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p),
                ("numel", C.c_int64), ("step_size", C.c_double), ("bias_correction2_sqrt", C.c_double)]


class TvPlane(C.Structure):
    """scr_tv_plane (include/splatco_raster.h)."""
    _fields_ = [("plane", C.c_void_p), ("grad", C.c_void_p), ("channels", C.c_int32), ("rows", C.c_int32),
                ("cols", C.c_int32), ("coef", C.c_float)]


class Settings(C.Structure):
    """struct scr_settings (include/splatco_raster.h): the 12 fields of
    GaussianRasterizationSettings, same order (gaussian_renderer/__init__.py:145-158)."""
    _fields_ = [
        ("image_height", C.c_int32), ("image_width", C.c_int32),
        ("tanfovx", C.c_float), ("tanfovy", C.c_float),
        ("bg", C.c_void_p), ("scale_modifier", C.c_float),
        ("viewmatrix", C.c_void_p), ("projmatrix", C.c_void_p),
        ("sh_degree", C.c_int32), ("campos", C.c_void_p),
        ("prefiltered", C.c_int32), ("debug", C.c_int32),
    ]


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP rasterizer library is not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C splatco_amd/csrc`). "
            "There is deliberately no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for s in SYMBOLS:
        if not hasattr(lib, s):
            raise ImportError(f"{LIB_PATH} does not export {s}")
    lib.scr_abi_version.restype = C.c_int
    if lib.scr_abi_version() != ABI_VERSION:
        raise ImportError(f"{LIB_PATH} has ABI version {lib.scr_abi_version()}, expected {ABI_VERSION}")
    lib.scr_last_error.restype = C.c_char_p
    for f in ("scr_geom_bytes", "scr_binning_bytes", "scr_image_bytes", "scr_backward_scratch_bytes",
              "scr_expand_scratch_bytes"):
        getattr(lib, f).restype = C.c_size_t
    lib.scr_geom_bytes.argtypes = [C.c_int64, C.c_int32, C.c_int32]
    lib.scr_binning_bytes.argtypes = [C.c_int64, C.c_int64]
    lib.scr_image_bytes.argtypes = [C.c_int32, C.c_int32]
    lib.scr_backward_scratch_bytes.argtypes = [C.c_int64]
    vp, i64, i32 = C.c_void_p, C.c_int64, C.c_int32
    sp = C.POINTER(Settings)
    lib.scr_visible_filter.argtypes = [i64, vp, vp, vp, vp, sp, vp, vp]
    lib.scr_mark_visible.argtypes = [i64, vp, vp, vp, vp]
    lib.scr_forward_plan.argtypes = [i64, i32, vp, vp, vp, vp, vp, vp, vp, sp, vp, vp, C.POINTER(C.c_int64), vp]
    lib.scr_forward_run.argtypes = [i64, i64, i64, i64, sp, vp, vp, vp, vp, vp]
    lib.scr_forward_plan_run.argtypes = [i64, i32, vp, vp, vp, vp, vp, vp, vp, sp, vp, vp, C.POINTER(C.c_int64), vp, C.c_size_t,
                                         vp, vp, vp]
    lib.scr_forward_plan_run.restype = C.c_int
    lib.scr_debug_force_deep_lists.argtypes = [C.c_int]
    lib.scr_debug_force_deep_lists.restype = C.c_int
    lib.scr_profile_stride.argtypes = [C.c_int]
    lib.scr_profile_stride.restype = C.c_int
    lib.scr_markers_enable.argtypes = [C.c_int]
    lib.scr_markers_enable.restype = C.c_int
    lib.scr_marker_push.argtypes = [C.c_char_p]
    lib.scr_marker_push.restype = C.c_int
    lib.scr_marker_pop.argtypes = []
    lib.scr_marker_pop.restype = C.c_int
    lib.scr_backward.argtypes = [i64, i32, i64, i64, vp, vp, vp, vp, vp, sp, vp, vp, vp, vp, vp, vp,
                                 vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.scr_debug_get.argtypes = [C.c_int, i64, i64, i32, i32, vp, vp, vp, vp, vp]
    lib.scr_expand_scratch_bytes.argtypes = [C.c_int64]
    lib.scr_expand_plan.argtypes = [i64, vp, vp, C.POINTER(C.c_int64), vp]
    lib.scr_expand_run.argtypes = [i64, i32, vp, vp, vp, vp, i32] + [vp] * 11
    lib.scr_expand_backward.argtypes = [i64, i32, vp, vp, i32] + [vp] * 14 + [i64, vp]
    for f in ("scr_expand_plan", "scr_expand_run", "scr_expand_backward"):
        getattr(lib, f).restype = C.c_int
    lib.scr_plane_sample_scratch_bytes.argtypes = [C.c_int64, C.c_int32, C.c_int32, C.c_int32]
    lib.scr_plane_sample_scratch_bytes.restype = C.c_size_t
    lib.scr_plane_sample_backward.argtypes = [i64, vp, i32, i32, i32, i32, i32, i32, i32, vp, vp, i32, vp, vp, vp, vp]
    lib.scr_triplane_forward.argtypes = [i64, vp, i32, vp, vp, vp, i32, i32, i32, i32, i32, vp, i32, i32, i32, i32, vp]
    lib.scr_triplane_forward.restype = C.c_int
    lib.scr_plane_row_pairs.argtypes = [i32, i32, i32, vp, vp, vp]
    lib.scr_triplane_backward_multi_scratch_bytes.argtypes = [i64, i32, vp, vp, vp, vp]
    lib.scr_triplane_backward_multi_scratch_bytes.restype = C.c_size_t
    lib.scr_triplane_backward_multi.argtypes = [i64, vp, i32, i32, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, i32, vp, i32, vp]
    lib.scr_triplane_backward_multi.restype = C.c_int
    lib.scr_plane_row_pairs.restype = C.c_int
    lib.scr_triplane_backward_scratch_bytes.argtypes = [C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32]
    lib.scr_triplane_backward_scratch_bytes.restype = C.c_size_t
    lib.scr_triplane_backward.argtypes = [i64, vp, i32, i32, i32, i32, i32, i32, vp, i32, vp, vp, vp, vp]
    lib.scr_triplane_backward.restype = C.c_int
    lib.scr_norm_linear_scratch_bytes.argtypes = [C.c_int64]
    lib.scr_norm_linear_scratch_bytes.restype = C.c_size_t
    lib.scr_norm_linear_forward.argtypes = [i64, i32, vp, i32, vp, vp, C.c_float, vp, vp, vp, vp, vp, vp, i32, vp]
    lib.scr_norm_linear_forward.restype = C.c_int
    lib.scr_norm_linear_backward.argtypes = [i64, i32, vp, i32, vp, i32, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp]
    lib.scr_norm_linear_backward.restype = C.c_int
    lib.scr_norm_linear_dx.argtypes = [i64, i32, vp, i32, vp, i32, vp, vp, i32, vp]
    lib.scr_norm_linear_dx.restype = C.c_int
    lib.scr_box_coords.argtypes = [i64, vp, vp, vp, vp, vp]
    lib.scr_box_coords.restype = C.c_int
    lib.scr_norm_fold.argtypes = [i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.scr_norm_fold_backward.argtypes = [i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.scr_norm_running_stats.argtypes = [i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, vp]
    for f in ("scr_norm_fold", "scr_norm_fold_backward", "scr_norm_running_stats"):
        getattr(lib, f).restype = C.c_int
    lib.scr_plane_sample_backward.restype = C.c_int
    lib.scr_l1_ssim_scratch_bytes.argtypes = [i32, i32, i32, i32]
    lib.scr_l1_ssim_scratch_bytes.restype = C.c_size_t
    lib.scr_l1_ssim_forward.argtypes = [i32, i32, i32, vp, vp, vp, i32, vp, vp]
    lib.scr_l1_ssim_backward.argtypes = [i32, i32, i32, vp, vp, vp, vp, vp, vp, vp]
    lib.scr_l1_ssim_forward.restype = lib.scr_l1_ssim_backward.restype = C.c_int
    lib.scr_scaling_reg_scratch_bytes.argtypes = [i64]
    lib.scr_scaling_reg_scratch_bytes.restype = C.c_size_t
    lib.scr_scaling_reg_forward.argtypes = [i64, vp, vp, vp, vp]
    lib.scr_scaling_reg_backward.argtypes = [i64, vp, vp, vp, vp]
    lib.scr_scaling_reg_forward.restype = lib.scr_scaling_reg_backward.restype = C.c_int
    lib.scr_pair_l1_scratch_bytes.argtypes = [i64]
    lib.scr_pair_l1_scratch_bytes.restype = C.c_size_t
    lib.scr_pair_l1_forward.argtypes = [i64, vp, vp, vp, vp, vp, vp, vp]
    lib.scr_pair_l1_backward.argtypes = [i64, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.scr_pair_l1_forward.restype = lib.scr_pair_l1_backward.restype = C.c_int
    lib.scr_mask_index_plan.argtypes = [i64, vp, vp, C.POINTER(C.c_int64), vp]
    lib.scr_mask_index_run.argtypes = [i64, vp, vp, vp, vp, vp]
    lib.scr_mask_index_plan.restype = lib.scr_mask_index_run.restype = C.c_int
    lib.scr_tpa_scratch_bytes.argtypes = [i32, i32, i32]
    lib.scr_tpa_scratch_bytes.restype = C.c_size_t
    lib.scr_tpa_stats.argtypes = [i32, i32, i32] + [vp] * 8
    lib.scr_tpa_forward.argtypes = [i32, i32, i32] + [vp] * 12
    lib.scr_tpa_backward.argtypes = [i32, i32, i32] + [vp] * 18
    lib.scr_tpa_backward_stats.argtypes = [i32, i32, i32] + [vp] * 7
    lib.scr_tpa_stats.restype = lib.scr_tpa_forward.restype = C.c_int
    lib.scr_tpa_backward.restype = lib.scr_tpa_backward_stats.restype = C.c_int
    lib.scr_statis_compute.argtypes = [i64, i32, vp, vp, vp, vp, i32, vp, vp, vp]
    lib.scr_statis_apply.argtypes = [i64, i32, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.scr_statis_compute.restype = lib.scr_statis_apply.restype = C.c_int
    lib.scr_mlp_heads_hidden_bytes.argtypes = lib.scr_mlp_heads_partial_bytes.argtypes = [i64]
    lib.scr_mlp_heads_hidden_bytes.restype = lib.scr_mlp_heads_partial_bytes.restype = C.c_size_t
    lib.scr_mlp_heads_forward.argtypes = [i64, vp, i32] + [vp] * 17
    lib.scr_mlp_heads_backward.argtypes = [i64, vp, i32] + [vp] * 28
    lib.scr_mlp_heads_forward.restype = lib.scr_mlp_heads_backward.restype = C.c_int
    lib.scr_anchor_gather.argtypes = [i64] + [vp] * 10 + [i32, vp, vp]
    lib.scr_anchor_gather_stat_rows.argtypes = [i64]
    lib.scr_anchor_gather_stat_rows.restype = C.c_int32
    lib.scr_anchor_gather_stat_buffer_rows.argtypes = [i64]
    lib.scr_anchor_gather_stat_buffer_rows.restype = C.c_int64
    lib.scr_anchor_gather_backward.argtypes = [i64, i64] + [vp] * 7 + [i32] + [vp] * 4 + [i32, vp, vp, i32, vp, i32, vp]
    lib.scr_anchor_gather.restype = lib.scr_anchor_gather_backward.restype = C.c_int
    lib.scr_knn.argtypes = [i64, i32, C.POINTER(C.c_float), vp, vp, vp, vp, vp]
    lib.scr_knn_curvature.argtypes = [i64, i32, vp, vp, vp, vp]
    lib.scr_knn.restype = lib.scr_knn_curvature.restype = C.c_int
    lib.scr_adam_step.argtypes = [i32, C.POINTER(AdamTensor), C.c_double, C.c_double, C.c_double, vp]
    lib.scr_adam_step.restype = C.c_int
    lib.scr_tv_add_grad.argtypes = [i32, C.POINTER(TvPlane), vp]
    lib.scr_tv_add_grad.restype = C.c_int
    lib.scr_copy_probe.argtypes = [vp, vp, C.c_size_t, vp]
    lib.scr_copy_probe.restype = C.c_int
    lib.scr_profile_enable.argtypes = [C.c_int]
    lib.scr_profile_read.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    lib.scr_profile_kernel_name.argtypes = [C.c_int]
    lib.scr_profile_kernel_name.restype = C.c_char_p
    for f in ("scr_visible_filter", "scr_mark_visible", "scr_forward_plan", "scr_forward_run",
              "scr_backward", "scr_debug_get"):
        getattr(lib, f).restype = C.c_int
    return lib


lib = _load()


def profile_enable(which, every=1):
    """True / -1: time every kernel class; False / 0: off; a kernel name: only that class.  every: bracket only every
    `every`-th launch of a timed class (the event pair costs ~6 us of stream time around the launch it times)."""
    check(lib.scr_profile_stride(max(int(every), 1)))
    if isinstance(which, str):
        names = [lib.scr_profile_kernel_name(i).decode() for i in range(PROF_COUNT)]
        mask = 1 << names.index(which)
    else:
        mask = -1 if which is True or which == -1 else int(which)
    lib.scr_profile_enable(mask)


MARKERS = False


def markers_enable(on=True):
    """Opt-in roctx stage markers (include/splatco_raster.h, scr_markers_enable): every C-ABI entry point, every kernel class
    and the host stages below open a range that `rocprofv3 --marker-trace` records.  Also switched on by SPLATCO_MARKERS=1
    in the environment when the library is loaded."""
    global MARKERS
    check(lib.scr_markers_enable(1 if on else 0))
    MARKERS = bool(on)


class stage:
    """with stage("rasterize"): ...   A host-side roctx range around a stage of the training iteration (the reference brackets
    the whole iteration with one event pair, train.py:136-137,163,245).  Off (the default): two attribute reads.  On: the
    range is closed behind a device synchronisation, so that the kernels it launched lie INSIDE it on the profiler's
    time line -- marker mode serialises host and device at stage boundaries on purpose; never use it for timing."""
    __slots__ = ("name", "on")

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        self.on = MARKERS
        if self.on:
            lib.scr_marker_push(self.name.encode())
        return self

    def __exit__(self, *exc):
        if self.on:
            import torch
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            lib.scr_marker_pop()
        return False


def profile_read():
    """{kernel name: (total ms, launches)} since the last read (HIP events on the launch stream)."""
    ms = (C.c_double * PROF_COUNT)()
    n = (C.c_int64 * PROF_COUNT)()
    check(lib.scr_profile_read(ms, n))
    return {lib.scr_profile_kernel_name(i).decode(): (ms[i], n[i]) for i in range(PROF_COUNT)}


def check(rc):
    if rc != 0:
        raise RuntimeError("splatco_raster: " + lib.scr_last_error().decode())


def scratch_size(nbytes):
    """Size class of a large scratch request: above 32 MiB the next multiple of 1/8 of the power of two below it (eight
    classes per octave, at most 12.5 % more than asked).  The buffers behind these requests scale with the number of
    visible anchors / Gaussians / tile instances, which drift from step to step while a scene trains: asked for by the
    byte, a slowly GROWING buffer misses the caching allocator's pool on every step and costs a device allocation each
    time (cfg3: one 1.9 GB hipMalloc per step, 0.4 ms, and 2 GB more reserved per step -- profiles/HISTORY.md, round 5)."""
    n = max(int(nbytes), 1)
    if n < (32 << 20):
        return n
    step = 1 << (n.bit_length() - 4)
    return (n + step - 1) // step * step


def scratch(nbytes, device):
    """Uninitialised device bytes for a kernel's scratch / saved state, in scratch_size() classes."""
    import torch
    return torch.empty(scratch_size(nbytes), dtype=torch.uint8, device=device)


if os.environ.get("SPLATCO_MARKERS", "") not in ("", "0"):
    markers_enable(True)
