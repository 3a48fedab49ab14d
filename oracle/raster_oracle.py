"""ctypes/numpy front-end of oracle/raster_oracle.c  --  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
the product (splatco_amd/) never does.  PARITY UNPINNED at the rasterizer boundary (see the
header of raster_oracle.c): the reference ships neither the rasterizer source nor tests for it.

The stage names follow the reference operator's call sites
(gaussian_renderer/__init__.py:160-171 forward, :239-242 visible_filter).
"""
import ctypes as C
import os
import subprocess
from dataclasses import dataclass, field

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
TILE = 16


class _Settings(C.Structure):
    _fields_ = [
        ("image_height", C.c_int32), ("image_width", C.c_int32),
        ("tanfovx", C.c_float), ("tanfovy", C.c_float),
        ("bg", C.c_float * 3), ("scale_modifier", C.c_float),
        ("viewmatrix", C.c_float * 16), ("projmatrix", C.c_float * 16),
        ("sh_degree", C.c_int32), ("campos", C.c_float * 3),
        ("prefiltered", C.c_int32), ("debug", C.c_int32),
    ]


@dataclass
class Settings:
    """Same 12 fields, same order, as GaussianRasterizationSettings
    (gaussian_renderer/__init__.py:145-158)."""
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: np.ndarray
    scale_modifier: float
    viewmatrix: np.ndarray
    projmatrix: np.ndarray
    sh_degree: int = 1
    campos: np.ndarray = field(default_factory=lambda: np.zeros(3, np.float32))
    prefiltered: bool = False
    debug: bool = False

    def c(self):
        s = _Settings()
        s.image_height, s.image_width = int(self.image_height), int(self.image_width)
        s.tanfovx, s.tanfovy = float(self.tanfovx), float(self.tanfovy)
        s.bg[:] = [float(v) for v in np.asarray(self.bg, np.float32).reshape(3)]
        s.scale_modifier = float(self.scale_modifier)
        s.viewmatrix[:] = [float(v) for v in np.asarray(self.viewmatrix, np.float32).reshape(16)]
        s.projmatrix[:] = [float(v) for v in np.asarray(self.projmatrix, np.float32).reshape(16)]
        s.sh_degree = int(self.sh_degree)
        s.campos[:] = [float(v) for v in np.asarray(self.campos, np.float32).reshape(3)]
        s.prefiltered, s.debug = int(self.prefiltered), int(self.debug)
        return s

    @property
    def grid(self):
        return ((self.image_width + TILE - 1) // TILE, (self.image_height + TILE - 1) // TILE)


def build(force=False):
    """(Re)build both oracle libraries with oracle/Makefile (gcc only)."""
    names = ["libraster_oracle_f32.so", "libraster_oracle_f64.so", "libraster_oracle_f32_omp.so"]
    src = os.path.join(_HERE, "raster_oracle.c")
    stale = force or any(
        not os.path.exists(os.path.join(_HERE, n)) or
        os.path.getmtime(os.path.join(_HERE, n)) < os.path.getmtime(src) for n in names)
    if stale:
        subprocess.run(["make", "-C", _HERE, "-B"], check=True, stdout=subprocess.DEVNULL)


_LIBS = {}
_THREADED = False
# "_asan": the AddressSanitizer + UBSan build of the same source (oracle/Makefile target `asan`; the interpreter must be
# started with libasan / libubsan preloaded).  Parity results are the same; the run is the memory-safety check of the checker.
_VARIANT = os.environ.get("SPLATCO_ORACLE_VARIANT", "")


def use_threads(on):
    """Route the fp32 calls to the OpenMP build (all host cores; atomic gradient sums, so the last bits
    vary from run to run).  For bench.py's timed cpu_baseline only -- the parity tests use the serial build."""
    global _THREADED
    _THREADED = bool(on)


def threads():
    return int(_lib().orc_threads())


def _lib(f64=False):
    key = "f64" if f64 else ("f32_omp" if _THREADED and not _VARIANT else "f32")
    if key not in _LIBS:
        build()
        if _VARIANT:
            subprocess.run(["make", "-C", _HERE, _VARIANT.lstrip("_")], check=True, stdout=subprocess.DEVNULL)
        lib = C.CDLL(os.path.join(_HERE, f"libraster_oracle_{key}{_VARIANT}.so"))
        lib.orc_scan.restype = C.c_int64
        assert lib.orc_real_size() == (8 if f64 else 4)
        _LIBS[key] = lib
    return _LIBS[key]


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def preprocess(st, means3D, scales=None, rotations=None, cov3D_precomp=None, opacities=None,
               shs=None, colors_precomp=None, f64=False):
    """A.2 steps 1-9 for every Gaussian; returns every intermediate."""
    lib, rt = _lib(f64), (np.float64 if f64 else np.float32)
    means3D = _f32(means3D)
    P = means3D.shape[0]
    scales, rotations, cov3D_precomp = _f32(scales), _f32(rotations), _f32(cov3D_precomp)
    opacities, shs, colors_precomp = _f32(opacities), _f32(shs), _f32(colors_precomp)
    M = 0 if shs is None else shs.shape[1]
    out = dict(
        radii=np.zeros(P, np.int32), xy=np.zeros((P, 2), rt), depth=np.zeros(P, rt),
        cov3D=np.zeros((P, 6), rt), conic_opacity=np.zeros((P, 4), rt), rgb=np.zeros((P, 3), rt),
        clamped=np.zeros((P, 3), np.int32), tiles_touched=np.zeros(P, np.uint32),
        rect=np.zeros((P, 4), np.int32))
    cs = st.c()
    lib.orc_preprocess(C.c_int(P), C.c_int(M), _p(means3D), _p(scales), _p(rotations),
                       _p(cov3D_precomp), _p(opacities), _p(shs), _p(colors_precomp), C.byref(cs),
                       _p(out["radii"]), _p(out["xy"]), _p(out["depth"]), _p(out["cov3D"]),
                       _p(out["conic_opacity"]), _p(out["rgb"]), _p(out["clamped"]),
                       _p(out["tiles_touched"]), _p(out["rect"]))
    return out


def visible_filter(st, means3D, scales=None, rotations=None, cov3D_precomp=None):
    """GaussianRasterizer.visible_filter -> radii[N] int32 (gaussian_renderer/__init__.py:239-242)."""
    lib = _lib(False)
    means3D = _f32(means3D)
    P = means3D.shape[0]
    radii = np.zeros(P, np.int32)
    cs = st.c()
    scales, rotations, cov3D_precomp = _f32(scales), _f32(rotations), _f32(cov3D_precomp)
    lib.orc_preprocess(C.c_int(P), C.c_int(0), _p(means3D), _p(scales), _p(rotations),
                       _p(cov3D_precomp), None, None, None, C.byref(cs), _p(radii),
                       None, None, None, None, None, None, None, None)
    return radii


def mark_visible(st, means3D):
    lib = _lib(False)
    means3D = _f32(means3D)
    out = np.zeros(means3D.shape[0], np.uint8)
    cs = st.c()
    lib.orc_mark_visible(C.c_int(means3D.shape[0]), _p(means3D), C.byref(cs), _p(out))
    return out.astype(bool)


def binning(st, pre):
    """A.3: inclusive scan, key emit, stable sort, tile ranges."""
    lib = _lib(False)
    P = pre["radii"].shape[0]
    gx, gy = st.grid
    tt = np.ascontiguousarray(pre["tiles_touched"], np.uint32)
    offsets = np.zeros(P, np.uint64)
    I = int(lib.orc_scan(C.c_int(P), _p(tt), _p(offsets))) if P else 0
    keys = np.zeros(max(I, 1), np.uint64)
    ids = np.zeros(max(I, 1), np.uint32)
    ranges = np.zeros((gx * gy, 2), np.uint32)
    depth32 = np.ascontiguousarray(pre["depth"], np.float32)
    rect = np.ascontiguousarray(pre["rect"], np.int32)
    lib.orc_bin(C.c_int(P), _p(tt), _p(rect), _p(depth32), C.c_int(gx), C.c_int(gy),
                C.c_int64(I), _p(keys), _p(ids), _p(ranges))
    return dict(point_offsets=offsets, num_rendered=I, keys_sorted=keys[:I], point_list=ids[:I],
                ranges=ranges)


def blend_forward(st, pre, bins, f64=False, want_margin=True):
    lib, rt = _lib(f64), (np.float64 if f64 else np.float32)
    H, W = st.image_height, st.image_width
    color = np.zeros((3, H, W), rt)
    final_T = np.zeros((H, W), rt)
    n_contrib = np.zeros((H, W), np.uint32)
    margin = np.zeros((H, W), rt) if want_margin else None
    ids = np.ascontiguousarray(bins["point_list"] if bins["num_rendered"] else np.zeros(1, np.uint32))
    bg = _f32(st.bg)
    lib.orc_blend_forward(C.c_int(H), C.c_int(W), _p(bins["ranges"]), _p(ids),
                          _p(np.ascontiguousarray(pre["xy"], rt)),
                          _p(np.ascontiguousarray(pre["conic_opacity"], rt)),
                          _p(np.ascontiguousarray(pre["rgb"], rt)), _p(bg), _p(color), _p(final_T),
                          _p(n_contrib), _p(margin))
    return dict(color=color, final_T=final_T, n_contrib=n_contrib, margin=margin)


def forward(st, means3D, opacities, scales=None, rotations=None, cov3D_precomp=None, shs=None,
            colors_precomp=None, f64=False):
    """GaussianRasterizer.forward (gaussian_renderer/__init__.py:163-171): returns
    color[3,H,W], radii[P] and every saved / debug intermediate."""
    pre = preprocess(st, means3D, scales, rotations, cov3D_precomp, np.asarray(opacities).reshape(-1),
                     shs, colors_precomp, f64=f64)
    bins = binning(st, pre)
    img = blend_forward(st, pre, bins, f64=f64)
    out = {}
    out.update(pre)
    out.update(bins)
    out.update(img)
    return out


def blend_backward(st, fwd, dL_dcolor_img, f64=False, raw_moments=False):
    """Screen-space gradient sums of the blend: (dL/dmean2D [P,2] in pixel units, dL/dconic [P,3], dL/dopacity [P],
    dL/dcolour [P,3]).  raw_moments: the same sums in the device kernel's formulation (raw moments per 8x8 quadrant, shifted
    to the splat's centre; orc_blend_backward_raw_moments; "reassociated": that function's control, per-quadrant partial
    sums of the centred terms without any shift) -- a study leg, see tools/exp/shift_study.py."""
    lib, rt = _lib(f64), (np.float64 if f64 else np.float32)
    H, W = st.image_height, st.image_width
    P = fwd["radii"].shape[0]
    dpix = np.ascontiguousarray(dL_dcolor_img, rt)
    g_m2 = np.zeros((P, 2), rt)
    g_conic = np.zeros((P, 3), rt)
    g_op = np.zeros(P, rt)
    g_col = np.zeros((P, 3), rt)
    ids = np.ascontiguousarray(fwd["point_list"] if fwd["num_rendered"] else np.zeros(1, np.uint32))
    bg = _f32(st.bg)
    args = (C.c_int(H), C.c_int(W), _p(fwd["ranges"]), _p(ids), _p(np.ascontiguousarray(fwd["xy"], rt)),
            _p(np.ascontiguousarray(fwd["conic_opacity"], rt)), _p(np.ascontiguousarray(fwd["rgb"], rt)), _p(bg),
            _p(np.ascontiguousarray(fwd["final_T"], rt)), _p(fwd["n_contrib"]), _p(dpix), C.c_int(P), _p(g_m2), _p(g_conic),
            _p(g_op), _p(g_col))
    if raw_moments:      # "reassociated": per-quadrant partial sums of the CENTRED terms, no shift (the control)
        lib.orc_blend_backward_raw_moments(*args, C.c_int(1 if raw_moments == "reassociated" else 0))
    else:
        lib.orc_blend_backward(*args)
    return g_m2, g_conic, g_op, g_col


def preprocess_backward(st, fwd, g_m2, g_conic, g_col, means3D, scales=None, rotations=None, cov3D_precomp=None,
                        shs=None, f64=False):
    """The per-Gaussian chain from the screen-space sums to the operator inputs (linear in g_m2 / g_conic / g_col)."""
    lib, rt = _lib(f64), (np.float64 if f64 else np.float32)
    means3D = _f32(means3D)
    P = means3D.shape[0]
    scales, rotations, cov3D_precomp, shs = _f32(scales), _f32(rotations), _f32(cov3D_precomp), _f32(shs)
    M = 0 if shs is None else shs.shape[1]
    g_m2, g_conic, g_col = (np.ascontiguousarray(a, rt) for a in (g_m2, g_conic, g_col))
    g_means3D = np.zeros((P, 3), rt)
    g_means2D = np.zeros((P, 3), rt)
    g_scales = np.zeros((P, 3), rt) if cov3D_precomp is None else None
    g_rot = np.zeros((P, 4), rt) if cov3D_precomp is None else None
    g_cov = np.zeros((P, 6), rt) if cov3D_precomp is not None else None
    g_sh = np.zeros((P, M, 3), rt) if shs is not None else None
    cs = st.c()
    lib.orc_preprocess_backward(C.c_int(P), C.c_int(M), _p(means3D), _p(scales), _p(rotations),
                                _p(cov3D_precomp), _p(shs), C.byref(cs), _p(fwd["radii"]),
                                _p(fwd["clamped"]), _p(g_m2), _p(g_conic), _p(g_col), _p(g_means3D),
                                _p(g_means2D), _p(g_scales), _p(g_rot), _p(g_cov), _p(g_sh))
    return dict(means3D=g_means3D, means2D=g_means2D, sh=g_sh, scales=g_scales, rotations=g_rot, cov3D_precomp=g_cov)


def backward(st, fwd, dL_dcolor_img, means3D, scales=None, rotations=None, cov3D_precomp=None,
             shs=None, colors_precomp=None, f64=False):
    """Rasterizer backward: gradients in the operator's input order."""
    P = np.asarray(means3D).shape[0]
    g_m2, g_conic, g_op, g_col = blend_backward(st, fwd, dL_dcolor_img, f64=f64)
    out = preprocess_backward(st, fwd, g_m2, g_conic, g_col, means3D, scales, rotations, cov3D_precomp, shs, f64=f64)
    out.update(colors_precomp=g_col if colors_precomp is not None else None, opacities=g_op.reshape(P, 1),
               _mean2D_px=g_m2, _conic=g_conic)
    return out
