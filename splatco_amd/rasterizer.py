"""Host side of the rasterizer operator: the reference's `diff_gaussian_rasterization` API
(GaussianRasterizationSettings / GaussianRasterizer, imported at gaussian_renderer/__init__.py:15)
over the C-ABI of include/splatco_raster.h.

Autograd contract: differentiable w.r.t. means3D, colors_precomp | shs, opacities,
scales + rotations | cov3D_precomp; `means2D` receives the NDC-space mean gradient although its
value is unused (read back through .grad at scene/gaussian_model.py:779); radii is int32 and
non-differentiable.  Saved state is per call, so the mv live graphs of train.py:171-240 coexist.
PyTorch is used for device memory and streams only.
"""
from typing import NamedTuple

import ctypes as C
import os

import torch
import torch.nn as nn

from . import _C


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


def _dev_f32(t, name):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a device tensor (the rasterizer has no CPU path)")
    if t.dtype != torch.float32 or not t.is_contiguous():
        t = t.contiguous().float()
    return t


def _ptr(t):
    return None if t is None or t.numel() == 0 else t.data_ptr()


class _CSettings:
    """Keeps the device tensors alive next to the C struct that points at them."""

    def __init__(self, rs: GaussianRasterizationSettings):
        self.bg = _dev_f32(rs.bg, "bg")
        self.view = _dev_f32(rs.viewmatrix, "viewmatrix")
        self.proj = _dev_f32(rs.projmatrix, "projmatrix")
        self.campos = _dev_f32(rs.campos, "campos")
        s = _C.Settings()
        s.image_height, s.image_width = int(rs.image_height), int(rs.image_width)
        s.tanfovx, s.tanfovy = float(rs.tanfovx), float(rs.tanfovy)
        s.bg, s.viewmatrix, s.projmatrix, s.campos = (self.bg.data_ptr(), self.view.data_ptr(),
                                                      self.proj.data_ptr(), self.campos.data_ptr())
        s.scale_modifier = float(rs.scale_modifier)
        s.sh_degree = int(rs.sh_degree)
        s.prefiltered, s.debug = int(bool(rs.prefiltered)), int(bool(rs.debug))
        self.c = s
        self.H, self.W = s.image_height, s.image_width

    def ref(self):
        return C.byref(self.c)


def _stream(device=None):
    """The caller's current stream ON THE TENSORS' DEVICE (the process may have another device current)."""
    return torch.cuda.current_stream(device).cuda_stream


def _bytes(n, device):
    return _C.scratch(n, device)


class RasterState:
    """Per-call saved buffers (opaque to Python) + what the debug getters need."""
    __slots__ = ("P", "M", "I", "max_tile", "flags", "cs", "geom", "binning", "image", "radii")

    def debug(self, which):
        """Integer / float intermediates for the parity tests (SCR_DBG_* selectors)."""
        dev = self.geom.device
        tiles = ((self.cs.W + 15) // 16) * ((self.cs.H + 15) // 16)
        shapes = {
            _C.DBG_TILES_TOUCHED: ((self.P,), torch.int32), _C.DBG_POINT_OFFSETS: ((self.P,), torch.int32),
            _C.DBG_RANGES: ((tiles, 2), torch.int32), _C.DBG_POINT_LIST: ((self.I,), torch.int32),
            _C.DBG_N_CONTRIB: ((self.cs.H, self.cs.W), torch.int32),
            _C.DBG_FINAL_T: ((self.cs.H, self.cs.W), torch.float32),
            _C.DBG_SPLAT_RECORDS: ((self.P, 12), torch.float32),
            _C.DBG_QMASK: ((self.I,), torch.uint8), _C.DBG_GM_INDEX: ((self.I,), torch.int32),
        }
        shape, dt = shapes[which]
        out = torch.empty(shape, dtype=dt, device=dev)
        if out.numel():
          with torch.cuda.device(dev):
            _C.check(_C.lib.scr_debug_get(which, self.P, self.I, self.cs.H, self.cs.W, _ptr(self.geom),
                                          _ptr(self.binning), _ptr(self.image), out.data_ptr(), _stream()))
        return out


SPECULATE = os.environ.get("SPLATCO_SPECULATIVE_BINNING", "1") != "0"
_plan_guess = {}           # (device, H, W) -> (P, instances, largest tile) of the last forward at that resolution: sizes the speculative binning buffer


def rasterize_forward(cs, means3D, opacities, scales, rotations, cov3D_precomp, shs, colors_precomp):
    """plan + run through the C-ABI.  Returns (color, radii, RasterState)."""
    dev = means3D.device
    P = means3D.shape[0]
    M = 0 if shs is None else shs.shape[1]
    st = RasterState()
    st.P, st.M, st.cs = P, M, cs
    st.geom = _bytes(_C.lib.scr_geom_bytes(P, cs.H, cs.W), dev)
    st.image = _bytes(_C.lib.scr_image_bytes(cs.H, cs.W), dev)
    radii = torch.empty(P, dtype=torch.int32, device=dev)      # every entry is written by preprocess_kernel
    color = torch.empty(3, cs.H, cs.W, dtype=torch.float32, device=dev)
    plan = (C.c_int64 * 4)(0, 0, 0, 0)   # (tile instances, largest per-tile instance count, phase 2 already ran, plan flags)
    # The binning buffer's size is only known after the plan phase.  A guess from the previous call of this size (the
    # instance count of a training loop moves by a few per cent per step) lets both phases go out in ONE call: the GPU
    # does not wait for this thread's allocation and second call in the middle of every forward pass.
    key = (dev.index, cs.H, cs.W)
    guess = _plan_guess.get(key)      # (P, instances, largest tile) of the last forward at this resolution
    spec = None
    with torch.cuda.device(dev):      # kernels launch on the CURRENT device: make it the tensors' device
        if SPECULATE and guess is not None and guess[0] > 0 and 0.5 <= P / guess[0] <= 2.0:
            scale = 1.06 * P / guess[0]
            cap = _C.lib.scr_binning_bytes(min(int(guess[1] * scale) + 4096, (1 << 32) - 2), max(int(guess[2] * 1.25), guess[2] + 64))
            spec = _bytes(cap, dev)
        _C.check(_C.lib.scr_forward_plan_run(P, M, _ptr(means3D), _ptr(scales), _ptr(rotations), _ptr(cov3D_precomp),
                                             _ptr(opacities), _ptr(shs), _ptr(colors_precomp), cs.ref(),
                                             st.geom.data_ptr(), _ptr(radii), plan, None if spec is None else spec.data_ptr(),
                                             0 if spec is None else spec.numel(), st.image.data_ptr(), color.data_ptr(), _stream()))
        st.I, st.max_tile, st.flags = int(plan[0]), int(plan[1]), int(plan[3])
        _plan_guess[key] = (P, st.I, st.max_tile)
        if plan[2]:
            st.binning = spec
        else:
            del spec
            st.binning = _bytes(_C.lib.scr_binning_bytes(st.I, st.max_tile), dev)
            _C.check(_C.lib.scr_forward_run(P, st.I, st.max_tile, st.flags, cs.ref(), st.geom.data_ptr(), st.binning.data_ptr(),
                                            st.image.data_ptr(), color.data_ptr(), _stream()))
    st.radii = radii
    return color, radii, st


def _debug_dump(path, raster_settings, **tensors):
    """debug=True (pipe.debug, switched on by --debug_from, train.py:179-180): the C-ABI synchronises and checks
    after every kernel; when a call fails the operator's inputs are written to `path` (snapshot_fw.dump /
    snapshot_bw.dump in the working directory, the operator family's convention) before the error propagates."""
    try:
        blob = {"raster_settings": {k: (v.detach().cpu() if isinstance(v, torch.Tensor) else v)
                                    for k, v in raster_settings._asdict().items()}}
        blob.update({k: (None if v is None else v.detach().cpu()) for k, v in tensors.items()})
        torch.save(blob, path)
        return f" (inputs written to {path})"
    except Exception as e:       # a faulted device may refuse the copies
        return f" (could not write {path}: {e})"


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                raster_settings):
        cs = _CSettings(raster_settings)
        # no zero tensors for outputs that received no gradient: autograd filled an int32 [P] "gradient" of radii with
        # zeros before every backward (88 M elements at configs[4]); backward() treats a missing dL/dcolor as zero
        ctx.set_materialize_grads(False)
        means3D = _dev_f32(means3D, "means3D")
        none_if_empty = lambda t, n: None if t is None or t.numel() == 0 else _dev_f32(t, n)
        sh, colors_precomp = none_if_empty(sh, "shs"), none_if_empty(colors_precomp, "colors_precomp")
        scales, rotations = none_if_empty(scales, "scales"), none_if_empty(rotations, "rotations")
        cov3Ds_precomp = none_if_empty(cov3Ds_precomp, "cov3D_precomp")
        opacities = _dev_f32(opacities, "opacities")
        P = means3D.shape[0]
        if P == 0:  # nothing to launch: background only
            color = cs.bg.reshape(3, 1, 1).expand(3, cs.H, cs.W).contiguous()
            radii = torch.zeros(0, dtype=torch.int32, device=means3D.device)
            ctx.state = None
            ctx.shapes = (means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp)
            ctx.mark_non_differentiable(radii)
            return color, radii
        try:
            color, radii, st = rasterize_forward(cs, means3D, opacities, scales, rotations, cov3Ds_precomp, sh,
                                                 colors_precomp)
        except RuntimeError as e:
            if raster_settings.debug:
                raise RuntimeError(str(e) + _debug_dump("snapshot_fw.dump", raster_settings, means3D=means3D, sh=sh,
                                                        colors_precomp=colors_precomp, opacities=opacities, scales=scales,
                                                        rotations=rotations, cov3Ds_precomp=cov3Ds_precomp)) from e
            raise
        ctx.state = st
        ctx.raster_settings = raster_settings
        ctx.save_for_backward(means3D, scales, rotations, cov3Ds_precomp, sh, colors_precomp, opacities)
        ctx.m2d_shape = tuple(means2D.shape)
        ctx.mark_non_differentiable(radii)
        return color, radii

    @staticmethod
    def backward(ctx, grad_out_color, _grad_radii):
        st = ctx.state
        if st is None:
            return tuple(None if t is None else torch.zeros_like(t) for t in ctx.shapes) + (None,)
        if grad_out_color is None:      # the image took no part in the loss: every gradient is zero
            return (None,) * 9
        means3D, scales, rotations, cov3D, sh, colors, opacities = ctx.saved_tensors
        dev, P, cs = means3D.device, st.P, st.cs
        g = _dev_f32(grad_out_color, "grad_out_color")
        # All per-Gaussian PARAMETER gradients are carved from ONE arena, adjacent (means2D, a per-view statistic, is a
        # tensor of its own): when the operator's inputs are leaves their .grad tensors
        # alias the arena, and multiview.allreduce_gradients reduces it in place without packing copies.
        widths = [3, 3 if colors is not None else 0, 3 * st.M if sh is not None else 0, 1,
                  3 if cov3D is None else 0, 4 if cov3D is None else 0, 6 if cov3D is not None else 0]
        arena = torch.empty(P * sum(widths), dtype=torch.float32, device=dev)
        parts, off = [], 0
        for w in widths:
            parts.append(arena[off:off + P * w].view(P, w) if w else None)
            off += P * w
        g_means3D, g_col, g_sh, g_op, g_scales, g_rot, g_cov = parts
        # a tensor of its own: the caller may KEEP it as viewspace_points.grad (renderer.keep_grad) without pinning the arena
        g_means2D = torch.empty(P, 3, dtype=torch.float32, device=dev)
        if g_sh is not None:
            g_sh = g_sh.view(P, st.M, 3)
        del arena, parts
        scratch = _bytes(_C.lib.scr_backward_scratch_bytes(st.I), dev)
        try:
          with torch.cuda.device(dev):
            _C.check(_C.lib.scr_backward(P, st.M, st.I, st.flags, _ptr(means3D), _ptr(scales), _ptr(rotations), _ptr(cov3D),
                                         _ptr(sh), cs.ref(), st.radii.data_ptr(), st.geom.data_ptr(),
                                         st.binning.data_ptr(), st.image.data_ptr(), g.data_ptr(), scratch.data_ptr(),
                                         g_means3D.data_ptr(), g_means2D.data_ptr(), _ptr(g_col), _ptr(g_sh),
                                         g_op.data_ptr(), _ptr(g_scales), _ptr(g_rot), _ptr(g_cov), _stream()))
        except RuntimeError as e:
            if ctx.raster_settings.debug:
                raise RuntimeError(str(e) + _debug_dump("snapshot_bw.dump", ctx.raster_settings, means3D=means3D, sh=sh,
                                                        colors_precomp=colors, opacities=opacities, scales=scales,
                                                        rotations=rotations, cov3Ds_precomp=cov3D, radii=st.radii,
                                                        grad_out_color=g)) from e
            raise
        g_op = g_op.reshape(opacities.shape)
        if ctx.m2d_shape != (P, 3):
            g_means2D = g_means2D[:, :ctx.m2d_shape[1]].reshape(ctx.m2d_shape) if len(ctx.m2d_shape) == 2 else None
        # order: means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, settings
        return g_means3D, g_means2D, g_sh, g_col, g_op, g_scales, g_rot, g_cov, None


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings):
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, raster_settings)


class GaussianRasterizer(nn.Module):
    """Same constructor / forward / visible_filter / markVisible as the reference operator."""

    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        with torch.no_grad():
            positions = _dev_f32(positions, "positions")
            P = positions.shape[0]
            out = torch.zeros(P, dtype=torch.uint8, device=positions.device)
            view = _dev_f32(self.raster_settings.viewmatrix, "viewmatrix")
            if P:
                with torch.cuda.device(positions.device):
                    _C.check(_C.lib.scr_mark_visible(P, positions.data_ptr(), view.data_ptr(), out.data_ptr(), _stream()))
            return out.bool()

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise ValueError('GaussianRasterizer: pass either shs or colors_precomp (one of them, not both, not neither)')
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise ValueError('GaussianRasterizer: pass either scales + rotations or cov3D_precomp (one form of the covariance, not both, not neither)')
        empty = torch.Tensor([])
        return rasterize_gaussians(
            means3D, means2D, empty if shs is None else shs, empty if colors_precomp is None else colors_precomp,
            opacities, empty if scales is None else scales, empty if rotations is None else rotations,
            empty if cov3D_precomp is None else cov3D_precomp, self.raster_settings)

    def visible_filter(self, means3D, scales=None, rotations=None, cov3D_precomp=None):
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise ValueError('GaussianRasterizer: pass either scales + rotations or cov3D_precomp (one form of the covariance, not both, not neither)')
        with torch.no_grad():
            cs = _CSettings(self.raster_settings)
            means3D = _dev_f32(means3D, "means3D")
            scales, rotations = _dev_f32(scales, "scales"), _dev_f32(rotations, "rotations")
            cov3D_precomp = _dev_f32(cov3D_precomp, "cov3D_precomp")
            P = means3D.shape[0]
            radii = torch.empty(P, dtype=torch.int32, device=means3D.device)  # filter_kernel writes every entry
            if P:
                with torch.cuda.device(means3D.device):
                    _C.check(_C.lib.scr_visible_filter(P, means3D.data_ptr(), _ptr(scales), _ptr(rotations),
                                                       _ptr(cov3D_precomp), cs.ref(), radii.data_ptr(), _stream()))
            return radii
