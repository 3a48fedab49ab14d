"""Fused neural-Gaussian expansion + opacity-mask compaction (host side of csrc/expand.hip).

One autograd op for gaussian_renderer/__init__.py:68-111: given the MLP outputs of the V visible
anchors (k offsets each) it returns the compacted per-Gaussian tensors the rasterizer consumes.
Parity: against the plain torch op chain of splatco_amd.renderer (itself pinned by the golden
fixture captured from the reference), tests/test_gpu_renderer.py.
"""
import ctypes as C

import torch

from . import _C
from .rasterizer import _ptr, _stream


class _ExpandCompact(torch.autograd.Function):
    @staticmethod
    def forward(ctx, neural_opacity, color, scale_rot, offsets, grid_scaling, anchor, k):
        # no zero "gradients" for mask / out_index (bool and int32 [V*k]: two fills of 184 M elements per backward at
        # configs[4]) nor for outputs the loss did not use: backward() takes None as zero
        ctx.set_materialize_grads(False)
        f = lambda t: t.contiguous().float()
        neural_opacity, color, scale_rot = f(neural_opacity).reshape(-1), f(color), f(scale_rot)
        grid_scaling, anchor = f(grid_scaling), f(anchor)
        V, dev = anchor.shape[0], anchor.device
        # offsets [V,k,3] may be columns 35..64 of the gather's [V,72] matrix (anchor_gather: not written a second time)
        if (offsets.dtype == torch.float32 and offsets.dim() == 3 and offsets.shape[1:] == (k, 3) and offsets.stride()[1:] == (3, 1)
                and offsets.stride(0) >= 3 * k and V > 1):
            offsets = offsets.detach()
        else:
            offsets = f(offsets).reshape(V, k, 3)
        n = V * k
        scratch = _C.scratch(_C.lib.scr_expand_scratch_bytes(n), dev)
        cnt = C.c_int64(0)
        with torch.cuda.device(dev):           # kernels launch on the CURRENT device: make it the tensors' device
            _C.check(_C.lib.scr_expand_plan(n, _ptr(neural_opacity), scratch.data_ptr(), C.byref(cnt), _stream(dev)))
        P = int(cnt.value)
        new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        xyz, col, opa, sca, rot = new(P, 3), new(P, 3), new(P, 1), new(P, 3), new(P, 4)
        out_index = torch.empty(n, dtype=torch.int32, device=dev)
        mask = torch.empty(n, dtype=torch.bool, device=dev)
        if n:
            with torch.cuda.device(dev):
                _C.check(_C.lib.scr_expand_run(V, k, _ptr(neural_opacity), _ptr(color), _ptr(scale_rot), _ptr(offsets),
                                               offsets.stride(0), _ptr(grid_scaling), _ptr(anchor), scratch.data_ptr(), out_index.data_ptr(),
                                               mask.data_ptr(), _ptr(xyz), _ptr(col), _ptr(opa), _ptr(sca), _ptr(rot),
                                               _stream(dev)))
        ctx.save_for_backward(scale_rot, offsets, grid_scaling, out_index)
        ctx.dims = (V, k)
        ctx.mark_non_differentiable(mask, out_index)
        # `tap`: a one-element output whose only purpose is its gradient -- losses.scaling_reg hands dL/dreg of the view
        # loss's regulariser mean(prod(scaling, 1)) to it, and the backward kernel adds the regulariser's gradient to
        # dL/dscaling on the fly (no [P,3] tensor, no accumulation pass)
        tap = torch.zeros(1, dtype=torch.float32, device=dev)
        return xyz, col, opa, sca, rot, tap, mask, out_index

    @staticmethod
    def backward(ctx, g_xyz, g_col, g_opa, g_sca, g_rot, g_tap, _g_mask, _g_index):
        scale_rot, offsets, grid_scaling, out_index = ctx.saved_tensors
        V, k = ctx.dims
        dev, n = scale_rot.device, V * k
        z = lambda t, *s: (torch.zeros(*s, dtype=torch.float32, device=dev) if t is None else t.contiguous().float())
        given = next((t for t in (g_xyz, g_col, g_opa, g_sca, g_rot) if t is not None), None)
        P = int((out_index >= 0).sum().item()) if given is None else given.shape[0]
        g_xyz, g_col, g_opa, g_sca, g_rot = z(g_xyz, P, 3), z(g_col, P, 3), z(g_opa, P, 1), z(g_sca, P, 3), z(g_rot, P, 4)
        g_tap = None if g_tap is None or P == 0 else g_tap.contiguous().float().reshape(1)
        new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        d_no, d_col, d_sr, d_off, d_gs, d_an = new(n, 1), new(n, 3), new(n, 7), new(V, k, 3), new(V, 6), new(V, 3)
        if n:
            with torch.cuda.device(dev):
                _C.check(_C.lib.scr_expand_backward(V, k, _ptr(scale_rot), _ptr(offsets), offsets.stride(0), _ptr(grid_scaling),
                                                    out_index.data_ptr(), _ptr(g_xyz), _ptr(g_col), _ptr(g_opa),
                                                    _ptr(g_sca), _ptr(g_rot), d_no.data_ptr(), d_col.data_ptr(),
                                                    d_sr.data_ptr(), d_off.data_ptr(), d_gs.data_ptr(), d_an.data_ptr(),
                                                    _ptr(g_tap), P, _stream(dev)))
        return d_no, d_col, d_sr, d_off, d_gs, d_an, None


def expand_compact(neural_opacity, color, scale_rot, grid_offsets, grid_scaling, anchor, n_offsets):
    """neural_opacity [V*k,1], color [V*k,3], scale_rot [V*k,7], grid_offsets [V,k,3], grid_scaling [V,6],
    anchor [V,3]  ->  xyz, color, opacity, scaling, rot (compacted, order preserved), mask [V*k] bool."""
    *out, tap, mask, out_index = _ExpandCompact.apply(neural_opacity, color, scale_rot, grid_offsets, grid_scaling, anchor,
                                                      int(n_offsets))
    if tap.requires_grad:
        out[3]._scr_reg_tap = tap      # scaling: losses.scaling_reg routes the regulariser's gradient through the tap
    # the compaction index rides along with the mask: the densification statistics (stats.selection_index) need the
    # position of every selected candidate among the Gaussians and would otherwise recompute it with a prefix sum
    mask._scr_out_index = out_index
    return (*out, mask)


def visible_indices(mask):
    """mask.nonzero().squeeze(1) for a 1-D mask: the HIP op on the GPU, torch elsewhere.  The list is remembered on the
    mask tensor (with the version it was built from): a training step asks for it twice -- the gather of render() and the
    densification statistics -- and each build is two kernels and a host read of the count.
    CONTRACT: the cache is keyed on the tensor's autograd version counter and storage address, so it follows every
    in-place torch operation on the mask -- but NOT writes that bypass the counter: a kernel writing through data_ptr()
    (e.g. scr_mark_visible into a reused buffer) or `mask.data[...] = ...`.  Masks handed to render() / training_statis
    must therefore be fresh tensors (prefilter_voxel returns one per call) or be modified through torch operations only;
    call `forget_indices(mask)` after rewriting one behind autograd's back."""
    if mask.is_cuda and mask.dim() == 1 and mask.dtype in (torch.bool, torch.uint8):
        key = (mask._version, mask.data_ptr(), mask.shape[0])
        cached = getattr(mask, "_scr_index", None)
        if cached is not None and cached[0] == key:
            return cached[1]
        idx = mask_indices(mask)
        mask._scr_index = (key, idx)
        return idx
    return mask.nonzero(as_tuple=False).squeeze(1)


def forget_indices(mask):
    """Drop the index list visible_indices() remembered on `mask` (after the mask's memory was rewritten without a torch
    in-place operation)."""
    if hasattr(mask, "_scr_index"):
        del mask._scr_index


def mask_indices(mask, inverse=True):
    """Ascending indices of the set entries of a 1-D bool / uint8 mask on the GPU -- mask.nonzero().squeeze(1) without
    torch's int64 reduction + select (0.33 ms for 20 M anchors; here one pass over the bytes and one over the index).
    inverse: the same pass also leaves the inverse map (position in the list, -1 where the mask is clear) on the result as
    `._scr_inverse`: the fused anchor gather's backward wants it and would otherwise build it with a fill, an arange and
    an index_put."""
    import ctypes as C
    m = mask.contiguous()
    m = m.view(torch.uint8) if m.dtype == torch.bool else m
    assert m.dim() == 1 and m.dtype == torch.uint8 and m.is_cuda
    n = m.shape[0]
    scratch = _C.scratch(_C.lib.scr_expand_scratch_bytes(n), m.device)
    cnt = C.c_int64(0)
    with torch.cuda.device(m.device):          # the plan call reads its count back on the host: one synchronisation
        _C.check(_C.lib.scr_mask_index_plan(n, m.data_ptr(), scratch.data_ptr(), C.byref(cnt), _stream(m.device)))
        idx = torch.empty(cnt.value, dtype=torch.int64, device=m.device)
        inv = torch.empty(n, dtype=torch.int64, device=m.device) if inverse and n else None
        if n and (cnt.value or inv is not None):
            _C.check(_C.lib.scr_mask_index_run(n, m.data_ptr(), scratch.data_ptr(), idx.data_ptr() if cnt.value else None,
                                               inv.data_ptr() if inv is not None else None, _stream(m.device)))
    if inv is not None:
        idx._scr_inverse = inv
    return idx
