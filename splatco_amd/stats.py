"""Densification statistics on the device: host side of csrc/densify.hip.

Consumer of the rasterizer's dL/dmeans2D output -- what GaussianModel.training_statis
(scene/gaussian_model.py:761-782, called at train.py:264-266) accumulates per anchor and per offset,
and therefore what pins the shape / units contract of that gradient (norm over [:, :2] of the
NDC-space gradient).  Two HIP kernels over the V visible anchors of the view:

    statis_increments(...) -> (inc_opacity[V], inc_grad[V*k])      the view's contribution, compact
    statis_apply(...)                                              adds it to the four accumulators

so that the sharded --mv step can broadcast the increments of the LAST view from the rank that
rendered it (train_step.sync_densification_stats).  Device tensors only: there is no CPU path.
Fixture: tests/golden/training_statis.npz (captured from the reference's own function).
"""
import torch

from . import _C
from .rasterizer import _stream


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("densification statistics run on the GPU only (csrc/densify.hip); got a CPU tensor")


def selection_index(offset_selection_mask):
    """Compaction index of every candidate (position among the selected ones, -1 when dropped).  The fused
    expansion kernel already produced it (expand.py attaches it to the mask it returns); any other mask gets
    one prefix sum."""
    idx = getattr(offset_selection_mask, "_scr_out_index", None)
    if idx is not None and idx.shape == offset_selection_mask.shape and idx.device == offset_selection_mask.device:
        return idx
    m = offset_selection_mask.reshape(-1)
    pos = torch.cumsum(m, 0, dtype=torch.int32) - 1
    return torch.where(m, pos, torch.full_like(pos, -1))


def statis_increments(n_offsets, viewspace_point_grad, opacity, update_filter, offset_selection_mask):
    """The view's contribution to the accumulators: (inc_opacity [V], inc_grad [V*k], -1 = not counted)."""
    _need_cuda(viewspace_point_grad, opacity, update_filter, offset_selection_mask)
    k = int(n_offsets)
    op = opacity.detach().reshape(-1).contiguous().float()
    V = op.numel() // k
    out_index = selection_index(offset_selection_mask).contiguous()
    grad = viewspace_point_grad.detach()
    if grad.dtype != torch.float32 or not grad.is_contiguous():
        grad = grad.contiguous().float()
    upd = update_filter.contiguous()
    if upd.dtype != torch.bool and upd.dtype != torch.uint8:
        upd = upd != 0
    dev = op.device
    inc_op = torch.empty(V, dtype=torch.float32, device=dev)
    inc_g = torch.empty(V * k, dtype=torch.float32, device=dev)
    if V:
        with torch.cuda.device(dev):
            _C.check(_C.lib.scr_statis_compute(V, k, op.data_ptr(), out_index.data_ptr(),
                                               upd.data_ptr() if upd.numel() else None,
                                               grad.data_ptr() if grad.numel() else None,
                                               grad.shape[1] if grad.dim() == 2 else 2,
                                               inc_op.data_ptr(), inc_g.data_ptr(), _stream()))
    return inc_op, inc_g


def statis_apply(opacity_accum, anchor_demon, offset_gradient_accum, offset_denom, n_offsets, visible_index,
                 inc_opacity, inc_grad):
    """Adds one view's increments to the accumulators, in place.  visible_index [V] int64: the anchor of
    every visible row (anchor_visible_mask.nonzero())."""
    accs = (opacity_accum, anchor_demon, offset_gradient_accum, offset_denom)
    _need_cuda(*accs, visible_index, inc_opacity, inc_grad)
    for a in accs:
        if a.dtype != torch.float32 or not a.is_contiguous():
            raise RuntimeError("accumulators must be contiguous float32 tensors")
    V = int(visible_index.numel())
    if V:
        visible_index = visible_index.contiguous().long()
        with torch.cuda.device(opacity_accum.device):
            _C.check(_C.lib.scr_statis_apply(V, int(n_offsets), visible_index.data_ptr(), inc_opacity.data_ptr(),
                                             inc_grad.data_ptr(), opacity_accum.data_ptr(), anchor_demon.data_ptr(),
                                             offset_gradient_accum.data_ptr(), offset_denom.data_ptr(), _stream()))
    return accs


def training_statis(opacity_accum, anchor_demon, offset_gradient_accum, offset_denom, n_offsets,
                    viewspace_point_grad, opacity, update_filter, offset_selection_mask, anchor_visible_mask):
    """GaussianModel.training_statis as a function over the four accumulators (updated in place, returned)."""
    inc_op, inc_g = statis_increments(n_offsets, viewspace_point_grad, opacity, update_filter, offset_selection_mask)
    from .expand import visible_indices
    vis_idx = visible_indices(anchor_visible_mask)
    return statis_apply(opacity_accum, anchor_demon, offset_gradient_accum, offset_denom, n_offsets, vis_idx,
                        inc_op, inc_g)
