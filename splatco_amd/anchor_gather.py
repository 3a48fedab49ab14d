"""The head of generate_neural_gaussians as one autograd op (host side of csrc/anchor_gather.hip):
gaussian_renderer/__init__.py:23-31 -- the four visible-anchor gathers, exp(_scaling) and the [V,71]
concatenation that feeds FeaturePlanes' attribute branch.  Device tensors only."""
import torch

from . import _C
from .rasterizer import _stream


import os

_DEBUG = bool(int(os.environ.get("SPLATCO_DEBUG", "0")))      # debug checks that cost a device pass + a host read


def supported(pc):
    try:
        return bool(pc._anchor_feat.is_cuda and pc._anchor_feat.shape[1] == 32 and pc._offset.shape[1:] == (10, 3)
                    and pc._scaling.shape[1] == 6 and pc._anchor_feat.dtype == torch.float32)
    except (AttributeError, IndexError):
        return False


def fused_gather_taken(pc, fused_heads=True):
    """THE predicate for "render() sends this model's per-anchor parameters through the fused gather": the renderer
    (generate_neural_gaussians) and whoever attaches a gradient sink (multiview.GradArena.sink) must decide alike -- a
    sink attached to a model that then takes the index_select path would have autograd ADD into gradient memory the
    arena did not clear.  The fused kernel reproduces get_scaling = exp(_scaling) (scene/gaussian_model.py:397-399)
    only, so a subclass that overrides the property is not eligible."""
    if not (fused_heads and supported(pc)):
        return False
    from .scene_model import AnchorGaussianModel
    return type(pc).get_scaling is AnchorGaussianModel.get_scaling


class GradSink:
    """Where the backward of the gather writes the gradients of (_anchor_feat, _anchor, _offset, _scaling) directly: the
    parameters' own .grad memory (views of multiview.GradArena's buffer).  Every gradient of those four parameters comes
    through this one op, and its kernel overwrites EVERY element (zeros for invisible anchors), so the first view of a
    step needs neither a zero-filled buffer nor autograd's `grad += new` pass over 71 floats per anchor (5.7 GB read
    twice and written once at 20 M anchors); further views of the same step add in the kernel.  `fresh` is set by the
    owner at the start of a step and cleared by the first write.
    `pending` counts the forward passes of the step whose backward has not run yet; the backward that brings it to zero
    is the one after which the four gradients are FINAL.  If `ranges` ([(n0, n1)] anchor ranges, 64-aligned starts) and
    `on_range` are set, that last backward runs one launch per range and calls on_range(r) behind each, so the owner can
    put range r's exchange on the wire while range r + 1 is computed (multiview.GradArena)."""

    def __init__(self, feat, anchor, offset, scaling):
        self.tensors = (feat, anchor, offset, scaling)
        self.fresh = True
        self.pending = 0
        self.ranges, self.on_range = None, None


class DeferredDx:
    """The hand-over of d g_fea (round 6).  g_fea has ONE consumer on the render path, the BatchNorm-Linear of
    FeaturePlanes' attribute branch (scene_model._NormLinearFn), whose backward would end by writing dx [V,72] for this
    op's backward to read back.  gather_anchors hangs one of these on g_fea; the BatchNorm-Linear's backward, finding it,
    runs only its reductions, leaves the three coefficient blocks of dx = k0 + x k1 + dy Gi here (`coef`, with `dy` and
    `x` kept alive) and returns a stride-0 ZERO tensor as g_fea's gradient -- a valid gradient that costs no memory and
    that autograd may add to any other consumer's gradient of g_fea; this op's backward then forms the rows of dx inside
    its kernel (csrc/anchor_gather.hip, DX) and adds whatever gradient did arrive for g_fea.  If the BatchNorm-Linear's
    backward never runs (g_fea unused, or the unfused fallback), `coef` stays None and nothing changes."""
    __slots__ = ("coef", "dy", "x", "width")

    def __init__(self, width=71):
        self.coef = self.dy = self.x = None
        self.width = width          # columns of the matrix this producer can form dx for

    def clear(self):
        self.coef = self.dy = self.x = None

    def materialise(self):
        """dx [V, width] as a real matrix (16-byte aligned rows) from the coefficients: for a producer that finds it cannot
        form the rows inside its own backward after all."""
        from . import _C
        from .rasterizer import _stream
        V, d = self.x.shape
        ld = (d + 3) // 4 * 4
        dx = torch.empty(V, ld, dtype=torch.float32, device=self.x.device)[:, :d]
        with torch.cuda.device(self.x.device):
            _C.check(_C.lib.scr_norm_linear_dx(V, d, self.x.data_ptr(), self.x.stride(0), self.dy.data_ptr(), self.dy.stride(0),
                                               self.coef.data_ptr(), dx.data_ptr(), ld, _stream(self.x.device)))
        return dx

    @staticmethod
    def is_token(t):
        return t is not None and t.dim() == 2 and t.stride() == (0, 0)


class _AnchorGather(torch.autograd.Function):
    @staticmethod
    def forward(ctx, idx, sink, box, anchor_feat, anchor, offset, scaling):
        c = lambda t: t.detach().contiguous()
        anchor_feat, anchor, offset, scaling = c(anchor_feat), c(anchor), c(offset), c(scaling)
        idx = idx.contiguous().long()
        V, N, dev = idx.numel(), anchor.shape[0], anchor.device
        new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        # g_fea rows padded to 72 floats: 16-byte aligned rows for the fused BatchNorm-Linear (csrc/normlinear.hip).
        # feat and the offsets are NOT written a second time: they are columns 0..31 / 35..64 of g_fea, and their readers
        # (csrc/mlp_heads.hip, csrc/expand.hip) take a row stride -- 62 of 143 floats per anchor less to write (1.1 GB at
        # configs[2]).  The two small ones (anchor 3, scaling 6) keep their packed copies.
        g72 = new(V, 72)
        anc, gs, g_fea = new(V, 3), new(V, 6), g72[:, :71]
        feat, off = g72[:, :32], g72[:, 35:65].unflatten(1, (10, 3))
        # the column statistics of g_fea for the BatchNorm that reads it, formed while the rows are in LDS (one row of partial
        # sums per workgroup): the fused BatchNorm-Linear skips its own pass over the matrix (gather_anchors hangs them on g_fea)
        stats = new(_C.lib.scr_anchor_gather_stat_buffer_rows(V), 2, 80) if V else new(0, 2, 80)
        if V:
            with torch.cuda.device(dev):
                _C.check(_C.lib.scr_anchor_gather(V, idx.data_ptr(), anchor_feat.data_ptr(), anchor.data_ptr(),
                                                  offset.data_ptr(), scaling.data_ptr(), None, anc.data_ptr(),
                                                  None, gs.data_ptr(), g72.data_ptr(), 72, stats.data_ptr(), _stream()))
        inv = getattr(idx, "_scr_inverse", None)       # left by expand.mask_indices: position of every anchor in idx, -1 = invisible
        if inv is not None and (inv.shape != (N,) or inv.device != dev):
            inv = None
        ctx.save_for_backward(idx, gs, *(() if inv is None else (inv,)))
        ctx.N, ctx.sink, ctx.box = N, sink, box
        if sink is not None:
            sink.pending += 1
        stats = stats[:_C.lib.scr_anchor_gather_stat_rows(V)] if V else stats      # what the consumer reads (per-tile rows behind it)
        ctx.mark_non_differentiable(stats)
        return feat, anc, off, gs, g_fea, stats

    @staticmethod
    def backward(ctx, d_feat, d_anc, d_off, d_gs, d_g_fea, _d_stats=None):
        idx, gs, *rest = ctx.saved_tensors
        N, V, dev = ctx.N, idx.numel(), idx.device
        if rest:
            inv = rest[0]
        else:
            inv = torch.full((N,), -1, dtype=torch.long, device=dev)
            inv[idx] = torch.arange(V, device=dev)
        p = lambda t: None if t is None else t.contiguous().float()
        ldg = 71
        box, nl = ctx.box, (None, None, 0, None, 0)
        if box is not None and box.coef is not None:
            # the BatchNorm-Linear's backward left its coefficients instead of dx: the kernel forms the rows (DeferredDx)
            nl = (box.coef.data_ptr(), box.dy.data_ptr(), box.dy.stride(0), box.x.data_ptr(), box.x.stride(0))
            if DeferredDx.is_token(d_g_fea):
                d_g_fea = None                      # the stride-0 zeros it returned for g_fea: nothing else arrived
        if d_g_fea is not None and d_g_fea.dtype == torch.float32 and d_g_fea.stride() == (72, 1):
            ldg = 72                                # rows as the fused BatchNorm-Linear backward leaves them: read in place
        else:
            d_g_fea = p(d_g_fea)
        d_feat, d_anc, d_off, d_gs = p(d_feat), p(d_anc), p(d_off), p(d_gs)
        ptr = lambda t: None if t is None or t.numel() == 0 else t.data_ptr()
        sink = ctx.sink
        ranges = [(0, N)]
        if sink is not None:
            g_feat, g_anchor, g_offset, g_scaling = sink.tensors
            accumulate = 0 if sink.fresh else 1
            sink.fresh = False
            sink.pending -= 1
            if sink.pending == 0 and sink.ranges and sink.on_range is not None:
                ranges = sink.ranges                # the gradients are final after this pass: hand them over range by range
        else:
            new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
            g_feat, g_anchor, g_offset, g_scaling = new(N, 32), new(N, 3), new(N, 10, 3), new(N, 6)
            accumulate = 0
        with torch.cuda.device(dev):
            for r, (n0, n1) in enumerate(ranges):
                # a range is the same kernel on offset pointers: a workgroup owns 64 consecutive anchors, their visible
                # ones own consecutive upstream rows wherever the range starts (n0 is a multiple of 64)
                if n1 > n0:
                    _C.check(_C.lib.scr_anchor_gather_backward(
                        n1 - n0, V, inv.data_ptr() + 8 * n0, ptr(gs), ptr(d_feat), ptr(d_anc), ptr(d_off), ptr(d_gs),
                        ptr(d_g_fea), ldg, g_feat.data_ptr() + 4 * 32 * n0, g_anchor.data_ptr() + 4 * 3 * n0,
                        g_offset.data_ptr() + 4 * 30 * n0, g_scaling.data_ptr() + 4 * 6 * n0, accumulate, *nl, _stream(dev)))
                if sink is not None and ranges is sink.ranges:
                    sink.on_range(r)
        if box is not None:
            box.clear()                             # (the box outlives the graph on g_fea: let the tensors go)
        if sink is not None:
            return None, None, None, None, None, None, None        # written where the optimiser reads them
        return None, None, None, g_feat, g_anchor, g_offset, g_scaling


def gather_anchors(pc, idx):
    """(feat [V,32], anchor [V,3], grid_offsets [V,10,3], grid_scaling [V,6] = exp(_scaling), g_fea [V,71]) of the
    visible anchors idx [V] (int64, STRICTLY ASCENDING and duplicate-free: what visible_indices returns; any other list
    gives wrong gradients -- SPLATCO_DEBUG=1 checks it, at the price of a host read).  With pc._grad_sink set (train_step.collaborative_step with a GradArena)
    the gradients of the four parameters go straight into their .grad memory."""
    if _DEBUG and idx.numel() > 1 and not bool((idx[1:] > idx[:-1]).all()):
        raise RuntimeError("gather_anchors: the index list must be strictly ascending (visible_indices of a mask); the "
                           "backward kernel hands the visible anchors of 64 consecutive anchors consecutive upstream rows")
    sink = getattr(pc, "_grad_sink", None)
    if sink is not None and not torch.is_grad_enabled():
        sink = None                                 # no backward will come
    if sink is not None:
        ok = all(t.is_cuda and t.is_contiguous() and t.dtype == torch.float32 and t.shape == q.shape
                 for t, q in zip(sink.tensors, (pc._anchor_feat, pc._anchor, pc._offset, pc._scaling)))
        if not ok:      # falling back to autograd here would ADD into gradient memory the arena did not clear
            raise RuntimeError("the gradient sink does not match the model's per-anchor parameters "
                               "(rebuild the GradArena after adjust_anchor / sort_anchors)")
    box = DeferredDx() if torch.is_grad_enabled() else None
    feat, anc, off, gs, g_fea, stats = _AnchorGather.apply(idx, sink, box, pc._anchor_feat, pc._anchor, pc._offset, pc._scaling)
    if stats.shape[0]:
        g_fea._scr_col_stats = stats          # scene_model._norm_linear hands them to the fused BatchNorm-Linear
        g_fea._scr_col_stats_version = g_fea._version      # ... while nobody has edited g_fea (or feat / offsets, its aliases) in place
    if box is not None:
        g_fea._scr_deferred_dx = box          # ... and lets its backward leave coefficients instead of dx (DeferredDx)
    return feat, anc, off, gs, g_fea
