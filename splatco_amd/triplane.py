"""Tri-plane feature sampling on MI355X (host side of csrc/triplane.hip).

`triplane_sample(ind, planes)` is the sampling part of `PlaneGrid.compute_planes_feat`
(scene/grids.py:146-182): every (xy, xz, yz) plane triple is sampled with
F.grid_sample(bilinear, align_corners=True, zeros padding) at the coordinate pairs [1,0], [2,0], [2,1]
and the results land side by side in one [V, n*R] matrix (what the reference builds with torch.cat).
Forward: one kernel launch per plane triple writes straight into the concatenated matrix.
Backward w.r.t. the planes: torch's one-global-atomic-per-(point, corner, channel) scatter is replaced
by tile-bucketed LDS accumulation, reading the column slice of the incoming gradient in place.
The sample positions get no gradient: the reference detaches them (scene/gaussian_model.py:210).
"""
import torch

from . import _C
from .rasterizer import _stream

# (column of ind holding grid-x -> last plane dim, column holding grid-y) for xy / xz / yz (scene/grids.py:148-150)
_PAIRS = ((1, 0), (2, 0), (2, 1))
CHANNEL_LAST_MIN_POINTS = 262144
FUSE_GRIDS = True        # the backward of all grids of an op in one pass over the points when the layout allows (False: grid by grid)


def _forward_into(out, ind, cols, planes):
    """Samples the plane triples of one grid at ind [V,3] into the columns cols[j].. of out [V, ld]."""
    V, R = ind.shape[0], planes[0].shape[1]
    cl = 1 if V >= CHANNEL_LAST_MIN_POINTS else 0
    for t in range(0, len(planes), 3):
        xy, xz, yz = planes[t:t + 3]
        X, Y, Z = xy.shape[2], xy.shape[3], xz.shape[3]
        assert xz.shape[2] == X and yz.shape[2] == Y and yz.shape[3] == Z, "plane shapes do not form a tri-plane"
        if cl:
            # many points: random gathers are bound by the number of cache lines requested -- the planes are first
            # rewritten as ROW PAIRS [A-1,B,2,R] (texels (a, b) and (a + 1, b) side by side: one streaming pass, twice the
            # plane's size), so that the four corners of a sample are 4 R consecutive floats: 1.6 cache lines per sample
            # at R = 5 where the reference layout needs 2 R and a channel-last copy 2.6
            cl = 2

            def pairs(p):
                p = p.contiguous()
                rp = torch.empty(p.shape[2] - 1, p.shape[3], 2, R, dtype=torch.float32, device=p.device)
                with torch.cuda.device(p.device):
                    _C.check(_C.lib.scr_plane_row_pairs(R, p.shape[2], p.shape[3], p.data_ptr(), rp.data_ptr(), _stream(p.device)))
                return rp
            xy, xz, yz = pairs(xy), pairs(xz), pairs(yz)
        else:
            xy, xz, yz = (p.contiguous() for p in (xy, xz, yz))
        _C.check(_C.lib.scr_triplane_forward(V, ind.data_ptr(), ind.stride(0), xy.data_ptr(), xz.data_ptr(), yz.data_ptr(),
                                             R, X, Y, Z, cl, out.data_ptr(), out.stride(0), cols[t], cols[t + 1], cols[t + 2],
                                             _stream()))


def _backward_from(g, ind, cols, shapes):
    """Gradients of the planes of one grid (3 or 6 planes; planes j and j + 3 -- plain / attended,
    scene/grids.py:174-181 -- are sampled at the same positions): the three projections x (1 or 2) planes go
    through ONE pass over the points.  g [V, ld] (unit column stride) is read in place."""
    import ctypes as C
    V, n = ind.shape[0], len(shapes)
    R, X, Y, Z = shapes[0][1], shapes[0][2], shapes[0][3], shapes[1][3]
    gp = [torch.empty(s, dtype=torch.float32, device=g.device) for s in shapes]
    scratch = torch.empty(_C.scratch_size(_C.lib.scr_triplane_backward_scratch_bytes(V, X, Y, Z, R * (n // 3))), dtype=torch.uint8,
                          device=g.device)
    c_cols = (C.c_int32 * n)(*cols)
    ptrs = (C.c_void_p * n)(*[t.data_ptr() for t in gp])
    _C.check(_C.lib.scr_triplane_backward(V, ind.data_ptr(), ind.stride(0), R, X, Y, Z, n // 3, g.data_ptr(), g.stride(0),
                                          c_cols, ptrs, scratch.data_ptr(), _stream()))
    return gp


class _TriPlaneSample(torch.autograd.Function):
    """Several grids at once: meta = ((n_planes, cols), ...) per grid, tensors = ind_0, planes of grid 0..., ind_1, ...
    The samples of all grids land in ONE matrix (the reference concatenates the grids' outputs,
    scene/gaussian_model.py:160-166) whose gradient is read in place by every grid's backward."""

    @staticmethod
    def forward(ctx, meta, width, box, *tensors):
        ctx.box = box
        k, inds, shapes = 0, [], []
        out = None
        for n, cols in meta:
            ind, planes = tensors[k], tensors[k + 1:k + 1 + n]
            if out is None:
                out = torch.empty(ind.shape[0], width, dtype=torch.float32, device=ind.device)
            with torch.cuda.device(ind.device):
                _forward_into(out, ind, cols, planes)
            inds.append(ind)
            shapes.append([tuple(p.shape) for p in planes])
            k += 1 + n
        ctx.save_for_backward(*inds)
        ctx.meta, ctx.shapes = meta, shapes
        return out

    @staticmethod
    def backward(ctx, g):
        from .anchor_gather import DeferredDx
        box = ctx.box
        nl = None
        if box is not None and box.coef is not None:
            # the BatchNorm-Linear that consumed the matrix left its coefficients instead of dx (anchor_gather.DeferredDx):
            # the fused pass over the points forms every point's gradient row itself
            nl = box
            if DeferredDx.is_token(g):
                g = None                                     # the stride-0 zeros it returned: nothing else arrived
        if g is not None and (g.dtype != torch.float32 or g.stride(1) != 1):     # a column block of a wider matrix is read in place
            g = g.contiguous().float()
        inds = ctx.saved_tensors
        try:
            fused = _backward_all(ctx, g, inds, nl)
            if fused is not None:
                return fused
            if nl is not None:                               # a layout outside the fused pass: the matrix after all
                dx = nl.materialise()
                g = dx if g is None else g + dx
        finally:
            if box is not None:
                box.clear()
        grads, k = [], 0
        for (n, cols), ind, shapes in zip(ctx.meta, inds, ctx.shapes):
            need = ctx.needs_input_grad[3 + k + 1:3 + k + 1 + n]
            with torch.cuda.device(ind.device):
                gp = _backward_from(g, ind, cols, shapes) if any(need) else [None] * n
            grads.append(None)
            grads.extend(t if nd else None for t, nd in zip(gp, need))
            k += 1 + n
        return (None, None, None, *grads)


def _backward_all(ctx, g, inds, nl=None):
    """All grids of the op in ONE pass over the points (scr_triplane_backward_multi) when they were sampled at the same
    coordinates (the same tensor), each is a plain triple, their column blocks lie back to back in standard order and
    the library knows the channel layout; None otherwise (the caller goes grid by grid)."""
    import ctypes as C
    ng = len(ctx.meta)
    if not FUSE_GRIDS or ng > 3 or any(i.data_ptr() != inds[0].data_ptr() or i.shape != inds[0].shape or i.stride() != inds[0].stride() for i in inds):
        return None
    R, X, Y, Z, col = [], [], [], [], []
    for (n, cols), shapes in zip(ctx.meta, ctx.shapes):
        r = shapes[0][1]
        if n != 3 or tuple(cols) != (cols[0], cols[0] + r, cols[0] + 2 * r):
            return None
        R.append(r); X.append(shapes[0][2]); Y.append(shapes[0][3]); Z.append(shapes[1][3]); col.append(cols[0])
    if not all(ctx.needs_input_grad[3:][j] for j in range(len(ctx.needs_input_grad) - 3) if j % 4 != 0):
        return None                                          # a plane without a gradient: the per-grid path skips whole grids
    ind, V = inds[0], inds[0].shape[0]
    arr = lambda v: (C.c_int32 * ng)(*v)
    cR, cX, cY, cZ, ccol = arr(R), arr(X), arr(Y), arr(Z), arr(col)
    dev = ind.device
    gp = [torch.empty(s, dtype=torch.float32, device=dev) for shapes in ctx.shapes for s in shapes]
    ptrs = (C.c_void_p * (3 * ng))(*[t.data_ptr() for t in gp])
    nlargs = (None, None, 0, None, 0) if nl is None else (nl.coef.data_ptr(), nl.dy.data_ptr(), nl.dy.stride(0), nl.x.data_ptr(),
                                                        nl.x.stride(0))
    with torch.cuda.device(dev):
        nbytes = _C.lib.scr_triplane_backward_multi_scratch_bytes(V, ng, cR, cX, cY, cZ)
        scratch = _C.scratch(max(int(nbytes), 16), dev)
        rc = _C.lib.scr_triplane_backward_multi(V, ind.data_ptr(), ind.stride(0), ng, cR, cX, cY, cZ, ccol,
                                                None if g is None else g.data_ptr(), 0 if g is None else g.stride(0), ptrs,
                                                scratch.data_ptr(), *nlargs, _stream())
    if rc == 3:
        return None
    _C.check(rc)
    out, k = [], 0
    for (n, _), shapes in zip(ctx.meta, ctx.shapes):
        out.append(None)
        out.extend(gp[k:k + n])
        k += n
    return (None, None, None, *out)


def _prep_ind(ind):
    ind = ind.detach().float()
    return ind if ind.stride(1) == 1 else ind.contiguous()


def multi_triplane_sample(grids):
    """grids: [(ind [V,3], planes (3 or 6 tensors), cols (first output column of every plane))...]; returns the
    [V, width] matrix with every grid's samples in its columns (width = the largest column end)."""
    meta, flat, width, prepped = [], [], 0, {}
    for ind, planes, cols in grids:
        R = planes[0].shape[1]
        meta.append((len(planes), tuple(int(c) for c in cols)))
        if id(ind) not in prepped:                           # the same coordinates for several grids stay ONE tensor
            prepped[id(ind)] = _prep_ind(ind)
        flat.append(prepped[id(ind)])
        flat.extend(planes)
        width = max(width, max(cols) + R)
    from .anchor_gather import DeferredDx
    box = DeferredDx(width) if torch.is_grad_enabled() else None
    out = _TriPlaneSample.apply(tuple(meta), width, box, *flat)
    if box is not None:
        out._scr_deferred_dx = box      # a fused BatchNorm-Linear that consumes the whole matrix leaves coefficients instead of dx
    return out


def triplane_sample(ind, planes, cols=None):
    """ind [V,3] in [-1,1] (detached); planes: 3 or 6 tensors (xy, xz, yz[, xyA, xzA, yzA]), R <= 16 channels each.
    Returns [V, len(planes)*R]; plane j occupies columns cols[j] .. cols[j]+R (default: in the order given)."""
    R = planes[0].shape[1]
    if cols is None:
        cols = tuple(R * j for j in range(len(planes)))
    return multi_triplane_sample([(ind, tuple(planes), tuple(cols))])


def plane_sample(plane, grid):
    """Single plane [1,R,A,B] sampled at grid [V,2] = (x -> dim B, y -> dim A): F.grid_sample(plane,
    grid.view(1,1,V,2), bilinear, align_corners=True).flatten(0,2).T with the LDS backward."""
    return _PlaneSample.apply(plane, grid.detach().contiguous().float())


class _PlaneSample(torch.autograd.Function):
    @staticmethod
    def forward(ctx, plane, grid):
        V = grid.shape[0]
        out = torch.nn.functional.grid_sample(plane, grid.view(1, 1, V, 2), mode="bilinear", align_corners=True).flatten(0, 2).T
        ctx.save_for_backward(grid)
        ctx.shape = tuple(plane.shape)
        return out.contiguous()

    @staticmethod
    def backward(ctx, g):
        (grid,) = ctx.saved_tensors
        _, R, A, B = ctx.shape
        V = grid.shape[0]
        g = g.contiguous().float()
        grad_plane = torch.empty(ctx.shape, dtype=torch.float32, device=g.device)
        scratch = _C.scratch(_C.lib.scr_plane_sample_scratch_bytes(V, A, B, R), g.device)
        with torch.cuda.device(g.device):
            _C.check(_C.lib.scr_plane_sample_backward(V, grid.data_ptr(), 2, 0, 1, R, A, B, 1, g.data_ptr(), g.data_ptr(),
                                                      g.stride(0), grad_plane.data_ptr(), grad_plane.data_ptr(),
                                                      scratch.data_ptr(), _stream(g.device)))
        return grad_plane, None
