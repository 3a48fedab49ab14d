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


class _TriPlaneSample(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ind, cols, *planes):
        # ind [V,3] normalised coordinates; planes = k triples (xy [1,R,X,Y], xz [1,R,X,Z], yz [1,R,Y,Z]);
        # cols[j] = first output column of plane j
        V, R = ind.shape[0], planes[0].shape[1]
        ld = R * len(planes)
        out = torch.empty(V, ld, dtype=torch.float32, device=ind.device)
        # random gathers are bound by the number of cache lines requested: for many points the planes are
        # first copied to channel-last [A,B,R] (a streaming pass), which cuts the lines per sampled row from R to 1-2
        cl = 1 if V >= CHANNEL_LAST_MIN_POINTS else 0
        for t in range(0, len(planes), 3):
            xy, xz, yz = planes[t:t + 3]
            X, Y, Z = xy.shape[2], xy.shape[3], xz.shape[3]
            assert xz.shape[2] == X and yz.shape[2] == Y and yz.shape[3] == Z, "plane shapes do not form a tri-plane"
            xy, xz, yz = ((p.permute(0, 2, 3, 1) if cl else p).contiguous() for p in (xy, xz, yz))
            _C.check(_C.lib.scr_triplane_forward(V, ind.data_ptr(), ind.stride(0), xy.data_ptr(), xz.data_ptr(), yz.data_ptr(),
                                                 R, X, Y, Z, cl, out.data_ptr(), ld, cols[t], cols[t + 1], cols[t + 2],
                                                 _stream()))
        ctx.save_for_backward(ind)
        ctx.cols, ctx.shapes = cols, [tuple(p.shape) for p in planes]
        return out

    @staticmethod
    def backward(ctx, g):
        (ind,) = ctx.saved_tensors
        V = ind.shape[0]
        g = g.contiguous().float()
        grads = []
        for j, shape in enumerate(ctx.shapes):
            if not ctx.needs_input_grad[2 + j]:
                grads.append(None)
                continue
            _, R, A, B = shape
            cx, cy = _PAIRS[j % 3]
            gp = torch.empty(shape, dtype=torch.float32, device=g.device)
            scratch = torch.empty(_C.lib.scr_plane_sample_scratch_bytes(V, A, B), dtype=torch.uint8, device=g.device)
            _C.check(_C.lib.scr_plane_sample_backward(V, ind.data_ptr(), ind.stride(0), cx, cy, R, A, B,
                                                      g.data_ptr() + 4 * ctx.cols[j], g.stride(0), gp.data_ptr(),
                                                      scratch.data_ptr(), _stream()))
            grads.append(gp)
        return (None, None, *grads)


def triplane_sample(ind, planes, cols=None):
    """ind [V,3] in [-1,1] (detached); planes: 3 or 6 tensors (xy, xz, yz[, xyA, xzA, yzA]), R <= 8 channels each.
    Returns [V, len(planes)*R]; plane j occupies columns cols[j] .. cols[j]+R (default: in the order given)."""
    R = planes[0].shape[1]
    if cols is None:
        cols = tuple(R * j for j in range(len(planes)))
    ind = ind.detach().float()
    if ind.stride(1) != 1:
        ind = ind.contiguous()
    return _TriPlaneSample.apply(ind, tuple(cols), *planes)


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
        scratch = torch.empty(_C.lib.scr_plane_sample_scratch_bytes(V, A, B), dtype=torch.uint8, device=g.device)
        _C.check(_C.lib.scr_plane_sample_backward(V, grid.data_ptr(), 2, 0, 1, R, A, B, g.data_ptr(), g.stride(0),
                                                  grad_plane.data_ptr(), scratch.data_ptr(), _stream()))
        return grad_plane, None
