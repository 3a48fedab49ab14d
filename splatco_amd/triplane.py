"""Tri-plane feature sampling with an MI355X-native backward (host side of csrc/triplane.hip).

plane_sample(plane [1,R,A,B], grid [V,2]) == F.grid_sample(plane, grid.view(1,1,V,2), bilinear,
align_corners=True).flatten(0,2).T  (scene/grids.py:148-150).  The forward IS torch's grid_sample
(a gather); the backward w.r.t. the plane replaces torch's one-global-atomic-per-(point, corner,
channel) scatter by tile-bucketed LDS accumulation.  The sample positions get no gradient: the
reference detaches them (scene/gaussian_model.py:210).
"""
import torch
import torch.nn.functional as F

from . import _C
from .rasterizer import _stream


class _PlaneSample(torch.autograd.Function):
    @staticmethod
    def forward(ctx, plane, grid):
        V = grid.shape[0]
        out = F.grid_sample(plane, grid.view(1, 1, V, 2), mode="bilinear", align_corners=True).flatten(0, 2).T
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
        _C.check(_C.lib.scr_plane_sample_backward(V, grid.data_ptr(), R, A, B, g.data_ptr(), grad_plane.data_ptr(),
                                                  scratch.data_ptr(), _stream()))
        return grad_plane, None


def plane_sample(plane, grid):
    """plane [1,R,A,B] (R <= 8), grid [V,2] = (x -> dim B, y -> dim A) in [-1,1], detached."""
    return _PlaneSample.apply(plane, grid.detach().contiguous().float())
