"""The tri-plane total-variation term of the reference's training step, added into the plane gradients in place.

Reference: train.py:242-243 -- `if gaussians.enable_net and iteration % 4 == 0 and not args.no_regularization:
gaussians.feat_planes.tv_loss(opt.tv_weight_a)` between backward() and optimizer.step(); enable_net is switched on at
iteration 1 (train.py:299-302), tv_weight_a = 4e-7 (arguments/__init__.py:169).  tv_loss (scene/gaussian_model.py:217-220)
gives grid `level` of the ACTIVE levels the weight w * 0.5^(2 - level); PlaneGrid.total_variation_add_grad
(scene/grids.py:240-250) builds six smooth-L1 sums over neighbour differences per grid, divides by 6 and calls
.backward(), which accumulates into the three planes' .grad.

Here: the closed-form derivative, one launch of scr_tv_add_grad (csrc/tv.hip) for all planes of all active grids, no
autograd graph.  The term is a function of the parameters only -- in a sharded step it must be added ONCE, after the
gradient SUM, identically on every rank (train_step.collaborative_step does that).  No CPU path."""
import ctypes as C

import numpy as np
import torch

TV_WEIGHT_A = 4e-7       # arguments/__init__.py:169
TV_EVERY = 4             # train.py:242


def tv_due(iteration, enable_net=True, no_regularization=False):
    """The reference's condition for the term at `iteration` (train.py:242)."""
    return bool(enable_net) and iteration % TV_EVERY == 0 and not no_regularization


def level_weight(w, level):
    """scene/gaussian_model.py:219 (a Python double, as there)."""
    return w * ((0.5) ** (2 - level))


def tv_coef(w):
    """What autograd multiplies the clamped difference with: DivBackward's fp32 1/6 times MulBackward's scalar w (cast
    to fp32 by the tensor-scalar multiply)."""
    return float(np.float32(1.0) / np.float32(6.0) * np.float32(w))


def grid_planes(grid):
    return (grid.xy_plane, grid.xz_plane, grid.yz_plane)


def tv_add_grad(entries):
    """entries: [(plane parameter [1, R, A, B], weight w)].  Adds d/dplane of  w / 6 * (smooth_l1(rows) + smooth_l1(cols))
    into plane.grad (allocated as zeros where a plane has none, as the reference's .backward() would leave it)."""
    from . import _C
    entries = [(p, w) for p, w in entries if p.numel() > 0]
    if not entries:
        return
    dev = entries[0][0].device
    table = (_C.TvPlane * len(entries))()
    for e, (p, w) in zip(table, entries):
        if not p.is_cuda:
            raise RuntimeError("tv_add_grad: the planes must live on the GPU (csrc/tv.hip; there is no CPU path)")
        if p.device != dev:
            raise ValueError("tv_add_grad: all planes of one call must live on one GPU")
        if p.dtype != torch.float32 or p.dim() != 4 or p.shape[0] != 1 or not p.is_contiguous():
            raise ValueError(f"tv_add_grad: expected a contiguous fp32 [1, R, A, B] plane, got {p.dtype} {tuple(p.shape)}")
        if p.grad is None:
            p.grad = torch.zeros_like(p)
        g = p.grad
        if g.dtype != torch.float32 or g.shape != p.shape or not g.is_contiguous() or g.device != dev:
            raise ValueError("tv_add_grad: plane.grad must be a contiguous fp32 tensor of the plane's shape on its device")
        e.plane, e.grad = p.data_ptr(), g.data_ptr()
        e.channels, e.rows, e.cols = p.shape[1], p.shape[2], p.shape[3]
        e.coef = tv_coef(w)
    with torch.cuda.device(dev):
        _C.check(_C.lib.scr_tv_add_grad(len(entries), table, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))


def feature_planes_tv(feat, w):
    """GaussianLearner.tv_loss(w) for a FeaturePlanes module: every grid k0s[0 .. activate_level] with its level weight,
    one launch.  Note the index: k0s = [attention grid, plain, plain (x2), plain (x4)], and the reference's loop runs over
    k0s[level] for level in range(activate_level + 1) -- the same grids FeaturePlanes.forward samples."""
    entries = []
    for level in range(feat.activate_level + 1):
        wl = level_weight(w, level)
        entries += [(p, wl) for p in grid_planes(feat.k0s[level])]
    tv_add_grad(entries)
