"""Minimal counterpart of the reference's training iteration around the hot path (train.py:147-312),
for the multi-view configurations of BASELINE.json (configs[3], configs[4]): render this rank's
shard of the mv views, per-view loss 0.8 L1 + 0.2 (1 - SSIM) + 0.01 mean(prod scaling) summed over
views (train.py:192-198), ONE backward (train.py:240), SUM all-reduce of the gradients, optimizer
step (train.py:310-312).  Densification, the cross-view consistency loss, logging and checkpoints
are out of scope."""
import torch

from .losses import view_loss
from .multiview import allreduce_gradients, shard_views
from .renderer import prefilter_voxel, render


def collaborative_step(pc, views, gt_images, pipe, bg_color, optimizer=None, bucket=None):
    """views / gt_images: the identically ordered mv view list every rank holds; gt_images[i] is the
    [3,H,W] target of views[i] (host or device).  Returns (local loss sum, last render dict, bucket)."""
    params = [p for p in pc.parameters() if p.requires_grad]
    for p in params:
        p.grad = None
    total, out = None, None
    for cam, gt in zip(shard_views(views), shard_views(gt_images)):
        vis = prefilter_voxel(cam, pc, pipe, bg_color)
        out = render(cam, pc, pipe, bg_color, visible_mask=vis, retain_grad=True)
        loss = view_loss(out["render"], gt.to(out["render"].device, non_blocking=True), out["scaling"])
        total = loss if total is None else total + loss
    if total is not None:
        total.backward()
    bucket = allreduce_gradients(params, bucket)
    if optimizer is not None:
        optimizer.step()
    return (total.detach() if total is not None else None), out, bucket
