"""Minimal counterpart of the reference's training iteration around the hot path (train.py:147-312),
for the multi-view configurations of BASELINE.json (configs[3], configs[4]): render this rank's
shard of the mv views, per-view loss 0.8 L1 + 0.2 (1 - SSIM) + 0.01 mean(prod scaling) summed over
views (train.py:192-198), optionally the pairwise cross-view consistency term (train.py:201-239, weight
0.05 for update_from < iteration < update_until), ONE backward (train.py:240), SUM all-reduce of the
gradients, the densification statistics of the last view (train.py:266) and the optimizer step
(train.py:310-312).  Learning-rate schedules, the key-point pruning of train.py:219-236, logging and
checkpoints are out of scope."""
import torch

from .losses import view_loss
from .multiview import allreduce_gradients, consistency_loss, shard_views, world_info
from .renderer import prefilter_voxel, render


def collaborative_step(pc, views, gt_images, pipe, bg_color, optimizer=None, bucket=None,
                       consistency_weight=0.0, densifier=None):
    """views / gt_images: the identically ordered mv view list every rank holds; gt_images[i] is the
    [3,H,W] target of views[i] (host or device).  densifier: a splatco_amd.densify.AnchorDensifier whose
    accumulators are fed from this rank's LAST rendered view (the reference uses the last view of the mv
    loop, train.py:266; with sharded views the caller applies them on the rank that owns it).
    Returns (local loss sum, last render dict, bucket)."""
    params = [p for p in pc.parameters() if p.requires_grad]
    for p in params:
        p.grad = None
    rank, world = world_info()
    total, out, vis, rendered = None, None, None, []
    for k, (cam, gt) in enumerate(zip(shard_views(views), shard_views(gt_images))):
        vis = prefilter_voxel(cam, pc, pipe, bg_color)
        out = render(cam, pc, pipe, bg_color, visible_mask=vis, retain_grad=True)
        gt = gt.to(out["render"].device, non_blocking=True)
        loss = view_loss(out["render"], gt, out["scaling"])
        total = loss if total is None else total + loss
        if consistency_weight:
            rendered.append((rank + k * world, out["render"], gt))
    if consistency_weight:
        term, _ = consistency_loss(rendered, consistency_weight)
        if term is not None:
            total = term if total is None else total + term
    if total is not None:
        total.backward()
    bucket = allreduce_gradients(params, bucket)
    if densifier is not None and out is not None:
        with torch.no_grad():
            densifier.training_statis(out["viewspace_points"], out["neural_opacity"], out["visibility_filter"],
                                      out["selection_mask"], vis)
    if optimizer is not None:
        optimizer.step()
    return (total.detach() if total is not None else None), out, bucket
