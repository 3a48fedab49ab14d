"""Minimal counterpart of the reference's training iteration around the hot path (train.py:147-312),
for the multi-view configurations of BASELINE.json (configs[3], configs[4]): render this rank's
shard of the mv views, per-view loss 0.8 L1 + 0.2 (1 - SSIM) + 0.01 mean(prod scaling) summed over
views (train.py:192-198), optionally the pairwise cross-view consistency term (train.py:201-239, weight
0.05 for update_from < iteration < update_until), ONE backward (train.py:240), SUM all-reduce of the
gradients, every 4th iteration the tri-plane total-variation term added into the summed plane gradients
(train.py:242-243; tv.py), the densification statistics of the LAST view of the mv loop (train.py:264-266) on every rank,
and the optimizer step (train.py:310-312).  Learning-rate schedules, the key-point pruning of
train.py:219-236, logging and checkpoints are out of scope."""
import torch
import torch.distributed as dist

from .losses import view_loss
from .multiview import GradArena, allreduce_gradients, collectives_on, consistency_loss, shard_views, world_info
from .renderer import prefilter_voxel, render
from .tv import tv_due
from ._C import stage


def sync_densification_stats(densifier, n_views, out, vis, device):
    """training_statis of the mv loop's LAST view (train.py:264-266 uses the variables the final loop pass left
    behind) applied on EVERY rank, so that the replicas keep identical accumulators and adjust_anchor (which every
    rank runs with an identically seeded generator) grows / prunes the same anchors everywhere.

    View mv-1 is rendered by rank (mv-1) % world as its last local view.  That rank computes the view's compact
    increments (stats.statis_increments: V + V*k floats, V = visible anchors) and broadcasts them with the visible
    anchor indices; every rank applies them (stats.statis_apply).  One header + two payload broadcasts; ~48 bytes per
    visible anchor instead of re-rendering the view or shipping the four accumulators (88 bytes per anchor)."""
    from . import stats
    rank, world = world_info()
    owner = (n_views - 1) % world if n_views > 0 else 0
    k = densifier.n_offsets
    if rank == owner:
        if out is None:
            raise RuntimeError("the rank that owns the last view rendered nothing")
        with torch.no_grad():
            inc_op, inc_g = stats.statis_increments(k, out["viewspace_points"].grad, out["neural_opacity"],
                                                    out["visibility_filter"], out["selection_mask"])
            from .expand import visible_indices
            vis_idx = visible_indices(vis)
    if collectives_on():
        head = torch.tensor([vis_idx.numel() if rank == owner else 0], dtype=torch.int64, device=device)
        dist.broadcast(head, src=owner)
        V = int(head.item())
        if rank != owner:
            vis_idx = torch.empty(V, dtype=torch.int64, device=device)
            payload = torch.empty(V + V * k, dtype=torch.float32, device=device)
        else:
            payload = torch.cat([inc_op, inc_g])
        dist.broadcast(vis_idx, src=owner)
        dist.broadcast(payload, src=owner)
        inc_op, inc_g = payload[:V], payload[V:]
    with torch.no_grad():
        densifier.apply_statis(vis_idx, inc_op, inc_g)


def collaborative_step(pc, views, gt_images, pipe, bg_color, optimizer=None, bucket=None,
                       consistency_weight=0.0, densifier=None, arena=None, iteration=None, tv_weight=0.0):
    """views / gt_images: the identically ordered mv view list every rank holds; gt_images[i] is the
    [3,H,W] target of views[i] (host or device).  densifier: a splatco_amd.densify.AnchorDensifier; its
    accumulators receive the statistics of the LAST view of the list on every rank (sync_densification_stats).
    arena: a multiview.GradArena over the trainable parameters (gradients live in one persistent buffer that is
    all-reduced in place, piecewise, overlapping the tail of the backward pass); without it the gradients are
    packed into `bucket`.
    iteration / tv_weight: the tri-plane total-variation term of train.py:242-243 -- with tv_weight > 0 (the reference's
    opt.tv_weight_a = 4e-7) and `iteration % 4 == 0` its gradient is added into the plane gradients AFTER the exchange,
    identically on every rank: the term is a function of the parameters alone, so adding it before the SUM would count it
    once per rank.  (The reference also waits for gaussians.enable_net, which is True from iteration 1 on.)
    optimizer: anything with .step() -- torch.optim.Adam, adam.FusedAdam, or adam.ShardedFusedAdam (Adam's work and moments
    divided by the number of ranks; needs the arena in mode "rs_ag").  With the sharded optimizer the gradients in the arena
    are complete on the owning rank only; the total-variation term is added over whole planes on every rank, which gives
    every owned slice its term exactly once.
    Returns (local loss sum, last render dict, bucket or arena buffer)."""
    params = arena.params if arena is not None else [p for p in pc.parameters() if p.requires_grad]
    if arena is not None:
        # the per-anchor gradients (99 % of the arena) are written in place by the gather's backward kernel; the sink is
        # attached to the model for the duration of this step only (a render() outside it must reach autograd as usual)
        if getattr(arena, "_sink_model", None) is not pc:
            arena.sink(pc)
            arena._sink_model = pc
        pc._grad_sink = getattr(arena, "_sink", None)
        if pc._grad_sink is not None and any(getattr(pc, n) is not arena.params[i] for n, i in
                                             zip(("_anchor_feat", "_anchor", "_offset", "_scaling"), arena._sink_ids)):
            pc._grad_sink = None
            raise RuntimeError("the GradArena was built from parameters the model no longer holds "
                               "(adjust_anchor / sort_anchors replace them): build a new arena")
        arena.zero()
    else:
        pc._grad_sink = None
        for p in params:
            p.grad = None
    rank, world = world_info()
    device = params[0].device
    total, out, vis, rendered = None, None, None, []
    from .adam import ShardedFusedAdam
    sharded = isinstance(optimizer, ShardedFusedAdam)
    want_union = arena is not None and arena.sparse_rows and arena.active      # (also under the sharded optimizer: set_row_union)
    union = None
    try:
        # (stage(): opt-in roctx ranges, _C.markers_enable / SPLATCO_MARKERS=1; nothing when off)
        for k, (cam, gt) in enumerate(zip(shard_views(views), shard_views(gt_images))):
            with stage("prefilter_voxel"):
                vis = prefilter_voxel(cam, pc, pipe, bg_color)
            if want_union:
                union = vis.clone() if union is None else union.logical_or_(vis)
            with stage("render"):
                out = render(cam, pc, pipe, bg_color, visible_mask=vis, retain_grad=True)
            with stage("loss"):
                gt = gt.to(out["render"].device, non_blocking=True)
                loss = view_loss(out["render"], gt, out["scaling"])
                total = loss if total is None else total + loss
            if consistency_weight:
                rendered.append((rank + k * world, out["render"], gt))
        if consistency_weight:
            with stage("consistency_loss"):
                term, _ = consistency_loss(rendered, consistency_weight, device=device)
                if term is not None:
                    total = term if total is None else total + term
        if want_union:
            # only the rows of anchors SOME view of the step sees need to travel (GradArena.set_row_union); every rank takes
            # part, also one without views
            arena.set_row_union(union)
        if total is not None:
            with stage("backward"):
                total.backward()
    finally:
        pc._grad_sink = None
    if sharded and optimizer.arena is not arena:
        raise ValueError("collaborative_step: the ShardedFusedAdam was built on another GradArena")
    with stage("gradient_exchange"):
        if arena is not None:
            # sharded optimizer: the exchange stops after the reduce-scatter; only this rank's slices of the arena hold the
            # summed gradient (the optimizer's all-gather distributes the updated PARAMETERS instead)
            bucket = arena.reduce(gather=not sharded)
        else:
            bucket = allreduce_gradients(params, bucket)
    if tv_weight and iteration is not None and tv_due(iteration):
        with stage("tv_loss"):
            pc.feat_planes.tv_loss(tv_weight)
    if densifier is not None:
        with stage("training_statis"):
            sync_densification_stats(densifier, len(views), out, vis, device)
    if optimizer is not None:
        with stage("optimizer_step"):
            optimizer.step()
    return (total.detach() if total is not None else None), out, bucket
