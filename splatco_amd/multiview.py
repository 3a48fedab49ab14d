"""Multi-view collaborative branch (--mv N) sharded over the GPUs of one node.

The reference renders the mv views one after another on a single GPU, SUMS the per-view losses
and calls backward once (train.py:171-240; mv default 4, arguments/__init__.py:100).  Views are
independent through render + per-view loss and the parameters are shared, so with one process
per GPU (rank r renders views r, r+world, ...) the only exchange is one all-reduce(SUM) of the
parameter gradients before optimizer.step() (train.py:311).  SUM, not MEAN, reproduces
`total_loss += loss` (train.py:198).

torch.distributed is used as plumbing: backend "nccl" is RCCL over xGMI on MI355X; "gloo" runs
the same code on CPU (tests).  Gradients travel as ONE flat fp32 bucket: xGMI is point to point
(7 links x ~153 GB/s per GPU), so a ring all-reduce is bound by a single link and a few large
messages beat many small ones.
"""
from typing import Callable, Iterable, List, Sequence

import torch
import torch.distributed as dist


def world_info():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_views(views: Sequence, rank: int = None, world: int = None) -> List:
    """Views this rank renders: rank, rank+world, ... (every rank must hold the identically
    ordered list, e.g. the identically seeded viewpoint stack of train.py:173-175)."""
    if rank is None:
        rank, world = world_info()
    return list(views[rank::world])


def _shared_arena(params):
    """If every parameter has a contiguous gradient and the gradients tile one gap-free range of one
    storage (the rasterizer's backward allocates them that way), return that range as a flat tensor."""
    grads = [p.grad for p in params]
    if any(g is None or not g.is_contiguous() or g.dtype != grads[0].dtype for g in grads):
        return None
    st = grads[0].untyped_storage()
    if any(g.untyped_storage().data_ptr() != st.data_ptr() for g in grads):
        return None
    order = sorted(grads, key=lambda g: g.storage_offset())
    pos = order[0].storage_offset()
    for g in order:
        if g.storage_offset() != pos:
            return None
        pos += g.numel()
    start = order[0].storage_offset()
    return torch.empty(0, dtype=grads[0].dtype, device=grads[0].device).set_(st, start, (pos - start,))


def allreduce_gradients(params: Iterable[torch.Tensor], bucket: torch.Tensor = None) -> torch.Tensor:
    """SUM-all-reduce the .grad of every tensor in `params` through one flat bucket, in place.
    Parameters without a gradient on this rank contribute zeros (a rank whose views do not see
    an anchor still takes part).  Returns the bucket (reusable)."""
    params = [p for p in params if p is not None and p.requires_grad]
    if not params:
        return bucket
    arena = _shared_arena(params)
    if arena is not None:      # the gradients already sit side by side in one buffer: reduce it in place
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(arena, op=dist.ReduceOp.SUM)
        return arena
    n = sum(p.numel() for p in params)
    dev, dt = params[0].device, params[0].dtype
    if bucket is None or bucket.numel() != n or bucket.device != dev:
        bucket = torch.empty(n, dtype=dt, device=dev)
    off = 0
    for p in params:
        k = p.numel()
        if p.grad is None:
            bucket[off:off + k].zero_()
        else:
            bucket[off:off + k].copy_(p.grad.reshape(-1))
        off += k
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(bucket, op=dist.ReduceOp.SUM)
    off = 0
    for p in params:
        k = p.numel()
        g = bucket[off:off + k].view_as(p)
        if p.grad is None:
            p.grad = g.clone()
        else:
            p.grad.copy_(g)
        off += k
    return bucket


def align_images(*imgs):                                        # train.py:79-96
    h, w = min(i.shape[1] for i in imgs), min(i.shape[2] for i in imgs)
    return tuple(i[:, :h, :w] for i in imgs)


def pair_consistency(gen1, real1, gen2, real2):
    """Cross-view consistency term of one view pair (train.py:208-217):
    ssim(real1, real2) * |mean |(real1 - real2) - (gen1 - gen2)||  if the two ground-truth views are alike
    (SSIM > 0.6), else 0."""
    from .losses import l1_loss, ssim
    gen1, gen2, real1, real2 = align_images(gen1, gen2, real1, real2)
    s = ssim(real1, real2)
    if not bool(s > 0.6):
        return None
    return s * torch.abs(l1_loss(real1 - real2, gen1 - gen2))


def consistency_loss(local: Sequence, weight: float = 0.05):
    """Sum over ALL view pairs of pair_consistency, times `weight` (train.py:201-239), in the sharded
    setting: `local` = [(global view index, rendered image, gt image), ...] of this rank.  The rendered
    and ground-truth images of the other ranks are all-gathered as constants; a pair with one remote
    view is evaluated on both owning ranks, each differentiating its own image only, so the gradients
    summed over ranks equal those of the single-process pairwise sum.  Returns (term to add to this
    rank's loss before backward, this rank's share of the loss VALUE -- cross-rank pairs count half)."""
    rank, world = world_info()
    items = [(int(i), g, r, True) for i, g, r in local]
    if world > 1:
        mine = [(int(i), g.detach(), r.detach()) for i, g, r in local]
        everyone = [None] * world
        dist.all_gather_object(everyone, [(i, tuple(g.shape)) for i, g, _ in mine])
        for src in range(world):
            for i, shape in everyone[src]:
                if src == rank:
                    g, r = next((g, r) for j, g, r in mine if j == i)
                    pair = torch.stack([g, r]).contiguous()
                else:
                    pair = torch.empty((2,) + shape, dtype=local[0][1].dtype if local else torch.float32,
                                       device=local[0][1].device if local else None)
                dist.broadcast(pair, src=src)
                if src != rank:
                    items.append((i, pair[0], pair[1], False))
    items.sort(key=lambda t: t[0])
    grad_term, value = None, 0.0
    for a in range(len(items)):
        for b in range(a + 1, len(items)):
            (_, g1, r1, own1), (_, g2, r2, own2) = items[a], items[b]
            if not (own1 or own2):
                continue
            t = pair_consistency(g1, r1, g2, r2)
            if t is None:
                continue
            grad_term = t if grad_term is None else grad_term + t
            value = value + t.detach() * (1.0 if (own1 and own2) else 0.5)
    if grad_term is None:
        return None, value
    return weight * grad_term, weight * value


def multiview_step(views: Sequence, params: Sequence[torch.Tensor],
                   render_loss: Callable[[object], torch.Tensor], bucket: torch.Tensor = None,
                   consistency_weight: float = 0.0):
    """One collaborative step: this rank renders its shard of `views`, sums the per-view losses,
    runs ONE backward (as train.py:240 does) and all-reduces the gradients.  After the call every
    rank holds d(sum over ALL views of loss)/d(params) -- identical to the sequential mv loop.
    With consistency_weight > 0 (train.py: 0.05 for update_from < iteration < update_until)
    `render_loss(view)` must return (loss, rendered image, gt image) and the pairwise cross-view term is
    added (see consistency_loss).  Returns (local loss sum, bucket)."""
    for p in params:
        p.grad = None
    total = None
    rank, world = world_info()
    rendered = []
    for k, v in enumerate(shard_views(views)):
        loss = render_loss(v)
        if consistency_weight:
            loss, img, gt = loss
            rendered.append((rank + k * world, img, gt))
        total = loss if total is None else total + loss
    if consistency_weight:
        term, _ = consistency_loss(rendered, consistency_weight)
        if term is not None:
            total = term if total is None else total + term
    if total is not None:
        total.backward()
    bucket = allreduce_gradients(params, bucket)
    return (total.detach() if total is not None else torch.zeros((), device=params[0].device)), bucket
