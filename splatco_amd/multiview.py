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


def allreduce_gradients(params: Iterable[torch.Tensor], bucket: torch.Tensor = None) -> torch.Tensor:
    """SUM-all-reduce the .grad of every tensor in `params` through one flat bucket, in place.
    Parameters without a gradient on this rank contribute zeros (a rank whose views do not see
    an anchor still takes part).  Returns the bucket (reusable)."""
    params = [p for p in params if p is not None and p.requires_grad]
    if not params:
        return bucket
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


def multiview_step(views: Sequence, params: Sequence[torch.Tensor],
                   render_loss: Callable[[object], torch.Tensor], bucket: torch.Tensor = None):
    """One collaborative step: this rank renders its shard of `views`, sums the per-view losses,
    runs ONE backward (as train.py:240 does) and all-reduces the gradients.  After the call every
    rank holds d(sum over ALL views of loss)/d(params) -- identical to the sequential mv loop.
    Returns (local loss sum, bucket)."""
    for p in params:
        p.grad = None
    total = None
    for v in shard_views(views):
        loss = render_loss(v)
        total = loss if total is None else total + loss
    if total is not None:
        total.backward()
    bucket = allreduce_gradients(params, bucket)
    return (total.detach() if total is not None else torch.zeros((), device=params[0].device)), bucket
