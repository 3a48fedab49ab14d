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
import os
from typing import Callable, Iterable, List, Sequence

import torch
import torch.distributed as dist


def world_info():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


_FORCE_COLLECTIVES = os.environ.get("SPLATCO_FORCE_COLLECTIVES", "0") == "1"


def force_collectives(on: bool):
    """A ONE-rank process group normally issues no collective at all (every exchange below is skipped when there is nobody
    to exchange with).  force_collectives(True) (or SPLATCO_FORCE_COLLECTIVES=1) makes a one-rank group run the very same
    sequence of collectives a larger one does -- each a copy onto itself -- so that the communication library (RCCL on
    MI355X) loads, builds its communicator and executes every call shape of the exchange on a single GPU.  Set it before
    building a GradArena (the arena fixes its behaviour at construction)."""
    global _FORCE_COLLECTIVES
    _FORCE_COLLECTIVES = bool(on)


def collectives_on() -> bool:
    """Do the exchange steps issue their collectives?  More than one rank -- or a one-rank group under force_collectives."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or _FORCE_COLLECTIVES


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


def _agree(flag: bool, signature: int, device) -> bool:
    """True only if `flag` is True and `signature` identical on EVERY rank (one 12-byte MIN all-reduce).  The
    in-place and the packed path move the same number of elements in different orders, so a split decision -- or
    two ranks whose gradients tile their buffers in different orders -- would sum misaligned data silently."""
    if not collectives_on():
        return flag
    t = torch.tensor([1 if flag else 0, signature, -signature], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    f, a, b = t.tolist()      # one host read
    return bool(f) and int(a) == -int(b)


class _Agreement:
    """What agree="once" remembers: the tensors it was made for (weakly -- densification and sort_anchors replace
    parameters and CPython reuses ids, so an id()-keyed table would hand a new parameter set a stale decision), this
    rank's (flag, signature) at the time, and the decision the ranks reached."""
    __slots__ = ("refs", "local", "decision", "__weakref__")

    def __init__(self, params, local, decision):
        import weakref
        self.refs = tuple(weakref.ref(p) for p in params)
        self.local, self.decision = local, decision

    def covers(self, params):
        return len(self.refs) == len(params) and all(r() is p for r, p in zip(self.refs, params))


def _sum_over_ranks(flat: torch.Tensor, shape: str = "all_reduce"):
    """SUM of a flat buffer over the ranks, in place.  shape "rs_ag": reduce_scatter_tensor + all_gather_into_tensor on the
    same memory (every rank owns 1/world of the sum in between; on the point-to-point xGMI mesh both phases use all links)
    when the length divides by the world size -- otherwise, and for "all_reduce", one all_reduce."""
    world = dist.get_world_size()
    if shape == "rs_ag" and flat.numel() % world == 0 and flat.numel() > 0:
        n = flat.numel() // world
        mine = flat[dist.get_rank() * n:(dist.get_rank() + 1) * n]
        dist.reduce_scatter_tensor(mine, flat, op=dist.ReduceOp.SUM)
        dist.all_gather_into_tensor(flat, mine)
    else:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)


def allreduce_gradients(params: Iterable[torch.Tensor], bucket: torch.Tensor = None, agree: str = "always",
                        shape: str = "all_reduce") -> torch.Tensor:
    """SUM-all-reduce the .grad of every tensor in `params` through one flat bucket, in place.
    Parameters without a gradient on this rank contribute zeros (a rank whose views do not see
    an anchor still takes part).  Returns the bucket (reusable).  Every rank must pass the same
    parameters in the same order.  (The training step uses GradArena instead: no packing at all.)
    agree: "always" -- the ranks agree on the in-place / packed path with a small MIN all-reduce and a host read on every
    call; "once" -- on the first call for these parameters only (a loop whose ranks run the same program: the host read
    would otherwise stand between every backward pass and its gradient exchange); a rank whose own situation changes
    afterwards raises instead of exchanging misaligned data.
    shape: "all_reduce" | "rs_ag" (see _sum_over_ranks); every rank must pass the same."""
    params = [p for p in params if p is not None and p.requires_grad]
    if not params:
        return bucket
    if not collectives_on():
        # nobody to exchange with: the gradients are final where they are (packing and unpacking them would copy
        # 2 x 1.4 GB per step at 5 M anchors for nothing); the shared arena is still what the caller gets when there is one
        arena = _shared_arena(params)
        return arena if arena is not None else bucket
    dev, dt = params[0].device, params[0].dtype
    arena = _shared_arena(params)
    # the gradients already sit side by side in one buffer (the rasterizer's backward allocates them that way):
    # reduce that buffer in place -- provided EVERY rank is in that situation with the same layout (which
    # parameter sits where), otherwise all ranks pack in the caller's parameter order
    sig = 0
    if arena is not None:
        order = sorted(range(len(params)), key=lambda i: params[i].grad.storage_offset())
        for i in order:
            sig = (sig * 1000003 + i + 1) % (1 << 61)
    if agree == "once":
        # the agreement lives ON the first parameter (an attribute of the tensor object): it dies with the tensors it was
        # made for, nothing grows when a caller passes fresh leaves every step
        seen = getattr(params[0], "_scr_agreement", None)
        if seen is None or not seen.covers(params):
            seen = _Agreement(params, (arena is not None, sig), _agree(arena is not None, sig, dev))
            params[0]._scr_agreement = seen
        elif seen.local != (arena is not None, sig):
            # rank-local: the peers are already inside their all_reduce and will block there until this process dies
            # (the launcher tears the job down); a caller whose ranks may diverge must use agree="always"
            raise RuntimeError("allreduce_gradients(agree='once'): this rank's gradient layout changed after the ranks agreed "
                               "on the exchange path; call with agree='always'")
        in_place = seen.decision
    else:
        in_place = _agree(arena is not None, sig, dev)
    if in_place:
        if collectives_on():
            _sum_over_ranks(arena, shape)
        return arena
    n = sum(p.numel() for p in params)
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
    if collectives_on():
        _sum_over_ranks(bucket, shape)
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


class GradArena:
    """ONE persistent flat buffer (fp32 in the product) that IS the .grad of every parameter of the training step.

    The per-anchor parameters of a SplatCo scene carry 71 floats of gradient per anchor (anchor 3 + offset 30 +
    feature 32 + scaling 6, scene/gaussian_model.py:502-507): 1.42 GB at 5 M anchors, 5.68 GB at 20 M.  Packing them
    into a bucket and unpacking after the collective would copy that twice per step.  Here every p.grad is a view
    of the arena from the start: autograd accumulates into it in place (AccumulateGrad adds into an existing
    .grad), the collective runs on slices of the same memory, and the optimiser reads the reduced values where
    they are.  Per step: zero() -> ONE backward() -> reduce() -> optimizer.step().  (A gradient that arrives for a unit whose
    exchange is already on the wire -- a second backward() in the same step -- raises instead of being silently left out.)

    The exchange is issued in UNITS: one per parameter (split into `chunk_bytes` pieces, every piece an independent
    collective that RCCL can spread over the xGMI links), consecutive small parameters sharing one; the four per-anchor parameters written through the gradient
    sink (see sink()) form `anchor_ranges` units instead, one per range of anchors, each holding that range's slice of
    all four.  A unit becomes READY when its gradient is final on this rank -- a parameter's post-accumulate hook, or
    the sink's per-range callback from the last anchor-gather backward of the step -- and reduce() declares the rest
    ready.  Collectives are matched across ranks by ISSUE ORDER, so the order must not depend on what happened on a
    rank: units are issued strictly in one agreed order, as far as the contiguous ready prefix reaches.  The first
    exchange runs in index order from reduce(); the ranks then agree on the order for all later steps (the order in which
    hooks fired and anchor ranges were reported, as observed, MIN over ranks of the positions, so every rank derives the
    same permutation from the same reduced vector; parameters nobody saw a gradient for last).  A rank without local views -- no hook fires,
    everything goes out from reduce() -- therefore issues exactly the sequence the others do.
    mode "all_reduce": dist.all_reduce(SUM) per piece.  mode "rs_ag": reduce_scatter_tensor + all_gather_into_tensor
    on a piece padded to a multiple of the world size (on the point-to-point xGMI mesh every rank then owns 1/world
    of the sum and all 7 links carry traffic in both phases).
    Every rank must build the arena from the same parameters in the same order (the layout is that order)."""

    SMALL = 1 << 16      # elements: parameters up to this size are merged with their small neighbours into one exchange unit

    def __init__(self, params: Sequence[torch.Tensor], chunk_bytes: int = 256 << 20, mode: str = "all_reduce",
                 overlap: bool = True, anchor_ranges: int = 8, merge_small: bool = True, sparse_rows: bool = False,
                 sparse_threshold: float = 0.6, check_rows: bool = False):
        self.params = [p for p in params if p is not None and p.requires_grad]
        assert self.params, "no trainable parameters"
        assert mode in ("all_reduce", "rs_ag")
        dev = self.params[0].device
        dt = self.params[0].dtype
        assert all(p.dtype == dt and p.device == dev for p in self.params)
        self.itemsize = self.params[0].element_size()
        self.mode, self.overlap = mode, overlap
        self.small = self.SMALL if merge_small else 0
        # row-sparse exchange of the per-anchor units (set_row_union): opt-in, used when the union of the ranks' visible
        # anchors is below `sparse_threshold` of all anchors
        self.sparse_rows, self.sparse_threshold = bool(sparse_rows), float(sparse_threshold)
        self._rows, self._packed, self.last_union_fraction = None, [], None
        self.check_rows, self._union = bool(check_rows), None      # debug: assert the zero-row invariant of the packed exchange
        rank, world = world_info()
        self.world = world
        self.active = collectives_on()                   # collectives are issued (a one-rank group: only under force_collectives)
        self.align = align = 64 * max(world, 1)          # elements: every parameter starts on a 256-byte, world-divisible boundary
        self.offsets, total = [], 0
        for p in self.params:
            self.offsets.append(total)
            total += (p.numel() + align - 1) // align * align
        self.flat = torch.zeros(total, dtype=dt, device=dev)
        self.views = [self.flat[o:o + p.numel()].view_as(p) for o, p in zip(self.offsets, self.params)]
        self.chunk = max(align, chunk_bytes // self.itemsize // align * align)
        self.anchor_ranges = max(1, int(anchor_ranges))
        self._sink, self._sink_ids = None, ()
        self._work, self._handles, self._pending = [], [], []
        self._layout()
        self.bind()
        if overlap and self.active:
            for i, p in enumerate(self.params):
                self._handles.append(p.register_post_accumulate_grad_hook(self._make_hook(i)))

    # ---- units ---------------------------------------------------------------------------------------------------
    def _split(self, a, end):
        return [(x, min(x + self.chunk, end)) for x in range(a, end, self.chunk)]

    def _layout(self):
        """Units of the exchange for the current sink: ("p", i) per parameter outside the sink, ("s", r) per anchor
        range of the sink.  Every quantity here depends on shapes only, so all ranks build the same table."""
        align = self.align
        padded = lambda p: (p.numel() + align - 1) // align * align
        self.units, self.unit_pieces, self.sink_ranges = [], [], []
        # consecutive SMALL parameters (the MLP / BatchNorm / attention tensors: four dozen of a few hundred floats each) share
        # one unit: every collective costs a launch and a rendezvous whatever its size, and a unit per tensor put 48 of them
        # on the wire per step for 64 KB of payload.  A merged unit is ready when the last of its members is.
        i, n_par = 0, len(self.params)
        while i < n_par:
            if i in self._sink_ids:
                i += 1
                continue
            members, span = [i], padded(self.params[i])
            if span <= self.small:
                j = i + 1
                while j < n_par and j not in self._sink_ids and padded(self.params[j]) <= self.small and span + padded(self.params[j]) <= 8 * self.small:
                    span += padded(self.params[j])
                    members.append(j)
                    j += 1
            self.units.append(("p", i) if len(members) == 1 else ("g", tuple(members)))
            self.unit_pieces.append(self._split(self.offsets[i], self.offsets[i] + span))
            i = members[-1] + 1
        if self._sink_ids:
            N = self.params[self._sink_ids[0]].shape[0]
            step = -(-N // self.anchor_ranges)
            step = max(align, (step + align - 1) // align * align)   # anchors per range: slices stay world-divisible
            n0 = 0
            while n0 < N:
                self.sink_ranges.append((n0, min(N, n0 + step)))
                n0 += step
            for r, (n0, n1) in enumerate(self.sink_ranges):
                pieces = []
                for i in self._sink_ids:
                    p, o = self.params[i], self.offsets[i]
                    w = p.numel() // max(N, 1)
                    end = o + padded(p) if n1 == N else o + n1 * w     # the last range takes the parameter's padding
                    pieces += self._split(o + n0 * w, end)
                self.units.append(("s", r))
                self.unit_pieces.append(pieces)
        self.layout_version = getattr(self, "layout_version", 0) + 1      # owned_slices() changed (adam.ShardedFusedAdam follows it)
        self._unit_of_param = {}
        for k, u in enumerate(self.units):
            for i in ((u[1],) if u[0] == "p" else u[1] if u[0] == "g" else ()):
                self._unit_of_param[i] = k
        self._unit_of_range = {u[1]: k for k, u in enumerate(self.units) if u[0] == "s"}
        self._order = None                                   # agreed issue order (unit numbers); None until agreed
        self._begin_step()

    def _begin_step(self):
        self._fired = set()                                  # parameters whose hook has fired in this step
        self._left = [len(u[1]) if u[0] == "g" else 1 for u in self.units]      # members of a unit still to report
        self._ready = [False] * len(self.units)
        self._issued = [False] * len(self.units)
        self._cursor, self._fire_log = 0, []

    @property
    def pieces(self):
        """Per unit: [(start, stop)] of its collectives in the flat buffer."""
        return self.unit_pieces

    def owned_slices(self):
        """[(parameter index, start, stop)] of the flat buffer this rank OWNS in mode "rs_ag": its 1/world of every piece, i.e.
        where reduce(gather=False) leaves the summed gradient (the whole piece when there is one rank).  A function of the
        shapes, the world size and the rank only -- the layout adam.ShardedFusedAdam keeps its moments in."""
        import bisect
        rank = dist.get_rank() if self.active else 0
        out = []
        for pieces in self.unit_pieces:
            for a, b in pieces:
                n = (b - a) // self.world
                lo, hi = a + rank * n, a + (rank + 1) * n
                i = bisect.bisect_right(self.offsets, lo) - 1         # a piece of a merged unit spans several parameters:
                while lo < hi:                                        # one entry per parameter (each has its own learning rate)
                    end = min(hi, self.offsets[i + 1] if i + 1 < len(self.offsets) else self.flat.numel())
                    out.append((i, lo, end))
                    lo, i = end, i + 1
        return out

    def bind(self):
        for p, v in zip(self.params, self.views):
            p.grad = v

    def sink(self, pc):
        """Lets the fused anchor gather write the gradients of pc's four per-anchor parameters straight into their
        arena views (anchor_gather.GradSink): zero() then skips those 71 floats per anchor and autograd's accumulation
        pass over them disappears; the LAST gather backward of a step runs range by range and reports every finished
        range, so that range's exchange is on the wire while the next is computed.  Returns the sink (also kept as
        self._sink), or None when the renderer would not take the fused gather for this model or a parameter is
        missing from the arena.  Every rank must call this alike (it changes the unit table)."""
        from . import anchor_gather as _ag
        ps = [getattr(pc, n, None) for n in ("_anchor_feat", "_anchor", "_offset", "_scaling")]
        return self.attach_sink(ps if _ag.fused_gather_taken(pc) else None)

    def attach_sink(self, per_anchor_params):
        """sink() for an explicit list of per-anchor parameters [N, ...] (all with the same N), or None to detach."""
        from .anchor_gather import GradSink
        self._sink, self._sink_ids = None, ()
        ids = {id(p): i for i, p in enumerate(self.params)}
        ps = per_anchor_params
        if ps is not None and all(p is not None and id(p) in ids for p in ps):
            self._sink_ids = tuple(ids[id(p)] for p in ps)
            self._sink = GradSink(*[self.views[i] for i in self._sink_ids])
        self._layout()
        if self._sink is not None:
            self._sink.ranges = list(self.sink_ranges) if (self.overlap and self.active) else None
            self._sink.on_range = self._range_done
        return self._sink

    def zero(self):
        """Start of a step: clear the arena and make sure every .grad still aliases it.  Parameters written through
        the sink are overwritten by the first view's backward: they are not cleared here."""
        sink = self._sink
        if sink is None:
            self.flat.zero_()
        else:
            for i, v in enumerate(self.views):
                if i not in self._sink_ids:
                    v.zero_()
            sink.fresh, sink.pending = True, 0
        self._rows, self._packed = None, []               # set_row_union() decides anew every step
        self._begin_step()
        self.bind()

    def _settle_sink(self):
        # no view of this step wrote through the sink (a rank without views): its parameters still hold the previous
        # step's gradients
        sink = self._sink
        if sink is not None and sink.fresh:
            for i in self._sink_ids:
                self.views[i].zero_()
            sink.fresh = False

    # ---- row-sparse exchange of the per-anchor units --------------------------------------------------------------
    def set_row_union(self, visible_local):
        """SURVEY.md 8e ("optional sparsification"): an anchor that NO view of the step sees has an all-zero gradient row
        on every rank (the gather's backward writes zeros for invisible anchors), so only the rows of the UNION of the ranks'
        visible anchors need to travel.  Call once per step, after zero() and before backward(), on every rank, with the OR of
        this rank's views' visibility masks ([N] bool / uint8; all zeros for a rank without views).  One MAX all-reduce of the
        byte mask + one host read of the per-range row counts (every rank derives the same message sizes from the same
        union); if the union holds less than `sparse_threshold` of the anchors, the sink's range units are exchanged as
        PACKED rows (gather -> all_reduce -> scatter back at reduce()) instead of whole ranges.  At MatrixCity scale a view
        sees a small part of the scene: 5.7 GB of per-anchor gradient shrink with the union.  Needs the gradient sink.
        With the sharded optimizer (reduce(gather=False)) the two combine without an owner map: the packed rows are summed by
        an ALL-reduce, so every rank ends up with the complete gradient of every union row -- a superset of the owned slices
        adam.ShardedFusedAdam reads (rows outside the union are zero everywhere, which is their sum) -- while the dense
        units (planes, MLPs) stop after their reduce-scatter as before.  On the wire per step: 2 f S for the gradients
        (f = union fraction, S = per-anchor bytes) + S for the parameter all-gather, against S + S dense: ahead below f = 0.5."""
        self._rows, self._packed, self.last_union_fraction = None, [], None
        if not self.sparse_rows or self._sink is None or not self.active:
            return False
        N = self.params[self._sink_ids[0]].shape[0]
        u = torch.zeros(N, dtype=torch.uint8, device=self.flat.device)      # (a copy: the caller's mask stays what it was)
        if visible_local is not None:
            assert visible_local.shape == (N,)
            u.copy_(visible_local)
        dist.all_reduce(u, op=dist.ReduceOp.MAX)
        # ONE compaction and ONE host read: the sorted row list of the whole union, cut at the range borders
        allrows = torch.nonzero(u).squeeze(1)
        edges = torch.tensor([n0 for n0, _ in self.sink_ranges] + [N], device=u.device)
        cuts = torch.searchsorted(allrows, edges).tolist()
        rows = [allrows[cuts[r]:cuts[r + 1]] for r in range(len(self.sink_ranges))]
        total = int(allrows.numel())
        self.last_union_fraction = total / max(N, 1)
        self._union = u if self.check_rows else None
        if self.last_union_fraction < self.sparse_threshold:
            self._rows = rows
        return self._rows is not None

    def _issue_packed(self, r):
        """Range r of the sink as packed rows: [U_r, 32 | 3 | 30 | 6] gathered into one buffer, one all_reduce."""
        rows = self._rows[r]
        if self.check_rows and self._union is not None:
            # THE INVARIANT the packed exchange rests on: a row outside the union of the ranks' visible anchors is exactly
            # zero on every rank (every per-anchor gradient reaches the sink through the visible gather, whose backward
            # writes zeros for invisible anchors).  A gradient that arrived another way would be dropped from the sum
            # without a trace -- check_rows=True (debug) looks.
            n0, n1 = self.sink_ranges[r]
            out = self._union[n0:n1] == 0
            for i in self._sink_ids:
                v = self.views[i].reshape(self.views[i].shape[0], -1)[n0:n1]
                if bool((v[out] != 0).any()):
                    raise RuntimeError(f"GradArena: per-anchor parameter {i} has a non-zero gradient row outside the union of the "
                                       f"visible anchors (range {r}): the row-sparse exchange would drop it")
        parts = [self.views[i].reshape(self.views[i].shape[0], -1).index_select(0, rows) for i in self._sink_ids]
        packed = torch.cat([p.reshape(-1) for p in parts]) if rows.numel() else self.flat.new_zeros(0)
        if packed.numel():
            self._work.append(dist.all_reduce(packed, op=dist.ReduceOp.SUM, async_op=True))
        self._packed.append((rows, packed, [p.shape[1] for p in parts]))

    def _unpack(self):
        for rows, packed, widths in self._packed:
            off = 0
            for i, w in zip(self._sink_ids, widths):
                n = rows.numel() * w
                if n:
                    self.views[i].reshape(self.views[i].shape[0], -1).index_copy_(0, rows, packed[off:off + n].view(rows.numel(), w))
                off += n
        self._packed = []

    # ---- issue ---------------------------------------------------------------------------------------------------
    def _issue(self, k):
        self._issued[k] = True
        if self._rows is not None and self.units[k][0] == "s":
            self._issue_packed(self.units[k][1])
            return
        for a, b in self.unit_pieces[k]:
            piece = self.flat[a:b]
            if self.mode == "all_reduce":
                self._work.append(dist.all_reduce(piece, op=dist.ReduceOp.SUM, async_op=True))
            else:                                         # phase 1 now, phase 2 (the gather) from reduce()
                n = (b - a) // self.world
                mine = piece[dist.get_rank() * n:(dist.get_rank() + 1) * n]
                self._work.append(dist.reduce_scatter_tensor(mine, piece, op=dist.ReduceOp.SUM, async_op=True))
                self._pending.append((piece, mine))

    def _flush(self):
        """Issue the contiguous ready prefix of the agreed order (nothing before the order is agreed)."""
        if self._order is None or not self.active:
            return
        while self._cursor < len(self._order) and self._ready[self._order[self._cursor]]:
            self._issue(self._order[self._cursor])
            self._cursor += 1

    def _make_hook(self, i):
        def hook(param):
            if param.grad is not self.views[i] and param.grad.data_ptr() != self.views[i].data_ptr():
                self.views[i].copy_(param.grad)           # autograd replaced the tensor (never seen; kept correct)
                param.grad = self.views[i]
            k = self._unit_of_param.get(i)
            if k is not None and self._issued[k]:
                # the unit's exchange is already on the wire: this accumulation would never reach the other ranks
                raise RuntimeError("GradArena: a gradient arrived for a parameter whose exchange was already issued in this step -- "
                                   "the arena expects ONE backward() per step (zero() -> backward -> reduce()); sum the views' losses "
                                   "and call backward once, or build the arena with overlap=False")
            if k is not None and i not in self._fired:
                self._fired.add(i)
                self._left[k] -= 1
                if self._left[k] == 0 and not self._ready[k]:
                    self._ready[k] = True
                    self._fire_log.append(k)
                    self._flush()
        return hook

    def _range_done(self, r):
        """anchor_gather's last backward of the step finished anchor range r (its kernel is queued on the current
        stream; the collective waits for it there)."""
        k = self._unit_of_range.get(r)
        if k is not None and self._issued[k]:
            raise RuntimeError("GradArena: an anchor range was reported final twice in one step (a second backward() after its "
                               "exchange was issued): the arena expects ONE backward() per step")
        if k is not None and not self._ready[k]:
            self._ready[k] = True
            self._fire_log.append(k)
            self._flush()

    def _agree_order(self):
        """One small MIN all-reduce + host read, after the FIRST exchange of a unit table: position of every parameter
        unit in this rank's hook log (len(units) where no hook fired) -> the same permutation on every rank; then the
        sink's ranges in order; then the units no rank saw a gradient for (they only ever go out from reduce())."""
        n = len(self.units)
        pos = [n] * n
        for j, k in enumerate(self._fire_log):
            pos[k] = j
        flags = [1 if u[0] == "s" else 0 for u in self.units]
        t = torch.tensor(pos + flags + [-f for f in flags], dtype=torch.int64, device=self.flat.device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        t = t.tolist()
        pos, fmin, fneg = t[:n], t[n:2 * n], t[2 * n:]
        if any(a != -b for a, b in zip(fmin, fneg)):
            raise RuntimeError("GradArena: the ranks attached different gradient sinks (the unit tables differ)")
        # every unit in the order it became final, the sink's anchor ranges among the parameters (round 6: with the tri-plane
        # features sampled before the gather, FeaturePlanes.presample, the ranges are final BEFORE the plane parameters and
        # must not queue behind them); ranges nobody reported (no rank had views this step) keep their place behind the
        # parameters that fired; parameters nobody saw a gradient for last
        fired = sorted((k for k in range(n) if pos[k] < n), key=lambda k: (pos[k], k))
        quiet_ranges = [k for k in range(n) if flags[k] and pos[k] >= n]
        silent = [k for k in range(n) if not flags[k] and pos[k] >= n]
        self._order = fired + quiet_ranges + silent

    def reduce(self, gather=True):
        """SUM over ranks of everything in the arena; returns when the reduced gradients are usable on the current
        stream.  Units that were not issued during the backward pass (no gradient on this rank: zeros; or behind one
        that was not ready) are exchanged now, in the agreed order.
        gather=False (mode "rs_ag" only): stop after the reduce-scatter phase -- every rank then holds the summed gradient
        of its owned_slices() and nothing usable elsewhere; the optimizer updates those slices of the PARAMETERS and
        all-gathers the parameters instead of the gradients (adam.ShardedFusedAdam: Adam's work and moments / world)."""
        if not gather and self.mode != "rs_ag":
            raise ValueError("GradArena.reduce(gather=False) needs mode='rs_ag'")
        # (gather=False after set_row_union(): the packed units were ALL-reduced -- complete on every rank, which covers the
        # owned slices; only the dense units stop after the reduce-scatter)
        self._settle_sink()
        if self.active:
            order = self._order if self._order is not None else list(range(len(self.units)))
            for k in order[self._cursor:] if self._order is not None else order:
                assert not self._issued[k]
                self._issue(k)
            self._cursor = len(order)
            for w in self._work:
                w.wait()
            self._work = [dist.all_gather_into_tensor(piece, mine, async_op=True) for piece, mine in self._pending] if gather else []
            for w in self._work:
                w.wait()
            self._work, self._pending = [], []
            self._unpack()
            self._rows = None
            if self._order is None and self.overlap:
                self._agree_order()
        return self.flat

    def nbytes(self):
        return self.flat.numel() * self.itemsize

    def close(self):
        for h in self._handles:
            h.remove()
        self._handles = []


def align_images(*imgs):                                        # train.py:79-96
    h, w = min(i.shape[1] for i in imgs), min(i.shape[2] for i in imgs)
    return tuple(i[:, :h, :w] for i in imgs)


def pair_similarity(real1, real2):
    """SSIM of two ground-truth views as a 0-dim tensor (train.py:210) -- no host read here, so that a caller with many pairs
    can read all the verdicts at once."""
    from .losses import l1_ssim, ssim
    real1, real2 = align_images(real1, real2)
    if real1.is_cuda and real1.dim() == 3 and not (real1.requires_grad or real2.requires_grad):
        # SSIM of two ground-truth images: a constant of the step.  Through the fused kernel (csrc/ssim.hip, forward only) --
        # the framework's five grouped 11x11 convolutions run as MIOpen's naive kernels on this stack: 12 ms EACH at 640 x 360
        # (profiles/HISTORY.md, round 5), a second per view pair at 1080p
        with torch.no_grad():
            return l1_ssim(real1.contiguous(), real2.contiguous())[1]
    return ssim(real1, real2)


def pair_consistency(gen1, real1, gen2, real2, similarity=None, alike=None):
    """Cross-view consistency term of one view pair (train.py:208-217):
    ssim(real1, real2) * |mean |(real1 - real2) - (gen1 - gen2)||  if the two ground-truth views are alike
    (SSIM > 0.6), else 0 (None).  similarity / alike: pair_similarity(real1, real2) and its verdict when the caller has
    them already (consistency_loss reads the verdicts of all pairs with one host read)."""
    from .losses import pair_l1
    gen1, gen2, real1, real2 = align_images(gen1, gen2, real1, real2)
    s = pair_similarity(real1, real2) if similarity is None else similarity
    if not (bool(s > 0.6) if alike is None else alike):
        return None
    return s * torch.abs(pair_l1(gen1, gen2, real1, real2))


def consistency_loss(local: Sequence, weight: float = 0.05, device=None):
    """Sum over ALL view pairs of pair_consistency, times `weight` (train.py:201-239), in the sharded
    setting: `local` = [(global view index, rendered image, gt image), ...] of this rank.  The rendered
    and ground-truth images of the other ranks are all-gathered as constants; a pair with one remote
    view is evaluated on both owning ranks, each differentiating its own image only, so the gradients
    summed over ranks equal those of the single-process pairwise sum.  Returns (term to add to this
    rank's loss before backward, this rank's share of the loss VALUE -- cross-rank pairs count half)."""
    rank, world = world_info()
    items = [(int(i), g, r, True) for i, g, r in local]
    if device is None and local:
        device = local[0][1].device
    if collectives_on():
        if device is None:
            raise ValueError("consistency_loss: a rank without local views must pass device= (the collectives need "
                             "buffers on the communicator's device)")
        mine = [(int(i), g.detach(), r.detach()) for i, g, r in local]
        everyone = [None] * world
        # (shape and dtype travel with the index: a rank without local views has no image to copy them from)
        dist.all_gather_object(everyone, [(i, tuple(g.shape), g.dtype) for i, g, _ in mine])
        for src in range(world):
            for i, shape, dtype in everyone[src]:
                if src == rank:
                    g, r = next((g, r) for j, g, r in mine if j == i)
                    pair = torch.stack([g, r]).contiguous()
                else:
                    pair = torch.empty((2,) + shape, dtype=dtype, device=device)
                dist.broadcast(pair, src=src)
                if src != rank:
                    items.append((i, pair[0], pair[1], False))
    items.sort(key=lambda t: t[0])
    grad_term, value = None, 0.0
    pairs = [(a, b) for a in range(len(items)) for b in range(a + 1, len(items)) if items[a][3] or items[b][3]]
    sims = [pair_similarity(*align_images(items[a][1], items[b][1], items[a][2], items[b][2])[2:]) for a, b in pairs]
    alike = (torch.stack([s.detach().reshape(()) for s in sims]) > 0.6).tolist() if sims else []   # ONE host read for all pairs
    for (a, b), s, ok in zip(pairs, sims, alike):
        (_, g1, r1, own1), (_, g2, r2, own2) = items[a], items[b]
        t = pair_consistency(g1, r1, g2, r2, similarity=s, alike=ok)
        if t is None:
            continue
        grad_term = t if grad_term is None else grad_term + t
        value = value + t.detach() * (1.0 if (own1 and own2) else 0.5)
    if grad_term is None:
        return None, value
    return weight * grad_term, weight * value


def multiview_step(views: Sequence, params: Sequence[torch.Tensor],
                   render_loss: Callable[[object], torch.Tensor], bucket: torch.Tensor = None,
                   consistency_weight: float = 0.0, after_reduce: Callable[[], None] = None):
    """One collaborative step: this rank renders its shard of `views`, sums the per-view losses,
    runs ONE backward (as train.py:240 does) and all-reduces the gradients.  After the call every
    rank holds d(sum over ALL views of loss)/d(params) -- identical to the sequential mv loop.
    With consistency_weight > 0 (train.py: 0.05 for update_from < iteration < update_until)
    `render_loss(view)` must return (loss, rendered image, gt image) and the pairwise cross-view term is
    added (see consistency_loss).
    after_reduce: called once the summed gradients are in place -- where terms that depend on the parameters only belong
    (the tri-plane total-variation term of train.py:242-243: added before the exchange it would count once per rank).
    Returns (local loss sum, bucket)."""
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
        term, _ = consistency_loss(rendered, consistency_weight, device=params[0].device)
        if term is not None:
            total = term if total is None else total + term
    if total is not None:
        total.backward()
    bucket = allreduce_gradients(params, bucket)
    if after_reduce is not None:
        after_reduce()
    return (total.detach() if total is not None else torch.zeros((), dtype=params[0].dtype, device=params[0].device)), bucket
