"""The optimizer of the training step as one streaming HIP pass: a drop-in for the reference's
`torch.optim.Adam(l, lr=0.0, eps=1e-15)` (scene/gaussian_model.py:575; stepped at train.py:310-312).

Same parameter groups, same state layout (`state[p] = {"step", "exp_avg", "exp_avg_sq"}`, `step` a CPU scalar tensor as in
torch's default Adam), so the optimizer surgery of densification (`densify.AnchorDensifier.replace_tensor_to_optimizer`,
`cat_tensors_to_optimizer`, `_prune_optimizer`: scene/gaussian_model.py:738-818) and `state_dict()` / `load_state_dict()`
work on it unchanged and checkpoints interchange with torch.optim.Adam.  What differs is the step: every group is one
launch of `scr_adam_step` (csrc/adam.hip) over parameter, gradient and moments -- at 20 M anchors that is 40 GB of
traffic, priced against the copy probe.  weight_decay / amsgrad / maximize are not part of the reference's optimizer and
are refused."""
import ctypes as C
import math

import torch

from . import _C


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False, maximize=False):
        if weight_decay != 0.0 or amsgrad or maximize:
            raise NotImplementedError("FusedAdam mirrors the reference's Adam(l, lr=0.0, eps=1e-15): no weight_decay / amsgrad / maximize")
        if not (0.0 <= betas[0] < 1.0 and 0.0 <= betas[1] < 1.0) or eps < 0.0 or lr < 0.0:
            raise ValueError(f"bad hyper-parameters: lr {lr}, betas {betas}, eps {eps}")
        # the keys torch.optim.Adam keeps in its groups ride along unchanged, so that a state_dict() of this optimizer
        # loads into torch's and back
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=0.0, amsgrad=False, maximize=False, foreach=None,
                                      capturable=False, differentiable=False, fused=None, decoupled_weight_decay=False))

    def _init_state(self, p):
        st = self.state[p]
        if len(st) == 0:
            st["step"] = torch.tensor(0.0, dtype=torch.float32)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return st

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            if group.get("weight_decay", 0.0) != 0.0 or group.get("amsgrad") or group.get("maximize"):
                raise NotImplementedError("FusedAdam: a group asks for weight_decay / amsgrad / maximize")
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            b1, b2 = group["betas"]
            lr = float(group["lr"])
            table = (_C.AdamTensor * len(ps))()
            dev = ps[0].device
            for e, p in zip(table, ps):
                g = p.grad
                st = self._init_state(p)
                m, v = st["exp_avg"], st["exp_avg_sq"]
                if not (p.is_cuda and p.device == dev and g.device == dev and m.device == dev and v.device == dev):
                    raise ValueError("FusedAdam: parameters, gradients and moments of a group must live on one GPU (no CPU path)")
                if not (p.dtype == g.dtype == m.dtype == v.dtype == torch.float32):
                    raise TypeError("FusedAdam: fp32 parameters, gradients and moments only")
                if g.is_sparse or not (p.is_contiguous() and g.is_contiguous() and m.is_contiguous() and v.is_contiguous()):
                    raise ValueError("FusedAdam: dense contiguous tensors only")
                if not (g.shape == p.shape == m.shape == v.shape):
                    raise ValueError(f"FusedAdam: shapes differ: param {tuple(p.shape)}, grad {tuple(g.shape)}, moments {tuple(m.shape)}")
                t = float(st["step"]) + 1.0
                e.param, e.grad, e.exp_avg, e.exp_avg_sq = p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr()
                e.numel = p.numel()
                e.step_size = lr / (1.0 - b1 ** t)                     # doubles, as torch's Python forms them
                e.bias_correction2_sqrt = math.sqrt(1.0 - b2 ** t)
            with torch.cuda.device(dev):
                _C.check(_C.lib.scr_adam_step(len(ps), table, b1, b2, float(group["eps"]), torch.cuda.current_stream(dev).cuda_stream))
            for p in ps:                        # the update is queued for every tensor of the group: the step counts move together
                self.state[p]["step"] += 1
        return loss


def _adam_apply(entries, beta1, beta2, eps, device):
    """entries: [(param slice, grad slice, exp_avg slice, exp_avg_sq slice, step_size, bias_correction2_sqrt)], flat fp32
    device tensors of equal length -> ONE scr_adam_step call (the library launches in chunks of its table size)."""
    table = (_C.AdamTensor * len(entries))()
    for e, (p, g, m, v, step_size, bc2) in zip(table, entries):
        if not (p.is_cuda and g.is_cuda and m.is_cuda and v.is_cuda):
            raise ValueError("ShardedFusedAdam: parameters, gradients and moments must live on the GPU (no CPU path)")
        e.param, e.grad, e.exp_avg, e.exp_avg_sq = p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr()
        e.numel, e.step_size, e.bias_correction2_sqrt = p.numel(), step_size, bc2
    with torch.cuda.device(device):
        _C.check(_C.lib.scr_adam_step(len(entries), table, beta1, beta2, eps, torch.cuda.current_stream(device).cuda_stream))


class ShardedFusedAdam:
    """The reference's Adam (torch.optim.Adam(l, lr=0.0, eps=1e-15), scene/gaussian_model.py:575) with its WORK and its
    MOMENTS divided by the number of ranks, for the sharded multi-view step (BASELINE.json configs[3] / configs[4]).

    The replicated step sums the gradients over the ranks (reduce-scatter + all-gather on the xGMI mesh) and then runs the
    same Adam over all parameters on every rank: at 20 M anchors 40 GB of optimizer traffic and 11.4 GB of moments per GPU,
    eight times over.  Here the exchange stops after the reduce-scatter (multiview.GradArena.reduce(gather=False)): every
    rank holds the summed gradient of its owned slices -- 1/world of every exchange piece --, runs scr_adam_step on exactly
    those slices of the PARAMETERS (which live in one flat buffer with the arena's layout), and the all-gather that would
    have completed the gradients distributes the updated parameters instead.  Same bytes on the wire, Adam / world,
    moments / world.  Adam is elementwise, so the result is the replicated step's, bit for bit.

    Parameter groups as torch's: [{"params": [...], "lr": ...}, ...]; lr may be changed between steps through
    .param_groups (the reference's schedulers do).  The parameters must be exactly the arena's.  Not a torch.optim.Optimizer:
    the moments are flat per-rank shards; full_state() / load_full_state() convert to and from per-parameter tensors in
    torch.optim.Adam's layout (all-gather), which is how densification's optimizer surgery and checkpoints reach them."""

    def __init__(self, param_groups, arena, betas=(0.9, 0.999), eps=1e-15):
        import torch.distributed as dist
        if arena.mode != "rs_ag":
            raise ValueError("ShardedFusedAdam needs a GradArena in mode 'rs_ag' (the exchange it splits)")
        self.arena, self.betas, self.eps = arena, (float(betas[0]), float(betas[1])), float(eps)
        self.param_groups = [dict(g) for g in param_groups]
        ids = {id(p): i for i, p in enumerate(arena.params)}
        self._group_of = {}
        for gi, g in enumerate(self.param_groups):
            for p in g["params"]:
                if id(p) not in ids:
                    raise ValueError("ShardedFusedAdam: a parameter of the groups is not in the arena")
                self._group_of[ids[id(p)]] = gi
        if len(self._group_of) != len(arena.params):
            raise ValueError("ShardedFusedAdam: the groups must cover exactly the arena's parameters")
        self.world = arena.world
        self.active = arena.active                 # collectives are issued (multiview.collectives_on() when the arena was built)
        self.rank = dist.get_rank() if self.active else 0
        # parameters move into ONE flat buffer with the arena's layout (p.data becomes a view of it)
        self.pflat = torch.zeros_like(arena.flat)
        for p, o in zip(arena.params, arena.offsets):
            view = self.pflat[o:o + p.numel()].view_as(p)
            view.copy_(p.data)
            p.data = view
        self.steps = [0] * len(arena.params)
        self._shard()

    def _shard(self):
        """(Re)build this rank's slice table and moment shards for the arena's CURRENT unit table.  The table changes when a
        gradient sink is attached or detached (GradArena.attach_sink: per-anchor parameters are then exchanged in anchor
        ranges); moments that already hold history are carried over through the per-parameter layout (a collective: every rank
        gets here at the same point of the same program)."""
        carried = self.full_state() if getattr(self, "slices", None) is not None and any(self.steps) else None
        arena = self.arena
        self.slices = arena.owned_slices()
        self._layout_seen = arena.layout_version
        self._state_off, total = [], 0
        for _, a, b in self.slices:
            self._state_off.append(total)
            total += b - a
        self.exp_avg = torch.zeros(total, dtype=self.pflat.dtype, device=self.pflat.device)
        self.exp_avg_sq = torch.zeros_like(self.exp_avg)
        if carried is not None:
            self.load_full_state(carried)

    def nbytes_state(self):
        return 2 * self.exp_avg.numel() * self.exp_avg.element_size()

    @property
    def state(self):
        # torch.optim.Optimizer.state does not exist here: code that edits per-parameter moments in place (the optimizer
        # surgery of densification, scene/gaussian_model.py:738-818) has to go through the per-parameter layout
        raise AttributeError("ShardedFusedAdam keeps its moments as flat per-rank shards: use full_state() to obtain them in "
                             "torch.optim.Adam's per-parameter layout (e.g. to run AnchorDensifier.adjust_anchor on a torch / FusedAdam "
                             "optimizer built from it), then build a new GradArena + ShardedFusedAdam and load_full_state()")

    @torch.no_grad()
    def step(self):
        """After arena.reduce(gather=False): update this rank's slices, then all-gather the parameters."""
        import torch.distributed as dist
        if self._layout_seen != self.arena.layout_version:
            self._shard()
        for p, o in zip(self.arena.params, self.arena.offsets):
            if p.data_ptr() != self.pflat.data_ptr() + o * self.pflat.element_size():
                # adjust_anchor / sort_anchors REPLACE parameters: the new tensor would never be stepped, silently
                raise RuntimeError("ShardedFusedAdam.step: a parameter no longer lives in the optimizer's flat buffer (it was replaced, "
                                   "e.g. by densification): build a new GradArena + ShardedFusedAdam and load_full_state()")
        b1, b2 = self.betas
        flat, entries = self.arena.flat, []
        for (i, a, b), so in zip(self.slices, self._state_off):
            if b <= a:
                continue
            t = self.steps[i] + 1
            lr = float(self.param_groups[self._group_of[i]]["lr"])
            entries.append((self.pflat[a:b], flat[a:b], self.exp_avg[so:so + b - a], self.exp_avg_sq[so:so + b - a],
                            lr / (1.0 - b1 ** t), math.sqrt(1.0 - b2 ** t)))
        if entries:
            _adam_apply(entries, b1, b2, self.eps, self.pflat.device)
        for i in range(len(self.steps)):
            self.steps[i] += 1
        if self.active:
            work = []
            for pieces in self.arena.unit_pieces:
                for a, b in pieces:
                    n = (b - a) // self.world
                    work.append(dist.all_gather_into_tensor(self.pflat[a:b], self.pflat[a + self.rank * n:a + (self.rank + 1) * n],
                                                            async_op=True))
            for w in work:
                w.wait()

    def zero_grad(self, set_to_none=True):
        """The arena owns the gradients (GradArena.zero() starts a step)."""

    # ---- conversion to / from torch.optim.Adam's per-parameter state (densification surgery, checkpoints)
    @torch.no_grad()
    def full_state(self):
        """{parameter index: {"step", "exp_avg", "exp_avg_sq"}} with full-size moments, identical on every rank.  One moment
        at a time: the shards are laid into ONE full-size buffer and completed by an all-gather per exchange piece (every
        rank's slice of a piece is its 1/world of it, exactly the layout of the parameter all-gather in step()); the result
        is handed out as VIEWS of that buffer -- one full-size transient per moment instead of four (two buffers all-reduced,
        then a clone of every parameter's moments: 4 x 5.7 GB per rank at 20 M anchors, at the point of the run -- a
        re-layout, densification, a checkpoint -- where head-room is smallest).
        VIEW SEMANTICS: the moments of different parameters share one transient buffer per key and are NOT copies of the
        optimizer's state -- editing one in place changes nothing in the optimizer and nothing on other ranks; a caller that
        wants to mutate (densification's state surgery) clones what it edits and hands the result to load_full_state().
        Exercised over gloo at world 2 / 4 / 8 (CPU and HIP path) and under RCCL in a one-rank group only: no multi-GPU
        hardware has run this path."""
        import torch.distributed as dist
        out = {i: {"step": torch.tensor(float(self.steps[i]))} for i in range(len(self.arena.params))}
        for key, shard in (("exp_avg", self.exp_avg), ("exp_avg_sq", self.exp_avg_sq)):
            full = torch.zeros_like(self.pflat)
            for (i, a, b), so in zip(self.slices, self._state_off):
                full[a:b] = shard[so:so + b - a]
            if self.active and self._layout_seen != self.arena.layout_version:
                # called from _shard() after the arena changed its unit table: the shards still follow the OLD table, the
                # arena's pieces the new one -- the ranks' slices are disjoint in any case, so a SUM assembles them
                dist.all_reduce(full)
            elif self.active:
                work = []
                for pieces in self.arena.unit_pieces:
                    for a, b in pieces:
                        n = (b - a) // self.world
                        work.append(dist.all_gather_into_tensor(full[a:b], full[a + self.rank * n:a + (self.rank + 1) * n], async_op=True))
                for w in work:
                    w.wait()
            for i, (p, o) in enumerate(zip(self.arena.params, self.arena.offsets)):
                out[i][key] = full[o:o + p.numel()].view_as(p)
        return out

    @torch.no_grad()
    def load_full_state(self, state):
        """Inverse of full_state(): keep this rank's slices of full-size moments."""
        for i, st in state.items():
            self.steps[i] = int(float(st["step"]))
        for (i, a, b), so in zip(self.slices, self._state_off):
            if i in state and b > a:
                o = self.arena.offsets[i]
                n = self.arena.params[i].numel()
                lo, hi = max(a, o), min(b, o + n)          # the slice may reach into the parameter's padding
                if hi > lo:
                    self.exp_avg[so + lo - a:so + hi - a] = state[i]["exp_avg"].reshape(-1)[lo - o:hi - o]
                    self.exp_avg_sq[so + lo - a:so + hi - a] = state[i]["exp_avg_sq"].reshape(-1)[lo - o:hi - o]
