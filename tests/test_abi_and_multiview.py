"""CPU tests: the C-ABI library exports what include/splatco_raster.h declares (no compute calls
without a GPU), and the multi-view sharding + gradient all-reduce equals the reference's
sequential mv loop (train.py:171-240) -- world_size 2, gloo backend."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from util import run_ranks

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_abi_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "splatco_raster.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(scr_[a-z_0-9]+)\s*\(", hdr))
    assert {"scr_visible_filter", "scr_forward_plan", "scr_forward_run", "scr_backward", "scr_mark_visible"} <= declared
    lib_path = os.path.join(ROOT, "splatco_amd", "csrc", "libsplatco_raster.so")
    assert os.path.exists(lib_path), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(lib_path)
    for sym in sorted(declared):
        assert hasattr(lib, sym), f"{sym} declared in the header but not exported"
    lib.scr_abi_version.restype = ctypes.c_int
    m = re.search(r"#define SCR_ABI_VERSION (\d+)", hdr)
    assert lib.scr_abi_version() == int(m.group(1))
    # pure size queries (no device needed): monotone, 256-byte aligned
    lib.scr_geom_bytes.restype = lib.scr_binning_bytes.restype = ctypes.c_size_t
    lib.scr_geom_bytes.argtypes = [ctypes.c_int64, ctypes.c_int32, ctypes.c_int32]
    lib.scr_binning_bytes.argtypes = [ctypes.c_int64, ctypes.c_int64]
    a, b = lib.scr_geom_bytes(1000, 1080, 1920), lib.scr_geom_bytes(2000, 1080, 1920)
    assert 0 < a < b and a % 256 == 0
    assert lib.scr_binning_bytes(10, 5) % 256 == 0
    assert lib.scr_binning_bytes(100000, 20000) > lib.scr_binning_bytes(100000, 500)   # merge buffers (tiles above 8192 instances)
    # the python binding loads the same symbols and refuses to run without the library
    from splatco_amd import _C
    assert set(_C.SYMBOLS) == declared


def test_stage_markers_are_off_by_default_and_load_roctx_on_demand():
    """scr_markers_enable (ABI 27): nothing is loaded before the first enable; enabled, the C-ABI ranges and the host-side
    stage() ranges push and pop in pairs (roctx without a profiler attached is a no-op library); disabled again, stage() is
    inert.  No device call is made."""
    from splatco_amd import _C
    assert _C.MARKERS is (os.environ.get("SPLATCO_MARKERS", "") not in ("", "0"))
    _C.markers_enable(True)
    try:
        assert _C.MARKERS
        assert _C.lib.scr_marker_push(b"unit-test") == 0 and _C.lib.scr_marker_pop() == 0
        with _C.stage("outer") as s:
            assert s.on
            with _C.stage("inner"):
                pass
    finally:
        _C.markers_enable(False)
    assert not _C.MARKERS
    with _C.stage("off") as s:
        assert not s.on


def test_large_scratch_comes_in_size_classes():
    """_C.scratch_size: at least what was asked, at most 12.5 % more, eight classes per octave above 32 MiB, untouched below
    -- and the plane backward's scratch query covers the partial sums of split tiles (a size query, no device)."""
    from splatco_amd import _C
    assert [_C.scratch_size(n) for n in (0, 1, 4096, (32 << 20) - 1)] == [1, 1, 4096, (32 << 20) - 1]
    prev = 0
    for n in sorted([(32 << 20) + k * 1_234_567 for k in range(0, 4000, 7)] + [2 ** 31 - 1, 2 ** 31, 2 ** 31 + 1, 3 * 2 ** 33 + 5]):
        c = _C.scratch_size(n)
        assert n <= c <= n + n // 8 + 1 and c >= prev, (n, c)
        assert c % (1 << (n.bit_length() - 4)) == 0
        prev = c
    assert len({_C.scratch_size(n) for n in range(1 << 30, 1 << 31, 1 << 20)}) == 9        # eight classes and the octave's end
    lib = _C.lib
    V = 4_600_000
    one = lib.scr_plane_sample_scratch_bytes(V, 700, 700, 5)
    # records (V x 32 B) + header + halos + (V / 8192 + V / 16384 + 2) slots of 4 x 5 x 1024 64-bit sums
    slots = V // 8192 + V // 16384 + 2
    assert one >= V * 32 + slots * 4 * 5 * 1024 * 8
    assert lib.scr_plane_sample_scratch_bytes(V, 700, 700, 10) > one


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under splatco_amd/ or the drop-in module may
    reference it."""
    for base in ("splatco_amd", "diff_gaussian_rasterization"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".hip", ".h", ".cpp")):
                    txt = open(os.path.join(dp, f)).read()
                    assert "raster_oracle" not in txt and "from oracle" not in txt and "import oracle" not in txt, f


WORKER = r'''
import os, sys, math, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
from util import small_scene
from oracle import torch_ref
from splatco_amd.cameras import look_at_camera
from splatco_amd.multiview import multiview_step, shard_views

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
S = world * (world + 1) // 2                      # sum over ranks of (rank + 1)
NV = int(sys.argv[2])                             # number of views of the mv loop (ranks beyond it render nothing)
cam0, g = small_scene(P=40, W=32, H=24, seed=4)
views = [look_at_camera(eye=(0.6 + 0.3 * i, -0.4, -4.0), target=(0.1, 0.05, 0.0), up=(0.05, -1.0, 0.1),
                        FoVx=math.radians(55.0), width=32, height=24, uid=i) for i in range(NV)]
t = lambda a: torch.tensor(a, dtype=torch.float64, requires_grad=True)
params = [t(g["means3D"]), t(g["opacities"]), t(g["scales"]), t(g["rotations"]), t(g["colors"])]
target = torch.rand(3, 24, 32, generator=torch.Generator().manual_seed(0), dtype=torch.float64)

def render_loss(cam):
    m, o, s, r, c = params
    img, _, _ = torch_ref.rasterize(24, 32, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), torch.tensor(g["bg"]),
                                    1.0, cam.world_view_transform, cam.full_proj_transform, 1, cam.camera_center,
                                    m, o, s, r, None, None, c)
    return (img - target).abs().mean() + 0.01 * s.prod(dim=1).mean()   # per-view loss, summed over views

assert [v.uid for v in shard_views(views)] == [v.uid for v in views[rank::world]]
loss, _ = multiview_step(views, params, render_loss)
sharded = [p.grad.clone() for p in params]
# the reference's sequential loop: sum of all view losses, one backward (train.py:198,240)
for p in params: p.grad = None
total = sum(render_loss(v) for v in views)
total.backward()
for a, p in zip(sharded, params):
    assert torch.allclose(a, p.grad, rtol=1e-10, atol=1e-12), (rank, (a - p.grad).abs().max())
tl = loss.clone(); dist.all_reduce(tl)
assert torch.allclose(tl, total.detach(), rtol=1e-12)
# ---- with the cross-view consistency term (train.py:201-239): views alike enough for SSIM > 0.6
from splatco_amd.multiview import pair_consistency
gts = [(target + 0.02 * i).clamp(0, 1) for i in range(NV)]
def render_loss3(cam):
    m, o, s, r, c = params
    img, _, _ = torch_ref.rasterize(24, 32, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), torch.tensor(g["bg"]),
                                    1.0, cam.world_view_transform, cam.full_proj_transform, 1, cam.camera_center,
                                    m, o, s, r, None, None, c)
    return (img - gts[cam.uid]).abs().mean(), img, gts[cam.uid]
multiview_step(views, params, render_loss3, consistency_weight=0.05)
sharded = [p.grad.clone() for p in params]
for p in params: p.grad = None
outs = [render_loss3(v) for v in views]
total = sum(o[0] for o in outs)
npairs = 0
for i in range(NV):
    for j in range(i + 1, NV):
        t_ = pair_consistency(outs[i][1], outs[i][2], outs[j][1], outs[j][2])
        if t_ is not None:
            total = total + 0.05 * t_; npairs += 1
assert npairs == NV * (NV - 1) // 2
total.backward()
for a, p in zip(sharded, params):
    assert torch.allclose(a, p.grad, rtol=1e-9, atol=1e-12), (rank, (a - p.grad).abs().max())
# ---- the tri-plane total-variation term (train.py:242-243) is a function of the parameters only: the sharded step adds
# it ONCE, after the SUM, on every rank (multiview_step's after_reduce; train_step.collaborative_step does the same)
from torch_restatements import tv_add_grad_torch
gen = torch.Generator().manual_seed(3)
plane = torch.randn(1, 5, 12, 9, generator=gen, dtype=torch.float64).mul_(0.8).requires_grad_(True)
plane_w = torch.randn(NV, 1, 5, 12, 9, generator=gen, dtype=torch.float64)
def render_loss_tv(cam):
    return render_loss(cam) + (plane * plane_w[cam.uid]).sum()          # a view-dependent gradient for the plane
multiview_step(views, params + [plane], render_loss_tv, after_reduce=lambda: tv_add_grad_torch([(plane, 0.3)]))
sharded = [p.grad.clone() for p in params + [plane]]
for p in params + [plane]: p.grad = None
sum(render_loss_tv(v) for v in views).backward()
no_tv = plane.grad.clone()
tv_add_grad_torch([(plane, 0.3)])                                           # the reference: after backward(), once
assert float((plane.grad - no_tv).norm() / plane.grad.norm()) > 1e-3        # the term is visible at the tolerance below
for a, p in zip(sharded, params + [plane]):
    assert torch.allclose(a, p.grad, rtol=1e-10, atol=1e-12), (rank, (a - p.grad).abs().max())
# the trap: the term added on every rank BEFORE the exchange is counted `world` times
wrong = no_tv + world * (plane.grad - no_tv)
assert not torch.allclose(wrong, plane.grad, rtol=1e-6, atol=0)
for p in params + [plane]: p.grad = None
# a rank with no gradient for a parameter still takes part
extra = torch.zeros(5, dtype=torch.float64, requires_grad=True)
if rank == 0: extra.grad = torch.ones(5, dtype=torch.float64)
from splatco_amd.multiview import allreduce_gradients
allreduce_gradients([extra])
assert torch.equal(extra.grad, torch.ones(5, dtype=torch.float64))
# gradients that already tile one buffer (the rasterizer's backward arena) are reduced in place, no packing
arena = torch.arange(12, dtype=torch.float64) * (rank + 1)
a_, b_, c_ = (torch.zeros(n, dtype=torch.float64, requires_grad=True) for n in (5, 3, 4))
b_.grad, a_.grad, c_.grad = arena[0:3], arena[3:8], arena[8:12]          # adjacent, not in parameter order
ptr = a_.grad.data_ptr()
out = allreduce_gradients([a_, b_, c_])
assert out.data_ptr() == arena.data_ptr() and out.numel() == 12 and a_.grad.data_ptr() == ptr
assert torch.equal(arena, torch.arange(12, dtype=torch.float64) * S)
# the same exchange as reduce_scatter + all_gather on the same memory (bench.py picks the faster shape on RCCL)
arena = torch.arange(12, dtype=torch.float64) * (rank + 1)
b_.grad, a_.grad, c_.grad = arena[0:3], arena[3:8], arena[8:12]
out = allreduce_gradients([a_, b_, c_], shape="rs_ag")
assert out.data_ptr() == arena.data_ptr() and torch.equal(arena, torch.arange(12, dtype=torch.float64) * S)
odd = torch.arange(7, dtype=torch.float64, requires_grad=True)            # a length the world size does not divide: one all_reduce
odd.grad = torch.ones(7, dtype=torch.float64) * (rank + 1)
allreduce_gradients([odd], shape="rs_ag")
assert torch.equal(odd.grad, torch.full((7,), float(S), dtype=torch.float64))
c_.grad = torch.ones(4, dtype=torch.float64)                               # a stranger breaks the tiling: packed path
allreduce_gradients([a_, b_, c_])
assert torch.equal(c_.grad, torch.full((4,), float(world), dtype=torch.float64)) and torch.equal(a_.grad, torch.arange(3, 8, dtype=torch.float64) * S * world)
# ranks that disagree -- rank 0 holds an arena, rank 1 separate tensors; then two arenas with different layouts --
# must all take the packed path in parameter order instead of summing misaligned buffers
vals = [torch.arange(n, dtype=torch.float64) + 10 * j for j, n in enumerate((5, 3, 4))]
for case in ("arena vs separate", "two layouts"):
    ar = torch.zeros(12, dtype=torch.float64)
    if case == "arena vs separate" and rank == 1:
        a_.grad, b_.grad, c_.grad = (v.clone() for v in vals)
    elif case == "two layouts" and rank == 1:
        a_.grad, b_.grad, c_.grad = ar[0:5], ar[5:8], ar[8:12]             # a, b, c
    else:
        b_.grad, a_.grad, c_.grad = ar[0:3], ar[3:8], ar[8:12]             # b, a, c
    for p_, v in zip((a_, b_, c_), vals):
        p_.grad.copy_(v)
    allreduce_gradients([a_, b_, c_])
    for p_, v in zip((a_, b_, c_), vals):
        assert torch.equal(p_.grad, world * v), (case, rank)
# ---- GradArena: gradients live in one persistent buffer, exchanged piecewise (hooks + both collective shapes)
from splatco_amd.multiview import GradArena
for mode, overlap in (("all_reduce", True), ("rs_ag", True), ("all_reduce", False)):
    arena = GradArena(params, chunk_bytes=256, mode=mode, overlap=overlap)      # tiny pieces: several per parameter
    for it in range(2):                                                         # the buffer is reused step after step
        arena.zero()
        assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(params, arena.views))
        local = None
        for v in shard_views(views):
            l = render_loss(v)
            local = l if local is None else local + l
        if local is not None:                                                   # a rank beyond the number of views renders nothing
            local.backward()
        arena.reduce()
        want = torch.autograd.grad(sum(render_loss(v) for v in views), params)
        for p, w in zip(params, want):
            assert torch.allclose(p.grad, w, rtol=1e-10, atol=1e-12), (mode, overlap, rank)
    arena.close()
for p in params: p.grad = None
# ---- a rank WITHOUT views (world > number of views): its hooks never fire, everything goes out from reduce(); the other
# rank's hooks fire in backward order [2, 1, 0].  Collectives are matched by issue order, so the arena must issue in one
# agreed order on both (three equal-size parameters: a mismatch would sum different parameters without any error)
eq = [torch.full((64,), float(j + 1), dtype=torch.float64, requires_grad=True) for j in range(3)]
for mode in ("all_reduce", "rs_ag"):
    arena = GradArena(eq, chunk_bytes=256, mode=mode, overlap=True, merge_small=False)      # one unit per parameter: the order matters
    for it in range(3):                                     # step 0 agrees on the order, steps 1.. issue from the hooks
        arena.zero()
        if rank == 0:
            h = eq[0] * 1.0
            h = h + eq[1] * 10.0                             # graph order makes the hook order 2, 1, 0
            h = h + eq[2] * 100.0
            h.sum().backward()
            if it:
                assert arena._cursor == 3, arena._cursor     # all three went out during the backward pass
        arena.reduce()
        for j, w in enumerate((1.0, 10.0, 100.0)):
            assert torch.equal(eq[j].grad, torch.full((64,), w, dtype=torch.float64)), (mode, it, rank, j, eq[j].grad[:2])
    assert arena._order is not None
    arena.close()
# ---- ONE backward per step: a second one whose gradients arrive after their unit's exchange went out must raise (it
# used to be left out of the exchange silently)
arena = GradArena(eq, chunk_bytes=256, mode="all_reduce", overlap=True, merge_small=False)
for it in range(3):
    arena.zero()
    sum(e.sum() for e in eq).backward()
    if it == 2:                                              # the order is agreed: the hooks have issued every unit
        assert arena._cursor == 3
        try:
            sum(e.sum() for e in eq).backward()
            raised = False
        except RuntimeError as e:
            raised = "ONE backward" in str(e)
        assert raised
    arena.reduce()
arena.close()
for p in eq: p.grad = None
# ---- optimiser-state sharding: reduce-scatter -> Adam on this rank's 1/world of every piece of a PARAMETER arena ->
# all-gather of the parameters.  Same parameters as the replicated step (all-reduce, Adam over everything on every rank)
# after three iterations: bit for bit with two ranks (a + b in either order); beyond two the collective's own summation
# order may differ between the two exchanges
import math as _math
import splatco_amd.adam as adam_mod
from torch_restatements import adam_apply_torch
adam_mod._adam_apply = adam_apply_torch                       # CPU stand-in of csrc/adam.hip (the product has no CPU path)
def fresh():
    gen = torch.Generator().manual_seed(123)
    shapes = ((300, 32), (300, 3), (300, 30), (300, 6), (64, 32), (5,), (33,))     # four per-anchor tensors; lengths the world size does not divide
    return [torch.randn(*sh, generator=gen, dtype=torch.float32).requires_grad_(True) for sh in shapes]
def loss_of(ps, view):
    return sum(((p * (1.0 + 0.1 * view.uid)) ** 2).sum() * (j + 1) * 1e-2 + (p * (view.uid + 1)).sum() * 1e-3 for j, p in enumerate(ps))
pa, pb = fresh(), fresh()
lrs = (1e-2, 3e-3, 1e-1, 2e-3, 5e-2, 7e-3, 2e-2)
sink_a = None
arena_a = GradArena(pa, chunk_bytes=1024, mode="rs_ag", overlap=True)
opt_a = adam_mod.ShardedFusedAdam([{"params": [p], "lr": lr} for p, lr in zip(pa, lrs)], arena_a, eps=1e-15)
assert all(p.data_ptr() == opt_a.pflat[o:].data_ptr() for p, o in zip(pa, arena_a.offsets))      # the parameters moved into the flat buffer
assert opt_a.exp_avg.numel() * world == arena_a.flat.numel()                                     # moments / world
assert any(u[0] == "g" for u in arena_a.units)        # the small tensors share one exchange unit: its slices span parameters with different learning rates
arena_b = GradArena(pb, chunk_bytes=1024, mode="all_reduce", overlap=True)
m_b, v_b = [torch.zeros_like(p) for p in pb], [torch.zeros_like(p) for p in pb]
for it in range(1, 4):
    # (one arena per backward pass: collectives are matched by issue order, and a rank without views issues everything
    # from reduce() -- two arenas fed by ONE backward would interleave differently on the ranks that do render)
    for arena_x, ps in ((arena_a, pa), (arena_b, pb)):
        arena_x.zero()
        if arena_x is arena_a and sink_a is not None:        # what the gather's backward kernel does for the sink's tensors:
            for t_ in sink_a.tensors:                        # the first write of a step overwrites (here: clear, autograd adds)
                t_.zero_()
            sink_a.fresh = False
        lx = None
        for v in shard_views(views):
            l = loss_of(ps, v)
            lx = l if lx is None else lx + l
        if lx is not None:
            lx.backward()
        if arena_x is arena_a:
            arena_a.reduce(gather=False)
            opt_a.step()
        else:
            arena_b.reduce()
    adam_apply_torch([(p.data, p.grad, m, v_, lr / (1.0 - 0.9 ** it), _math.sqrt(1.0 - 0.999 ** it))
                      for p, m, v_, lr in zip(pb, m_b, v_b, lrs)], 0.9, 0.999, 1e-15)
    if it == 1:
        # the arena's unit table changes under the optimizer (a gradient sink makes the [N, .] parameters travel in anchor
        # ranges): the moment shards follow, history included
        sink_a = arena_a.attach_sink(pa[:4])
        assert sink_a is not None and opt_a._layout_seen != arena_a.layout_version
    if it == 2:
        opt_a.param_groups[0]["lr"] = lrs[0] * 0.5            # a scheduler moves a group's learning rate
        lrs = (lrs[0] * 0.5,) + lrs[1:]
full = opt_a.full_state()
for i, (m, v_) in enumerate(zip(m_b, v_b)):
    ok = (torch.equal if world == 2 else (lambda x, y: torch.allclose(x, y, rtol=1e-5, atol=1e-8)))
    assert ok(pa[i].data, pb[i].data), ("parameters", i, world, (pa[i].data - pb[i].data).abs().max())
    assert ok(full[i]["exp_avg"], m) and ok(full[i]["exp_avg_sq"], v_) and float(full[i]["step"]) == 3.0, ("moments", i)
# round trip of the sharded moments through torch.optim.Adam's per-parameter layout
opt_a2 = adam_mod.ShardedFusedAdam([{"params": [p], "lr": lr} for p, lr in zip(pa, lrs)], arena_a, eps=1e-15)
opt_a2.load_full_state(full)
assert torch.equal(opt_a2.exp_avg, opt_a.exp_avg) and torch.equal(opt_a2.exp_avg_sq, opt_a.exp_avg_sq) and opt_a2.steps == opt_a.steps
arena_a.close(); arena_b.close()
# ---- the per-anchor exchange in anchor RANGES (the sink's units) equals the unchunked exchange bit for bit, whether the
# ranges are reported during the step (rank 0) or only declared final by reduce() (rank 1), and in both collective shapes
Na = 1000
pa = [torch.zeros(Na, w, dtype=torch.float32, requires_grad=True) for w in (32, 3, 30, 6)]
other = torch.zeros(77, dtype=torch.float32, requires_grad=True)
gen = torch.Generator().manual_seed(7 + rank)
vals = [torch.randn(Na, w, generator=gen) for w in (32, 3, 30, 6)]
oval = torch.randn(77, generator=gen)
results = {}
for mode in ("all_reduce", "rs_ag"):
    for nr in (1, 4):
        arena = GradArena([other] + pa, chunk_bytes=4096, mode=mode, overlap=True, anchor_ranges=nr)
        sink = arena.attach_sink(pa)
        # (range boundaries are multiples of 64 x world anchors, so that every slice divides by the world size)
        assert sink is not None and (len(arena.sink_ranges) == 1 if nr == 1 else 1 < len(arena.sink_ranges) <= 4) and sink.ranges == arena.sink_ranges
        assert all(n0 % 64 == 0 for n0, _ in sink.ranges) and sink.ranges[-1][1] == Na
        for it in range(2):
            arena.zero()
            (other * oval).sum().backward()                  # an ordinary parameter: its hook fires
            for r, (n0, n1) in enumerate(sink.ranges):       # what anchor_gather's last backward does, range by range
                for t, v in zip(sink.tensors, vals):
                    t[n0:n1] = v[n0:n1]
                sink.fresh = False
                if rank == 0:
                    sink.on_range(r)
            if it and rank == 0:
                assert arena._cursor == len(arena.units)     # everything was on the wire before reduce()
            arena.reduce()
        results[(mode, nr)] = arena.flat.clone()
        arena.close()
both = [torch.zeros_like(v) for v in vals]
for b, v in zip(both, vals):
    g2 = [torch.zeros_like(v) for _ in range(world)]
    dist.all_gather(g2, v)
    b.copy_(sum(g2[1:], g2[0]))
# two ranks: a + b whatever the chunking -- bit for bit; more: the order of a three-term fp32 sum may depend on where a
# chunk boundary falls in the collective's own schedule
same = torch.equal if world == 2 else (lambda x, y: torch.allclose(x, y, rtol=1e-6, atol=1e-6))
for key, flat in results.items():
    assert same(flat, results[("all_reduce", 1)]), key
# ---- row-sparse exchange (SURVEY.md 8e): only the rows of anchors SOME rank sees travel -- same result as the dense exchange
mask = torch.rand(Na, generator=torch.Generator().manual_seed(50 + rank)) < 0.5 / world      # this rank's visible anchors (union: ~40 %)
mvals = [v * mask[:, None] for v in vals]                                                # zero rows where invisible (the gather's backward)
sparse_out = {}
for sparse in (False, True):
    for mode in ("all_reduce", "rs_ag"):
        arena = GradArena([other] + pa, chunk_bytes=4096, mode=mode, overlap=True, anchor_ranges=4, sparse_rows=sparse, sparse_threshold=0.9,
                          check_rows=True)
        sink = arena.attach_sink(pa)
        for it in range(2):
            arena.zero()
            took = arena.set_row_union(mask if rank != world - 1 or it == 0 else None)      # (second step: the last rank has no views)
            assert took == sparse and (not sparse or 0.1 < arena.last_union_fraction < 0.9)
            (other * oval).sum().backward()
            last_has_views = not (rank == world - 1 and it == 1)
            for r, (n0, n1) in enumerate(sink.ranges):
                for t, v in zip(sink.tensors, mvals):
                    t[n0:n1] = v[n0:n1] if last_has_views else 0.0
                sink.fresh = False
                if rank == 0:
                    sink.on_range(r)
            arena.reduce()
            sparse_out[(sparse, mode, it)] = arena.flat.clone()
        arena.close()
for mode in ("all_reduce", "rs_ag"):
    for it in range(2):
        assert same(sparse_out[(True, mode, it)], sparse_out[(False, mode, it)]), ("row-sparse exchange", mode, it)
# ---- ... combined with the sharded optimizer (round 6): reduce(gather=False) after set_row_union() -- the packed rows are
# ALL-reduced (complete on every rank), the dense unit stops after its reduce-scatter; what the sharded Adam reads, the
# owned slices, equals the dense exchange's there
arena = GradArena([other] + pa, chunk_bytes=4096, mode="rs_ag", overlap=True, anchor_ranges=4, sparse_rows=True, sparse_threshold=0.9,
                  check_rows=True)
sink = arena.attach_sink(pa)
for it in range(2):
    arena.zero()
    assert arena.set_row_union(mask if rank != world - 1 or it == 0 else None)
    (other * oval).sum().backward()
    last_has_views = not (rank == world - 1 and it == 1)
    for r, (n0, n1) in enumerate(sink.ranges):
        for t, v in zip(sink.tensors, mvals):
            t[n0:n1] = v[n0:n1] if last_has_views else 0.0
        sink.fresh = False
        if rank == 0:
            sink.on_range(r)
    arena.reduce(gather=False)
    want = sparse_out[(False, "rs_ag", it)]
    for i, lo, hi in arena.owned_slices():
        assert same(arena.flat[lo:hi], want[lo:hi]), ("row-sparse x sharded: owned slice", i, lo, hi, it)
    for i in arena._sink_ids:          # the per-anchor gradients are complete everywhere (all-reduced packed rows)
        o = arena.offsets[i]
        assert same(arena.flat[o:o + arena.params[i].numel()], want[o:o + arena.params[i].numel()]), ("row-sparse x sharded: sink", i, it)
arena.close()
# check_rows (debug): a gradient row OUTSIDE the union of the visible anchors -- which the packed exchange would drop from the sum
# without a trace -- is refused.  Every rank plants one (the check is local; all ranks raise at the same range, none is left
# inside a collective)
arena = GradArena([other] + pa, chunk_bytes=4096, mode="all_reduce", overlap=False, anchor_ranges=4, sparse_rows=True, sparse_threshold=0.9,
                  check_rows=True)
sink = arena.attach_sink(pa)
arena.zero()
union = torch.zeros(Na, dtype=torch.uint8)
union[:300] = 1
assert arena.set_row_union(union)
for t, v in zip(sink.tensors, vals):
    t.zero_()
    t[:300] = v[:300]
sink.tensors[2][900, 5] = 1.0
sink.fresh = False
try:
    arena.reduce()
    raise SystemExit("a non-zero row outside the union went through the packed exchange")
except RuntimeError as e:
    assert "outside the union" in str(e)
arena.close()
dist.barrier()
a1 = GradArena([other] + pa, anchor_ranges=1)
for i, b in zip(range(1, 5), both):
    assert same(results[("all_reduce", 4)][a1.offsets[i]:a1.offsets[i] + b.numel()].view_as(b), b)
a1.close()
# ---- densification statistics of the LAST view reach every rank; identically seeded growth -> identical anchors
import splatco_amd.stats as stats
from torch_restatements import statis_increments_torch, statis_apply_torch
stats.statis_increments, stats.statis_apply = statis_increments_torch, statis_apply_torch   # CPU stand-ins of the HIP kernels
from splatco_amd.densify import AnchorDensifier
from splatco_amd.scene_model import AnchorGaussianModel
from splatco_amd.train_step import sync_densification_stats
torch.manual_seed(11)                                    # the same replica on both ranks
N, k = 300, 10
pc = AnchorGaussianModel(plane_size=16, num_channels=15)
pc.set_anchors(torch.rand(N, 3) * 2 - 1, torch.randn(N, k, 3) * 0.1, torch.randn(N, 32), torch.randn(N, 6) * 0.1 - 3)
groups = [{"params": [getattr(pc, "_" + n)], "lr": 1e-3, "name": n} for n in ("anchor", "offset", "anchor_feat", "scaling", "opacity", "rotation")]
opt = torch.optim.Adam(groups)
den = AnchorDensifier(pc, opt, voxel_size=0.05, seed=1234)
for it in range(1, 5):
    # every rank "renders" ITS OWN view (different random tensors per rank); only the last view's must count
    gen = torch.Generator().manual_seed(100 * it + rank)
    vis = torch.rand(N, generator=gen) < 0.7
    V = int(vis.sum())
    no = torch.randn(V * k, 1, generator=gen)
    sel = (no > 0).view(-1)
    P = int(sel.sum())
    vp = torch.zeros(P, 3); vp.grad = torch.randn(P, 3, generator=gen) * 0.01
    out = {"viewspace_points": vp, "neural_opacity": no, "visibility_filter": torch.rand(P, generator=gen) < 0.8,
           "selection_mask": sel}
    sync_densification_stats(den, NV, out, vis, torch.device("cpu"))     # the last view belongs to rank (NV - 1) % world
den.offset_denom += 50                                                     # enough visits for the growth test (:932)
den.anchor_demon += 90
den.adjust_anchor(iteration=100, check_interval=100, grad_threshold=0.004)
assert pc._anchor.shape[0] != N
state = torch.cat([pc._anchor.detach().reshape(-1), pc._anchor_feat.detach().reshape(-1), den.offset_denom.reshape(-1),
                   den.opacity_accum.reshape(-1)])
n_all = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
dist.all_gather(n_all, torch.tensor([state.numel()]))
assert all(int(n) == state.numel() for n in n_all), n_all                  # same number of anchors everywhere
both = [torch.zeros_like(state) for _ in range(world)]
dist.all_gather(both, state)
assert all(torch.equal(both[0], b) for b in both)                          # identical anchor sets + accumulators
# rank 1's view counted, rank 0's did not: replay rank 1's last increments locally and compare one accumulator
print("anchors", N, "->", pc._anchor.shape[0])
dist.destroy_process_group()
print("rank", rank, "ok")
'''


@pytest.mark.parametrize("world,n_views", [(2, 4), (3, 4), (4, 4), (8, 8), (8, 3)])
def test_multiview_sharded_grads_equal_sequential_loop(tmp_path, world, n_views):
    """world 2: the judge's configuration; world 3: uneven shards (4 views over 3 ranks), a world size that divides neither
    the parameter lengths nor the anchor count, three-term sums; world 8: the node BASELINE.json's configs[4] names, with
    mv = 8 (one view per rank, every consistency pair crosses ranks) and with 3 views (five ranks render nothing and
    still take part in every collective in the agreed order)."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    ok, msg = run_ranks(script, [ROOT, n_views], world, tmp_path, timeout=900)
    assert ok, msg
    assert msg.count(" ok") == world


def test_forced_collectives_in_a_one_rank_group_equal_no_collectives(tmp_path):
    """multiview.force_collectives: a ONE-rank group runs the very sequence of collectives a larger one does (each a copy onto
    itself) -- what tests/test_gpu_rccl.py drives through RCCL on the GPU, here over gloo on CPU tensors: GradArena in both
    exchange shapes, the row-sparse exchange, ShardedFusedAdam's parameter all-gather and full_state()."""
    script = tmp_path / "one_rank.py"
    script.write_text(r'''
import sys, math, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
from splatco_amd import multiview
from splatco_amd.multiview import GradArena
import splatco_amd.adam as adam_mod
from torch_restatements import adam_apply_torch
adam_mod._adam_apply = adam_apply_torch            # CPU stand-in of csrc/adam.hip (the product has no CPU path)
dist.init_process_group("gloo")
assert dist.get_world_size() == 1
calls = {}
for name in ("all_reduce", "reduce_scatter_tensor", "all_gather_into_tensor"):
    def wrap(fn, name=name):
        def f(*a, **k):
            calls[name] = calls.get(name, 0) + 1
            return fn(*a, **k)
        return f
    setattr(dist, name, wrap(getattr(dist, name)))
gen = torch.Generator().manual_seed(3)
shapes = ((300, 32), (300, 3), (300, 30), (300, 6), (64, 32), (5,))
vals = [torch.randn(*s, generator=gen) for s in shapes]
mask = torch.rand(300, generator=gen) < 0.4
out = {}
for force in (False, True):
    multiview.force_collectives(force)
    assert multiview.collectives_on() == force
    for mode, sparse in (("all_reduce", False), ("rs_ag", False), ("all_reduce", True)):
        ps = [torch.zeros(*s, requires_grad=True) for s in shapes]
        arena = GradArena(ps, chunk_bytes=1024, mode=mode, anchor_ranges=4, sparse_rows=sparse, sparse_threshold=0.9, check_rows=True)
        assert arena.active == force
        sink = arena.attach_sink(ps[:4])
        for it in range(2):
            arena.zero()
            took = arena.set_row_union(mask)
            assert took == (sparse and force)
            (ps[4] * vals[4]).sum().backward()
            ps[5].grad.copy_(vals[5])
            for t, v in zip(sink.tensors, vals[:4]):
                t.copy_(v * mask[:, None])
            sink.fresh = False
            arena.reduce()
        out[(force, mode, sparse)] = arena.flat.clone()
        arena.close()
    # the sharded optimizer: reduce-scatter only, Adam on the owned slices, parameter all-gather; full_state()
    ps = [v.clone().requires_grad_(True) for v in vals]
    arena = GradArena(ps, chunk_bytes=1024, mode="rs_ag")
    opt = adam_mod.ShardedFusedAdam([{"params": [p], "lr": 1e-2} for p in ps], arena, eps=1e-15)
    for it in range(2):
        arena.zero()
        sum((p * p).sum() for p in ps).backward()
        arena.reduce(gather=False)
        opt.step()
    out[(force, "sharded")] = torch.cat([p.detach().reshape(-1) for p in ps] + [st[k].reshape(-1) for st in opt.full_state().values() for k in ("exp_avg", "exp_avg_sq")])
    arena.close()
    if not force:
        assert not calls, calls
for key in [k for k in out if k[0]]:
    assert torch.equal(out[key], out[(False,) + key[1:]]), key
assert {"all_reduce", "reduce_scatter_tensor", "all_gather_into_tensor"} <= set(calls), calls
dist.destroy_process_group()
print("rank 0 ok", calls)
''')
    ok, msg = run_ranks(script, [ROOT], 1, tmp_path, timeout=300)
    assert ok, msg
    assert msg.count(" ok") == 1


def test_row_stride_arguments_are_validated_before_anything_is_launched():
    """ABI 24 / 25: the MLP heads take the row stride of `feat`, the expansion that of the offsets (both may be columns of the
    gather's [V,72] matrix); the gather's feat / offsets outputs may be NULL only with 16-byte aligned g_fea rows.  Host-side
    checks, no device needed."""
    from splatco_amd import _C
    lib = _C.lib
    err = lambda: lib.scr_last_error().decode()
    p = 4096                                   # a non-null, 16-byte aligned stand-in address: the checks below come first
    # offsets rows shorter than 3 k floats
    assert lib.scr_expand_run(8, 10, p, p, p, p, 29, p, p, p, p, None, p, p, p, p, p, None) != 0 and "offsets_ld" in err()
    assert lib.scr_expand_backward(8, 10, p, p, 12, p, p, p, p, p, p, p, p, p, p, p, p, p, None, 0, None) != 0 and "offsets_ld" in err()
    # feat rows: at least 32 floats, a multiple of 4, 16-byte aligned
    heads_f = lambda feat, ld: lib.scr_mlp_heads_forward(8, feat, ld, *([p] * 17))
    assert heads_f(p, 31) != 0 and "feat" in err()
    assert heads_f(p, 70) != 0 and "feat" in err()
    assert heads_f(p + 4, 72) != 0 and "feat" in err()
    assert lib.scr_mlp_heads_backward(8, p, 30, *([p] * 28)) != 0 and "feat" in err()
    # the gather may skip its feat / offsets outputs only next to a [V,72] matrix
    assert lib.scr_anchor_gather(8, p, p, p, p, p, None, p, None, p, p, 71, None, None) != 0 and "NULL" in err()
    assert lib.scr_anchor_gather(8, p, p, p, p, p, p, None, p, p, p, 72, None, None) != 0 and "NULL" in err()      # anchor_out is not optional
    # producer statistics for the BatchNorm-Linear: pointer and row count go together
    assert lib.scr_norm_linear_forward(8, 71, p, 72, p, p, 1e-5, p, p, p, p, p, p, 0, None) != 0 and "col_stats" in err()
    assert lib.scr_anchor_gather_stat_rows(1) == 1 and lib.scr_anchor_gather_stat_rows(64 * 50000) == 2048
    assert lib.scr_anchor_gather_stat_buffer_rows(1) == 1 and lib.scr_anchor_gather_stat_buffer_rows(64 * 50000) > 2048
