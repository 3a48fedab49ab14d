"""GPU tests of the multi-view branch at the sizes BASELINE.json names for it, on the ONE GPU of the test box:

  * sharded == sequential ON THE HIP PATH: two ranks on one device over gloo run train_step.collaborative_step with a
    GradArena (gradient sink, hook-issued pieces, per-anchor exchange in anchor ranges), the cross-view consistency term,
    the tri-plane total-variation term (added once, after the exchange) and a densifier; every rank then replays the reference's sequential mv loop (train.py:171-240: all views, summed
    losses + pairwise term, ONE backward; training_statis of the last view, train.py:264-266) with the same kernels.
  * configs[3] per-GPU reality: 5 M anchors, mv = 4 views rendered one after another on one GPU -- four live rasterizer
    graphs, one backward -- which is exactly how the reference executes --mv 4.
  * configs[4] per-rank size: 20 M anchors, one view: prefilter radii bit-exact against the OpenMP oracle at N = 20 M and
    the full training step.
  * num_rendered >= 2^32 is refused (capi.hip, scr_forward_plan) instead of wrapping.
"""
import os
import sys
import types

import numpy as np
import pytest
import torch

from util import oracle_settings, run_ranks

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SHARDED_WORKER = r'''
import os, sys, types, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from splatco_amd import stats
from splatco_amd.densify import AnchorDensifier
from splatco_amd.expand import visible_indices
from splatco_amd.losses import view_loss
from splatco_amd.multiview import GradArena, pair_consistency
from splatco_amd.renderer import prefilter_voxel, render
from splatco_amd.synthetic import synthetic_anchor_model, synthetic_views
from splatco_amd.train_step import collaborative_step

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
pipe = types.SimpleNamespace(debug=False, compute_cov3D_python=False)
bg = torch.ones(3, device=dev)
W, H, N, MV = 640, 360, 200_000, int(sys.argv[3])
TVW = 1e-3          # the reference's 4e-7 would vanish next to this scene's plane gradients: the test wants the term visible
views = [v.to(dev) for v in synthetic_views(MV, W, H)]
g = torch.Generator(device=dev).manual_seed(5)
base = torch.rand(3, H, W, device=dev, generator=g)
gts = [(base + 0.02 * i).clamp(0, 1) for i in range(MV)]           # alike enough for SSIM > 0.6: every pair counts


def make():
    pc = synthetic_anchor_model(N, 9, dev, plane_size=256)
    idle = {id(p) for p in pc.feat_planes._feat.inactive_parameters()}
    groups = [{"params": [getattr(pc, "_" + n)], "lr": 1e-4, "name": n} for n in ("anchor", "offset", "anchor_feat", "scaling")]
    rest = [p for n, p in pc.named_parameters() if not n.startswith("_") and p.requires_grad and id(p) not in idle]
    groups.append({"params": rest, "lr": 1e-3, "name": "mlp_and_feat_planes"})
    opt = torch.optim.Adam(groups, eps=1e-15)
    den = AnchorDensifier(pc, opt, voxel_size=0.01, seed=77)
    return pc, [p for grp in groups for p in grp["params"]], den, groups


ULP = 2.0 ** -23


def sequential(pc, params, den, cw, perturb=None):
    """The reference's loop: every view on this one process, one backward, statistics of the last view.
    perturb = a generator: dL/dpixel of every view is multiplied by (1 + ULP * U(-1, 1)) on its way back -- the place where
    the sharded step's other summation order enters; what that does to a tensor is the yardstick of the comparison."""
    for p in params:
        p.grad = None
    total, outs = None, []
    for cam, gt in zip(views, gts):
        vis = prefilter_voxel(cam, pc, pipe, bg)
        out = render(cam, pc, pipe, bg, visible_mask=vis, retain_grad=True)
        if perturb is not None:
            out["render"].register_hook(lambda g: g * (1.0 + ULP * (2.0 * torch.rand(g.shape, device=g.device, generator=perturb) - 1.0)))
        loss = view_loss(out["render"], gt, out["scaling"])
        total = loss if total is None else total + loss
        outs.append((out, vis))
    own, cross = 0.0, 0.0
    if cw:
        for i in range(MV):
            for j in range(i + 1, MV):
                t = pair_consistency(outs[i][0]["render"], gts[i], outs[j][0]["render"], gts[j])
                assert t is not None
                total = total + cw * t
                if i % world == j % world:
                    own = own + cw * t.detach()
                else:
                    cross = cross + cw * t.detach()
    total.backward()
    # the tri-plane total-variation term (train.py:242-243): once, whatever the number of ranks
    plane = pc.feat_planes._feat.k0s[0].xy_plane
    before = plane.grad.clone()
    pc.feat_planes.tv_loss(TVW)
    share = float((plane.grad - before).norm() / plane.grad.norm())
    assert share > 1e-2, share            # large enough that adding it once per rank would fail the 1e-6 comparison below
    out, vis = outs[-1]
    with torch.no_grad():
        inc_op, inc_g = stats.statis_increments(den.n_offsets, out["viewspace_points"].grad, out["neural_opacity"],
                                                out["visibility_filter"], out["selection_mask"])
        den.apply_statis(visible_indices(vis), inc_op, inc_g)
    return total.detach(), own, cross


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-300))


def bar(noise):
    """8 x the larger of one ulp and the measured response of the same quantity to a one-ulp perturbation of dL/dpixel"""
    return 8.0 * max(ULP, noise)


for mode, cw in (("all_reduce", 0.05), ("rs_ag", 0.0))[:int(sys.argv[2])]:
    pc_a, params_a, den_a, _ = make()
    pc_b, params_b, den_b, _ = make()
    pc_p, params_p, den_p, _ = make()          # the yardstick: the sequential loop under a one-ulp perturbation of dL/dpixel
    assert all(torch.equal(a, b) for a, b in zip(params_a, params_b))
    # several pieces per large parameter; the rs_ag case also exchanges the per-anchor gradients ROW-SPARSE (only the rows of
    # the union of the ranks' visible anchors travel, GradArena.set_row_union; the threshold is lifted so that it always packs)
    arena = GradArena(params_a, chunk_bytes=4 << 20, mode=mode, anchor_ranges=4, sparse_rows=(mode == "rs_ag"), sparse_threshold=1.01, check_rows=True)
    gen_p = torch.Generator(device=dev).manual_seed(1234)
    for it in range(2):         # step 0 goes out from reduce() and agrees on the order; step 1 issues from the hooks / ranges
        loss_a, out_a, _ = collaborative_step(pc_a, views, gts, pipe, bg, consistency_weight=cw, densifier=den_a, arena=arena,
                                                  iteration=4 * (it + 1), tv_weight=TVW)
        loss_b, own, cross = sequential(pc_b, params_b, den_b, cw)
        sequential(pc_p, params_p, den_p, cw, perturb=gen_p)
    # the sequential loop is bit-reproducible (deterministic plane gradients included): a second pass leaves the same bits
    keep = [p.grad.clone() for p in params_b]
    sequential(pc_b, params_b, AnchorDensifier(pc_b, torch.optim.Adam([pc_b._anchor]), voxel_size=0.01, seed=77), cw)
    fails = [f"sequential loop not bit-reproducible: parameter {i} {tuple(k.shape)} rel {rel(p.grad, k):.2e}"
             for i, (p, k) in enumerate(zip(params_b, keep)) if not torch.equal(p.grad, k)]
    assert arena._order is not None and arena._sink is not None and len(arena.sink_ranges) == 4
    assert (arena.last_union_fraction is not None and 0.5 < arena.last_union_fraction <= 1.0) == (mode == "rs_ag")
    assert arena._cursor == len(arena.units)
    worst = 0.0
    for i, (pa, pb, pp) in enumerate(zip(params_a, params_b, params_p)):
        assert pa.grad is not None and pb.grad is not None and pa.grad.data_ptr() == arena.views[i].data_ptr()
        r, noise = rel(pa.grad, pb.grad), rel(pp.grad, pb.grad)
        worst = max(worst, r)
        # the sum over views is formed in another order (per rank, then across ranks) and, with the pairwise term, dL/dpixel
        # too; nothing else differs.  How far that may move a tensor depends on its conditioning (a bias gradient is a sum
        # over millions of signed terms), which is MEASURED here: the response of the same tensor to a one-ulp perturbation.
        line = f"[{mode}] parameter {i} {tuple(pa.shape)}: sharded vs sequential {r:.2e}, one-ulp response {noise:.2e}, bar {bar(noise):.2e}"
        print(f"[rank {rank}] " + line, flush=True)
        if not r <= bar(noise):
            fails.append(line)
    # losses: a cross-rank pair is evaluated by both owners (each differentiates its own image): it counts twice in the
    # sum of the local losses
    tl = loss_a.clone().double()
    dist.all_reduce(tl)
    want = float(loss_b) + float(cross)
    # fp32 sums of MV view losses + MV (MV - 1) / 2 pair terms (cross pairs twice) in two different orders: every partial
    # sum rounds once, all terms are positive
    n_terms = MV + MV * (MV - 1)
    line = f"[{mode}] loss: sharded {float(tl):.9g}, sequential {want:.9g}, rel {abs(float(tl) - want) / abs(want):.2e}, bar {n_terms * 2.0 ** -24:.2e}"
    print(f"[rank {rank}] " + line, flush=True)
    if not abs(float(tl) - want) <= n_terms * 2.0 ** -24 * abs(want):
        fails.append(line)
    # densification statistics: the last view's on every rank
    for name in ("opacity_accum", "anchor_demon", "offset_gradient_accum", "offset_denom"):
        a, b, p = getattr(den_a, name), getattr(den_b, name), getattr(den_p, name)
        if cw == 0.0:
            if not torch.equal(a, b):                           # same kernels, same dL/dpixel: bit for bit
                fails.append(f"[{mode}] {name}: not bit-identical, rel {rel(a, b):.2e}")
        else:                                                   # the pairwise term reaches the image in another order
            r, noise = rel(a, b), rel(p, b)
            line = f"[{mode}] {name}: sharded vs sequential {r:.2e}, one-ulp response {noise:.2e}, bar {bar(noise):.2e}"
            print(f"[rank {rank}] " + line, flush=True)
            if not r <= bar(noise):
                fails.append(line)
    assert not fails, "\n".join(["comparisons out of bounds:"] + fails)
    for d in (den_a, den_b):
        d.offset_denom += 50
        d.anchor_demon += 90
    den_a.adjust_anchor(iteration=100, check_interval=100, grad_threshold=float(den_a.offset_gradient_accum.mean() / 52))
    n_new = pc_a._anchor.shape[0]
    assert n_new != N, "adjust_anchor changed nothing: the comparison below would be empty"
    state = torch.cat([t.detach().reshape(-1).float() for t in (pc_a._anchor, pc_a._anchor_feat, pc_a._offset, pc_a._scaling,
                                                                den_a.offset_denom, den_a.opacity_accum, den_a.anchor_demon,
                                                                den_a.offset_gradient_accum)])
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([state.numel()], device=dev))
    assert all(int(s) == state.numel() for s in sizes), sizes
    both = [torch.zeros_like(state) for _ in range(world)]
    dist.all_gather(both, state)
    assert all(torch.equal(both[0], b) for b in both), "replicas diverged after adjust_anchor"
    if cw == 0.0:
        den_b.adjust_anchor(iteration=100, check_interval=100, grad_threshold=float(den_b.offset_gradient_accum.mean() / 52))
        assert pc_b._anchor.shape[0] == n_new and torch.equal(pc_a._anchor, pc_b._anchor) and torch.equal(pc_a._offset, pc_b._offset)
    print(f"[rank {rank}] {mode} consistency {cw}: worst gradient rel-L2 {worst:.2e}, anchors {N} -> {n_new}", flush=True)
    arena.close()
if int(sys.argv[2]) >= 2:
    # ---- optimiser-state sharding (adam.ShardedFusedAdam: reduce-scatter -> scr_adam_step on this rank's 1/world of every
    # piece of the parameter arena -> all-gather of the parameters) against the replicated step (all-reduce, FusedAdam over
    # everything on every rank): the same parameters after three full training steps with the total-variation term
    from splatco_amd.adam import FusedAdam, ShardedFusedAdam
    pc_c, params_c, _, groups_c = make()
    pc_d, params_d, _, groups_d = make()
    # (round 6: the sharded side also exchanges its per-anchor gradients ROW-SPARSE -- packed rows all-reduced, i.e. complete on
    # every rank, dense units reduce-scattered: the combination GradArena.set_row_union describes; threshold lifted so it packs)
    arena_c = GradArena(params_c, chunk_bytes=4 << 20, mode="rs_ag", anchor_ranges=4, sparse_rows=True, sparse_threshold=1.01, check_rows=True)
    opt_c = ShardedFusedAdam(groups_c, arena_c, eps=1e-15)
    arena_d = GradArena(params_d, chunk_bytes=4 << 20, mode="all_reduce", anchor_ranges=4)
    opt_d = FusedAdam(groups_d, eps=1e-15)
    assert opt_c.nbytes_state() * world == 2 * arena_c.nbytes()
    # (1) the optimizer in isolation, BIT FOR BIT: both are handed the same complete gradient (the plane gradients of two
    # replicas differ in their last bits from run to run -- tools/exp/determinism_probe.py -- so two independently computed
    # steps cannot be compared exactly): ownership slices, moment shards, the parameter all-gather, through csrc/adam.hip
    for it in range(3):
        collaborative_step(pc_c, views, gts, pipe, bg, arena=arena_c, iteration=4 * (it + 1), tv_weight=TVW)    # full exchange, no optimizer
        arena_d.flat.copy_(arena_c.flat)
        arena_d.bind()
        opt_c.step()
        opt_d.step()
        for i, (a, b) in enumerate(zip(params_c, params_d)):
            assert a.data_ptr() != b.data_ptr() and torch.equal(a, b), ("sharded optimizer vs replicated", it, i, tuple(a.shape), rel(a, b))
    full = opt_c.full_state()
    for i, p in enumerate(params_d):
        assert torch.equal(full[i]["exp_avg"], opt_d.state[p]["exp_avg"]) and torch.equal(full[i]["exp_avg_sq"], opt_d.state[p]["exp_avg_sq"])
        assert float(full[i]["step"]) == float(opt_d.state[p]["step"]) == 3.0
    # (2) the whole path (reduce-scatter only, total-variation term on the owned slices, sharded step, parameter gather)
    # against the replicated step, each computing its own gradients: equal to the run-to-run noise of the plane gradients
    worst = 0.0
    for it in range(3, 5):
        for pc_x, opt_x, arena_x in ((pc_c, opt_c, arena_c), (pc_d, opt_d, arena_d)):
            collaborative_step(pc_x, views, gts, pipe, bg, optimizer=opt_x, arena=arena_x, iteration=4 * (it + 1), tv_weight=TVW)
    assert arena_c.last_union_fraction is not None and 0.5 < arena_c.last_union_fraction <= 1.0      # the packed path was taken
    for i, (a, b) in enumerate(zip(params_c, params_d)):
        worst = max(worst, rel(a, b))
        # (two replicated replicas drift apart like this too: 3.7e-6 max-abs on a weight matrix after three steps in the probe)
        assert rel(a, b) <= 1e-4, (i, tuple(a.shape), rel(a, b))
    print(f"[rank {rank}] sharded optimizer: bit-identical to the replicated one on the same gradients; after two more independent steps worst "
          f"parameter rel-L2 {worst:.1e} (moments {opt_c.nbytes_state() >> 20} MiB per rank instead of {2 * arena_c.nbytes() >> 20})", flush=True)
    arena_c.close(); arena_d.close()
dist.destroy_process_group()
print("rank", rank, "ok")
'''


@pytest.mark.parametrize("world", [2, 4, 8])
def test_sharded_step_equals_the_sequential_loop_on_the_hip_path(tmp_path, world):
    """world 2: two views per rank (the gather's backward overwrites for the first view and adds for the second), both
    exchange shapes, then the sharded optimizer against the replicated one; world 4: one view per rank, every consistency
    pair crosses ranks; world 8 (the node of configs[4]): eight views, one per rank, eight processes on the one device."""
    script = tmp_path / "sharded_worker.py"
    script.write_text(SHARDED_WORKER)
    ok, msg = run_ranks(script, [ROOT, "2" if world == 2 else "1",       # world 4 / 8 run the all_reduce + consistency case only
                                 "8" if world == 8 else "4"], world, tmp_path, timeout=1500)
    if ok:
        print("\n".join(l for l in msg.splitlines() if l.startswith("[rank 0]")))
    assert ok, msg                                                      # (ends with the failing rank's own stderr)
    assert msg.count(" ok") == world


def _train_setup(pc, seed, mode="all_reduce"):
    from splatco_amd.densify import AnchorDensifier
    from splatco_amd.multiview import GradArena
    groups = [{"params": [getattr(pc, "_" + n)], "lr": 1e-4, "name": n} for n in ("anchor", "offset", "anchor_feat", "scaling")]
    idle = {id(p) for p in pc.feat_planes._feat.inactive_parameters()}
    rest = [p for n, p in pc.named_parameters() if not n.startswith("_") and p.requires_grad and id(p) not in idle]
    groups.append({"params": rest, "lr": 1e-3, "name": "mlp_and_feat_planes"})
    opt = torch.optim.Adam(groups, eps=1e-15, fused=True)
    den = AnchorDensifier(pc, opt, seed=seed)
    arena = GradArena([p for grp in groups for p in grp["params"]], mode=mode)
    return opt, den, arena


def _check_param_grads(pc):
    planes = pc.feat_planes._feat.k0s
    for name, p in [("anchor", pc._anchor), ("offset", pc._offset), ("feat", pc._anchor_feat), ("scaling", pc._scaling),
                    ("mlp_opacity", pc.mlp_opacity[0].weight), ("mlp_cov", pc.mlp_cov[0].weight),
                    ("mlp_color", pc.mlp_color[2].weight), ("plane L0 (attention)", planes[0].xy_plane),
                    ("plane L1", planes[1].xz_plane), ("plane L2 (1400)", planes[2].yz_plane)]:
        assert p.grad is not None and torch.isfinite(p.grad).all() and p.grad.abs().sum() > 0, name
    assert planes[3].xy_plane.grad is None            # the 2800^2 level is never sampled


def test_per_anchor_gradients_are_final_before_the_triplane_backward_runs():
    """Round 6, DESIGN.md section 7: in a multi-rank step (the gradient sink reports anchor RANGES) the tri-plane features are
    sampled before the anchor gather is applied (FeaturePlanes.presample), so autograd runs the gather's backward -- which
    forms the attribute branch's dx itself and finishes the per-anchor gradients range by range -- BEFORE the tri-plane and
    attention backward passes: their exchange is on the wire while those still compute.  Checked on one GPU with a
    stand-in for the arena: every range is reported before the first plane gradient appears, and all gradients equal
    those of the ordinary order bit for bit (same kernels on the same values)."""
    import math
    from splatco_amd import anchor_gather as ag
    from splatco_amd.cameras import look_at_camera
    from splatco_amd.renderer import prefilter_voxel, render
    from splatco_amd.synthetic import synthetic_anchor_model
    dev = torch.device("cuda:0")
    pipe = types.SimpleNamespace(debug=False, compute_cov3D_python=False)
    bg = torch.ones(3, device=dev)
    cam = look_at_camera(eye=(0.2, -0.1, -5.0), target=(0, 0, 0), up=(0, -1, 0), FoVx=math.radians(60), width=320, height=200).to(dev)
    target = torch.rand(3, 200, 320, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
    N = 40_000
    results = []
    for ranged in (False, True):
        pc = synthetic_anchor_model(N, 5, dev, plane_size=280)
        pc.train()
        names = ("_anchor_feat", "_anchor", "_offset", "_scaling")
        grads = [torch.full_like(getattr(pc, n), float("nan")) for n in names]
        sink = ag.GradSink(*grads)
        events = []
        if ranged:
            step = (N // 4 + 63) // 64 * 64
            sink.ranges = [(a, min(N, a + step)) for a in range(0, N, step)]
            sink.on_range = lambda r: events.append(("range", r))
        pc._grad_sink = sink
        planes = [p for p in pc.feat_planes._feat.parameters() if p.dim() == 4 and p.requires_grad]
        for p in planes[:3]:
            p.register_post_accumulate_grad_hook(lambda _p: events.append(("plane", 0)))
        vis = prefilter_voxel(cam, pc, pipe, bg)
        out = render(cam, pc, pipe, bg, visible_mask=vis, retain_grad=True)
        ((out["render"] - target).abs().mean() + 0.01 * out["scaling"].prod(dim=1).mean()).backward()
        pc._grad_sink = None
        if ranged:
            kinds = [e[0] for e in events]
            assert kinds.count("range") == len(sink.ranges) and "plane" in kinds
            assert max(i for i, k in enumerate(kinds) if k == "range") < min(i for i, k in enumerate(kinds) if k == "plane"), kinds
        results.append([g.clone() for g in grads] + [p.grad.clone() for p in pc.parameters() if p.grad is not None and not any(p is getattr(pc, n) for n in names)])
    assert len(results[0]) == len(results[1]) > 8
    for a, b in zip(*results):
        assert torch.isfinite(a).all() and torch.equal(a, b)


def test_cfg3_5M_anchors_mv4_sequential_views_on_one_gpu():
    """configs[3] as the REFERENCE executes it (train.py:171-240): the four views of --mv 4 rendered one after another on
    one GPU, four rasterizer graphs alive, one backward; here through collaborative_step at world size 1 with the gradient
    arena, the densifier and the optimiser."""
    from splatco_amd.renderer import prefilter_voxel, render
    from splatco_amd.synthetic import ANCHOR_CONFIGS, synthetic_anchor_model, synthetic_views
    from splatco_amd.train_step import collaborative_step
    dev = torch.device("cuda:0")
    N, mv, seed = ANCHOR_CONFIGS["cfg3"]
    assert (N, mv) == (5_000_000, 4)
    pc = synthetic_anchor_model(N, seed, dev)
    opt, den, arena = _train_setup(pc, seed)
    pipe = types.SimpleNamespace(debug=False, compute_cov3D_python=False)
    bg = torch.ones(3, device=dev)
    views = [v.to(dev) for v in synthetic_views(mv)]
    g = torch.Generator(device=dev).manual_seed(100 + seed)
    gts = [torch.rand(3, 1080, 1920, device=dev, generator=g) for _ in views]
    torch.cuda.reset_peak_memory_stats()
    before = pc._anchor.detach().clone()
    loss, out, _ = collaborative_step(pc, views, gts, pipe, bg, densifier=den, arena=arena)       # no optimiser: look at the gradients
    torch.cuda.synchronize()
    peak = torch.cuda.max_memory_allocated() / 2 ** 30
    P = out["radii"].shape[0]
    print(f"[cfg3] 5 M anchors, mv = 4 sequential views: last view {P} Gaussians, loss {float(loss):.4f}, "
          f"peak memory {peak:.1f} GiB (four live graphs + 1.42 GB arena)")
    assert torch.isfinite(loss) and out["render"].shape == (3, 1080, 1920) and P > 10_000_000
    _check_param_grads(pc)
    assert arena._sink is not None and pc._anchor.grad.data_ptr() == arena._sink.tensors[1].data_ptr()
    assert float(den.anchor_demon.sum()) > 0 and float(den.offset_denom.sum()) > 0        # the last view's statistics arrived
    # the gradient of four views is the sum of four single-view gradients (a second arena step per view)
    total = torch.zeros_like(pc._anchor_feat)
    for v, gt in zip(views, gts):
        collaborative_step(pc, [v], [gt], pipe, bg, arena=arena)
        total += pc._anchor_feat.grad
    collaborative_step(pc, views, gts, pipe, bg, arena=arena)
    err = float((pc._anchor_feat.grad - total).norm() / total.norm())
    assert err <= 1e-6, err
    # bit-reproducible image of the last view
    with torch.no_grad():
        vis = prefilter_voxel(views[-1], pc, pipe, bg)
        again = render(views[-1], pc, pipe, bg, visible_mask=vis)["render"]
    assert torch.equal(again, out["render"].detach())
    # and the full step with the optimiser moves the anchors
    collaborative_step(pc, views, gts, pipe, bg, optimizer=opt, densifier=den, arena=arena)
    torch.cuda.synchronize()
    assert torch.isfinite(pc._anchor).all() and not torch.equal(pc._anchor.detach(), before)


def test_cfg4_20M_anchors_one_view_per_rank(oracle):
    """configs[4] per-rank work: 20 M anchors, one 1080p view: anchor visibility bit-exact against the oracle at N = 20 M,
    then the full training step (prefilter, render, loss, backward into the arena, statistics, Adam)."""
    from splatco_amd.rasterizer import GaussianRasterizer
    from splatco_amd import rasterizer as R
    from splatco_amd.renderer import _settings
    from splatco_amd.synthetic import ANCHOR_CONFIGS, synthetic_anchor_model, synthetic_views
    from splatco_amd.train_step import collaborative_step
    dev = torch.device("cuda:0")
    N, mv, seed = ANCHOR_CONFIGS["cfg4"]
    assert (N, mv) == (20_000_000, 8)
    pc = synthetic_anchor_model(N, seed, dev)
    cam = synthetic_views(1)[0]
    bg = torch.ones(3, device=dev)
    pipe = types.SimpleNamespace(debug=False, compute_cov3D_python=False)
    st = oracle_settings(oracle, cam, np.ones(3, np.float32))
    oracle.use_threads(True)
    try:
        with torch.no_grad():
            want = oracle.visible_filter(st, pc.get_anchor.cpu().numpy(), pc.get_scaling[:, :3].cpu().numpy(),
                                         pc.get_rotation.cpu().numpy())
            got = GaussianRasterizer(_settings(cam.to(dev), bg, 1.0, False)).visible_filter(
                means3D=pc.get_anchor, scales=pc.get_scaling[:, :3], rotations=pc.get_rotation)
    finally:
        oracle.use_threads(False)
    assert np.array_equal(got.cpu().numpy(), want), "visible_filter radii at N = 20M"
    assert int((want > 0).sum()) > 15_000_000
    del want, got
    opt, den, arena = _train_setup(pc, seed)
    gt = torch.rand(3, 1080, 1920, device=dev, generator=torch.Generator(device=dev).manual_seed(100 + seed))
    before = pc._anchor_feat.detach().clone()
    torch.cuda.reset_peak_memory_stats()
    plans, inner = [], R.rasterize_forward         # test bookkeeping: the instance count lives in the per-call RasterState

    def recording(*a, **k):
        res = inner(*a, **k)
        plans.append((res[2].P, res[2].I, res[2].max_tile))
        return res
    R.rasterize_forward = recording
    try:
        loss, out, _ = collaborative_step(pc, [cam.to(dev)], [gt], pipe, bg, densifier=den, arena=arena)
    finally:
        R.rasterize_forward = inner
    torch.cuda.synchronize()
    P, I = out["radii"].shape[0], plans[-1][1]
    print(f"[cfg4] 20 M anchors, one view: {P} Gaussians, {I} tile instances ({I / 2**32:.3f} of the 2^32 index space), "
          f"largest tile {plans[-1][2]}, loss {float(loss):.4f}, peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
    assert P > 50_000_000 and 100_000_000 < I < 2 ** 32
    assert torch.isfinite(loss) and torch.isfinite(out["render"]).all()
    _check_param_grads(pc)
    gvs = out["viewspace_points"].grad
    assert gvs is not None and torch.isfinite(gvs).all() and gvs[:, :2].abs().sum() > 0
    assert float(den.anchor_demon.sum()) > 0
    loss2, _, _ = collaborative_step(pc, [cam.to(dev)], [gt], pipe, bg, optimizer=opt, densifier=den, arena=arena)
    torch.cuda.synchronize()
    assert float(loss2) == float(loss)                       # same parameters, deterministic forward: the same loss bit for bit
    assert torch.isfinite(pc._anchor_feat).all() and not torch.equal(pc._anchor_feat.detach(), before)


def test_more_than_2_32_tile_instances_is_an_error_not_a_wrap():
    """capi.hip (scr_forward_plan): instance indices are 32-bit; a scene whose (Gaussian, tile) count reaches 2^32 must be
    refused.  4 160 giant splats over a 16 384 x 16 384 image = 1 048 576 tiles each: 4.36e9 instances -- one workgroup
    of 4 096 Gaussians alone passes 2^32 (the per-workgroup sums saturate instead of wrapping)."""
    from splatco_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
    from splatco_amd.synthetic import synthetic_camera
    dev = torch.device("cuda:0")
    S, P = 16384, 4160
    cam = synthetic_camera(S, S)
    import math
    tx = math.tan(cam.FoVx * 0.5)
    rs = GaussianRasterizationSettings(S, S, tx, tx, torch.ones(3, device=dev), 1.0, cam.world_view_transform.to(dev),
                                       cam.full_proj_transform.to(dev), 1, cam.camera_center.to(dev), False, False)
    means = torch.tensor([[0.0, 0.0, 5.0]], device=dev).repeat(P, 1)
    scales = torch.full((P, 3), 50.0, device=dev)
    rots = torch.tensor([[1.0, 0.0, 0.0, 0.0]], device=dev).repeat(P, 1)
    op = torch.full((P, 1), 0.5, device=dev)
    col = torch.rand(P, 3, device=dev)
    with pytest.raises(RuntimeError, match="num_rendered"):
        GaussianRasterizer(rs)(means3D=means, means2D=torch.zeros(P, 3, device=dev), opacities=op, colors_precomp=col,
                               scales=scales, rotations=rots)
    # one Gaussian fewer than a workgroup's worth below the limit still plans and renders (4 095 x 2^20 < 2^32 - 1)
    # -- not run: 4.29e9 instances would need 73 GB of binning state; the planning arithmetic is what is under test
    # a second call after the refusal works (no state was left behind)
    rs2 = rs._replace(image_height=64, image_width=64)
    img, radii = GaussianRasterizer(rs2)(means3D=means[:8], means2D=torch.zeros(8, 3, device=dev), opacities=op[:8],
                                         colors_precomp=col[:8], scales=scales[:8], rotations=rots[:8])
    assert img.shape == (3, 64, 64) and torch.isfinite(img).all() and int((radii > 0).sum()) == 8
