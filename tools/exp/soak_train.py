"""Developer soak (GPU box): a few hundred collaborative steps with everything on -- mv views drawn at random, the consistency and
total-variation terms, densification statistics every step, adjust_anchor (grow + prune) every 100 iterations, FusedAdam -- against
ground-truth images rendered from a TEACHER scene (another seed).  Reports per 50 iterations: loss, PSNR of view 0 against its
target, step time, anchors, reserved memory, device allocations, and whether every parameter is still finite.
usage: python tools/exp/soak_train.py [iterations] [anchors] [mv] [WxH] [arena]
arena: the gradients live in a multiview.GradArena (the anchor gather's backward writes straight into it), rebuilt after every
adjust_anchor -- the layout the sharded step uses, here with one rank."""
import math, sys, time, types, random
import torch
sys.path.insert(0, ".")
from splatco_amd.adam import FusedAdam
from splatco_amd.densify import AnchorDensifier
from splatco_amd.renderer import prefilter_voxel, render
from splatco_amd.synthetic import synthetic_anchor_model, synthetic_views
from splatco_amd.train_step import collaborative_step

ITERS = int(sys.argv[1]) if len(sys.argv) > 1 else 400
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200_000
MV = int(sys.argv[3]) if len(sys.argv) > 3 else 2
W, H = (int(x) for x in (sys.argv[4] if len(sys.argv) > 4 else "640x360").split("x"))
ARENA = len(sys.argv) > 5 and sys.argv[5] == "arena"
dev = torch.device("cuda:0")
pipe = types.SimpleNamespace(debug=False, compute_cov3D_python=False)
bg = torch.ones(3, device=dev)
views = [v.to(dev) for v in synthetic_views(6, W, H)]
teacher = synthetic_anchor_model(N, 101, dev, plane_size=512)
teacher.eval()
with torch.no_grad():
    gts = [render(v, teacher, pipe, bg, visible_mask=prefilter_voxel(v, teacher, pipe, bg))["render"].clamp(0, 1).clone() for v in views]
del teacher
pc = synthetic_anchor_model(N, 7, dev, plane_size=512)
groups = [{"params": [getattr(pc, "_" + n)], "lr": lr, "name": n} for n, lr in (("anchor", 0.0), ("offset", 1e-3), ("anchor_feat", 7.5e-3), ("scaling", 7e-3))]
groups.append({"params": [p for n, p in pc.named_parameters() if not n.startswith("_") and p.requires_grad], "lr": 2e-3, "name": "mlp_and_feat_planes"})
opt = FusedAdam(groups, eps=1e-15)
den = AnchorDensifier(pc, opt, voxel_size=0.01, seed=3)
from splatco_amd.multiview import GradArena
make_arena = lambda: GradArena([p for grp in opt.param_groups for p in grp["params"]]) if ARENA else None
arena = make_arena()
rng = random.Random(0)


def psnr0():
    with torch.no_grad():
        img = render(views[0], pc, pipe, bg, visible_mask=prefilter_voxel(views[0], pc, pipe, bg))["render"].clamp(0, 1)
        return float(10 * torch.log10(1.0 / ((img - gts[0]) ** 2).mean()))


print(f"start: PSNR(view 0) {psnr0():.2f} dB, {pc._anchor.shape[0]} anchors")
t_blk, losses = time.perf_counter(), []
for it in range(1, ITERS + 1):
    pick = rng.sample(range(len(views)), MV)
    cw = 0.05 if 100 < it < 300 else 0.0
    loss, out, _ = collaborative_step(pc, [views[i] for i in pick], [gts[i] for i in pick], pipe, bg, optimizer=opt, densifier=den, arena=arena,
                                      consistency_weight=cw, iteration=it, tv_weight=4e-7)
    losses.append(loss.detach())
    if it % 100 == 0 and it >= 100:
        den.adjust_anchor(iteration=it, check_interval=100, grad_threshold=0.0002)
        if arena is not None:           # the per-anchor parameters are new tensors now: a new arena over the optimizer's current ones
            arena.close()
            arena = make_arena()
    if it % 50 == 0:
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t_blk) / 50
        ms = torch.cuda.memory_stats(dev)
        finite = all(bool(torch.isfinite(p).all()) for p in pc.parameters())
        import resource
        rss = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 2**20
        print(f"it {it}: host peak RSS {rss:.2f} GiB; loss {float(torch.stack(losses).mean()):.4f}, PSNR(view 0) {psnr0():.2f} dB, {dt * 1e3:.1f} ms per step (mv = {MV}), "
              f"{pc._anchor.shape[0]} anchors, reserved {ms['reserved_bytes.all.current'] / 2**30:.2f} GiB, device allocs {ms['num_device_alloc']}, "
              f"all parameters finite: {finite}")
        losses, t_blk = [], time.perf_counter()
