"""Developer probe (GPU box): where does the HOST spend a collaborative step of a small scene (the launch-bound regime)?"""
import cProfile, pstats, sys, types, io, time, torch
sys.path.insert(0, ".")
from splatco_amd.adam import FusedAdam
from splatco_amd.densify import AnchorDensifier
from splatco_amd.synthetic import synthetic_anchor_model, synthetic_views
from splatco_amd.train_step import collaborative_step
dev = torch.device("cuda:0")
N, W, H, MV = 150_000, 640, 360, 2
pipe = types.SimpleNamespace(debug=False, compute_cov3D_python=False)
bg = torch.ones(3, device=dev)
views = [v.to(dev) for v in synthetic_views(MV, W, H)]
gts = [torch.rand(3, H, W, device=dev) for _ in views]
pc = synthetic_anchor_model(N, 7, dev, plane_size=512)
groups = [{"params": [getattr(pc, "_" + n)], "lr": 1e-3, "name": n} for n in ("anchor", "offset", "anchor_feat", "scaling")]
groups.append({"params": [p for n, p in pc.named_parameters() if not n.startswith("_") and p.requires_grad], "lr": 2e-3, "name": "rest"})
opt = FusedAdam(groups, eps=1e-15)
den = AnchorDensifier(pc, opt, voxel_size=0.01, seed=3)
step = lambda it: collaborative_step(pc, views, gts, pipe, bg, optimizer=opt, densifier=den, iteration=it, tv_weight=4e-7)
for it in range(1, 31):
    step(it)
torch.cuda.synchronize()
t0 = time.perf_counter()
for it in range(31, 81):
    step(it)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"50 steps (mv = {MV}, {N} anchors, {W}x{H}): host {1e3 * (t1 - t0) / 50:.2f} ms per step, with the GPU {1e3 * (t2 - t0) / 50:.2f} ms")
pr = cProfile.Profile()
pr.enable()
for it in range(81, 131):
    step(it)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print("\n".join(l[:150] for l in s.getvalue().splitlines()[:45]))
