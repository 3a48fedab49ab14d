"""Developer probe (GPU box): which kernels make the EARLY steps of the teacher-scene soak slow (13 ms per mv = 2 step at iteration 50 against
5.5 ms after 500)?  Per-kernel device time of one step at iteration 20 and at iteration 400."""
import sys, types, random, torch
sys.path.insert(0, ".")
from torch.profiler import profile, ProfilerActivity
from splatco_amd.adam import FusedAdam
from splatco_amd.densify import AnchorDensifier
from splatco_amd.renderer import prefilter_voxel, render
from splatco_amd.synthetic import synthetic_anchor_model, synthetic_views
from splatco_amd.train_step import collaborative_step
dev = torch.device("cuda:0")
W, H, N = 640, 360, 200_000
pipe = types.SimpleNamespace(debug=False, compute_cov3D_python=False)
bg = torch.ones(3, device=dev)
views = [v.to(dev) for v in synthetic_views(6, W, H)]
teacher = synthetic_anchor_model(N, 101, dev, plane_size=512); teacher.eval()
with torch.no_grad():
    gts = [render(v, teacher, pipe, bg, visible_mask=prefilter_voxel(v, teacher, pipe, bg))["render"].clamp(0, 1).clone() for v in views]
del teacher
pc = synthetic_anchor_model(N, 7, dev, plane_size=512)
groups = [{"params": [getattr(pc, "_" + n)], "lr": lr, "name": n} for n, lr in (("anchor", 0.0), ("offset", 1e-3), ("anchor_feat", 7.5e-3), ("scaling", 7e-3))]
groups.append({"params": [p for n, p in pc.named_parameters() if not n.startswith("_") and p.requires_grad], "lr": 2e-3, "name": "rest"})
opt = FusedAdam(groups, eps=1e-15)
den = AnchorDensifier(pc, opt, voxel_size=0.01, seed=3)
rng = random.Random(0)
def step(it):
    pick = rng.sample(range(6), 2)
    return collaborative_step(pc, [views[i] for i in pick], [gts[i] for i in pick], pipe, bg, optimizer=opt, densifier=den, iteration=it, tv_weight=4e-7)
import cProfile, pstats, io, time
pr = cProfile.Profile()
for it in range(1, 401):
    if it == 21:
        torch.cuda.synchronize(); t_a = time.perf_counter(); pr.enable()
    if it == 41:
        pr.disable(); torch.cuda.synchronize()
        print(f"iterations 21 - 40: {(time.perf_counter() - t_a) * 50:.2f} ms per step")
        st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(14)
        print("\n".join(l[:140] for l in st.getvalue().splitlines()[:26]))
    if it in (20, 400):
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            loss, out, _ = step(it)
            torch.cuda.synchronize()
        rows = [(ev.key, ev.count, ev.self_device_time_total) for ev in prof.key_averages() if ev.self_device_time_total > 0]
        tot = sum(r[2] for r in rows)
        print(f"--- iteration {it}: {tot / 1e3:.2f} ms of kernels, {out['radii'].shape[0]} Gaussians in the last view")
        for k, n, t in sorted(rows, key=lambda r: -r[2])[:9]:
            print(f"   {n:4d} {t / 1e3:8.3f} ms  {k[:100]}")
    else:
        step(it)
    if it % 100 == 0:
        den.adjust_anchor(iteration=it, check_interval=100, grad_threshold=0.0002)
