"""Developer probe (GPU box): what does the cross-view consistency term (train.py:201-239) add to a --mv 4 step at 1080p?"""
import sys, time, types, torch
sys.path.insert(0, ".")
from splatco_amd.synthetic import synthetic_anchor_model, synthetic_views
from splatco_amd.train_step import collaborative_step
dev = torch.device("cuda:0")
pipe = types.SimpleNamespace(debug=False, compute_cov3D_python=False)
bg = torch.ones(3, device=dev)
MV, W, H, N = 4, 1920, 1080, 500_000
views = [v.to(dev) for v in synthetic_views(MV, W, H)]
g = torch.Generator(device=dev).manual_seed(5)
base = torch.rand(3, H, W, device=dev, generator=g)
gts = [(base + 0.02 * i).clamp(0, 1) for i in range(MV)]
pc = synthetic_anchor_model(N, 9, dev, plane_size=256)
for cw in (0.0, 0.05):
    for _ in range(3):
        collaborative_step(pc, views, gts, pipe, bg, consistency_weight=cw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        collaborative_step(pc, views, gts, pipe, bg, consistency_weight=cw)
    torch.cuda.synchronize()
    print(f"consistency weight {cw}: {(time.perf_counter() - t0) * 1000 / 30:.2f} ms per step (mv = {MV}, {W}x{H}, {N} anchors)")
from torch.profiler import profile, ProfilerActivity
for cw in (0.0, 0.05):
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        collaborative_step(pc, views, gts, pipe, bg, consistency_weight=cw)
        torch.cuda.synchronize()
    evs = prof.key_averages()
    dev_total = sum(getattr(ev, "self_device_time_total", 0) for ev in evs)
    print(f"consistency weight {cw}: {dev_total / 1e3:.3f} ms of device time in one step, {sum(ev.count for ev in evs if getattr(ev, 'self_device_time_total', 0) > 0)} device ops")
    if cw:
        rows = [(ev.key, ev.count, ev.self_device_time_total) for ev in evs if getattr(ev, "self_device_time_total", 0) > 0]
        for k, n, t in sorted(rows, key=lambda r: -r[2])[:40]:
            print(f"{n:4d} {t / 1e3:8.3f} ms  {k[:110]}")
