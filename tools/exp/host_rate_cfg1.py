"""Developer probe (GPU box): how far ahead of the GPU does the host run in the cfg1 loop?  Enqueue time of K steps against their GPU time."""
import sys, time, math, torch
sys.path.insert(0, ".")
import bench
from splatco_amd.rasterizer import GaussianRasterizer
from splatco_amd.synthetic import synthetic_gaussians
dev = torch.device("cuda:0")
P, W, H = bench.P_CFG1, bench.W_CFG1, bench.H_CFG1
g = synthetic_gaussians(P, W, H, seed=0)
cam = bench.make_view(0, W, H)
rast = GaussianRasterizer(bench.settings_for(cam, g["bg"], dev))
t = lambda a: torch.tensor(a, device=dev, requires_grad=True)
params = dict(means3D=t(g["means3D"]), opacities=t(g["opacities"]), colors_precomp=t(g["colors"]), scales=t(g["scales"]), rotations=t(g["rotations"]))
dL = torch.randn(3, H, W, device=dev)
means2D = torch.zeros(P, 3, device=dev, requires_grad=True)
import os
KEEP = os.environ.get("KEEP", "0") == "1"
PROF = os.environ.get("PROF", "0") == "1"
state = {}
def step():
    for p in params.values():
        p.grad = None
    means2D.grad = None
    img, radii = rast(means2D=means2D, **params)
    img.backward(dL)
    if KEEP:
        state["radii"], state["img"] = radii, img
if PROF:
    from splatco_amd import _C
    _C.profile_enable("blend_backward_kernel", every=4)
for _ in range(10):
    step()
torch.cuda.synchronize()
for K in (20, 20, 100, 100):
    t0 = time.perf_counter()
    per = []
    for _ in range(K):
        s0 = time.perf_counter()
        step()
        per.append(time.perf_counter() - s0)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    per.sort()
    print(f"K = {K}: host enqueue {1e3 * (t1 - t0) / K:.3f} ms per step (median {1e3 * per[K // 2]:.3f}, max {1e3 * per[-1]:.3f}), with the GPU {1e3 * (t2 - t0) / K:.3f} ms per step")
import gc
for pause, label in ((0.0, "no pause"), (0.02, "20 ms idle"), (0.1, "100 ms idle"), (0.5, "500 ms idle"), (0.0, "gc.collect()")):
    torch.cuda.synchronize()
    if label == "gc.collect()":
        g0 = time.perf_counter(); gc.collect(); label += f" = {1e3 * (time.perf_counter() - g0):.0f} ms"
    time.sleep(pause)
    t0 = time.perf_counter()
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    print(f"after {label}: 20 steps at {1e3 * (time.perf_counter() - t0) / 20:.3f} ms per step")
