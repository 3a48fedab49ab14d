"""Developer probe (GPU box): single-step host stalls of the cfg1 loop -- how often, how long, in which half of the step?"""
import sys, time, gc, torch
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
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
if len(sys.argv) > 2 and sys.argv[2] == "nogc":
    gc.collect(); gc.disable()
rec = []
for i in range(N + 50):
    a = time.perf_counter()
    for p in params.values():
        p.grad = None
    means2D.grad = None
    img, radii = rast(means2D=means2D, **params)
    b = time.perf_counter()
    img.backward(dL)
    c = time.perf_counter()
    if i >= 50:
        rec.append((c - a, b - a, c - b, i))
torch.cuda.synchronize()
tot = sorted(r[0] for r in rec)
print(f"{N} steps: median {1e3 * tot[N // 2]:.3f} ms, mean {1e3 * sum(tot) / N:.3f} ms, p99 {1e3 * tot[int(N * 0.99)]:.3f}, max {1e3 * tot[-1]:.3f}")
slow = [r for r in rec if r[0] > 2 * tot[N // 2]]
print(f"{len(slow)} steps over twice the median:")
for r in slow[:20]:
    print(f"  step {r[3]}: {1e3 * r[0]:.2f} ms (forward call {1e3 * r[1]:.2f}, backward call {1e3 * r[2]:.2f})")
