"""Per-kernel times of the tri-plane forward / backward at cfg2's point count and grid set (uniform random points):
attention grid 700 (6 planes) + plain 700 + plain 1400 into one [V,60] matrix.
usage: [SPLATCO_RASTER_LIB=variant.so] python tools/exp/tp_backward_probe.py [V] [order: random|lex] [shape: uniform|sheet|centre]
sheet: |z| < 0.02 (a city seen from above: two of the three projections collapse onto a strip of tiles);
centre: 80 % of the points in the central [-0.25, 0.25]^3 (a contracted scene)."""
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, ".")
from splatco_amd.triplane import multi_triplane_sample

V = int(sys.argv[1]) if len(sys.argv) > 1 else 4_600_000
order = sys.argv[2] if len(sys.argv) > 2 else "random"
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
ind = torch.rand(V, 3, device=dev, generator=g) * 2 - 1
shape = sys.argv[3] if len(sys.argv) > 3 else "uniform"
if shape == "sheet":
    ind[:, 2] *= 0.02
elif shape == "centre":
    m = torch.rand(V, device=dev, generator=g) < 0.8
    ind[m] *= 0.25
if order == "lex":
    q = ((ind + 1) / 2 * (1 << 20)).long()
    ind = ind[torch.argsort((q[:, 0] << 40) | (q[:, 1] << 20) | q[:, 2])].contiguous()
mk = lambda S, n: [torch.randn(1, 5, S, S, device=dev, generator=g).requires_grad_() for _ in range(n)]
grids = [(mk(700, 6), (0, 10, 20, 5, 15, 25)), (mk(700, 3), (30, 35, 40)), (mk(1400, 3), (45, 50, 55))]
up = torch.randn(V, 60, device=dev, generator=g)


def step():
    for pl, _ in grids:
        for p in pl:
            p.grad = None
    out = multi_triplane_sample([(ind, tuple(pl), cols) for pl, cols in grids])
    out.backward(up)


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(5):
        step()
    torch.cuda.synchronize()
tot = 0.0
for e in sorted(prof.key_averages(), key=lambda e: -e.self_device_time_total):
    if e.self_device_time_total > 0:
        tot += e.self_device_time_total / 5 / 1e3
        print(f"{e.key[:90]:90s} {e.self_device_time_total / 5 / 1e3:8.3f} ms/step  {e.count / 5:5.1f} calls")
print(f"total {tot:.3f} ms/step")
