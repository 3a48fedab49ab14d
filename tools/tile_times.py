"""Per-tile durations of blend_backward_kernel at cfg1 (developer build with -DSCR_TILE_TIMING, see blend.hip): is the
kernel's time the sum of its tiles' work, or the tail of the last, longest tiles?
build:  cd splatco_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize \
        -DSCR_TILE_TIMING -c blend.hip -o exp/blend_timing.o && hipcc --offload-arch=gfx950 -shared -fPIC preprocess.o \
        binning.o exp/blend_timing.o expand.o triplane.o ssim.o densify.o mlp_heads.o anchor_gather.o capi.o -o exp/lib_timing.so
run:    SPLATCO_RASTER_LIB=$PWD/splatco_amd/csrc/exp/lib_timing.so python tools/tile_times.py > profiles/<round>_tile_times.txt"""
import ctypes as C
import math
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from splatco_amd import _C, rasterizer as R
from splatco_amd.synthetic import synthetic_camera, synthetic_gaussians

P, W, H = 1_000_000, 1920, 1080
dev = torch.device("cuda:0")
cam, g = synthetic_camera(W, H), synthetic_gaussians(P, W, H, 0)
rs = R.GaussianRasterizationSettings(H, W, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), torch.tensor(g["bg"], device=dev), 1.0,
                                     cam.world_view_transform.to(dev), cam.full_proj_transform.to(dev), 1,
                                     cam.camera_center.to(dev), False, False)
t = lambda a: torch.tensor(a, device=dev, requires_grad=True)
m, o, c, s, r = t(g["means3D"]), t(g["opacities"]), t(g["colors"]), t(g["scales"]), t(g["rotations"])
m2 = torch.zeros(P, 3, device=dev, requires_grad=True)
dL = torch.randn(3, H, W, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
rast = R.GaussianRasterizer(rs)
for it in range(3):
    for p in (m, o, c, s, r, m2):
        p.grad = None
    img, radii = rast(means3D=m, means2D=m2, opacities=o, colors_precomp=c, scales=s, rotations=r)
    img.backward(dL)
torch.cuda.synchronize()
tiles = ((W + 15) // 16) * ((H + 15) // 16)
buf = (C.c_ulonglong * (2 * tiles))()
assert _C.lib.scr_debug_tile_ticks(buf, tiles) == 0
tk = np.frombuffer(buf, dtype=np.uint64).reshape(tiles, 2).astype(np.int64)
ok = tk[:, 1] > tk[:, 0]
start, end = tk[ok, 0], tk[ok, 1]
dur = (end - start) / 100.0          # wall_clock64: 100 MHz -> microseconds
span = (end.max() - start.min()) / 100.0
print(f"# blend_backward_kernel, cfg1 (1M Gaussians, 1920x1080): {ok.sum()} tiles with work, kernel span {span:.1f} us")
print(f"# per-tile duration (us): mean {dur.mean():.1f}  median {np.median(dur):.1f}  p90 {np.percentile(dur, 90):.1f}  "
      f"p99 {np.percentile(dur, 99):.1f}  max {dur.max():.1f}")
conc = dur.sum() / span
print(f"# sum of tile durations / span = {conc:.0f} workgroups resident on average (256 CUs x 5 = 1280 slots)")
rel = (end - start.min()) / 100.0
for q in (50, 90, 99, 99.9, 100):
    print(f"# {q:5.1f} % of the tiles have finished after {np.percentile(rel, q):7.1f} us ({np.percentile(rel, q) / span:.1%} of the span)")
# how long the machine runs below half occupancy at the end: time from the moment fewer than 640 tiles remain
order = np.sort(rel)
tail_start = order[-640] if len(order) > 640 else 0.0
print(f"# the last 640 tiles (half of the resident slots) finish in the final {span - tail_start:.1f} us = {(span - tail_start) / span:.1%} of the span")
hist, edges = np.histogram(dur, bins=12)
for hcount, lo_, hi_ in zip(hist, edges[:-1], edges[1:]):
    print(f"{lo_:7.1f} .. {hi_:7.1f} us  {hcount:5d}  " + "#" * int(60 * hcount / hist.max()))
