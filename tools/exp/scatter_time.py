"""cfg1 forward only, per-kernel times through the library's profile scopes (for experiment builds whose results are wrong)."""
import math, sys
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
t = lambda a: torch.tensor(a, device=dev)
m, o, c, s, r = t(g["means3D"]), t(g["opacities"]), t(g["colors"]), t(g["scales"]), t(g["rotations"])
rast = R.GaussianRasterizer(rs)
with torch.no_grad():
    for it in range(3):
        rast(means3D=m, means2D=torch.zeros(P, 3, device=dev), opacities=o, colors_precomp=c, scales=s, rotations=r)
    torch.cuda.synchronize()
    _C.profile_enable(True); _C.profile_read()
    for it in range(10):
        rast(means3D=m, means2D=torch.zeros(P, 3, device=dev), opacities=o, colors_precomp=c, scales=s, rotations=r)
    torch.cuda.synchronize()
    print({k: round(ms / n, 4) for k, (ms, n) in _C.profile_read().items() if n})
import ctypes as C
if hasattr(_C.lib, "scr_debug_sc_ticks"):
    buf = (C.c_ulonglong * 8)()
    _C.lib.scr_debug_sc_ticks(buf, 1)
    with torch.no_grad():
        rast(means3D=m, means2D=torch.zeros(P, 3, device=dev), opacities=o, colors_precomp=c, scales=s, rotations=r)
    torch.cuda.synchronize()
    _C.lib.scr_debug_sc_ticks(buf, 1)
    n = buf[7]
    print("scatter phases, microseconds per workgroup:", {k: round(buf[i] / 100.0 / n, 1) for i, k in enumerate(("zero hist", "scan+loads+count", "reserve", "place"))}, "workgroups", n)
