"""Developer probe: how many (wave, splat) iterations of the forward blend touch no pixel at all
(needs the instrumented library built from csrc/exp/blend_stats.hip)."""
import ctypes, math, os, sys
import torch
sys.path.insert(0, ".")
os.environ["SPLATCO_RASTER_LIB"] = os.path.abspath("splatco_amd/csrc/exp/lib_stats.so")
from splatco_amd import _C, rasterizer as R
from splatco_amd.synthetic import synthetic_camera, synthetic_gaussians
P, W, H = 1000000, 1920, 1080
d = torch.device("cuda:0")
cam, g = synthetic_camera(W, H), synthetic_gaussians(P, W, H, 0)
rs = R.GaussianRasterizationSettings(H, W, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), torch.tensor(g["bg"], device=d), 1.0,
                                     cam.world_view_transform.to(d), cam.full_proj_transform.to(d), 1, cam.camera_center.to(d), False, False)
t = lambda a: torch.tensor(a, device=d)
rast = R.GaussianRasterizer(rs)
out = (ctypes.c_ulonglong * 8)()
_C.lib.scr_debug_stats(out, 1)
img, radii = rast(means3D=t(g["means3D"]), means2D=torch.zeros(P, 3, device=d), opacities=t(g["opacities"]), colors_precomp=t(g["colors"]), scales=t(g["scales"]), rotations=t(g["rotations"]))
torch.cuda.synchronize()
_C.lib.scr_debug_stats(out, 0)
it = out[0]
print(f"(wave,splat) iterations {it}; top/bottom halves: sum of max {out[1]} ({100*out[1]/it:.1f} %), half-entries {out[2]} ({out[2]/it:.2f} per entry); "
      f"left/right halves: sum of max {out[3]} ({100*out[3]/it:.1f} %), half-entries {out[4]} ({out[4]/it:.2f} per entry)")
