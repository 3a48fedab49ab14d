"""Developer probe: per-kernel HIP-event times of the rasterizer at a given synthetic size."""
import math, sys
import torch
sys.path.insert(0, ".")
from splatco_amd import _C, rasterizer as R
from splatco_amd.synthetic import synthetic_camera, synthetic_gaussians
P, W, H = (int(a) for a in sys.argv[1:4])
d = torch.device("cuda:0")
cam, g = synthetic_camera(W, H), synthetic_gaussians(P, W, H, 0)
rs = R.GaussianRasterizationSettings(H, W, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), torch.tensor(g["bg"], device=d), 1.0,
                                     cam.world_view_transform.to(d), cam.full_proj_transform.to(d), 1, cam.camera_center.to(d), False, False)
t = lambda a: torch.tensor(a, device=d, requires_grad=True)
m, o, s, r, c = t(g["means3D"]), t(g["opacities"]), t(g["scales"]), t(g["rotations"]), t(g["colors"])
rast = R.GaussianRasterizer(rs)
dL = torch.randn(3, H, W, device=d)
m2d = torch.zeros_like(m, requires_grad=True)
for it in range(6):
    if it == 2:
        torch.cuda.synchronize(); _C.profile_enable(True); _C.profile_read()
    img, radii = rast(means3D=m, means2D=m2d, opacities=o, colors_precomp=c, scales=s, rotations=r)
    img.backward(dL)
torch.cuda.synchronize()
prof = _C.profile_read()
print({k: round(v[0] / v[1], 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0]) if v[1]})
