"""Developer timing probe (not the benchmark contract -- see bench.py): per-stage wall times of the
rasterizer on the synthetic configs, with stream events."""
import math
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from splatco_amd import rasterizer as R
from splatco_amd.synthetic import synthetic_camera, synthetic_gaussians


def main(P=1_000_000, W=1920, H=1080, iters=10):
    d = torch.device("cuda:0")
    cam, g = synthetic_camera(W, H), synthetic_gaussians(P, W, H, 0)
    rs = R.GaussianRasterizationSettings(H, W, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2),
                                         torch.tensor(g["bg"], device=d), 1.0, cam.world_view_transform.to(d),
                                         cam.full_proj_transform.to(d), 1, cam.camera_center.to(d), False, False)
    t = lambda a: torch.tensor(a, device=d, requires_grad=True)
    m, o, s, r, c = t(g["means3D"]), t(g["opacities"]), t(g["scales"]), t(g["rotations"]), t(g["colors"])
    rast = R.GaussianRasterizer(rs)
    dL = torch.randn(3, H, W, device=d)
    m2d = torch.zeros_like(m, requires_grad=True)
    for it in range(iters + 3):
        if it == 3:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            tf = tb = 0.0
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        img, radii = rast(means3D=m, means2D=m2d, opacities=o, colors_precomp=c, scales=s, rotations=r)
        e1.record()
        img.backward(dL)
        e2.record()
        torch.cuda.synchronize()
        if it >= 3:
            tf += e0.elapsed_time(e1)
            tb += e1.elapsed_time(e2)
    dt = (time.perf_counter() - t0) / iters
    print(f"P={P} {W}x{H}: fwd {tf/iters:.3f} ms  bwd {tb/iters:.3f} ms  wall/iter {dt*1e3:.3f} ms "
          f"-> {P/(tf+tb)*iters/1e3:.1f} Msplats/s (events)  visible={(radii>0).sum().item()}")


if __name__ == "__main__":
    main(*(int(a) for a in sys.argv[1:]))
