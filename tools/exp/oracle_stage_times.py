"""Developer probe: per-stage times of the threaded CPU oracle on cfg1."""
import sys, time, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from oracle import raster_oracle as orc
from splatco_amd.synthetic import synthetic_camera, synthetic_gaussians
from util import oracle_settings
W, H, P = 1920, 1080, 1000000
cam, g = synthetic_camera(W, H), synthetic_gaussians(P, W, H, 0)
st = oracle_settings(orc, cam, g["bg"])
dL = np.random.default_rng(1).standard_normal((3, H, W)).astype(np.float32)
orc.use_threads(True)
print("threads", orc.threads())
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
f = orc.forward(st, g["means3D"], g["opacities"], g["scales"], g["rotations"], colors_precomp=g["colors"])
b = orc.backward(st, f, dL, g["means3D"], g["scales"], g["rotations"], colors_precomp=g["colors"])
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
