"""Seeded synthetic inputs of SURVEY.md section 8(d) (the configurations BASELINE.json names).

Camera: identity rotation at the origin looking down +z, FoVx = 60 deg, FoVy from the aspect
ratio, matrices built as scene/cameras.py:54-58.  Gaussians (numpy default_rng(seed), draw
order fixed below): pixel position uniform over the image, depth U[2,20], per-axis screen
sigma logU[0.5,5] px, random unit quaternion, opacity U[0.05,0.95], colour U[0,1]^3, white
background (arguments/__init__.py:62).
"""
import math

import numpy as np

from .cameras import make_camera

CONFIGS = {
    # name: (P, width, height, seed)
    "cfg0_10k_400x400": (10_000, 400, 400, 0),
    "cfg1_1M_1080p": (1_000_000, 1920, 1080, 0),
}


def synthetic_camera(width, height, fovx_deg=60.0):
    FoVx = math.radians(fovx_deg)
    FoVy = 2.0 * math.atan(math.tan(FoVx / 2) * height / width)
    return make_camera(np.eye(3), np.zeros(3), FoVx, FoVy, width, height)


def synthetic_gaussians(P, width, height, seed=0, fovx_deg=60.0):
    """Returns float32 numpy arrays: means3D[P,3], scales[P,3], rotations[P,4], opacities[P,1],
    colors[P,3] and bg[3]."""
    rng = np.random.default_rng(seed)
    tanfovx = math.tan(math.radians(fovx_deg) / 2)
    tanfovy = tanfovx * height / width
    f = width / (2.0 * tanfovx)
    px = rng.uniform(0.0, width, P)
    py = rng.uniform(0.0, height, P)
    z = rng.uniform(2.0, 20.0, P)
    sig = np.exp(rng.uniform(math.log(0.5), math.log(5.0), (P, 3)))
    q = rng.standard_normal((P, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    opac = rng.uniform(0.05, 0.95, (P, 1))
    col = rng.uniform(0.0, 1.0, (P, 3))
    ndc_x = (2.0 * px + 1.0) / width - 1.0
    ndc_y = (2.0 * py + 1.0) / height - 1.0
    means = np.stack([ndc_x * tanfovx * z, ndc_y * tanfovy * z, z], axis=1)
    scales = sig * (z / f)[:, None]
    f32 = np.float32
    return dict(means3D=means.astype(f32), scales=scales.astype(f32), rotations=q.astype(f32),
                opacities=opac.astype(f32), colors=col.astype(f32), bg=np.ones(3, f32))
