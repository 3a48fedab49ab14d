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


def synthetic_gaussians(P, width, height, seed=0, fovx_deg=60.0, sigma_scale=1.0, scene="uniform"):
    """Returns float32 numpy arrays: means3D[P,3], scales[P,3], rotations[P,4], opacities[P,1],
    colors[P,3] and bg[3].  sigma_scale: multiplies every screen-space sigma (1 = the benchmark scene of SURVEY.md 8d;
    smaller = the same Gaussians, same positions, sparser tile lists -- bench.py's --sigma-scale sweep).
    scene: "uniform" (the benchmark scene: pixel positions uniform over the image) or "clustered" (a developer scene, same
    draws: 80 % of the Gaussians moved into the central eighth of the image -- a quarter of its width x half of its height --
    so that tile lists differ by an order of magnitude: what a real capture looks like to the tile scheduler and the sort)."""
    rng = np.random.default_rng(seed)
    tanfovx = math.tan(math.radians(fovx_deg) / 2)
    tanfovy = tanfovx * height / width
    f = width / (2.0 * tanfovx)
    px = rng.uniform(0.0, width, P)
    py = rng.uniform(0.0, height, P)
    if scene == "clustered":
        inner = np.arange(P) % 5 != 0          # 80 %, decided by the index: the other draws stay those of the uniform scene
        px = np.where(inner, width * 0.375 + px * 0.25, px)
        py = np.where(inner, height * 0.25 + py * 0.5, py)
    elif scene != "uniform":
        raise ValueError(f"unknown scene {scene!r}")
    z = rng.uniform(2.0, 20.0, P)
    sig = np.exp(rng.uniform(math.log(0.5), math.log(5.0), (P, 3))) * float(sigma_scale)
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


# ---- anchor scenes of BASELINE.json configs[2..4]
ANCHOR_CONFIGS = {
    # name: (anchors, views = mv, seed)
    "cfg2": (5_000_000, 1, 1),
    "cfg3": (5_000_000, 4, 2),
    "cfg4": (20_000_000, 8, 3),
}


def synthetic_anchor_model(N, seed, device, plane_size=2800, num_channels=15, activate_level=2, n_offsets=10):
    """N anchors uniform in [-2,2]^3 (the tri-plane's fixed box, scene/gaussian_model.py:185), k = 10 offsets,
    32 features, tri-plane features at `plane_size` with levels 0..activate_level active, seeded MLP / plane
    weights, plane noise off (Q0 = 0, as render.py:79).  Everything is drawn from torch generators seeded with
    `seed` (weights on the host, the per-anchor tensors on `device`)."""
    import torch
    from .scene_model import AnchorGaussianModel
    torch.manual_seed(seed)                                   # module initialisers (host generator)
    pc = AnchorGaussianModel(plane_size=plane_size, num_channels=num_channels, n_offsets=n_offsets).to(device)
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    r = lambda *s: torch.rand(*s, device=device, generator=g)
    n = lambda *s: torch.randn(*s, device=device, generator=g)
    pc.set_anchors(r(N, 3) * 4 - 2, n(N, n_offsets, 3) * 0.5, n(N, 32) * 0.5, n(N, 6) * 0.3 - 5.0)
    pc.feat_planes.Q0 = 0
    pc.feat_planes._feat.activate_level = activate_level
    pc.train()
    return pc


def synthetic_views(n, width=1920, height=1080, fovx_deg=60.0):
    """n cameras outside the [-2,2]^3 box looking at its centre (view i is shifted sideways by 0.3 i)."""
    from .cameras import look_at_camera
    return [look_at_camera((0.5 + 0.3 * i, -0.4, -6.0), (0.0, 0.0, 0.0), (0.0, -1.0, 0.0), math.radians(fovx_deg),
                           width, height, uid=i) for i in range(n)]
