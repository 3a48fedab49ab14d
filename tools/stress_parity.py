"""Developer stress run (GPU box): randomised scenes / image sizes / cameras / anisotropy against the
CPU oracle -- integers bit-exact, n_contrib exact away from thresholds, image 1e-4, grads 2e-4."""
import math
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from oracle import raster_oracle as orc
from splatco_amd.cameras import look_at_camera
from test_gpu_parity import _check_forward, _check_grads, _run_gpu
from util import oracle_settings, rel_l2


def scene(rng):
    W, H = int(rng.integers(17, 700)), int(rng.integers(17, 500))
    P = int(rng.integers(1, 30000))
    eye = rng.uniform(-1, 1, 3) + np.array([0, 0, -rng.uniform(2.5, 7)])
    cam = look_at_camera(eye, rng.uniform(-0.3, 0.3, 3), (rng.uniform(-0.2, 0.2), -1.0, rng.uniform(-0.2, 0.2)),
                         math.radians(rng.uniform(30, 100)), W, H)
    spread = rng.uniform(0.3, 3.0)
    means = rng.normal(0, spread, (P, 3))
    if rng.random() < 0.3:                       # a dense clump -> large tiles / merge passes
        k = P // 2
        means[:k] = rng.normal(0, 0.03, (k, 3)) + rng.uniform(-0.5, 0.5, 3)
    smax = rng.choice([0.02, 0.1, 0.6, 3.0])
    scales = np.exp(rng.uniform(math.log(smax / 300), math.log(smax), (P, 3)))   # extreme anisotropy included
    q = rng.standard_normal((P, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    q *= rng.uniform(0.7, 1.3, (P, 1))
    op = rng.uniform(0.0, 1.0, (P, 1)) ** rng.choice([0.3, 1.0, 3.0])
    col = rng.uniform(0, 1, (P, 3))
    f = np.float32
    g = dict(means3D=means.astype(f), scales=scales.astype(f), rotations=q.astype(f), opacities=op.astype(f),
             colors=col.astype(f), bg=rng.uniform(0, 1, 3).astype(f))
    return cam, g, float(rng.choice([0.5, 1.0, 1.7]))


def main(n=40, seed=0):
    rng = np.random.default_rng(seed)
    bad = 0
    for it in range(n):
        cam, g, sm = scene(rng)
        st = oracle_settings(orc, cam, g["bg"], scale_modifier=sm)
        f = orc.forward(st, g["means3D"], g["opacities"], g["scales"], g["rotations"], colors_precomp=g["colors"])
        dL = rng.standard_normal((3, cam.image_height, cam.image_width)).astype(np.float32)
        b = orc.backward(st, f, dL, g["means3D"], g["scales"], g["rotations"], colors_precomp=g["colors"])
        f64 = orc.forward(st, g["means3D"], g["opacities"], g["scales"], g["rotations"], colors_precomp=g["colors"], f64=True)
        b64 = orc.backward(st, f64, dL, g["means3D"], g["scales"], g["rotations"], colors_precomp=g["colors"], f64=True)
        try:
            o = _run_gpu(cam, g, scale_modifier=sm, dL=dL)
            _check_forward(f, o, st)
            if np.array_equal(o["n_contrib"], f["n_contrib"]) and np.array_equal(f["n_contrib"], f64["n_contrib"]) \
                    and np.array_equal(f["point_list"], f64["point_list"]):
                # ill-conditioned scenes: the fp32 oracle itself drifts from fp64; the device result
                # must be as close to fp64 as the fp32 oracle is (x3) or within the 1e-4 bar
                for k in ("means3D", "means2D", "colors_precomp", "opacities", "scales", "rotations"):
                    e_gpu, e_o32 = rel_l2(o["grads"][k], b64[k]), rel_l2(b[k], b64[k])
                    assert e_gpu <= 3.0 * e_o32 + 1e-4, (k, e_gpu, e_o32)
            status = "ok"
        except AssertionError as e:
            bad += 1
            status = "FAIL " + str(e)[:200]
        ranges = f["ranges"]
        print(f"[{it}] P={g['means3D'].shape[0]} {cam.image_width}x{cam.image_height} I={f['num_rendered']} "
              f"max tile={(ranges[:, 1].astype(np.int64) - ranges[:, 0]).max()} vis={(f['radii'] > 0).sum()} {status}", flush=True)
    print("failures:", bad)
    return bad


if __name__ == "__main__":
    sys.exit(1 if main(*(int(a) for a in sys.argv[1:])) else 0)
