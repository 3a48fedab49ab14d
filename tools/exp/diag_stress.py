"""Developer probe: one stress scene (seed, index) -- per-tensor gradient errors of GPU / fp32 oracle vs fp64 oracle,
and the Gaussians that carry the discrepancy."""
import sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests"); sys.path.insert(0, "tools")
from oracle import raster_oracle as orc
from util import stress_scene as scene
from test_gpu_parity import _run_gpu
from util import oracle_settings, rel_l2
seed, idx = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
for it in range(idx + 1):
    cam, g, sm = scene(rng)
    dL = rng.standard_normal((3, cam.image_height, cam.image_width)).astype(np.float32)
st = oracle_settings(orc, cam, g["bg"], scale_modifier=sm)
args = (g["means3D"], g["opacities"], g["scales"], g["rotations"])
f = orc.forward(st, *args, colors_precomp=g["colors"]); b = orc.backward(st, f, dL, g["means3D"], g["scales"], g["rotations"], colors_precomp=g["colors"])
f64 = orc.forward(st, *args, colors_precomp=g["colors"], f64=True); b64 = orc.backward(st, f64, dL, g["means3D"], g["scales"], g["rotations"], colors_precomp=g["colors"], f64=True)
o = _run_gpu(cam, g, scale_modifier=sm, dL=dL)
print("scale_modifier", sm, "P", len(g["means3D"]), "n_contrib equal", np.array_equal(o["n_contrib"], f["n_contrib"]))
for k in ("means3D", "means2D", "colors_precomp", "opacities", "scales", "rotations"):
    print(f"{k:15s} gpu-vs-f64 {rel_l2(o['grads'][k], b64[k]):.3e}  o32-vs-f64 {rel_l2(b[k], b64[k]):.3e}  gpu-vs-o32 {rel_l2(o['grads'][k], b[k]):.3e}")
for k in (sys.argv[3:] or ["means3D", "scales"]):
    eg = np.abs(o["grads"][k] - b64[k]).reshape(len(b64[k]), -1).sum(1); eo = np.abs(b[k] - b64[k]).reshape(len(b64[k]), -1).sum(1)
    top = np.argsort(-eg)[:5]
    print(k, "norm", np.abs(b64[k]).sum(), "top offenders:")
    for i in top:
        s = g["scales"][i] * sm
        print(f"  id {i} err_gpu {eg[i]:.3e} err_o32 {eo[i]:.3e} |g64| {np.abs(b64[k][i]).sum():.3e} gpu {o['grads'][k][i]} o32 {b[k][i]} f64 {b64[k][i]} scales {s} aniso {s.max()/s.min():.0f} radius {f['radii'][i]} tiles {f['tiles_touched'][i]} op {g['opacities'][i,0]:.3f} m2d gpu {o['grads']['means2D'][i]} o32 {b['means2D'][i]} f64 {b64['means2D'][i]}")
