"""Developer probe: accuracy of the blend-backward moments (before the projection chain) for one stress scene.
Sums the device's per-instance gradient records per Gaussian (fp64 on the host) and compares sxx/sxy/syy/sx/sy with the
fp64 oracle's blend backward on the same fp32 records; fp32 oracle alongside.  usage: moment_check.py seed index"""
import ctypes as C
import sys
import numpy as np
import torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from oracle import raster_oracle as orc
from util import stress_scene, oracle_settings
from splatco_amd import _C, rasterizer as R
import test_gpu_parity as T

seed, idx = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
for it in range(idx + 1):
    cam, g, sm = stress_scene(rng)
    dL = rng.standard_normal((3, cam.image_height, cam.image_width)).astype(np.float32)
st = oracle_settings(orc, cam, g["bg"], scale_modifier=sm)
f = orc.forward(st, g["means3D"], g["opacities"], g["scales"], g["rotations"], colors_precomp=g["colors"])
f64 = T._fp64_on_fp32_records(orc, st, f)
bad = f64["n_contrib"] != f["n_contrib"]
dev = torch.device("cuda:0")
rs = T._settings(cam, g["bg"], sm)
cs = R._CSettings(rs)
t = lambda a: torch.tensor(a, device=dev)
m, o, s, r, c = t(g["means3D"]), t(g["opacities"]), t(g["scales"]), t(g["rotations"]), t(g["colors"])
color, radii, stt = R.rasterize_forward(cs, m, o, s, r, None, None, c)
nc = stt.debug(_C.DBG_N_CONTRIB).cpu().numpy().view(np.uint32)
bad |= nc != f["n_contrib"]
dL[:, bad] = 0
print("excused pixels", bad.sum())
P, I = stt.P, stt.I
scratch = torch.zeros(_C.lib.scr_backward_scratch_bytes(I), dtype=torch.uint8, device=dev)
outs = [torch.empty(P, w, device=dev) for w in (3, 3, 3, 1, 3, 4)]
_C.check(_C.lib.scr_backward(P, 0, I, stt.flags, m.data_ptr(), s.data_ptr(), r.data_ptr(), None, None, cs.ref(), stt.radii.data_ptr(),
                             stt.geom.data_ptr(), stt.binning.data_ptr(), stt.image.data_ptr(), t(dL).data_ptr(), scratch.data_ptr(),
                             outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr(), None, outs[3].data_ptr(), outs[4].data_ptr(),
                             outs[5].data_ptr(), None, R._stream()))
torch.cuda.synchronize()
rec = scratch.cpu().numpy().view(np.float32)[:I * 9].reshape(I, 9).astype(np.float64)
tt = f["tiles_touched"].astype(np.int64)
off = np.concatenate([[0], np.cumsum(tt)])[:-1]
vis = tt > 0
dev_m = np.zeros((P, 9))
dev_m[vis] = np.add.reduceat(rec[:, :9], off[vis], axis=0)
b32 = orc.blend_backward(st, f, dL)
b64 = orc.blend_backward(st, f64, dL, f64=True)
op = f["conic_opacity"][:, 3].astype(np.float64)
def moments(b):   # (sx, sy, sxx, sxy, syy) from the oracle's screen-space sums is not direct for sx, sy; use conic only
    g_m2, g_conic, g_op, g_col = b
    return np.stack([-2 * g_conic[:, 0] / op, -g_conic[:, 1] / op, -2 * g_conic[:, 2] / op, g_op.reshape(-1)], 1)
m32, m64 = moments([x.astype(np.float64) for x in b32]), moments(b64)
md = np.stack([dev_m[:, 2], dev_m[:, 3], dev_m[:, 4], dev_m[:, 5]], 1)
for k, name in enumerate(["sxx", "sxy", "syy", "sum Y"]):
    ref = m64[vis, k]
    ed, eo = np.abs(md[vis, k] - ref), np.abs(m32[vis, k] - ref)
    sc = np.abs(ref) + 1e-300
    print(f"{name}: rel-L2 device {np.linalg.norm(ed) / np.linalg.norm(ref):.2e} oracle32 {np.linalg.norm(eo) / np.linalg.norm(ref):.2e}; "
          f"median row-rel device {np.median(ed / sc):.2e} oracle32 {np.median(eo / sc):.2e}; max row-rel device {np.max(ed / sc):.2e} oracle32 {np.max(eo / sc):.2e}")
ids = np.nonzero(vis)[0]
worst = ids[np.argsort(-(np.abs(md[vis, 0] - m64[vis, 0]) / (np.abs(m64[vis, 0]) + 1e-300)))[:4]]
for i in worst:
    print("id", i, "radius", f["radii"][i], "tiles", tt[i], "mean px", f["xy"][i], "sxx dev/o32/f64", md[i, 0], m32[i, 0], m64[i, 0], "sumY", md[i, 3], m64[i, 3])
gm2 = b64[0]
for i in [int(a) for a in sys.argv[3:]]:
    print("== id", i, "radius", f["radii"][i], "tiles", tt[i], "mean px", f["xy"][i], "op", op[i])
    print("   device moments (sx sy sxx sxy syy sumY c0 c1 c2):", dev_m[i])
    print("   f64: sxx sxy syy sumY", m64[i], " o32:", m32[i])
    print("   f64 mean2D grad px", gm2[i], " device means3D grad", outs[0][i].cpu().numpy())
# ---- chain sensitivity: the fp32 oracle chain on (a) the device's moments, (b) the fp32 oracle's own, (c) jiggled
def chain32(mom9):      # mom9 [P,9] -> oracle fp32 preprocess_backward outputs
    gm2 = np.zeros((P, 2)); gcon = np.zeros((P, 3))
    rec_ = f  # fp32 forward dict
    co = f["conic_opacity"].astype(np.float64)
    # invert the device's factoring: gmx = op (2 cA sx + cB sy) with cA = -Qxx/2, cB = -Qxy, cC = -Qyy/2
    cA, cB, cC = -0.5 * co[:, 0], -co[:, 1], -0.5 * co[:, 2]
    gm2[:, 0] = op * (2 * cA * mom9[:, 0] + cB * mom9[:, 1]); gm2[:, 1] = op * (2 * cC * mom9[:, 1] + cB * mom9[:, 0])
    gcon[:, 0] = -0.5 * op * mom9[:, 2]; gcon[:, 1] = -op * mom9[:, 3]; gcon[:, 2] = -0.5 * op * mom9[:, 4]
    return orc.preprocess_backward(st, f, gm2.astype(np.float32), gcon.astype(np.float32), np.zeros((P, 3), np.float32),
                                   g["means3D"], g["scales"], g["rotations"])
ref9 = np.zeros((P, 9)); 
o32c = chain32(dev_m)
for i in [int(a) for a in sys.argv[3:]]:
    print("   fp32 oracle chain on the device's moments ->", o32c["means3D"][i])
    pr = np.random.default_rng(1)
    for trial in range(3):
        jm = dev_m * (1 + 2.0 ** -23 * pr.choice([-1.0, 1.0], dev_m.shape))
        print("   ... on 2-ulp jiggled moments ->", chain32(jm)["means3D"][i])
