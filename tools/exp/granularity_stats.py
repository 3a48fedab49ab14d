"""Developer probe (CPU, oracle): (wave, splat) iteration counts of the blend backward for sub-quadrant granularities
-- how many list entries can reach an 8x8 quadrant / 8x4 half / 8x2 strip / 4x4 block at all (exact: some pixel passes
the alpha test before that pixel's last contributor), and what lock-stepping the sub-lists of one wave costs (sum over
quadrants of the LONGEST sub-list).  cfg1 scene, a sample of tiles."""
import sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from oracle import raster_oracle as orc
from util import oracle_settings
from splatco_amd.synthetic import synthetic_camera, synthetic_gaussians

W, H, P = 1920, 1080, 1_000_000
cam, g = synthetic_camera(W, H), synthetic_gaussians(P, W, H, 0)
st = oracle_settings(orc, cam, g["bg"])
orc.use_threads(True)
f = orc.forward(st, g["means3D"], g["opacities"], g["scales"], g["rotations"], colors_precomp=g["colors"])
gx = (W + 15) // 16
rng = np.random.default_rng(0)
tiles = rng.choice(f["ranges"].shape[0], 150, replace=False)
tot = dict(entries=0, quad=0, half=0, half_max=0, strip=0, strip_max=0, blk=0, blk_max=0, lanes=0)
for t in tiles:
    lo, hi = f["ranges"][t]
    n = int(hi) - int(lo)
    if n == 0:
        continue
    ids = f["point_list"][lo:hi]
    tx, ty = t % gx, t // gx
    px = (tx * 16 + np.arange(16))[None, :].repeat(16, 0)
    py = (ty * 16 + np.arange(16))[:, None].repeat(16, 1)
    inside = (px < W) & (py < H)
    xy, co = f["xy"][ids], f["conic_opacity"][ids]
    dx = xy[:, 0, None, None] - px[None]
    dy = xy[:, 1, None, None] - py[None]
    power = -0.5 * (co[:, 0, None, None] * dx * dx + co[:, 2, None, None] * dy * dy) - co[:, 1, None, None] * dx * dy
    alpha = np.minimum(0.99, co[:, 3, None, None] * np.exp(power))
    last = np.zeros((16, 16), np.int64)
    last[inside] = f["n_contrib"][py[inside], px[inside]]
    hit = (power <= 0) & (alpha >= 1 / 255) & (np.arange(n)[:, None, None] < last[None]) & inside[None]   # [n,16,16]
    tot["entries"] += n
    for qy in range(2):
        for qx in range(2):
            q = hit[:, qy * 8:qy * 8 + 8, qx * 8:qx * 8 + 8]               # [n,8,8]
            anyq = q.any(axis=(1, 2))
            tot["quad"] += anyq.sum()
            tot["lanes"] += q.sum()
            halves = [q[:, 0:4].any(axis=(1, 2)), q[:, 4:8].any(axis=(1, 2))]
            tot["half"] += sum(h.sum() for h in halves); tot["half_max"] += max(h.sum() for h in halves)
            strips = [q[:, 2 * k:2 * k + 2].any(axis=(1, 2)) for k in range(4)]
            tot["strip"] += sum(s.sum() for s in strips); tot["strip_max"] += max(s.sum() for s in strips)
            blks = [q[:, 4 * a:4 * a + 4, 4 * b:4 * b + 4].any(axis=(1, 2)) for a in range(2) for b in range(2)]
            tot["blk"] += sum(s.sum() for s in blks); tot["blk_max"] += max(s.sum() for s in blks)
q = tot["quad"]
print(f"list entries {tot['entries']}; (quadrant, entry) pairs that touch a pixel {q} ({q / tot['entries']:.2f} per entry), "
      f"lanes hit per pair {tot['lanes'] / q:.1f} of 64")
print(f"8x4 halves : {tot['half'] / q:.2f} sub-entries per pair, iterations with 2 lock-stepped sub-lists {tot['half_max'] / q:.3f} of the pairs")
print(f"8x2 strips : {tot['strip'] / q:.2f} sub-entries per pair, iterations with 4 lock-stepped sub-lists {tot['strip_max'] / q:.3f}")
print(f"4x4 blocks : {tot['blk'] / q:.2f} sub-entries per pair, iterations with 4 lock-stepped sub-lists {tot['blk_max'] / q:.3f}")
