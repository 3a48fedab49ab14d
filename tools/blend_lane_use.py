"""Developer tool (GPU box): how many of the 64 lanes of every staged (wave, splat) pair of the blend kernels do useful
work?  Needs the counting variant of blend.hip:
    tools/build_variant.sh count blend.hip -DSCR_BLEND_COUNT
    SPLATCO_RASTER_LIB=$PWD/splatco_amd/csrc/exp/libvar_count.so python tools/blend_lane_use.py [sigma_scale ...]
Writes one table per scene (BASELINE.json configs[1]: 1 M Gaussians, 1920x1080; sigma_scale shrinks every splat, the
sparse sweep of profiles/r04_sparse_sweep.txt)."""
import ctypes
import math
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from splatco_amd import _C
from splatco_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
from splatco_amd.synthetic import synthetic_camera, synthetic_gaussians


def run(sigma_scale, P=1_000_000, W=1920, H=1080):
    dev = torch.device("cuda:0")
    cam, g = synthetic_camera(W, H), synthetic_gaussians(P, W, H, seed=0, sigma_scale=sigma_scale)
    tx, ty = math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5)
    rs = GaussianRasterizationSettings(H, W, tx, ty, torch.tensor(g["bg"], device=dev), 1.0, cam.world_view_transform.to(dev),
                                       cam.full_proj_transform.to(dev), 1, cam.camera_center.to(dev), False, False)
    t = lambda a: torch.tensor(a, device=dev, requires_grad=True)
    m, o, c, s, r = t(g["means3D"]), t(g["opacities"]), t(g["colors"]), t(g["scales"]), t(g["rotations"])
    m2d = torch.zeros(P, 3, device=dev, requires_grad=True)
    fn = _C.lib.scr_tool_blend_counters
    fn.argtypes, fn.restype = [ctypes.c_void_p, ctypes.c_int], ctypes.c_int
    buf = (ctypes.c_ulonglong * 16)()
    torch.cuda.synchronize()
    assert fn(buf, 1) == 0
    img, radii = GaussianRasterizer(rs)(means3D=m, means2D=m2d, opacities=o, colors_precomp=c, scales=s, rotations=r)
    dL = torch.randn(3, H, W, device=dev, generator=torch.Generator(device=dev).manual_seed(0))
    (img * dL).sum().backward()
    torch.cuda.synchronize()
    assert fn(buf, 0) == 0
    k = [int(x) for x in buf]
    I = int(img.grad_fn.state.I)
    f_pairs, f_live, f_done, f_skip = k[0], k[1], k[2], k[3]
    b_pairs, b_hit, b_past, b_geo, b_skip, b_tail = k[8], k[9], k[10], k[11], k[12], k[13]
    f_sub, f_submax, b_sub, b_submax = k[4], k[5], k[14], k[15]
    f_chunkmax, b_chunkmax = k[6], k[7]
    lines = [f"scene: P = {P}, {W}x{H}, sigma_scale = {sigma_scale}: {I} (Gaussian, tile) instances, {int((radii > 0).sum())} visible",
             "forward  (blend_forward_kernel, one wave per 8x8 quadrant, 64 lanes per staged splat):",
             f"  staged (wave, splat) pairs            {f_pairs:>14,d}   = {f_pairs / max(I, 1):.2f} per instance (of 4 quadrants)",
             f"  lane slots                            {64 * f_pairs:>14,d}",
             f"  lanes that blend the splat            {f_live:>14,d}   = {f_live / max(64 * f_pairs, 1):.1%} of the lane slots",
             f"  lanes whose pixel is already finished {f_done:>14,d}   = {f_done / max(64 * f_pairs, 1):.1%}",
             f"  pairs in groups skipped whole         {f_skip:>14,d}   = {f_skip / max(f_pairs, 1):.1%} of the pairs",
             "backward (blend_backward_kernel, one wave per quadrant, four splats per reduction):",
             f"  staged (wave, splat) pairs            {b_pairs:>14,d}   = {b_pairs / max(I, 1):.2f} per instance",
             f"  lane slots                            {64 * b_pairs:>14,d}",
             f"  lanes that contribute                 {b_hit:>14,d}   = {b_hit / max(64 * b_pairs, 1):.1%} of the lane slots",
             f"  lanes inside the alpha>=1/255 ellipse {b_geo:>14,d}   = {b_geo / max(64 * b_pairs, 1):.1%}",
             f"  lanes behind the pixel's last contrib {b_past:>14,d}   = {b_past / max(64 * b_pairs, 1):.1%}",
             f"  pairs in groups without any hit       {b_skip:>14,d}   = {b_skip / max(b_pairs, 1):.1%} of the pairs (reduction skipped)",
             f"  empty slots of partial groups         {b_tail:>14,d}   = {b_tail / max(b_pairs + b_tail, 1):.1%} of the group slots"]
    lines += ["4x4-pixel sub-blocks (four per wave, each with its own list: what 16-bit sub-block masks could pack):",
              f"  forward : (pair, sub-block) combinations at work {f_sub:>14,d} = {f_sub / max(4 * f_pairs, 1):.1%} of 4 per pair; "
              f"lanes at work inside them {f_live / max(16 * f_sub, 1):.1%}; rounds if every wave walked its longest sub-block list: "
              f"{f_submax:,d} = {f_submax / max(f_pairs, 1):.2f} of today's; in step chunk by chunk (64 list entries): {f_chunkmax:,d} = {f_chunkmax / max(f_pairs, 1):.2f}",
              f"  backward: (pair, sub-block) combinations at work {b_sub:>14,d} = {b_sub / max(4 * b_pairs, 1):.1%} of 4 per pair; "
              f"lanes at work inside them {b_hit / max(16 * b_sub, 1):.1%}; rounds if every wave walked its longest sub-block list: "
              f"{b_submax:,d} = {b_submax / max(b_pairs, 1):.2f} of today's; in step round by round (64 list entries): {b_chunkmax:,d} = {b_chunkmax / max(b_pairs, 1):.2f}"]
    return "\n".join(lines)


if __name__ == "__main__":
    scales = [float(a) for a in sys.argv[1:]] or [1.0]
    for sc in scales:
        print(run(sc))
        print()
