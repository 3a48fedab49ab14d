"""Developer probe (GPU): how many list entries can never produce a gradient record, per configuration.

For the view of cfg1 / cfg2 / cfg4 (reduced with --anchors): the fraction of (Gaussian, tile) instances whose quadrant
mask is 0 (the splat's rect touches the tile but no quadrant can reach alpha >= 1/255: known statically, in the forward),
the fraction of instances behind their tile's last contributor (known after the forward blend: n_contrib), and the
fraction of all-zero 36-byte gradient records after a backward pass with a random dL/dpixel (what the blend backward
writes and preprocess_backward reads back for nothing).  Prints one line per configuration.

usage: python tools/exp/record_stats.py [cfg1] [cfg2[:anchors]] [cfg4[:anchors]]"""
import math
import sys
import types

import torch

sys.path.insert(0, ".")
from splatco_amd import _C, rasterizer as R


def stats(name, rs, means3D, opacities, colors, scales, rotations):
    dev = means3D.device
    cs = R._CSettings(rs)
    color, radii, st = R.rasterize_forward(cs, means3D, opacities, scales, rotations, None, None, colors)
    qm = st.debug(_C.DBG_QMASK)
    gm = st.debug(_C.DBG_GM_INDEX).long()
    ranges = st.debug(_C.DBG_RANGES).long()
    ncon = st.debug(_C.DBG_N_CONTRIB).long()
    I, P = st.I, st.P
    H, W = cs.H, cs.W
    gx, gy = (W + 15) // 16, (H + 15) // 16
    # last contributor per tile = max n_contrib over its pixels
    pad = torch.zeros(gy * 16, gx * 16, dtype=torch.long, device=dev)
    pad[:H, :W] = ncon
    tile_last = pad.view(gy, 16, gx, 16).permute(0, 2, 1, 3).reshape(gy * gx, 256).amax(dim=1)
    n = ranges[:, 1] - ranges[:, 0]
    pos = torch.arange(I, device=dev) - torch.repeat_interleave(ranges[:, 0], n)
    behind = pos >= torch.repeat_interleave(tile_last, n)
    dead = qm == 0
    # one backward with random dL: which records come out all-zero
    g = torch.randn(3, H, W, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
    scratch = torch.full((_C.lib.scr_backward_scratch_bytes(I),), 0x7f, dtype=torch.uint8, device=dev)
    outs = [torch.empty(P, w, device=dev) for w in (3, 3, 3, 1, 3, 4)]
    _C.check(_C.lib.scr_backward(P, 0, I, st.flags, means3D.data_ptr(), scales.data_ptr(), rotations.data_ptr(), None, None, cs.ref(),
                                 st.radii.data_ptr(), st.geom.data_ptr(), st.binning.data_ptr(), st.image.data_ptr(), g.data_ptr(),
                                 scratch.data_ptr(), outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr(), None,
                                 outs[3].data_ptr(), outs[4].data_ptr(), outs[5].data_ptr(), None, R._stream(dev)))
    torch.cuda.synchronize()
    recs = scratch[:I * 36].view(I, 36)
    written = ~(recs == 0x7f).all(dim=1)
    zero = (recs == 0).all(dim=1)
    f = lambda m: f"{100.0 * float(m.sum()) / I:5.1f} %"
    print(f"{name}: P = {P}, I = {I}, largest tile {int(n.max())}: quadrant mask 0: {f(dead)} of the instances; behind the tile's last "
          f"contributor: {f(behind)}; either: {f(dead | behind)}; records written {f(written)}, of them all-zero: "
          f"{100.0 * float((zero & written).sum()) / max(float(written.sum()), 1):5.1f} %", flush=True)
    # by Gaussian-major index: which of the zero records were predictable from the mask alone
    zero_gm = torch.zeros(I, dtype=torch.bool, device=dev)
    zero_gm[:] = zero & written
    dead_gm = torch.zeros(I, dtype=torch.bool, device=dev)
    dead_gm[gm] = dead
    behind_gm = torch.zeros(I, dtype=torch.bool, device=dev)
    behind_gm[gm] = behind
    print(f"    zero records explained by mask 0: {f(zero_gm & dead_gm)}; by 'behind' only: {f(zero_gm & behind_gm & ~dead_gm)}; "
          f"neither (every hit pixel stopped earlier / alpha < 1/255 everywhere): {f(zero_gm & ~dead_gm & ~behind_gm)}; "
          f"records NOT written (cut): {f(~written)}", flush=True)


def main():
    dev = torch.device("cuda:0")
    for arg in (sys.argv[1:] or ["cfg1"]):
        cfg, _, anchors = arg.partition(":")
        if cfg == "cfg1":
            from splatco_amd.synthetic import synthetic_camera, synthetic_gaussians
            P, W, H = 1_000_000, 1920, 1080
            cam, g = synthetic_camera(W, H), synthetic_gaussians(P, W, H, 0)
            rs = R.GaussianRasterizationSettings(H, W, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), torch.tensor(g["bg"], device=dev), 1.0,
                                                 cam.world_view_transform.to(dev), cam.full_proj_transform.to(dev), 1,
                                                 cam.camera_center.to(dev), False, False)
            t = lambda a: torch.tensor(a, device=dev)
            stats("cfg1", rs, t(g["means3D"]), t(g["opacities"]), t(g["colors"]), t(g["scales"]), t(g["rotations"]))
        else:
            from splatco_amd.renderer import _settings, generate_neural_gaussians, prefilter_voxel
            from splatco_amd.synthetic import ANCHOR_CONFIGS, synthetic_anchor_model, synthetic_views
            N, _, seed = ANCHOR_CONFIGS[cfg]
            N = int(anchors) if anchors else N
            pc = synthetic_anchor_model(N, seed, dev)
            cam = synthetic_views(1)[0].to(dev)
            pipe = types.SimpleNamespace(debug=False, compute_cov3D_python=False)
            bg = torch.ones(3, device=dev)
            with torch.no_grad():
                vis = prefilter_voxel(cam, pc, pipe, bg)
                xyz, color, opacity, scaling, rot = generate_neural_gaussians(cam, pc, vis, is_training=False)
            del pc
            torch.cuda.empty_cache()
            stats(f"{cfg} ({N} anchors)", _settings(cam, bg, 1.0, False), xyz.contiguous(), opacity.contiguous(), color.contiguous(),
                  scaling.contiguous(), rot.contiguous())
            del xyz, color, opacity, scaling, rot
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
