"""BASELINE.json configs[2] probe: N anchors uniform in [-2,2]^3 viewed from outside, k=10, feat 32,
tri-plane features (plane_size, num_channels=15, activate_level=2, Q0=0), 1 view 1080p.  Reports
anchors/s for a2+a3 (prefilter_voxel + generate_neural_gaussians) and splats/s for a5/a6 (rasterizer
forward+backward) separately, as SURVEY.md 8(d) asks.  Developer tool, not the bench contract."""
import math
import sys
import time
import types

import torch

sys.path.insert(0, ".")
from splatco_amd import _C
from splatco_amd.cameras import look_at_camera
from splatco_amd.renderer import prefilter_voxel, render
from splatco_amd.scene_model import AnchorGaussianModel


def morton_order(xyz, lo=-2.0, hi=2.0, bits=10):
    """Permutation that sorts points along a Z-order curve (developer probe of anchor-order locality)."""
    q = ((xyz - lo) / (hi - lo) * (2 ** bits - 1)).clamp(0, 2 ** bits - 1).long()
    code = torch.zeros(xyz.shape[0], dtype=torch.long, device=xyz.device)
    for b in range(bits):
        for a in range(3):
            code |= ((q[:, a] >> b) & 1) << (3 * b + a)
    return torch.argsort(code)


def main(N=5_000_000, plane=2800, iters=3, sort_anchors=False):
    dev = torch.device("cuda:0")
    torch.manual_seed(2)
    pc = AnchorGaussianModel(plane_size=plane, num_channels=15).to(dev)
    anchors = torch.rand(N, 3, device=dev) * 4 - 2
    if sort_anchors:
        anchors = anchors[morton_order(anchors)]
    pc.set_anchors(anchors, torch.randn(N, 10, 3, device=dev) * 0.5,
                   torch.randn(N, 32, device=dev) * 0.5, torch.randn(N, 6, device=dev) * 0.3 - 5.0)
    pc.feat_planes.Q0 = 0
    pc.feat_planes._feat.activate_level = 2
    pc.train()
    cam = look_at_camera((0.5, -0.4, -6.0), (0, 0, 0), (0, -1, 0), math.radians(60), 1920, 1080).to(dev)
    pipe = types.SimpleNamespace(debug=False, compute_cov3D_python=False)
    bg = torch.ones(3, device=dev)
    target = torch.rand(3, 1080, 1920, device=dev)
    ev = lambda: torch.cuda.Event(enable_timing=True)
    for it in range(iters + 1):
        for p in pc.parameters():
            p.grad = None
        e = [ev() for _ in range(4)]
        _C.profile_enable(True); _C.profile_read()
        e[0].record()
        vis = prefilter_voxel(cam, pc, pipe, bg)
        out = render(cam, pc, pipe, bg, visible_mask=vis, retain_grad=True)
        e[1].record()
        loss = (out["render"] - target).abs().mean() + 0.01 * out["scaling"].prod(dim=1).mean()
        loss.backward()
        e[2].record()
        torch.cuda.synchronize()
        prof = _C.profile_read(); _C.profile_enable(False)
        if it == 0:
            continue
        ras = {k: v[0] / max(v[1], 1) for k, v in prof.items() if v[1]}
        ras_fwd = sum(ras.get(k, 0) for k in ("preprocess_kernel", "plan_scan_kernel", "scatter_kernel", "tile_sort_kernel", "blend_forward_kernel"))
        ras_bwd = sum(ras.get(k, 0) for k in ("blend_backward_kernel", "preprocess_backward_kernel"))
        P = out["radii"].shape[0]
        t_fwd, t_all = e[0].elapsed_time(e[1]), e[0].elapsed_time(e[2])
        print(f"N={N} visible anchors={int(vis.sum())} Gaussians P={P} rendered={(out['radii'] > 0).sum().item()}  "
              f"render() fwd {t_fwd:.1f} ms, fwd+bwd {t_all:.1f} ms | rasterizer kernels fwd {ras_fwd:.2f} ms bwd {ras_bwd:.2f} ms "
              f"-> {P / (ras_fwd + ras_bwd) / 1e3:.0f} Msplats/s | anchor path (rest) {t_all - ras_fwd - ras_bwd:.1f} ms "
              f"-> {N / (t_all - ras_fwd - ras_bwd) / 1e3:.1f} Manchors/s | peak {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
        print("   ", {k: round(v, 3) for k, v in sorted(ras.items(), key=lambda kv: -kv[1])})


if __name__ == "__main__":
    if "--sorted" in sys.argv:
        sys.argv.remove("--sorted")
        main(*[int(a) for a in sys.argv[1:]], sort_anchors=True)
        sys.exit(0)
    main(*(int(a) for a in sys.argv[1:]))
