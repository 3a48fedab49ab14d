"""BASELINE.json configs[3]/[4] probe on ONE GPU: the per-rank work of the sharded mv step -- prefilter,
render, loss (fused L1+SSIM + scaling regulariser), backward, gradient all-reduce (no-op at world size 1),
densification statistics, Adam step -- on N synthetic anchors with tri-plane features, 1080p.
usage: run_train_step.py [N anchors] [views rendered by this rank] [iters].  Developer tool, not the bench contract."""
import math
import sys
import time
import types

import torch

sys.path.insert(0, ".")
from splatco_amd.cameras import look_at_camera
from splatco_amd.densify import AnchorDensifier
from splatco_amd.scene_model import AnchorGaussianModel
from splatco_amd.train_step import collaborative_step


def main(N=5_000_000, views=1, iters=3):
    dev = torch.device("cuda:0")
    torch.manual_seed(2)
    pc = AnchorGaussianModel(plane_size=2800, num_channels=15).to(dev)
    pc.set_anchors(torch.rand(N, 3, device=dev) * 4 - 2, torch.randn(N, 10, 3, device=dev) * 0.5,
                   torch.randn(N, 32, device=dev) * 0.5, torch.randn(N, 6, device=dev) * 0.3 - 5.0)
    pc.feat_planes.Q0 = 0
    pc.feat_planes._feat.activate_level = 2
    pc.train()
    groups = [{"params": [getattr(pc, "_" + n)], "lr": 1e-4, "name": n} for n in ("anchor", "offset", "anchor_feat", "scaling")]
    groups.append({"params": [p for n, p in pc.named_parameters() if not n.startswith("_") and p.requires_grad], "lr": 1e-3, "name": "mlp_and_feat_planes"})
    opt = torch.optim.Adam(groups, eps=1e-15)
    den = AnchorDensifier(pc, opt)
    cams = [look_at_camera((0.5 + 0.3 * i, -0.4, -6.0), (0, 0, 0), (0, -1, 0), math.radians(60), 1920, 1080, uid=i).to(dev)
            for i in range(views)]
    gts = [torch.rand(3, 1080, 1920, device=dev) for _ in cams]
    pipe = types.SimpleNamespace(debug=False, compute_cov3D_python=False)
    bg = torch.ones(3, device=dev)
    for it in range(iters + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loss, out, _ = collaborative_step(pc, cams, gts, pipe, bg, optimizer=opt, densifier=den)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if it:
            print(f"N={N} anchors, {views} view(s)/rank: step {dt * 1e3:.1f} ms = {1 / dt:.2f} iter/s, loss {loss.item():.4f}, "
                  f"Gaussians in last view {out['radii'].shape[0]}, peak {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")


if __name__ == "__main__":
    main(*(int(a) for a in sys.argv[1:]))
