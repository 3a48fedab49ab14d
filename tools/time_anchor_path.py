"""Developer timing probe for the anchor path (a2 + a3): prefilter_voxel and
generate_neural_gaussians, torch op chain vs fused expansion, at a given anchor count."""
import math
import sys
import time
import types

import torch

sys.path.insert(0, ".")
from splatco_amd.cameras import look_at_camera
from splatco_amd.renderer import generate_neural_gaussians, prefilter_voxel
sys.path.insert(0, 'tests')
from torch_restatements import expand_torch_chain
from splatco_amd.scene_model import AnchorGaussianModel


def main(N=1_000_000, plane=700):
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    pc = AnchorGaussianModel(plane_size=plane, num_channels=15).to(dev)
    pc.set_anchors(torch.rand(N, 3, device=dev) * 3.6 - 1.8, torch.randn(N, 10, 3, device=dev) * 0.5,
                   torch.randn(N, 32, device=dev) * 0.5, torch.randn(N, 6, device=dev) * 0.3 - 4.5)
    pc.feat_planes.Q0 = 0
    pc.train()
    cam = look_at_camera((0.3, -0.2, -5.5), (0, 0, 0), (0, -1, 0), math.radians(60), 1920, 1080).to(dev)
    pipe = types.SimpleNamespace(debug=False, compute_cov3D_python=False)
    bg = torch.ones(3, device=dev)

    def timeit(fn, n=5):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    t_pre = timeit(lambda: prefilter_voxel(cam, pc, pipe, bg))
    vis = prefilter_voxel(cam, pc, pipe, bg)
    print(f"N={N} visible={int(vis.sum())}  prefilter_voxel {t_pre:.3f} ms")
    for fused in (False, True):
        def fwd_bwd():
            out = generate_neural_gaussians(cam, pc, vis, is_training=True, expand=None if fused else expand_torch_chain)
            sum(t.sum() for t in out[:5]).backward()
        def fwd():
            with torch.no_grad():
                generate_neural_gaussians(cam, pc, vis, is_training=True, expand=None if fused else expand_torch_chain)
        print(f"  fused={fused}: forward {timeit(fwd):.2f} ms  forward+backward {timeit(fwd_bwd):.2f} ms  "
              f"peak mem {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB")
        torch.cuda.reset_peak_memory_stats()


if __name__ == "__main__":
    main(*(int(a) for a in sys.argv[1:]))
