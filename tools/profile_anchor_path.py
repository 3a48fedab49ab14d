"""Developer probe: torch profiler breakdown of generate_neural_gaussians fwd+bwd."""
import math, sys, types
import torch
sys.path.insert(0, ".")
from splatco_amd.cameras import look_at_camera
from splatco_amd.renderer import generate_neural_gaussians, prefilter_voxel
from splatco_amd.scene_model import AnchorGaussianModel
N, plane = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda:0")
torch.manual_seed(0)
pc = AnchorGaussianModel(plane_size=plane, num_channels=15).to(dev)
pc.set_anchors(torch.rand(N, 3, device=dev) * 3.6 - 1.8, torch.randn(N, 10, 3, device=dev) * 0.5,
               torch.randn(N, 32, device=dev) * 0.5, torch.randn(N, 6, device=dev) * 0.3 - 4.5)
pc.feat_planes.Q0 = 0; pc.train(); pc.feat_planes._feat.activate_level = int(sys.argv[3]) if len(sys.argv) > 3 else 0
cam = look_at_camera((0.3, -0.2, -5.5), (0, 0, 0), (0, -1, 0), math.radians(60), 1920, 1080).to(dev)
pipe = types.SimpleNamespace(debug=False, compute_cov3D_python=False)
vis = prefilter_voxel(cam, pc, pipe, torch.ones(3, device=dev))
def step():
    out = generate_neural_gaussians(cam, pc, vis, is_training=True)
    sum(t.sum() for t in out[:5]).backward()
step(); torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    for _ in range(3): step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=28, max_name_column_width=60))
