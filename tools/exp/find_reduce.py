"""Which torch ops launch the int64 reduce kernel in the cfg3 train step (developer probe)."""
import sys, types
import torch
from torch.profiler import ProfilerActivity, profile
sys.path.insert(0, ".")
from splatco_amd.densify import AnchorDensifier
from splatco_amd.multiview import GradArena
from splatco_amd.synthetic import ANCHOR_CONFIGS, synthetic_anchor_model, synthetic_views
from splatco_amd.train_step import collaborative_step
dev = torch.device("cuda:0")
N = 1_000_000
pc = synthetic_anchor_model(N, 1, dev)
pipe = types.SimpleNamespace(debug=False, compute_cov3D_python=False)
bg = torch.ones(3, device=dev)
views = [v.to(dev) for v in synthetic_views(1)]
gts = [torch.rand(3, 1080, 1920, device=dev)]
params = [p for p in pc.parameters() if p.requires_grad]
opt = torch.optim.Adam(params, lr=1e-4, eps=1e-15, fused=True)
arena = GradArena(params)
dens = AnchorDensifier(pc, opt, seed=0)
def step():
    collaborative_step(pc, views, gts, pipe, bg, optimizer=opt, densifier=dens, arena=arena)
for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.self_device_time_total > 30 and not e.key.startswith(("void", "scr::", "_"))]
rows.sort(key=lambda e: -e.self_device_time_total)
for e in rows[:40]:
    print(f"{e.key[:40]:40s} {e.self_device_time_total / 1e3:7.3f} ms x{e.count:3d}  {str(e.input_shapes)[:120]}")
sys.exit(0)
