"""Which torch ops (with input shapes) remain in the cfg2 step, by device time."""
import sys, types
import torch
from torch.profiler import ProfilerActivity, profile
sys.path.insert(0, ".")
from splatco_amd.losses import scaling_reg
from splatco_amd.renderer import prefilter_voxel, render
from splatco_amd.synthetic import ANCHOR_CONFIGS, synthetic_anchor_model, synthetic_views
dev = torch.device("cuda:0")
N, _, seed = ANCHOR_CONFIGS["cfg2"]
pc = synthetic_anchor_model(N, seed, dev)
pipe = types.SimpleNamespace(debug=False, compute_cov3D_python=False)
bg = torch.ones(3, device=dev)
view = synthetic_views(1)[0].to(dev)
gt = torch.rand(3, 1080, 1920, device=dev)
def step():
    for p in pc.parameters():
        p.grad = None
    vis = prefilter_voxel(view, pc, pipe, bg)
    out = render(view, pc, pipe, bg, visible_mask=vis, retain_grad=True)
    ((out["render"] - gt).abs().mean() + 0.01 * scaling_reg(out["scaling"])).backward()
for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.self_device_time_total > 8 and not e.key.startswith(("void", "scr::", "_", "Memcpy", "Memset"))]
rows.sort(key=lambda e: -e.self_device_time_total)
for e in rows[:90]:
    print(f"{e.key[:42]:42s} {e.self_device_time_total / 1e3:7.3f} ms x{e.count:3d}  {str(e.input_shapes)[:110]}")
