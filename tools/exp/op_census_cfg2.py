"""Developer tool: framework ops left in the cfg2 step (prefilter + render forward + backward), by shape."""
import sys, types, collections
import torch
sys.path.insert(0, ".")
from splatco_amd.losses import scaling_reg
from splatco_amd.renderer import prefilter_voxel, render
from splatco_amd.synthetic import ANCHOR_CONFIGS, synthetic_anchor_model, synthetic_views
dev = torch.device("cuda:0")
N, _, seed = ANCHOR_CONFIGS["cfg2"]
pc = synthetic_anchor_model(N, seed, dev)
pc.sort_anchors()
pipe = types.SimpleNamespace(debug=False, compute_cov3D_python=False)
bg = torch.ones(3, device=dev)
view = synthetic_views(1, 1920, 1080)[0].to(dev)
target = torch.rand(3, 1080, 1920, device=dev)
def step():
    for p in pc.parameters():
        p.grad = None
    vis = prefilter_voxel(view, pc, pipe, bg)
    out = render(view, pc, pipe, bg, visible_mask=vis, retain_grad=True)
    loss = (out["render"] - target).abs().mean() + 0.01 * scaling_reg(out["scaling"])
    loss.backward()
for _ in range(4):
    step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.key_averages(group_by_input_shape=True):
    t = getattr(ev, "self_device_time_total", 0)
    if t <= 0 or not ev.key.startswith("aten::"):
        continue
    agg[(ev.key, str(ev.input_shapes)[:70])][0] += ev.count
    agg[(ev.key, str(ev.input_shapes)[:70])][1] += t
print(f"aten ops with own device time: {sum(v[0] for v in agg.values())} calls, {sum(v[1] for v in agg.values()) / 1e3:.3f} ms")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{v[0]:4d} {v[1]:9.1f} us  {k[0]:28s} {k[1]}")
