"""Developer tool: which framework ops (fills, copies, small elementwise kernels) are left in a training step, by call
site.  usage: python tools/exp/op_census.py [anchors]"""
import sys, types
import torch
sys.path.insert(0, ".")
from splatco_amd.adam import FusedAdam
from splatco_amd.densify import AnchorDensifier
from splatco_amd.multiview import GradArena
from splatco_amd.synthetic import synthetic_anchor_model, synthetic_views
from splatco_amd.train_step import collaborative_step
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
pc = synthetic_anchor_model(N, 3, dev)
pc.sort_anchors()
pipe = types.SimpleNamespace(debug=False, compute_cov3D_python=False)
bg = torch.ones(3, device=dev)
views = [v.to(dev) for v in synthetic_views(1, 1920, 1080)]
gts = [torch.rand(3, 1080, 1920, device=dev)]
groups = [{"params": [getattr(pc, "_" + n)], "lr": 1e-4, "name": n} for n in ("anchor", "offset", "anchor_feat", "scaling")]
idle = {id(p) for p in pc.feat_planes._feat.inactive_parameters()}
groups.append({"params": [p for n, p in pc.named_parameters() if not n.startswith("_") and p.requires_grad and id(p) not in idle], "lr": 1e-3, "name": "rest"})
opt = FusedAdam(groups, eps=1e-15)
den = AnchorDensifier(pc, opt, seed=3)
arena = GradArena([p for grp in groups for p in grp["params"]])
for _ in range(4):
    collaborative_step(pc, views, gts, pipe, bg, optimizer=opt, densifier=den, arena=arena)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    collaborative_step(pc, views, gts, pipe, bg, optimizer=opt, densifier=den, arena=arena)
    torch.cuda.synchronize()
import collections
agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.key_averages(group_by_input_shape=True, group_by_stack_n=12):
    t = getattr(ev, "self_device_time_total", 0) or getattr(ev, "self_cuda_time_total", 0)
    if t <= 0 or not ev.key.startswith("aten::"):
        continue
    site = next((s for s in ev.stack if "/splatco_amd/" in s or "bench.py" in s), ev.stack[0] if ev.stack else "?")
    key = (ev.key, site.split("/root/repo/")[-1][:90], str(ev.input_shapes)[:60])
    agg[key][0] += ev.count
    agg[key][1] += t
tot = sum(v[1] for v in agg.values())
print(f"aten ops with own device time: {sum(v[0] for v in agg.values())} calls, {tot / 1e3:.3f} ms")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:70]:
    print(f"{v[0]:4d} {v[1]:9.1f} us  {k[0]:28s} {k[1]:90s} {k[2]}")
