"""Developer probe (GPU box): which aten::copy_ / clone calls does one collaborative_step without a GradArena make, and how large?"""
import sys, types, torch
sys.path.insert(0, ".")
from splatco_amd.synthetic import synthetic_anchor_model, synthetic_views
from splatco_amd.train_step import collaborative_step
from torch.profiler import profile, ProfilerActivity
N = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
dev = torch.device("cuda:0")
pipe = types.SimpleNamespace(debug=False, compute_cov3D_python=False)
bg = torch.ones(3, device=dev)
views = [v.to(dev) for v in synthetic_views(1, 1920, 1080)]
gts = [torch.rand(3, 1080, 1920, device=dev)]
pc = synthetic_anchor_model(N, 2, dev)
for _ in range(2):
    collaborative_step(pc, views, gts, pipe, bg)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    collaborative_step(pc, views, gts, pipe, bg)
    torch.cuda.synchronize()
rows = []
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::add_", "aten::fill_", "aten::zero_") and ev.device_time_total > 20:
        st = [s for s in (ev.stack or []) if "splatco_amd" in s or "torch/autograd" in s][:3]
        rows.append((ev.device_time_total, ev.name, str(ev.input_shapes)[:80], " <- ".join(s.split("/")[-1][:60] for s in st)))
for t, n, sh, st in sorted(rows, reverse=True)[:30]:
    print(f"{t / 1e3:7.3f} ms {n:16s} {sh:80s} {st}")
