"""Steady-state kernel breakdown of one BASELINE.json configuration (torch profiler over N steps after the warm-up,
so MIOpen's solver search and the allocator's growth stay out of the numbers -- `rocprofv3 --stats` of the whole
process mixes those in).  Writes the per-step table committed under profiles/.
usage: python tools/step_breakdown.py cfg2|cfg3|cfg4 [anchors] [steps] [random|morton] > profiles/<round>_<cfg>_step_breakdown.txt"""
import sys
import types

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, ".")
from splatco_amd.densify import AnchorDensifier
from splatco_amd.multiview import GradArena
from splatco_amd.losses import scaling_reg
from splatco_amd.renderer import prefilter_voxel, render
from splatco_amd.synthetic import ANCHOR_CONFIGS, synthetic_anchor_model, synthetic_views
from splatco_amd.train_step import collaborative_step


def main(cfg="cfg2", anchors=0, steps=3, order="random"):
    dev = torch.device("cuda:0")
    N, _, seed = ANCHOR_CONFIGS[cfg]
    N = anchors or N
    pc = synthetic_anchor_model(N, seed, dev)
    if order == "morton":
        pc.sort_anchors()
    pipe = types.SimpleNamespace(debug=False, compute_cov3D_python=False)
    bg = torch.ones(3, device=dev)
    views = [v.to(dev) for v in synthetic_views(1)]
    g = torch.Generator(device=dev)
    g.manual_seed(100 + seed)
    gts = [torch.rand(3, 1080, 1920, device=dev, generator=g)]
    if cfg == "cfg2":
        def step():
            for p in pc.parameters():
                p.grad = None
            vis = prefilter_voxel(views[0], pc, pipe, bg)
            out = render(views[0], pc, pipe, bg, visible_mask=vis, retain_grad=True)
            ((out["render"] - gts[0]).abs().mean() + 0.01 * scaling_reg(out["scaling"])).backward()
    else:
        groups = [{"params": [getattr(pc, "_" + n)], "lr": 1e-4, "name": n} for n in ("anchor", "offset", "anchor_feat", "scaling")]
        idle = {id(p) for p in pc.feat_planes._feat.inactive_parameters()}
        groups.append({"params": [p for n, p in pc.named_parameters() if not n.startswith("_") and p.requires_grad and id(p) not in idle], "lr": 1e-3, "name": "mlp_and_feat_planes"})
        opt = torch.optim.Adam(groups, eps=1e-15, fused=True)
        den = AnchorDensifier(pc, opt, seed=seed)
        arena = GradArena([p for grp in groups for p in grp["params"]])

        def step():
            collaborative_step(pc, views, gts, pipe, bg, optimizer=opt, densifier=den, arena=arena)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        step()
    e1.record()
    torch.cuda.synchronize()
    wall = e0.elapsed_time(e1) / steps
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
    rows = [(e.key, e.self_device_time_total / steps / 1e3, e.count / steps) for e in prof.key_averages() if e.self_device_time_total > 0]
    rows.sort(key=lambda r: -r[1])
    total = sum(r[1] for r in rows)
    print(f"# {cfg}: {N} anchors ({order} order), 1 view 1920x1080, steady state ({steps} steps after 3 warm-up steps), MI355X")
    print(f"# step wall time (un-profiled) {wall:.2f} ms; sum of kernel times under the profiler {total:.2f} ms")
    print(f"# {'kernel':88s} {'ms/step':>9s} {'calls/step':>10s} {'pct':>6s}")
    for k, ms, n in rows[:60]:
        print(f"{k[:90]:90s} {ms:9.3f} {n:10.1f} {100 * ms / total:6.2f}")


if __name__ == "__main__":
    a = sys.argv[1:]
    main(a[0] if a else "cfg2", int(a[1]) if len(a) > 1 else 0, int(a[2]) if len(a) > 2 else 3, a[3] if len(a) > 3 else "random")
