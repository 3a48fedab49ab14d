"""Developer probe (GPU box): is a full training step bit-reproducible between two identical model instances in one
process?  Prints, per parameter, whether three steps of two replicas left identical bits (single rank)."""
import sys, types, torch
sys.path.insert(0, ".")
from splatco_amd.adam import FusedAdam
from splatco_amd.multiview import GradArena
from splatco_amd.synthetic import synthetic_anchor_model, synthetic_views
from splatco_amd.train_step import collaborative_step

dev = torch.device("cuda:0")
pipe = types.SimpleNamespace(debug=False, compute_cov3D_python=False)
bg = torch.ones(3, device=dev)
W, H, N, MV = 640, 360, 200_000, 2
views = [v.to(dev) for v in synthetic_views(MV, W, H)]
g = torch.Generator(device=dev).manual_seed(5)
gts = [torch.rand(3, H, W, device=dev, generator=g) for _ in range(MV)]


def make():
    pc = synthetic_anchor_model(N, 9, dev, plane_size=256)
    idle = {id(p) for p in pc.feat_planes._feat.inactive_parameters()}
    groups = [{"params": [getattr(pc, "_" + n)], "lr": 1e-4, "name": n} for n in ("anchor", "offset", "anchor_feat", "scaling")]
    rest = [(n, p) for n, p in pc.named_parameters() if not n.startswith("_") and p.requires_grad and id(p) not in idle]
    groups.append({"params": [p for _, p in rest], "lr": 1e-3, "name": "rest"})
    names = ["anchor", "offset", "anchor_feat", "scaling"] + [n for n, _ in rest]
    params = [p for grp in groups for p in grp["params"]]
    return pc, params, names, GradArena(params), FusedAdam(groups, eps=1e-15)


a, b = make(), make()
for it in range(3):
    for pc, params, names, arena, opt in (a, b):
        collaborative_step(pc, views, gts, pipe, bg, optimizer=opt, arena=arena, iteration=4 * (it + 1), tv_weight=1e-3)
    bad = [(n, float((x.double() - y.double()).abs().max())) for n, x, y in zip(a[2], a[1], b[1]) if not torch.equal(x, y)]
    print(f"after step {it + 1}: {len(bad)} of {len(a[1])} parameters differ", bad[:8])
    if it == 0:
        gbad = [(n, float((x.grad.double() - y.grad.double()).abs().max())) for n, x, y in zip(a[2], a[1], b[1]) if not torch.equal(x.grad, y.grad)]
        print("   gradients of step 1 that differ:", gbad[:12])
