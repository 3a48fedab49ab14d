"""Developer tool: where do the large fills of a training step come from?  Wraps the allocating-and-filling entry points
and prints the call site of every fill of more than 50 M elements during one step."""
import sys, types, traceback
import torch
sys.path.insert(0, ".")
from splatco_amd.adam import FusedAdam
from splatco_amd.densify import AnchorDensifier
from splatco_amd.multiview import GradArena
from splatco_amd.synthetic import synthetic_anchor_model, synthetic_views
from splatco_amd.train_step import collaborative_step
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
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
for _ in range(3):
    collaborative_step(pc, views, gts, pipe, bg, optimizer=opt, densifier=den, arena=arena)
torch.cuda.synchronize()
BIG = 50_000_000
def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "/splatco_amd/" in fr.filename:
            return f"{fr.filename.split('/splatco_amd/')[-1]}:{fr.lineno} {fr.line}"
    return "?"
def wrap_fn(mod, name):
    orig = getattr(mod, name)
    def f(*a, **k):
        r = orig(*a, **k)
        if isinstance(r, torch.Tensor) and r.numel() > BIG:
            print(f"{name:12s} {tuple(r.shape)} {r.dtype}  <- {site()}")
        return r
    setattr(mod, name, f)
for n in ("zeros", "full", "zeros_like", "full_like", "ones", "ones_like"):
    wrap_fn(torch, n)
for n in ("zero_", "fill_", "new_zeros", "new_full"):
    orig = getattr(torch.Tensor, n)
    def mk(orig, n):
        def f(self, *a, **k):
            r = orig(self, *a, **k)
            if isinstance(r, torch.Tensor) and r.numel() > BIG:
                print(f"Tensor.{n:9s} {tuple(r.shape)} {r.dtype}  <- {site()}")
            return r
        return f
    setattr(torch.Tensor, n, mk(orig, n))
collaborative_step(pc, views, gts, pipe, bg, optimizer=opt, densifier=den, arena=arena)
torch.cuda.synchronize()
