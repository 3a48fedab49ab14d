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
from torch.utils._python_dispatch import TorchDispatchMode
import traceback
class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        big_int = [a for a in args if isinstance(a, torch.Tensor) and a.is_cuda and a.numel() > 100000 and a.dtype in (torch.bool, torch.int64, torch.int32, torch.uint8)]
        name = str(func)
        if big_int and any(s in name for s in ("sum", "nonzero", "any", "all", "max", "min", "count", "cumsum", "unique", "sort")):
            print("SPY", name, [(tuple(a.shape), a.dtype) for a in big_int], "".join(traceback.format_stack(limit=7)[:-1])[-700:])
        return func(*args, **(kwargs or {}))
with Spy():
    step()
torch.cuda.synchronize()
sys.exit(0)
