import sys, time, types, os
import torch
sys.path.insert(0, ".")
from splatco_amd import _C
from splatco_amd.densify import AnchorDensifier
from splatco_amd.multiview import GradArena
from splatco_amd.synthetic import ANCHOR_CONFIGS, synthetic_anchor_model, synthetic_views
from splatco_amd.train_step import collaborative_step
dev = torch.device("cuda:0")
N, _, seed = ANCHOR_CONFIGS["cfg4"]
pc = synthetic_anchor_model(N, seed, dev)
pipe = types.SimpleNamespace(debug=False, compute_cov3D_python=False)
bg = torch.ones(3, device=dev)
views = [v.to(dev) for v in synthetic_views(1)]
gts = [torch.rand(3, 1080, 1920, device=dev)]
groups = [{"params": [getattr(pc, "_" + n)], "lr": 1e-4, "name": n} for n in ("anchor", "offset", "anchor_feat", "scaling")]
groups.append({"params": [p for n, p in pc.named_parameters() if not n.startswith("_") and p.requires_grad], "lr": 1e-3, "name": "mlp_and_feat_planes"})
opt = torch.optim.Adam(groups, eps=1e-15, fused=True)
den = AnchorDensifier(pc, opt, seed=seed)
arena = GradArena([p for grp in groups for p in grp["params"]])
mode = sys.argv[1] if len(sys.argv) > 1 else "none"
for i in range(14):
    if mode == "prof" and i == 1:
        _C.profile_enable(True)
    if mode == "prof" and i == 3:
        _C.profile_read(); _C.profile_enable("mlp_heads_backward_kernel"); _C.profile_read()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    collaborative_step(pc, views, gts, pipe, bg, optimizer=opt, densifier=den, arena=arena)
    torch.cuda.synchronize()
    print(f"{mode} step {i}: {(time.perf_counter() - t0) * 1e3:.1f} ms  reserved {torch.cuda.memory_reserved() / 2**30:.1f} GiB allocated {torch.cuda.memory_allocated() / 2**30:.1f}", flush=True)
