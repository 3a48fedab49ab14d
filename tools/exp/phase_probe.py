"""Per-phase wall-clock shares of the instrumented kernels (-DSCR_PHASE_TIMING, csrc/common.h) on the cfg2 workload:
the MLP heads backward and forward kernels and the plane-gradient cell gather.
build:  cd splatco_amd/csrc && mkdir -p exp && for f in mlp_heads triplane capi; do hipcc --offload-arch=gfx950 -O3 -std=c++17 \
        -fPIC -ffp-contract=off -fno-slp-vectorize -DSCR_PHASE_TIMING -c $f.hip -o exp/${f}_pt.o; done && hipcc --offload-arch=gfx950 \
        -shared -fPIC preprocess.o binning.o blend.o expand.o exp/triplane_pt.o ssim.o densify.o exp/mlp_heads_pt.o \
        anchor_gather.o normlinear.o exp/capi_pt.o -o exp/lib_phase.so
run:    SPLATCO_RASTER_LIB=$PWD/splatco_amd/csrc/exp/lib_phase.so python tools/exp/phase_probe.py"""
import ctypes as C
import sys
import types

import torch

sys.path.insert(0, ".")
from splatco_amd import _C
from splatco_amd.renderer import prefilter_voxel, render
from splatco_amd.synthetic import ANCHOR_CONFIGS, synthetic_anchor_model, synthetic_views

if not hasattr(_C.lib, "scr_debug_phase_ticks"):
    sys.exit("the loaded library was not built with -DSCR_PHASE_TIMING (see the docstring)")
_C.lib.scr_debug_phase_ticks.argtypes = [C.c_int32, C.c_void_p]
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
    ((out["render"] - gt).abs().mean() + 0.01 * out["scaling"].prod(dim=1).mean()).backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
buf = (C.c_ulonglong * 16)()
for which in (0, 1, 2):
    _C.lib.scr_debug_phase_ticks(which, buf)          # clear
step()
torch.cuda.synchronize()
NAMES = {0: ("mlp_heads_backward_kernel (per wave)", ["stage dZ / H", "dH MFMAs + relu", "dW2 MFMAs + db2", "issue loads", "dX MFMAs",
                                                       "1/|o| + stores", "restage + dW1 MFMAs", "loop top"]),
         1: ("mlp_heads_forward_kernel (per wave)", ["loads + ob_view", "layer-1 MFMAs", "relu + hidden store", "layer-2 MFMAs",
                                                     "activations + stores"]),
         2: ("tp_cell_gather_kernel (per workgroup, all 9 launches of a step)", ["copy to LDS", "cells + ranks", "scan", "index list",
                                                                                "cell loop", "corner exchange + output"])}
for which, (title, names) in NAMES.items():
    _C.lib.scr_debug_phase_ticks(which, buf)
    n, tot = buf[15], sum(buf[i] for i in range(len(names)))
    if not n:
        continue
    print(f"{title}: {n} leaders, {tot / n / 100:.1f} us each; " + ", ".join(f"{nm} {buf[i] / tot:.1%}" for i, nm in enumerate(names)))
