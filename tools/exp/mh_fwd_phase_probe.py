import ctypes as C, sys
import torch
sys.path.insert(0, ".")
from splatco_amd import _C
exec(open("tools/exp/op_shapes.py").read().split("for _ in range(3):")[0])     # model + step() of cfg2
for _ in range(3):
    step()
torch.cuda.synchronize()
buf = (C.c_ulonglong * 16)()
_C.lib.scr_debug_mh_ticks(buf, 1)
with torch.no_grad():
    vis = prefilter_voxel(view, pc, pipe, bg)
    render(view, pc, pipe, bg, visible_mask=vis)
torch.cuda.synchronize()
_C.lib.scr_debug_mh_ticks(buf, 1)
n = buf[8]
names = ["loads + ob", "layer 1 MFMAs", "relu + hidden store", "layer 2 MFMAs", "activations + stores"]
tot = sum(buf[i] for i in range(5))
print(f"{n} waves; per wave total {tot / n / 100:.1f} us; share per phase:", {nm: f"{buf[i] / tot:.1%}" for i, nm in enumerate(names)})
