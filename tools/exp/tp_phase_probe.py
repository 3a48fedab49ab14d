import ctypes as C, sys
import torch
sys.path.insert(0, ".")
from splatco_amd import _C
from splatco_amd.triplane import multi_triplane_sample
V = 4_600_000
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
ind = torch.rand(V, 3, device=dev, generator=g) * 2 - 1
for S, n, cols in ((700, 3, (0, 5, 10, 15)), (1400, 3, (0, 5, 10, 15)), (700, 6, (0, 10, 20, 5, 15, 25))):
    pl = [torch.randn(1, 5, S, S, device=dev, generator=g).requires_grad_() for _ in range(n)]
    up = torch.randn(V, 32, device=dev, generator=g)
    buf = (C.c_ulonglong * 16)()
    for it in range(3):
        out = multi_triplane_sample([(ind, tuple(pl), cols[:n])])
        torch.cuda.synchronize()
        _C.lib.scr_debug_tp_ticks(buf, 1)
        out.backward(up[:, :out.shape[1]] if False else torch.randn_like(out))
        torch.cuda.synchronize()
        _C.lib.scr_debug_tp_ticks(buf, 1)
    t = [buf[i] / 100.0 for i in range(7)]
    wgs = buf[6]
    names = ["copy", "cells+atomics", "scan", "order", "cell loop", "exchange+out"]
    print(f"size {S} planes {n}: {wgs} workgroups; per-workgroup microseconds: " + ", ".join(f"{nm} {t[i] / wgs:.1f}" for i, nm in enumerate(names)) + f"; total {sum(t[:6]) / wgs:.1f}")
