"""Developer probe: which torch (ATen) operators of a bench configuration launch elementwise kernels, with shapes.
usage: python tools/exp/elementwise_probe.py cfg3"""
import sys
import torch
from torch.profiler import profile, ProfilerActivity
sys.path.insert(0, ".")
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
sys.argv = ["bench.py", "--config", cfg, "--steps", "3", "--warmup", "2", "--no-cpu-baseline"]
import bench
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU], record_shapes=True) as prof:
    bench.main()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    dt = getattr(e, "device_time_total", None)
    if dt is None:
        dt = e.cuda_time_total
    if e.key.startswith("aten::") and dt > 0:
        rows.append((dt, e.count, e.key, str(e.input_shapes)[:150]))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"aten operators with device time: {tot / 1e3:.1f} ms over the run")
for dt, n, k, sh in rows[:40]:
    print(f"{dt / 1e3:9.3f} ms  x{n:<5d} {dt / n:8.1f} us  {k:28s} {sh}")
