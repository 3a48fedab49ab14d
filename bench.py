#!/usr/bin/env python3
"""bench.py -- the hot path on MI355X (BASELINE.json metric), one JSON line.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg1|cfg2|cfg3|cfg4]

--config (default cfg1) names the BASELINE.json configuration:
  cfg1  configs[1]: 1M synthetic Gaussians, 1 view 1920x1080 per GPU, GaussianRasterizer forward+backward.  One
        "step" = one pass of the operator over one view.  With N > 1 the N ranks render N different views of the SAME
        Gaussians (the --mv N branch, train.py:171) and SUM-all-reduce the per-Gaussian gradients over RCCL inside the
        step; value = N * P / step time (weak scaling).  THIS is the line the driver records.
  cfg2  configs[2]: 5M anchors + tri-plane features (plane_size 2800, 15 channels, levels 0..2), 1 view 1080p, 1 GPU:
        prefilter_voxel + render() forward + backward (gaussian_renderer/__init__.py:118-244).  value = Gaussians
        through the whole path per second; `stages` reports anchors/s for a2+a3 and splats/s for a5/a6 separately.
  cfg3  configs[3]: the same scene, mv = N views (4 on 4 GPUs), one per rank: full sharded training step (prefilter,
        render, fused loss, ONE backward, in-place piecewise gradient all-reduce, densification statistics of the last
        view on every rank, Adam).  value = Gaussians rasterised per second over all ranks; iter/s alongside.
  cfg4  configs[4]: 20M anchors, mv = N views (8 on 8 GPUs): the same step; metric = train-step iter/s.

--gpus N with WORLD_SIZE unset starts the N ranks itself (python -m torch.distributed.run, one process per GPU) BEFORE
anything touches the GPU and exits with the launcher's code; under an external launcher WORLD_SIZE must equal N.

`roofline` is for the dominant kernel class of the step, timed with HIP events on the launch stream inside the timed
region (scr_profile_*); `cpu_baseline` is the CPU oracle (oracle/, test infrastructure) timed on the host cores (rank 0,
N = 1 only).  Nothing here reads /root/reference.

Timing protocol: W untimed warm-up steps; untimed settle steps, each kind counted in the line (`allocator_settle_steps`:
until torch's caching allocator stops asking the device for memory; `time_settle_steps`: first-use costs of a fresh box;
`clock_settle_steps`: until the chip is back at the clocks it holds under this load -- it drops them within 20 ms of
idling, which the bookkeeping between warm-up and timed region is; profiles/r05_clock_ramp.txt); then EXACTLY K steps
between barrier + synchronize on both sides, the interpreter's cyclic collector parked.  `host_step_ms` (when each step's
calls returned to the host) and `device_allocs_in_timed_region` tell a disturbed run from a clean one.
"""
import argparse
import gc
import json
import math
import os
import socket
import subprocess
import sys
import time
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "rasterizer fwd+bwd Msplats/s @1080p; PSNR-match vs ref"
P_CFG1, W_CFG1, H_CFG1 = 1_000_000, 1920, 1080
MFMA_F32_PEAK_TFLOPS = 157.3   # v_mfma_f32_16x16x4_f32 / 32x32x2_f32, f32 in / f32 accumulate (MI355X_MICROARCH.md; 155 measured)
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md (datasheet; a float4 copy reaches ~6.3 TB/s)
XGMI_LINK_GBS = 153.0


def algorithmic_bytes(kernel, P, I, npix, extra=None):
    """SURVEY.md 8(d) per-unit figures (bytes one launch must move), also stated in DESIGN.md section 3.
    P = Gaussians, I = (Gaussian, tile) instances, npix = pixels; extra: N anchors, V visible anchors, n = V*k."""
    e = extra or {}
    N, V, n = e.get("N", 0), e.get("V", 0), e.get("n", 0)
    table = {
        "preprocess_kernel": P * (56 + 36) + P * 8 + P * 20,
        "plan_scan_kernel": P // 256 * 8,
        "scatter_kernel": P * 20 + I * 12,
        "tile_sort_kernel": I * 16,
        "blend_forward_kernel": 40 * I + 20 * npix,
        "blend_backward_kernel": 40 * I + 20 * npix + 44 * P,
        "preprocess_backward_kernel": P * (56 + 24 + 44 + 4) + P * 40,
        "filter_kernel": 44 * N,
        # anchor path (per step; V visible anchors, n = V k candidates, P kept Gaussians)
        "expand_kernel": 4 * n + 56 * n + 36 * V + 56 * P + 5 * n,   # count pass + candidates read + Gaussians written + index / mask
        "expand_backward_kernel": 56 * P + 60 * n + 36 * V + 4 * n,
        "triplane_forward_kernel": 4 * (240 * V + 12 * V + 60 * V),  # four sampled grids (attention grid twice): corner gathers + coords + 15 outputs
        # twelve planes: corner scatter + coords + 5 gradient columns -- formed by the binning pass from the sampled matrix (the same
        # 20 bytes per plane) and the BatchNorm-Linear's upstream gradient dy [V,32], read once (round 6)
        "plane_sample_backward_kernels": 12 * (240 * V / 3 + 8 * V + 20 * V) + 4 * V * 32,
        "l1_ssim_forward_kernel": 2 * 12 * npix + 3 * 12 * npix,
        "l1_ssim_backward_kernel": 2 * 12 * npix + 3 * 12 * npix + 12 * npix,
        "mlp_heads_kernel": 4 * V * (32 + 3 + 64) + 4 * V * 110 + 4 * V * 96,             # inputs + outputs + saved hidden layer
        "mlp_heads_backward_kernel": 4 * V * (32 + 3 + 64 + 96 + 40 + 110) + 4 * V * (32 + 3 + 64),   # inputs, hidden, outputs(y), upstream; input gradients
        # BatchNorm-Linear pair of FeaturePlanes (d = 60 and 71 columns): statistics pass + GEMM pass forward;
        # dy^T x pass backward (x re-read by every pass: the algorithm needs the statistics first).  The dx pass is gone from
        # this class since round 6: the producers of the two matrices form its rows (anchor gather backward, tri-plane binning)
        "norm_linear_kernels": 4 * V * (2 * (60 + 71) + 2 * 32),
        "norm_linear_backward_kernels": 4 * V * ((60 + 71) + 2 * 32),
        # attention of the level-0 grid (C stacked channels, HW pixels; forward + backward of one step): the planes are read
        # by the pools, the channel reduction and the apply pass, and written twice as pair planes; backward reads the
        # planes twice, the pair-plane gradients three halves, writes the plane gradients and updates them once more
        "plane_attention_kernels": 4 * e.get("HW", 0) * (13 * e.get("C", 0) + 12),
    }
    return float(table.get(kernel, 0))


MLP_HEADS_MAC = 3 * 99 * 32 + 32 * (10 + 70 + 30)      # 13 024 multiply-adds per anchor: three 99 -> 32 layers, 32 -> {10, 70, 30}


def algorithmic_flops(kernel, V):
    """ALGORITHMIC fp32 flops of the kernels bound by the matrix pipe (what roofline.frac is computed from): the MLP heads
    99 -> 32 -> {10, 70, 30} (scene/gaussian_model.py:315-337) are 13 024 MAC = 26 048 flop per anchor forward; the backward
    forms an input gradient and a weight gradient per layer: twice that."""
    per_anchor = {"mlp_heads_kernel": 2.0 * MLP_HEADS_MAC, "mlp_heads_backward_kernel": 4.0 * MLP_HEADS_MAC}.get(kernel)
    return None if per_anchor is None else per_anchor * V


def issued_mfma_flops(kernel, V):
    """What the kernels ISSUE to the matrix pipe: MFMA instructions per 16-anchor tile x 2048 flop (v_mfma_f32_16x16x4_f32),
    counted in the kernels' ISA (profiles/r02_isa_mix.txt): forward 28 k-steps x 6 hidden tiles + 8 k-steps x 8 output tiles
    = 232; backward dH 64 + dW2 64 + dX 168 + dW1 168 = 464.  More than the algorithmic count by the K padding 99 -> 112,
    the output padding 110 -> 128 and the hidden-layer recompute of the backward."""
    per_tile = {"mlp_heads_kernel": 232, "mlp_heads_backward_kernel": 464}.get(kernel)
    return None if per_tile is None else per_tile * 2048.0 * ((V + 15) // 16)


# ------------------------------------------------------------------ launch
def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args):
    """--gpus N without a launcher: start N ranks as CHILD processes (never re-exec a process that touched the GPU;
    nothing has touched it yet) and propagate the exit code."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    if args.dry_run_ranks:      # every rank on device 0, gloo in RCCL's place: the call sequence is what is looked at
        env.update(SPLATCO_BENCH_ONE_DEVICE="1", SPLATCO_BENCH_BACKEND="gloo")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    # the ranks get their own session = their own process group, so that a hang (a rank that died inside a collective
    # leaves its peers blocked forever) ends in a non-zero exit after --timeout instead of holding the node
    import signal
    proc = subprocess.Popen(cmd, env=env, start_new_session=True)
    try:
        sys.exit(proc.wait(timeout=args.timeout))
    except subprocess.TimeoutExpired:
        print(f"bench.py: --gpus {args.gpus} did not finish within {args.timeout:.0f} s: killing the ranks", file=sys.stderr)
        try:
            os.killpg(proc.pid, signal.SIGKILL)      # exactly the group started above
        except ProcessLookupError:
            pass
        proc.wait()
        sys.exit(124)


class CollectiveLog:
    """--dry-run-ranks: every torch.distributed collective the step issues, in issue order, with its payload.  Installed by
    replacing the module-level functions (the product calls them as dist.<name>), so what is logged is exactly what an RCCL
    run would issue; the ranks of a dry run execute them over gloo on one device."""
    NAMES = ("all_reduce", "reduce_scatter_tensor", "all_gather_into_tensor", "broadcast", "all_gather", "all_gather_object", "barrier")

    def __init__(self):
        self.log, self.on = [], False

    def install(self):
        import torch.distributed as dist
        for name in self.NAMES:
            inner = getattr(dist, name)

            def wrapper(*a, __inner=inner, __name=name, **k):
                if self.on:
                    tens = [x for x in a if isinstance(x, torch.Tensor)] + [x for x in k.values() if isinstance(x, torch.Tensor)]
                    big = max(tens, key=lambda t: t.numel()) if tens else None
                    self.log.append({"collective": __name, "bytes": int(big.numel() * big.element_size()) if big is not None else 0,
                                     "dtype": str(big.dtype).replace("torch.", "") if big is not None else None,
                                     "async": bool(k.get("async_op", False)),
                                     **({"src": k["src"]} if "src" in k else {})})
                return __inner(*a, **k)
            setattr(dist, name, wrapper)

    def capture(self, fn):
        self.log, self.on = [], True
        try:
            fn()
        finally:
            self.on = False
        return list(self.log)


DRY = None      # a CollectiveLog in --dry-run-ranks runs


def dry_run_report(args, step, rank, world, dev, what):
    """One step under the collective log; every rank's sequence must be the same (collectives are matched by issue order);
    rank 0 prints it."""
    import torch.distributed as dist
    DRY.on = False
    seq = DRY.capture(step)
    torch.cuda.synchronize()
    everyone = [None] * world
    dist.all_gather_object(everyone, seq)
    if rank != 0:
        return
    same = all(e == everyone[0] for e in everyone)
    groups, out = [], []
    for e in seq:                                            # run-length encode identical consecutive calls
        if groups and {k: v for k, v in groups[-1].items() if k != "count"} == e:
            groups[-1]["count"] += 1
        else:
            groups.append(dict(e, count=1))
    print(json.dumps({
        "dry_run_ranks": world, "status": "UNMEASURED ON HARDWARE: call sequence only (gloo on one device executes what RCCL would be handed)",
        "what": what, "exchange": args.exchange, "optimizer": getattr(args, "optimizer", None),
        "identical_on_all_ranks": same, "collectives_per_step": len(seq), "bytes_per_step": sum(e["bytes"] for e in seq),
        "sequence": groups}), flush=True)
    if not same:
        raise SystemExit("bench.py --dry-run-ranks: the ranks issued DIFFERENT collective sequences")


def make_view(rank, W, H):
    """Rank r's camera: the synthetic camera shifted sideways by 0.02*r (a different view of the
    same scene, comparable work)."""
    from splatco_amd.cameras import make_camera
    FoVx = math.radians(60.0)
    FoVy = 2.0 * math.atan(math.tan(FoVx / 2) * H / W)
    return make_camera(np.eye(3), np.array([-0.02 * rank, 0.0, 0.0]), FoVx, FoVy, W, H, uid=rank)


def settings_for(cam, bg, dev):
    from splatco_amd.rasterizer import GaussianRasterizationSettings
    return GaussianRasterizationSettings(
        image_height=cam.image_height, image_width=cam.image_width, tanfovx=math.tan(cam.FoVx * 0.5),
        tanfovy=math.tan(cam.FoVy * 0.5), bg=torch.tensor(bg, device=dev), scale_modifier=1.0,
        viewmatrix=cam.world_view_transform.to(dev), projmatrix=cam.full_proj_transform.to(dev), sh_degree=1,
        campos=cam.camera_center.to(dev), prefiltered=False, debug=False)


# ------------------------------------------------------------------ measurement helpers
def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def hbm_copy_peak(dev):
    """Achievable HBM bandwidth on this box: float4 copy of 1 GiB (read + write bytes / time), best of 5."""
    from splatco_amd import _C
    n = 1 << 30
    a = torch.empty(n, dtype=torch.uint8, device=dev)
    b = torch.empty_like(a)
    a.zero_()
    st = torch.cuda.current_stream().cuda_stream
    best = 0.0
    for it in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _C.check(_C.lib.scr_copy_probe(a.data_ptr(), b.data_ptr(), n, st))
        e1.record()
        torch.cuda.synchronize()
        if it:
            best = max(best, 2 * n / (e0.elapsed_time(e1) * 1e-3) / 1e9)
    return best


def cpu_baseline(g, cam, dev, hip_image):
    """Oracle (kind "port": this repo's CPU restatement; the reference has no CPU path) on the whole cfg1 scene, on
    all host cores (the oracle's OpenMP build): one warm-up run, then the median of five forward+backward runs
    (SURVEY.md 8d).  cfg0 (10 k Gaussians, 400x400: BASELINE.json configs[0], the CPU-only plumbing case) is timed the
    same way.  Also returns PSNR(HIP image, oracle image)."""
    from oracle import raster_oracle as orc
    from splatco_amd.synthetic import synthetic_camera, synthetic_gaussians
    orc.build()
    orc.use_threads(True)
    cores = orc.threads()

    def timed(cam_, g_, runs):
        st = orc.Settings(cam_.image_height, cam_.image_width, math.tan(cam_.FoVx * 0.5), math.tan(cam_.FoVy * 0.5),
                          g_["bg"], 1.0, cam_.world_view_transform.numpy(), cam_.full_proj_transform.numpy(), 1,
                          cam_.camera_center.numpy())
        dL = np.random.default_rng(1).standard_normal((3, cam_.image_height, cam_.image_width)).astype(np.float32)
        ts, f, it = [], None, 0
        while len(ts) < runs:                            # run 0 = warm-up (page faults, thread pool)
            t0 = time.perf_counter()
            f = orc.forward(st, g_["means3D"], g_["opacities"], g_["scales"], g_["rotations"], colors_precomp=g_["colors"])
            orc.backward(st, f, dL, g_["means3D"], g_["scales"], g_["rotations"], colors_precomp=g_["colors"])
            dt = time.perf_counter() - t0
            if it:
                ts.append(dt)
            elif dt * (runs + 1) > 40.0:                 # a host with few cores: keep the leg near 30 s of CPU work
                runs = max(1, int(30.0 / dt) - 1)
            it += 1
        return float(np.median(ts)), f, len(ts)

    dt1, f, runs1 = timed(cam, g, 5)
    cam0, g0 = synthetic_camera(400, 400), synthetic_gaussians(10_000, 400, 400, 0)
    dt0, _, _ = timed(cam0, g0, 5)
    orc.use_threads(False)
    a, b = hip_image.astype(np.float64), f["color"].astype(np.float64)
    mse = ((a - b) ** 2).reshape(3, -1).mean(1)
    psnr = float(np.mean(20 * np.log10(1.0 / np.sqrt(np.maximum(mse, 1e-300)))))
    P = g["means3D"].shape[0]
    base = {"value": P / dt1 / 1e6, "unit": "Msplats/s", "cores": cores, "kind": "port", "cpu": cpu_model(),
            "sample": f"the whole cfg1 scene ({P} Gaussians, 1 view 1920x1080), forward+backward: 1 warm-up + median of {runs1} "
                      f"run(s) = {dt1:.2f} s per pass on {cores} OpenMP threads ({os.cpu_count()} host cores)",
            "cfg0": {"value": 10_000 / dt0 / 1e6, "unit": "Msplats/s", "ms_per_pass": dt0 * 1e3,
                     "sample": "configs[0]: 10k Gaussians, 1 view 400x400, same protocol"}}
    return base, psnr


def time_allreduce(fn, dev, reps=5):
    import torch.distributed as dist
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    dist.barrier()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    t = torch.tensor([e0.elapsed_time(e1) / reps], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def allreduce_report(nbytes, ms, world, **more):
    alg = nbytes / (ms * 1e-3) / 1e9
    bus = alg * 2 * (world - 1) / world
    # xGMI is a point-to-point mesh: a ring sends a GPU's whole bus traffic over ONE link (bus rate = that link's rate),
    # a direct / all-to-all schedule spreads it over the world - 1 links to its peers
    out = {"bytes": int(nbytes), "ms": ms, "algbw_GBps": alg, "busbw_GBps": bus,
           "xgmi_link_peak_GBps": XGMI_LINK_GBS, "per_link_GBps_if_ring": bus, "frac_of_one_link": bus / XGMI_LINK_GBS,
           "per_link_GBps_if_all_links": bus / max(world - 1, 1),
           "frac_of_all_links": bus / (XGMI_LINK_GBS * max(world - 1, 1))}
    out.update(more)
    return out


def exposed_exchange(step, state, steps, ms_with, ms_alone, dev):
    """How much of the gradient exchange the step could NOT hide: the same K steps once more with the exchange switched
    off (state["exchange"] = False; every rank the same program, same barriers), MAX over ranks.
    exposed = step with - step without; overlapped = the exchange's stand-alone time - exposed (>= 0)."""
    import torch.distributed as dist
    state["exchange"] = False
    for _ in range(2):
        step()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dist.barrier()
    t = torch.tensor([(time.perf_counter() - t0) / steps * 1e3], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    state["exchange"] = True
    ms_without = float(t.item())
    exp_ms = ms_with - ms_without
    return {"ms_per_step_with_exchange": ms_with, "ms_per_step_without_exchange": ms_without, "exposed_ms": exp_ms,
            "exchange_alone_ms": ms_alone, "overlapped_ms": max(0.0, ms_alone - max(exp_ms, 0.0)) if ms_alone else None,
            "note": "without = the same steps with the gradient exchange skipped (timed after the headline region)"}


def pick_dominant(warm_prof):
    kern = {k: ms / n for k, (ms, n) in warm_prof.items() if n}
    per_step = {k: ms for k, (ms, n) in warm_prof.items() if n}
    return (max(per_step, key=per_step.get) if per_step else "blend_backward_kernel"), kern


# kernel class -> the sources its code comes from (csrc/): a PMC reading is only valid for the code it was taken on
KERNEL_SOURCES = {
    "blend_forward_kernel": ("blend.hip", "common.h"), "blend_backward_kernel": ("blend.hip", "common.h"),
    "preprocess_kernel": ("preprocess.hip", "common.h"), "preprocess_backward_kernel": ("preprocess.hip", "common.h"),
    "filter_kernel": ("preprocess.hip", "common.h"), "plan_scan_kernel": ("binning.hip", "common.h"),
    "scatter_kernel": ("binning.hip", "common.h"), "tile_sort_kernel": ("binning.hip", "common.h"),
    "expand_kernel": ("expand.hip", "common.h"), "expand_backward_kernel": ("expand.hip", "common.h"),
    "triplane_forward_kernel": ("triplane.hip", "common.h"), "plane_sample_backward_kernels": ("triplane.hip", "common.h"),
    "l1_ssim_forward_kernel": ("ssim.hip", "common.h"), "l1_ssim_backward_kernel": ("ssim.hip", "common.h"),
    "mlp_heads_kernel": ("mlp_heads.hip", "common.h"), "mlp_heads_backward_kernel": ("mlp_heads.hip", "common.h"),
    "norm_linear_kernels": ("normlinear.hip", "common.h"), "norm_linear_backward_kernels": ("normlinear.hip", "common.h"),
    "plane_attention_kernels": ("attention.hip", "common.h"),
}


def source_hashes(names=None):
    """sha256[:16] of the kernel sources (they travel with the repository, so the GPU box can check them)."""
    import hashlib
    d = os.path.join(ROOT, "splatco_amd", "csrc")
    names = names or sorted(f for f in os.listdir(d) if f.endswith((".hip", ".h")))
    return {n: hashlib.sha256(open(os.path.join(d, n), "rb").read()).hexdigest()[:16] for n in names if os.path.exists(os.path.join(d, n))}


def profile_value(fname, kernel):
    """A number a profiling run left under profiles/ (PMC traffic, VALU counters), WITH its provenance -- or (None, why)
    when the file is missing, carries no provenance (profiles/provenance.json, written by the tools that produce the
    file), or was collected on other kernel code than the one that is running now."""
    path = os.path.join(ROOT, "profiles", fname)
    if not os.path.exists(path):
        return None, None
    table = json.load(open(path))
    prov = table.get("_provenance")                  # written by tools/pmc_summary.py / pmc_cfg_summary.py (tools/provenance.py)
    if prov is None:
        return None, {"file": f"profiles/{fname}", "dropped": "no provenance recorded in this file"}
    need = KERNEL_SOURCES.get(kernel, ("common.h",))
    now = source_hashes(need)
    then = prov.get("sources", {})
    stale = [n for n in need if then.get(n) != now.get(n)]
    if stale:
        return None, {"file": f"profiles/{fname}", "collected_at_git": prov.get("git"), "dropped":
                      f"stale: {', '.join(stale)} changed since the counters were collected"}
    return table.get(kernel), {"file": f"profiles/{fname}", "collected_at_git": prov.get("git"), "collected": prov.get("date"),
                               "command": prov.get("command"), "sources": {n: then[n] for n in need}}


def roofline_object(dom, avg_ms, ab, peak_measured, note, flops=None, traffic_file="hbm_traffic.json", launches=1.0, issued=None):
    achieved = ab / (avg_ms * 1e-3) / 1e9 if avg_ms else 0.0
    # PMC-derived bytes per launch (cfg1) / per step (cfg2): only if collected on THIS kernel code
    traffic, traffic_src = profile_value(traffic_file, dom)
    if traffic is not None and launches != 1.0:
        traffic = int(traffic / launches)
    out = {"kernel": dom, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
           "algorithmic_bytes_per_launch": ab,
           "avg_launch_ms": avg_ms, "avg_launch_ms_source": "HIP events on the launch stream inside this run's timed region (every 4th step when K >= 8)",
           "peak_measured": peak_measured,
           "frac_of_measured": (achieved / peak_measured) if peak_measured else None, "note": note}
    if flops:      # a kernel whose floor is the matrix pipe, not HBM: the roofline that bounds it (the HBM view stays beside it)
        tf = flops / (avg_ms * 1e-3) / 1e12 if avg_ms else 0.0
        out.update({"bound": "mfma", "achieved": tf, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": tf / MFMA_F32_PEAK_TFLOPS, "algorithmic_flops_per_launch": flops,
                    "frac_is": "ALGORITHMIC flops (2 flop per multiply-add of the layers, x2 for the backward) / time / dense fp32-MFMA peak",
                    **({"mfma_issue_frac": issued / (avg_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS if avg_ms else 0.0,
                        "issued_mfma_flops_per_launch": issued,
                        "mfma_issue_frac_is": "flops of the MFMA instructions the kernel issues (K 99 -> 112 and output 110 -> 128 "
                                              "padding, hidden-layer recompute) / time / peak"} if issued else {}),
                    "hbm_view": {"achieved_GBps": achieved, "frac": achieved / HBM_PEAK_GBS,
                                 "frac_of_measured": (achieved / peak_measured) if peak_measured else None}})
        out.pop("frac_of_measured", None)
    out["valu"] = valu_object(dom, avg_ms)
    if out["valu"] is None:
        out.pop("valu")
    return out


def valu_object(dom, avg_ms):
    """The VALU view of a kernel that is not HBM-bound, from counters collected on this kernel code (else dropped) and the
    calibration of profiles/valu_calibration.json (tools/exp/valu_calib.hip, bare and under rocprofv3 --pmc).  What the
    calibration established: SQ_ACTIVE_INST_VALU counts issue PASSES, not busy cycles -- 1 per plain / DPP / compare /
    select instruction, 2 per transcendental or permlane swap -- and a SIMD sustains 0.44 passes per shader cycle of pure
    v_fma_f32 (2.3 cycles each) but only 0.23 - 0.25 of the 4-cycle forms (DPP, v_cmp -> SGPR pair, v_cndmask with an SGPR
    mask) and of the two-pass forms; the clock a kernel runs at comes from GRBM_GUI_ACTIVE (DVFS: 1.8 - 2.4 GHz).
      valu_frac        = modelled issue cycles / kernel cycles, with the calibrated cost of each class and the kernel's
                         counts: two-pass instructions = ACTIVE - INSTS, 4-cycle forms = their share of the inner loop's
                         ISA (profiles/*isa_mix*), the rest plain.  <= 1: the classes' costs are the saturated ones.
      vs_mix_probe     = the kernel's passes per SIMD-cycle / those of a pure-VALU loop with the same instruction mix at
                         saturating occupancy: how close the kernel is to a loop that does nothing but issue VALU work."""
    n_inst, src_i = profile_value("valu_insts.json", dom)
    passes, src_b = profile_value("valu_busy.json", dom)
    gui, _ = profile_value("gui_active.json", dom)
    if not avg_ms or n_inst is None or passes is None:
        why = (src_i or src_b or {}).get("dropped")
        return {"dropped": why} if why else None
    cpath = os.path.join(ROOT, "profiles", "valu_calibration.json")
    cal = json.load(open(cpath)) if os.path.exists(cpath) else None
    out = {"insts_per_launch": n_inst, "issue_passes_per_launch": passes, "source": src_i or src_b}
    if not cal:
        return out
    f = cal["forms"]
    rate = lambda key: f[key]["busy_quads_per_cycle_per_simd"]      # passes per SIMD and shader cycle of a saturated probe
    mhz = gui / 8.0 / (avg_ms * 1e3) if gui else cal["shader_clock_MHz"]
    cycles = avg_ms * 1e-3 * mhz * 1e6                                   # per SIMD
    c_plain = 1.0 / rate("v_fma_f32 @8")
    c_four = 1.0 / min(rate("v_add_f32 dpp row_shr @5"), rate("v_cmp_lt_f32 -> sgpr pair @5"), rate("v_cndmask_b32_e64 sgpr mask @5"))
    c_two_pass = 2.0 / min(rate("v_exp_f32 @5"), rate("v_permlane32_swap @5"))
    # DPP + compares + selects among the VALU instructions of the inner loop (tools/isa_mix.py, profiles/r06_isa_mix.txt: 12 + 12 + 15 of 182
    # since the round-6 reduction; round 2-5: 41 of 217, with 13 two-pass permlane swaps that are gone)
    four_share = {"blend_backward_kernel": 39.0 / 182.0, "blend_forward_kernel": 0.12}.get(dom, 0.0)
    n_two = max(passes - n_inst, 0)
    n_four = four_share * n_inst
    n_plain = max(n_inst - n_two - n_four, 0)
    modelled = (n_plain * c_plain + n_four * c_four + n_two * c_two_pass) / 1024.0
    mix = max(v["busy_quads_per_cycle_per_simd"] for k, v in f.items() if k.startswith("blend-backward mix"))
    out.update({
        "shader_clock_MHz": round(mhz, 0), "clock_source": "GRBM_GUI_ACTIVE / 8 XCDs / launch time" if gui else "calibration probes",
        "valu_frac": modelled / cycles, "modelled_issue_cycles_per_simd": modelled, "kernel_cycles_per_simd": cycles,
        "class_cycles": {"plain": round(c_plain, 2), "dpp_cmp_select": round(c_four, 2), "transcendental_permlane": round(c_two_pass, 2)},
        "class_counts": {"plain": int(n_plain), "dpp_cmp_select": int(n_four), "transcendental_permlane": int(n_two)},
        "passes_per_simd_cycle": passes / (1024.0 * cycles), "passes_per_simd_cycle_pure_fma": rate("v_fma_f32 @8"),
        "passes_per_simd_cycle_mix_probe": mix, "vs_mix_probe": passes / (1024.0 * cycles) / mix,
        "calibration": "profiles/valu_calibration.json (+ .txt)",
        "note": "VALU-issue bound when valu_frac is near 1: the remaining time is the issue slots the SALU / LDS / waits take "
                "from the vector pipe; SQ_ACTIVE_INST_VALU is an instruction-pass count (round 2 read it as busy quad-cycles: "
                "hence its 1.10)"})
    return out


# ------------------------------------------------------------------ cfg1: the operator
def settle_clocks(step, world, dev, block_ms=10.0, max_blocks=30):
    """Untimed steps until the GPU runs at the clocks it HOLDS under this load, right before the timed region.
    The chip drops its clocks within 20 ms of idling and takes tens of milliseconds of work to bring them back
    (tools/exp/host_rate_cfg1.py, profiles/r05_clock_ramp.txt: 20 cfg1 steps straight after 100 of them 0.963 ms each; the
    same 20 steps after 20 ms / 100 ms / 500 ms of idling, or after a 54 ms gc.collect(): 1.059 / 1.068 / 1.067 / 1.076).
    The bookkeeping between the warm-up and the timed region (reading the warm-up's kernel times, picking the dominant
    class, the collector pass) IS such a pause, so W = 5 warm-up steps of 1 ms followed by it measured the ramp, not the
    kernels: 1.08 against 0.96 ms.  Blocks of >= block_ms of steps are run until a block is no longer 1 % faster than the
    one before it (every rank the same count); the timed region follows without another host pause.  A training loop
    never idles between steps: the sustained clocks are the ones it sees."""
    import torch.distributed as dist
    done, prev, per_block = 0, None, 1
    for _ in range(max_blocks):
        torch.cuda.synchronize()
        ts = time.perf_counter()
        for _ in range(per_block):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - ts) / per_block
        done += per_block
        if os.environ.get("SPLATCO_BENCH_TRACE_SETTLE"):     # developer aid (SPLATCO_BENCH_TRACE would add a sync to every timed step)
            print(f"[trace] clock settle: block of {per_block} step(s) at {dt * 1e3:.4f} ms per step", file=sys.stderr)
        calm = torch.tensor([float(prev is not None and dt >= 0.99 * prev)], device=dev)
        if world > 1:
            dist.all_reduce(calm, op=dist.ReduceOp.MIN)
        prev = dt
        per_block = max(1, int(math.ceil(block_ms * 1e-3 / dt)))       # (the first block is one step: it only sizes the rest)
        if world > 1:
            nb = torch.tensor([float(per_block)], device=dev)
            dist.all_reduce(nb, op=dist.ReduceOp.MAX)
            per_block = int(nb.item())
        if calm.item():
            break
    return done


def run_cfg1(args, rank, world, dev):
    import torch.distributed as dist
    from splatco_amd import _C
    from splatco_amd import rasterizer as R
    from splatco_amd.multiview import allreduce_gradients
    from splatco_amd.rasterizer import GaussianRasterizer
    from splatco_amd.synthetic import synthetic_gaussians

    P, W, H = P_CFG1, W_CFG1, H_CFG1
    g = synthetic_gaussians(P, W, H, seed=0, sigma_scale=args.sigma_scale, scene=args.scene)
    cam = make_view(rank, W, H)
    rast = GaussianRasterizer(settings_for(cam, g["bg"], dev))
    t = lambda a: torch.tensor(a, device=dev, requires_grad=True)
    params = dict(means3D=t(g["means3D"]), opacities=t(g["opacities"]), colors_precomp=t(g["colors"]),
                  scales=t(g["scales"]), rotations=t(g["rotations"]))
    gen = torch.Generator(device=dev)
    gen.manual_seed(1 + rank)
    dL = torch.randn(3, H, W, device=dev, generator=gen)     # dL/dcolor ~ N(0,1), seeded
    means2D = torch.zeros(P, 3, device=dev, requires_grad=True)
    leaves = list(params.values())
    state = {}

    def step():
        for p in leaves:
            p.grad = None
        means2D.grad = None          # the reference makes a fresh screenspace tensor per render (:133)
        img, radii = rast(means2D=means2D, **params)
        img.backward(dL)
        if world > 1 and state.get("exchange", True):
            allreduce_gradients(leaves, agree="once", shape=state.get("shape", "all_reduce"))         # SUM, in place on the operator's gradient arena (train.py:198,240);
                                                              # the ranks agree on the path once, not with a host read per step
        state["radii"], state["img"] = radii, img
        return radii

    if DRY is not None:
        step()
        dry_run_report(args, step, rank, world, dev, f"cfg1: {P} Gaussians, one view per rank, forward + backward + gradient exchange")
        return None
    shapes_ms = None
    if world > 1 and dist.get_backend() == "nccl" and args.warmup:
        # which shape of the SUM exchange is faster on this machine's links is measured, not assumed: one all_reduce of the
        # 56 MB arena against reduce_scatter + all_gather on the same memory, three repetitions each after one untimed
        # step; the timings are MAX-reduced over the ranks, so every rank picks the same one (before the timed region)
        step()
        shapes_ms = {}
        for shp in ("all_reduce", "rs_ag"):
            shapes_ms[shp] = time_allreduce(lambda: allreduce_gradients(leaves, agree="once", shape=shp), dev, reps=3)
        state["shape"] = min(shapes_ms, key=shapes_ms.get)
    # warm-up: the last warm-up steps are timed per kernel class (HIP events around every launch)
    # to find the dominant kernel; the timed region then brackets ONLY that class, because each
    # event pair costs a few microseconds of stream time.
    for w in range(args.warmup):
        step()
    torch.cuda.synchronize()
    # three more untimed steps with every launch bracketed (the per-class table), at the clocks the chip holds under load
    clock_settle, warm_prof = 0, {}
    if args.warmup:
        clock_settle = settle_clocks(step, world, dev)
        _C.profile_enable(True)
        _C.profile_read()
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        warm_prof = _C.profile_read()
    dominant, warm_kern = pick_dominant(warm_prof)
    # the dominant class is timed inside the timed region on every 4th step only: an event pair costs ~6 us of stream time
    # on each side of the launch it brackets (1 % of a cfg1 step); the average is over ceil(K / 4) launches
    _C.profile_enable(dominant, every=4 if args.steps >= 8 else 1)
    _C.profile_read()
    # the interpreter's cyclic collector is parked for the timed region: a generation-2 pass over this process's objects takes
    # milliseconds, and one of them inside 20 steps of 1 ms is a quarter of a millisecond per step (seen once in round 5:
    # 1.28 ms per step with the kernels summing to 1.06 -- profiles/HISTORY.md).  Nothing is skipped.
    gc.collect()
    gc.disable()
    if args.warmup:
        clock_settle += settle_clocks(step, world, dev)       # the bookkeeping above was a pause: once more, then straight into the timed region
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    host_ms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        host_ms.append(time.perf_counter())
        if os.environ.get("SPLATCO_BENCH_TRACE"):        # developer aid: per-step times (adds a sync per step)
            torch.cuda.synchronize()
            print(f"[trace] step done at {(time.perf_counter() - t0) * 1e3:.1f} ms", file=sys.stderr)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    prof = _C.profile_read()
    # Straight behind the timed region, same process, same clocks (no host pause in between): K more steps with EVERY kernel
    # class bracketed by HIP events on every launch -- the per-class table and its sum (`kernel_sum_ms`) beside this run's own
    # `ms_per_step`.  Each event pair costs ~6 us of stream time on both sides of its launch, which is why these steps are
    # not the timed ones; the kernels' own durations are not affected.
    _C.profile_enable(True)
    same_run = {}
    if args.steps:
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        same_run = {k: ms / n for k, (ms, n) in _C.profile_read().items() if n}
    _C.profile_enable(False)
    # ... and what the same K steps take when the chip has idled first (50 ms: the clocks have dropped, profiles/r05_clock_ramp.txt):
    # the figure a caller sees who renders now and then, next to the sustained one a training loop sees
    ramp_ms = None
    if args.steps and args.warmup:
        time.sleep(0.05)
        tr0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        ramp_ms = (time.perf_counter() - tr0) / args.steps * 1e3
    gc.enable()
    allreduce_info, exposed = None, None
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        # bookkeeping (outside the timed region): the gradient all-reduce on its own, SURVEY.md 8(e)
        shp = state.get("shape", "all_reduce")
        bucket = allreduce_gradients(leaves, agree="once", shape=shp)
        ms = time_allreduce(lambda: allreduce_gradients(leaves, agree="once", shape=shp), dev)
        allreduce_info = allreduce_report(bucket.numel() * bucket.element_size(), ms, world, shape=shp,
                                          shapes_timed_in_warmup_ms=shapes_ms,
                                          what=f"{P}x14 fp32 per-Gaussian gradients, exchanged in place on the operator's arena")
        exposed = exposed_exchange(step, state, args.steps, elapsed / args.steps * 1e3, ms, dev)
    if rank != 0:
        return None
    with torch.no_grad():
        _, _, st = R.rasterize_forward(R._CSettings(rast.raster_settings), params["means3D"].detach(),
                                       params["opacities"].detach(), params["scales"].detach(),
                                       params["rotations"].detach(), None, None, params["colors_precomp"].detach())
    I, npix = st.I, W * H
    kern = dict(warm_kern)                                   # all classes: from the warm-up steps ...
    kern.update(same_run)                                    # ... replaced by the K bracketed steps behind the timed region
    kern.update({k: (ms / max(n, 1)) for k, (ms, n) in prof.items() if n})   # dominant: timed region
    peak = hbm_copy_peak(dev)
    ab = algorithmic_bytes(dominant, P, I, npix)
    out = {
        "metric": METRIC, "value": world * P / (elapsed / args.steps) / 1e6, "unit": "Msplats/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "clock_settle_steps": clock_settle,      # untimed, after the W warm-up steps: see settle_clocks()
        # EVERY untimed step before the timed region: the W asked for + the clock-settling blocks + the three bracketed ones
        "effective_warmup_steps": args.warmup + clock_settle + (3 if args.warmup else 0),
        # the same K steps after 50 ms of idling (clocks dropped), measured behind the timed region; `ms_per_step` is the sustained figure
        "ramp_ms_per_step": None if ramp_ms is None else round(ramp_ms, 4),
        # sum of the per-class kernel times of THIS run (K steps straight behind the timed region, every launch bracketed) beside
        # this run's wall time per step: the difference is launch gaps and host-side time, not another box
        "kernel_sum_ms": round(sum(same_run.values()), 4) if same_run else None,
        # when each step's calls RETURNED to the host, as differences (no synchronisation involved): a host-side stall shows here
        "host_step_ms": {"median": round(1e3 * float(np.median(np.diff([t0] + host_ms))), 4), "max": round(1e3 * float(np.max(np.diff([t0] + host_ms))), 4)},
        "config": {"workload": "cfg1: 1M synthetic Gaussians (seed 0), 1 view 1920x1080 per GPU, "
                               "GaussianRasterizer forward+backward, colors_precomp + scale/rotation path",
                   "gaussians": P, "image": f"{W}x{H}", "tile_instances": I,
                   "visible": int((state["radii"] > 0).sum().item()),
                   "parallelism": "1 view per GPU (mv sharding)" + (
                       f", RCCL all-reduce(SUM) of {P}x14 fp32 per-Gaussian grads per step" if world > 1 else "")},
        "roofline": roofline_object(dominant, kern[dominant], ab, peak,
                                    "blend kernels are FP32-VALU/exp-issue bound at this density (SURVEY.md 8d); "
                                    "the HBM fraction is reported as the contract asks"),
        "kernel_ms": {k: round(v, 4) for k, v in sorted(kern.items(), key=lambda kv: -kv[1])},
        "kernel_ms_source": f"{dominant}: HIP events inside the timed region (every 4th step); the other classes: HIP events around "
                            "every launch of K untimed steps straight behind the timed region (same clocks), each pair adding ~6 us of "
                            "stream time on both sides of its launch -- the kernels' own durations are not affected; ms_per_step is "
                            "wall time over K undisturbed steps",
        "hbm_gbs_all_kernels": sum(algorithmic_bytes(k, P, I, npix) for k in kern) / (sum(kern.values()) * 1e-3) / 1e9,
        # every kernel class against ITS byte model, from the per-launch times above
        "kernel_rooflines": {k: {"ms": round(ms, 4), "algorithmic_MB": round(algorithmic_bytes(k, P, I, npix) / 1e6, 1),
                                 "GBps": round(algorithmic_bytes(k, P, I, npix) / (ms * 1e-3) / 1e9, 1),
                                 "frac_of_8TBps": round(algorithmic_bytes(k, P, I, npix) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 3),
                                 "frac_of_measured_peak": round(algorithmic_bytes(k, P, I, npix) / (ms * 1e-3) / 1e9 / peak, 3)}
                             for k, ms in sorted(kern.items(), key=lambda kv: -kv[1]) if ms > 0},
    }
    if args.scene != "uniform":
        out["config"]["workload"] += (f" -- DEVELOPER SCENE '{args.scene}': 80 % of the Gaussians in the central eighth of the image "
                                      "(not the headline: longest-tile-first scheduling and the merge-path sort at work)")
        out["config"]["scene"] = args.scene
        out["config"]["largest_tile_entries"] = int(st.max_tile)
        out["config"]["mean_list_entries_per_tile"] = round(I / (((W + 15) // 16) * ((H + 15) // 16)), 1)
    if args.sigma_scale != 1.0:
        out["config"]["workload"] += f" -- DEVELOPER SWEEP POINT: every screen-space sigma x {args.sigma_scale} (sparser tile lists)"
        out["config"]["sigma_scale"] = args.sigma_scale
        out["config"]["mean_list_entries_per_tile"] = round(I / (((W + 15) // 16) * ((H + 15) // 16)), 1)
    if allreduce_info is not None:
        out["allreduce"] = allreduce_info
    if exposed is not None:
        out["exchange"] = exposed
    hip_image = state["img"].detach().cpu().numpy()
    if world == 1 and not args.no_cfg2 and args.sigma_scale == 1.0 and args.scene == "uniform":
        # the largest single-GPU configuration (BASELINE.json configs[2]) measured by the same process, after the headline:
        # cfg1 stays the line's metric / value, cfg2 rides along so that it is timed under the driver's clock too
        del params, leaves, means2D, dL, st
        state.clear()
        torch.cuda.empty_cache()
        sub = argparse.Namespace(**vars(args))
        sub.config, sub.steps, sub.warmup, sub.anchors = "cfg2", 10, 3, 0
        c2 = run_anchor_config(sub, rank, world, dev)
        out["cfg2"] = {k: c2[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "warmup", "config", "stages", "roofline",
                                          "kernel_ms_per_step", "peak_mem_GiB", "time_settle_steps")}
        out["cfg2"]["kernel_rooflines"] = {k: {kk: v[kk] for kk in ("ms_per_step", "GBps", "frac_of_measured_peak") if kk in v}
                                           for k, v in c2["kernel_rooflines"].items()}
    if world == 1 and not args.no_cpu_baseline and args.sigma_scale == 1.0 and args.scene == "uniform":
        out["cpu_baseline"], out["psnr_match_db"] = cpu_baseline(g, cam, dev, hip_image)
    return out


# ------------------------------------------------------------------ cfg2..4: the anchor scenes
LAST_PLAN = [0, 0, 0]      # (P, tile instances, largest tile) of the most recent rasterizer forward of THIS bench process


def record_plans():
    """Bench bookkeeping, kept out of the product: wrap splatco_amd.rasterizer.rasterize_forward so that the instance count
    of the last forward (part of the per-call RasterState) is visible to the byte model."""
    from splatco_amd import rasterizer as R
    if getattr(R.rasterize_forward, "_recording", False):
        return
    inner = R.rasterize_forward

    def rasterize_forward(*a, **k):
        color, radii, st = inner(*a, **k)
        LAST_PLAN[:] = (st.P, st.I, st.max_tile)
        return color, radii, st
    rasterize_forward._recording = True
    R.rasterize_forward = rasterize_forward


def run_anchor_config(args, rank, world, dev):
    record_plans()
    import torch.distributed as dist
    from splatco_amd import _C
    from splatco_amd.densify import AnchorDensifier
    from splatco_amd.multiview import GradArena
    from splatco_amd.losses import scaling_reg
    from splatco_amd.renderer import prefilter_voxel, render
    from splatco_amd.synthetic import ANCHOR_CONFIGS, synthetic_anchor_model, synthetic_views
    from splatco_amd.train_step import collaborative_step

    N, mv_named, seed = ANCHOR_CONFIGS[args.config]
    N = args.anchors or N
    W, H = 1920, 1080
    pc = synthetic_anchor_model(N, seed, dev)
    if args.anchor_order == "morton":       # the layout AnchorDensifier.sort_anchors keeps (before the optimizer exists)
        pc.sort_anchors()
    pipe = types.SimpleNamespace(debug=False, compute_cov3D_python=False)
    bg = torch.ones(3, device=dev)
    train = args.config in ("cfg3", "cfg4")
    mv = world if train else 1
    views = [v.to(dev) for v in synthetic_views(mv, W, H)]
    gen = torch.Generator(device=dev)
    gen.manual_seed(100 + seed)
    gts = [torch.rand(3, H, W, device=dev, generator=gen) for _ in views]     # identical on every rank
    stats = {}

    if train:
        groups = [{"params": [getattr(pc, "_" + n)], "lr": 1e-4, "name": n} for n in ("anchor", "offset", "anchor_feat", "scaling")]
        idle = {id(p) for p in pc.feat_planes._feat.inactive_parameters()}     # plane levels above activate_level: grad None in the reference
        rest = [p for n, p in pc.named_parameters() if not n.startswith("_") and p.requires_grad and id(p) not in idle]
        groups.append({"params": rest, "lr": 1e-3, "name": "mlp_and_feat_planes"})
        arena = GradArena([p for grp in groups for p in grp["params"]], mode=args.exchange, sparse_rows=args.sparse_exchange)
        if args.optimizer == "hip":         # csrc/adam.hip: the reference's Adam(l, lr=0.0, eps=1e-15) as one streaming pass per group
            from splatco_amd.adam import FusedAdam
            opt = FusedAdam(groups, eps=1e-15)
        elif args.optimizer == "sharded":   # the same update on this rank's 1/world of the parameters; the parameters are gathered
            from splatco_amd.adam import ShardedFusedAdam
            opt = ShardedFusedAdam(groups, arena, eps=1e-15)
        else:                               # torch's fused multi-tensor Adam, for comparison
            opt = torch.optim.Adam(groups, eps=1e-15, fused=True)
        # (the densifier only accumulates statistics here; its optimizer surgery is not exercised by the bench)
        den = AnchorDensifier(pc, opt if args.optimizer != "sharded" else torch.optim.Adam(groups[:1], eps=1e-15), seed=seed)

        from splatco_amd.tv import TV_EVERY, TV_WEIGHT_A

        def step():
            arena.world = world if stats.get("exchange", True) else 1      # 1: reduce() and the hooks issue no collective
            stats["iteration"] = stats.get("iteration", 0) + 1             # train.py:147: iterations count from 1
            # the tri-plane total-variation term rides on every 4th iteration (train.py:242-243, opt.tv_weight_a = 4e-7)
            loss, out, _ = collaborative_step(pc, views, gts, pipe, bg, optimizer=opt, densifier=den, arena=arena,
                                              iteration=stats["iteration"], tv_weight=TV_WEIGHT_A)
            stats["P"], stats["V"] = out["radii"].shape[0], int(out["selection_mask"].numel() // pc.n_offsets)
            stats["rendered"] = out["radii"]
    else:
        target = gts[0]

        def step():
            for p in pc.parameters():
                p.grad = None
            vis = prefilter_voxel(views[0], pc, pipe, bg)
            out = render(views[0], pc, pipe, bg, visible_mask=vis, retain_grad=True)
            loss = (out["render"] - target).abs().mean() + 0.01 * scaling_reg(out["scaling"])
            loss.backward()
            stats["P"], stats["V"] = out["radii"].shape[0], int(out["selection_mask"].numel() // pc.n_offsets)
            stats["rendered"] = out["radii"]

    if DRY is not None:
        step(); step()                                       # the first exchange agrees on the issue order; the second uses it
        dry_run_report(args, step, rank, world, dev,
                       f"{args.config}: {N} anchors, mv = {mv} views, one per rank: one full training step "
                       f"(iteration {stats.get('iteration', 0) + 1}: {'with' if (stats.get('iteration', 0) + 1) % 4 == 0 else 'without'} the total-variation term)")
        step(); step()
        dry_run_report(args, step, rank, world, dev, f"{args.config}: the step of iteration {stats.get('iteration', 0) + 1}")
        return None
    for w in range(args.warmup):
        if w == max(args.warmup - 2, 0):
            torch.cuda.synchronize()
            _C.profile_enable(True)
            _C.profile_read()
        step()
    torch.cuda.synchronize()
    warm_prof = _C.profile_read() if args.warmup else {}
    # The anchors move with every Adam step, so the number of visible anchors / kept Gaussians -- and with it the size of
    # every intermediate -- changes from step to step, and torch's caching allocator keeps requesting fresh device memory
    # (hundreds of ms per step on a fresh box) until its pool covers the pattern: at 20 M anchors the pool grows for
    # about nine steps (56 -> 149 GiB reserved, 26 GiB live).  Steady state = the pool has stopped growing: settle
    # (untimed, every rank the same count) before the K timed steps.
    settle = 0
    if train and args.warmup:
        _C.profile_enable(False)
        last, quiet = torch.cuda.memory_reserved(), 0
        while settle < 24:
            step()
            torch.cuda.synchronize()
            settle += 1
            grew = torch.tensor([float(torch.cuda.memory_reserved() > last)], device=dev)
            last = torch.cuda.memory_reserved()
            if world > 1:
                dist.all_reduce(grew, op=dist.ReduceOp.MAX)
            quiet = 0 if grew.item() else quiet + 1
            if quiet >= 3:       # three steps in a row without a new device allocation (one quiet step proved too few: a
                break            # later step's sizes can still miss the pool, 180 ms of hipMalloc inside the timed region)
    # A fresh box also pays one-time costs that outlast a three-step warm-up (MIOpen compiles and caches the attention
    # grids' convolution kernels on first use; measured 34 instead of 27 ms per cfg2 step in the first process on a box):
    # untimed steps until a step takes no longer than 1.05 x the one before it (every rank the same count).
    time_settle = 0
    if args.warmup:
        _C.profile_enable(False)
        prev = None
        while time_settle < 12:
            torch.cuda.synchronize()
            t_s = time.perf_counter()
            step()
            torch.cuda.synchronize()
            dt_s = time.perf_counter() - t_s
            time_settle += 1
            calm = torch.tensor([float(prev is not None and dt_s <= 1.05 * prev and prev <= 1.05 * dt_s)], device=dev)
            prev = dt_s
            if world > 1:
                dist.all_reduce(calm, op=dist.ReduceOp.MIN)
            if calm.item():
                break
    nwarm = min(args.warmup, 2) or 1
    dominant, warm_kern = pick_dominant(warm_prof)
    warm_step_ms = {k: ms / nwarm for k, (ms, n) in warm_prof.items() if n}
    # the dominant class is timed inside the timed region on every 4th step only: an event pair costs ~6 us of stream time
    # on each side of the launch it brackets (1 % of a cfg1 step); the average is over ceil(K / 4) launches
    _C.profile_enable(dominant, every=4 if args.steps >= 8 else 1)
    _C.profile_read()
    # the interpreter's cyclic collector is parked for the timed region: a generation-2 pass over this process's objects takes
    # milliseconds, and one of them inside 20 steps of 1 ms is a quarter of a millisecond per step (seen once in round 5:
    # 1.28 ms per step with the kernels summing to 1.06 -- profiles/HISTORY.md).  Nothing is skipped.
    gc.collect()
    gc.disable()
    clock_settle = settle_clocks(step, world, dev, max_blocks=6) if args.warmup else 0     # the pause above cooled the chip's clocks
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dev_allocs0 = torch.cuda.memory_stats(dev).get("num_device_alloc", 0)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        if os.environ.get("SPLATCO_BENCH_TRACE"):        # developer aid: per-step times (adds a sync per step)
            torch.cuda.synchronize()
            ms_ = torch.cuda.memory_stats(dev)
            print(f"[trace] step done at {(time.perf_counter() - t0) * 1e3:.1f} ms; reserved {ms_.get('reserved_bytes.all.current', 0) / 2**20:.0f} MiB, "
                  f"device allocs {ms_.get('num_device_alloc', 0)}, frees {ms_.get('num_device_free', 0)}, visible {stats.get('V')}", file=sys.stderr)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    gc.enable()
    dev_allocs = torch.cuda.memory_stats(dev).get("num_device_alloc", 0) - dev_allocs0    # hipMalloc calls inside the timed region (0: the pool covered it)
    prof = _C.profile_read()
    _C.profile_enable(False)
    Pt = torch.tensor([stats["P"]], device=dev, dtype=torch.float64)
    allreduce_info, exposed = None, None
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        dist.all_reduce(Pt)                                  # Gaussians rasterised by all ranks in one step
        if train:
            def exchange(mode):
                a2 = GradArena(arena.params, mode=mode, overlap=False)
                ms = time_allreduce(lambda: (a2.zero(), a2.reduce()), dev, reps=3)
                a2.close()
                return ms
            ms_ar = exchange("all_reduce")
            try:
                ms_rs = exchange("rs_ag")
            except RuntimeError:                             # a backend without reduce_scatter_tensor (older gloo builds)
                ms_rs = None
            arena.bind()
            allreduce_info = allreduce_report(
                arena.nbytes(), ms_ar if args.exchange == "all_reduce" else ms_rs, world,
                what=f"gradient arena: {N} anchors x 71 fp32 + planes + MLPs, exchanged in place in "
                     f"{sum(len(p) for p in arena.pieces)} pieces of <= 256 MiB, issued from autograd hooks",
                mode=args.exchange, ms_all_reduce_pieces=ms_ar, ms_reduce_scatter_all_gather=ms_rs,
                units=len(arena.units), anchor_ranges=len(arena.sink_ranges),
                issue_order_agreed=arena._order is not None)
            if args.optimizer != "sharded":      # (the sharded optimizer's parameter all-gather IS half of the exchange: it cannot be switched off)
                exposed = exposed_exchange(step, stats, args.steps, elapsed / args.steps * 1e3,
                                           ms_ar if args.exchange == "all_reduce" else ms_rs, dev)
    if rank != 0:
        return None
    step_s = elapsed / args.steps
    P_all = float(Pt.item())
    kern = dict(warm_kern)
    kern.update({k: (ms / max(n, 1)) for k, (ms, n) in prof.items() if n})
    ras_f = sum(warm_step_ms.get(k, 0) for k in ("preprocess_kernel", "plan_scan_kernel", "scatter_kernel", "tile_sort_kernel", "blend_forward_kernel"))
    ras_b = sum(warm_step_ms.get(k, 0) for k in ("blend_backward_kernel", "preprocess_backward_kernel"))
    P1 = stats["P"]
    peak = hbm_copy_peak(dev)
    V = stats.get("V", 0)
    lvl0 = pc.feat_planes._feat.k0s[0]
    extra = {"N": N, "V": V, "n": V * pc.n_offsets, "C": lvl0.channels,
             "HW": lvl0.xy_plane.shape[2] * lvl0.xy_plane.shape[3]}
    I = LAST_PLAN[1]                                         # (Gaussian, tile) instances of the last rasterised view
    launches = {k: n / nwarm for k, (ms, n) in warm_prof.items() if n}
    ab = algorithmic_bytes(dominant, P1, I, W * H, extra) / max(launches.get(dominant, 1.0), 1.0)     # per launch
    out = {
        "metric": METRIC if args.config != "cfg4" else "full train-step iter/s (BASELINE.json configs[4])",
        "value": (P_all / step_s / 1e6) if args.config != "cfg4" else 1.0 / step_s,
        "unit": "Msplats/s" if args.config != "cfg4" else "iter/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": step_s * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "allocator_settle_steps": settle, "time_settle_steps": time_settle, "clock_settle_steps": clock_settle, "device_allocs_in_timed_region": int(dev_allocs),
        "reserved_gib": round(torch.cuda.memory_reserved() / 2**30, 1),
        "config": {"workload": {
            "cfg2": f"cfg2: {N} anchors uniform in [-2,2]^3 (seed {seed}), k=10, tri-planes 700/700/1400 active "
                    f"(plane_size 2800, 15 channels, activate_level 2), 1 view 1920x1080: prefilter_voxel + render() "
                    f"forward + backward",
            "cfg3": f"cfg3: {N} anchors (seed {seed}), mv = {mv} views 1080p, 1 per GPU: full sharded training step",
            "cfg4": f"cfg4: {N} anchors (seed {seed}), mv = {mv} views 1080p, 1 per GPU: full training step "
                    f"(prefilter, render, loss, backward, gradient exchange, densification statistics, Adam)"}[args.config],
                   "anchors": N, "anchor_order": {"random": "random (as drawn: the worst case for every gather)",
                                                  "morton": "Morton (AnchorDensifier.sort_anchors)"}[args.anchor_order],
                   "visible_anchors_last_view": V, "gaussians_last_view": P1, "tile_instances_last_view": I,
                   "image": f"{W}x{H}",
                   "rendered_last_view": int((stats["rendered"] > 0).sum().item()),
                   "parallelism": f"{mv} view(s), 1 per GPU" + (", RCCL gradient exchange in place" if world > 1 else ""),
                   **({"optimizer": {"hip": "splatco_amd.adam.FusedAdam (csrc/adam.hip)",
                                     "torch": "torch.optim.Adam(fused=True)",
                                     "sharded": "splatco_amd.adam.ShardedFusedAdam (Adam and its moments / ranks; parameters all-gathered)"}[args.optimizer]}
                      if train else {})},
        "iter_per_s": 1.0 / step_s,
        "stages": {
            "rasterizer_kernels_ms": {"forward": ras_f, "backward": ras_b},
            "rasterizer_Msplats_per_s": P1 / (ras_f + ras_b) / 1e3 if ras_f + ras_b else None,
            "anchor_path_ms": step_s * 1e3 - ras_f - ras_b if not train else None,
            "anchor_path_Manchors_per_s": N / (step_s * 1e3 - ras_f - ras_b) / 1e3 if not train else None,
        },
        "roofline": roofline_object(dominant, kern.get(dominant, 0.0), ab, peak,
                                    "dominant among this library's kernel classes by time per step",
                                    flops=algorithmic_flops(dominant, V), traffic_file=f"hbm_traffic_{args.config}.json",
                                    launches=max(launches.get(dominant, 1.0), 1.0), issued=issued_mfma_flops(dominant, V)),
        "kernel_ms_per_step": {k: round(v, 4) for k, v in sorted(warm_step_ms.items(), key=lambda kv: -kv[1])},
        # every kernel class of this library against ITS byte model (per step, all launches of the class)
        "kernel_rooflines": {k: {"ms_per_step": round(ms, 4),
                                 "algorithmic_MB_per_step": round(algorithmic_bytes(k, P1, I, W * H, extra) / 1e6, 1),
                                 "GBps": round(algorithmic_bytes(k, P1, I, W * H, extra) / (ms * 1e-3) / 1e9, 1),
                                 "frac_of_measured_peak": round(algorithmic_bytes(k, P1, I, W * H, extra) / (ms * 1e-3) / 1e9 / peak, 3),
                                 **({"mfma_TFLOPs": round(algorithmic_flops(k, V) / (ms * 1e-3) / 1e12, 1),
                                     "frac_of_f32_mfma_peak": round(algorithmic_flops(k, V) / (ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, 3),
                                     "mfma_issue_frac": round(issued_mfma_flops(k, V) / (ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, 3)}
                                    if algorithmic_flops(k, V) else {})}
                             for k, ms in sorted(warm_step_ms.items(), key=lambda kv: -kv[1]) if ms > 0},
        "peak_mem_GiB": torch.cuda.max_memory_allocated() / 2 ** 30,
    }
    for k, v in out["kernel_rooflines"].items():
        # PMC traffic of the class per step (tools/profile_cfg_pmc.sh) at the default scene of the config -- only if it
        # was collected on the kernel code that is running now
        t_pmc, _src = profile_value(f"hbm_traffic_{args.config}.json", k) if not args.anchors else (None, None)
        if t_pmc is not None:
            v["pmc_traffic_MB_per_step"] = round(t_pmc / 1e6, 1)
            v["pmc_GBps"] = round(t_pmc / (v["ms_per_step"] * 1e-3) / 1e9, 1)
            v["pmc_collected_at_git"] = _src.get("collected_at_git")
        if k == "blend_forward_kernel" and v["frac_of_measured_peak"] > 1.0:
            v["note"] = ("the byte model counts every entry of every tile list; the forward stops reading a tile's list once "
                         "all of its pixels are opaque (T < 1e-4), which at this density is long before the end")
    if train:
        # the total-variation pass on its own (it runs on every 4th timed step): HBM-bound, 12 algorithmic bytes per plane
        # element (read plane, read + write gradient) over the planes of the active grids
        feat = pc.feat_planes._feat
        elems = sum(p.numel() for lvl in range(feat.activate_level + 1)
                    for p in (feat.k0s[lvl].xy_plane, feat.k0s[lvl].xz_plane, feat.k0s[lvl].yz_plane))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        pc.feat_planes.tv_loss(TV_WEIGHT_A)
        e0.record()
        for _ in range(10):
            pc.feat_planes.tv_loss(TV_WEIGHT_A)
        e1.record()
        torch.cuda.synchronize()
        tv_ms = e0.elapsed_time(e1) / 10
        out["config"]["tv"] = f"every {TV_EVERY}th step (train.py:242-243), weight {TV_WEIGHT_A}, after the gradient exchange, once"
        out["tv_pass"] = {"kernel": "tv_add_grad_kernel", "ms": round(tv_ms, 4), "plane_elements": elems,
                          "algorithmic_MB": round(12 * elems / 1e6, 1), "GBps": round(12 * elems / (tv_ms * 1e-3) / 1e9, 1),
                          "frac_of_measured_peak": round(12 * elems / (tv_ms * 1e-3) / 1e9 / peak, 3),
                          "timed_steps_with_the_term": sum(1 for i in range(stats["iteration"] - args.steps + 1, stats["iteration"] + 1)
                                                           if i % TV_EVERY == 0)}
    if allreduce_info is not None:
        out["allreduce"] = allreduce_info
    if exposed is not None:
        out["exchange"] = exposed
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", choices=["cfg1", "cfg2", "cfg3", "cfg4"], default="cfg1")
    ap.add_argument("--anchors", type=int, default=0, help="override the anchor count of cfg2..4 (developer runs)")
    ap.add_argument("--anchor-order", choices=["random", "morton"], default="morton",
                    help="memory order of the anchors of cfg2..4: Morton-sorted (what AnchorDensifier.sort_anchors keeps; the "
                         "reference's own initial order is the sorted np.unique order of create_from_pcd, "
                         "scene/gaussian_model.py:449) or as drawn (random in space: the stated worst case for every gather)")
    ap.add_argument("--exchange", choices=["all_reduce", "rs_ag"], default="all_reduce",
                    help="shape of the gradient exchange of cfg3/cfg4 (GradArena)")
    ap.add_argument("--optimizer", choices=["hip", "torch", "sharded"], default="hip",
                    help="cfg3/cfg4: splatco_amd.adam.FusedAdam (csrc/adam.hip), torch.optim.Adam(fused=True), or "
                         "splatco_amd.adam.ShardedFusedAdam (needs --exchange rs_ag: reduce-scatter, Adam on 1/N of the parameters, "
                         "all-gather of the parameters)")
    ap.add_argument("--sparse-exchange", action="store_true",
                    help="cfg3/cfg4: exchange the per-anchor gradients row-sparse (only the anchors some view of the step sees; used "
                         "when that union is below 60 %% of the anchors -- never at the synthetic scenes, where every view sees ~92 %%)")
    ap.add_argument("--dry-run-ranks", action="store_true",
                    help="with --gpus N: run the N ranks on ONE device over gloo, log every collective of one step (name, bytes, "
                         "order), check that all ranks issue the same sequence and print it -- the sequence the first RCCL run "
                         "can be diffed against.  No timing is reported.")
    ap.add_argument("--scene", choices=("uniform", "clustered"), default="uniform",
                    help="cfg1 developer scene: 'clustered' moves 80 %% of the Gaussians into the central eighth of the image "
                         "(uneven tile lists: the tile scheduler and the deep-list sort are timed, not just tested); the headline is 'uniform'")
    ap.add_argument("--sigma-scale", type=float, default=1.0,
                    help="cfg1 developer sweep: multiply every screen-space sigma (sparser tile lists; the headline is 1.0)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cfg2", action="store_true", help="cfg1 at N = 1: skip the cfg2 block that rides along on the line")
    ap.add_argument("--timeout", type=float, default=1500.0,
                    help="--gpus N > 1 started by this script: kill the ranks and exit non-zero after this many seconds")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 20 if args.config == "cfg1" else 5
    if args.warmup is None:
        args.warmup = 5 if args.config == "cfg1" else 3

    if args.optimizer == "sharded" and args.exchange != "rs_ag":
        raise SystemExit("bench.py: --optimizer sharded needs --exchange rs_ag")
    if args.dry_run_ranks:
        if args.gpus < 2:
            raise SystemExit("bench.py: --dry-run-ranks needs --gpus N with N > 1")
        if args.config != "cfg1" and not args.anchors:
            args.anchors = 200_000                           # N ranks share one device
            sys.argv += ["--anchors", "200000"]
        args.steps, args.warmup = 1, 1
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args)                                   # does not return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if args.config == "cfg2" and world != 1:
        raise SystemExit("bench.py: cfg2 is the single-GPU configuration (BASELINE.json configs[2])")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the rasterizer has no CPU path")
    # developer overrides to exercise the N > 1 code path on a 1-GPU box (never set by the driver):
    # every rank on one device, gloo instead of RCCL
    if os.environ.get("SPLATCO_BENCH_ONE_DEVICE"):
        local = 0
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    import torch.distributed as dist
    if world > 1:
        backend = os.environ.get("SPLATCO_BENCH_BACKEND", "nccl")   # "nccl" is RCCL over xGMI on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        if args.dry_run_ranks:
            global DRY
            DRY = CollectiveLog()
            DRY.install()
    if args.config == "cfg1":
        out = run_cfg1(args, rank, world, dev)
    else:
        out = run_anchor_config(args, rank, world, dev)
    if out is not None:                                      # rank 0
        if world > 1:
            rccl = None
            if dist.get_backend() == "nccl":
                try:                                         # bookkeeping must never cost the line
                    v = torch.cuda.nccl.version()
                    rccl = ".".join(str(x) for x in v) if isinstance(v, (tuple, list)) else str(v)
                except Exception as e:
                    rccl = f"unknown ({type(e).__name__})"
            out["ranks"] = {"world_size": dist.get_world_size(), "backend": dist.get_backend(), "rccl_version": rccl,
                            "devices": torch.cuda.device_count()}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
