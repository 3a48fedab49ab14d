#!/usr/bin/env python3
"""bench.py -- rasterizer forward+backward throughput on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over one view: GaussianRasterizer forward (projection,
bucketing, depth sort, blend) + backward (blend backward, per-Gaussian reduce, projection
backward) on BASELINE.json configs[1]: 1M synthetic Gaussians, 1920x1080, inputs resident in
HBM.  With --gpus N (launched by torch.distributed.run, one rank per GPU) the N ranks render N
different views of the SAME Gaussians (the --mv N branch, train.py:171) and SUM-all-reduce the
per-Gaussian gradients over RCCL inside the step; value = N * P / step time (weak scaling).

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel, timed with HIP events on
the launch stream inside the timed region (scr_profile_*); `cpu_baseline` is the CPU oracle
(oracle/, test infrastructure) timed on the host cores for a bounded sample of the same workload.
Nothing here reads /root/reference.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "rasterizer fwd+bwd Msplats/s @1080p; PSNR-match vs ref"
P_CFG1, W_CFG1, H_CFG1 = 1_000_000, 1920, 1080
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md (spec; ~6.3 TB/s achievable)
CPU_SAMPLE_P = 1_000_000       # cpu_baseline: the whole cfg1 scene (all host cores; ~20 s of CPU work)


def algorithmic_bytes(kernel, P, I, npix):
    """SURVEY.md 8(d) per-unit figures (bytes one launch must move), also stated in DESIGN.md."""
    return {
        "preprocess_kernel": P * (56 + 36) + P * 8 + P * 20,
        "plan_scan_kernel": P // 256 * 8,
        "scatter_kernel": P * 20 + I * 12,
        "tile_sort_kernel": I * 16,
        "blend_forward_kernel": 40 * I + 20 * npix,
        "blend_backward_kernel": 40 * I + 20 * npix + 44 * P,
        "preprocess_backward_kernel": P * (56 + 24 + 44 + 4) + P * 40,
        "filter_kernel": 44 * P,
    }[kernel]


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


def cpu_baseline(g, cam, dev):
    """Oracle (kind "port": this repo's CPU restatement; the reference has no CPU path) on the
    first CPU_SAMPLE_P Gaussians, on all host cores (the oracle's OpenMP build).  Also returns
    PSNR(HIP image, oracle image)."""
    from oracle import raster_oracle as orc
    from splatco_amd.rasterizer import GaussianRasterizer
    n = CPU_SAMPLE_P
    sub = {k: (v[:n] if k != "bg" else v) for k, v in g.items()}
    st = orc.Settings(cam.image_height, cam.image_width, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5),
                      sub["bg"], 1.0, cam.world_view_transform.numpy(), cam.full_proj_transform.numpy(), 1,
                      cam.camera_center.numpy())
    rng = np.random.default_rng(1)
    dL = rng.standard_normal((3, cam.image_height, cam.image_width)).astype(np.float32)
    orc.build()
    orc.use_threads(True)
    cores = orc.threads()
    t0 = time.perf_counter()
    f = orc.forward(st, sub["means3D"], sub["opacities"], sub["scales"], sub["rotations"], colors_precomp=sub["colors"])
    orc.backward(st, f, dL, sub["means3D"], sub["scales"], sub["rotations"], colors_precomp=sub["colors"])
    dt = time.perf_counter() - t0
    orc.use_threads(False)
    t = lambda a: torch.tensor(a, device=dev)
    with torch.no_grad():
        img, _ = GaussianRasterizer(settings_for(cam, sub["bg"], dev))(
            means3D=t(sub["means3D"]), means2D=torch.zeros(n, 3, device=dev), opacities=t(sub["opacities"]),
            colors_precomp=t(sub["colors"]), scales=t(sub["scales"]), rotations=t(sub["rotations"]))
    a, b = img.cpu().numpy().astype(np.float64), f["color"].astype(np.float64)
    mse = ((a - b) ** 2).reshape(3, -1).mean(1)
    psnr = float(np.mean(20 * np.log10(1.0 / np.sqrt(np.maximum(mse, 1e-300)))))
    base = {"value": n / dt / 1e6, "unit": "Msplats/s", "cores": cores, "kind": "port",
            "sample": f"first {n} of the {P_CFG1} cfg1 Gaussians, 1 view 1920x1080, fwd+bwd once, "
                      f"{dt:.1f} s on {cores} OpenMP threads ({os.cpu_count()} host cores)"}
    return base, psnr


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
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

    from splatco_amd import _C
    from splatco_amd.multiview import allreduce_gradients
    from splatco_amd.rasterizer import GaussianRasterizer
    from splatco_amd.synthetic import synthetic_gaussians

    P, W, H = P_CFG1, W_CFG1, H_CFG1
    g = synthetic_gaussians(P, W, H, seed=0)
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

    def step():
        for p in leaves:
            p.grad = None
        means2D.grad = None          # the reference makes a fresh screenspace tensor per render (:133)
        img, radii = rast(means2D=means2D, **params)
        img.backward(dL)
        if world > 1:
            allreduce_gradients(leaves)                       # SUM, one flat bucket (train.py:198,240)
        return radii

    # warm-up: the last warm-up steps are timed per kernel class (HIP events around every launch)
    # to find the dominant kernel; the timed region then brackets ONLY that class, because each
    # event pair costs a few microseconds of stream time.
    for w in range(args.warmup):
        if w == max(args.warmup - 3, 0):
            torch.cuda.synchronize()
            _C.profile_enable(True)
            _C.profile_read()
        step()
    torch.cuda.synchronize()
    warm_prof = _C.profile_read() if args.warmup else {}
    warm_kern = {k: ms / n for k, (ms, n) in warm_prof.items() if n}
    dominant = max(warm_kern, key=warm_kern.get) if warm_kern else "blend_backward_kernel"
    _C.profile_enable(dominant)
    _C.profile_read()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        radii = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    prof = _C.profile_read()
    _C.profile_enable(False)
    allreduce_info = None
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        # bookkeeping (outside the timed region): the gradient all-reduce on its own, SURVEY.md 8(e)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        dist.barrier()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(5):
            bucket = allreduce_gradients(leaves)
        e1.record()
        torch.cuda.synchronize()
        ar = torch.tensor([e0.elapsed_time(e1) / 5], device=dev, dtype=torch.float64)
        dist.all_reduce(ar, op=dist.ReduceOp.MAX)
        nbytes = bucket.numel() * bucket.element_size()
        allreduce_info = {"bytes": nbytes, "ms": float(ar.item()),
                          "algbw_GBps": nbytes / (float(ar.item()) * 1e-3) / 1e9,
                          "busbw_GBps": nbytes / (float(ar.item()) * 1e-3) / 1e9 * 2 * (world - 1) / world,
                          "xgmi_link_peak_GBps": 153.0}

    if rank == 0:
        # instance count of this view (the units the blend / sort kernels process)
        from splatco_amd import rasterizer as R
        with torch.no_grad():
            _, _, st = R.rasterize_forward(R._CSettings(rast.raster_settings), params["means3D"].detach(),
                                           params["opacities"].detach(), params["scales"].detach(),
                                           params["rotations"].detach(), None, None, params["colors_precomp"].detach())
        I, npix = st.I, W * H
        kern = dict(warm_kern)                                   # all classes: from the warm-up steps
        kern.update({k: (ms / max(n, 1)) for k, (ms, n) in prof.items() if n})   # dominant: timed region
        dom = dominant
        ab = algorithmic_bytes(dom, P, I, npix)
        achieved = ab / (kern[dom] * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")   # PMC-derived bytes per launch, if measured
        if os.path.exists(tpath):
            traffic = json.load(open(tpath)).get(dom)
        # VALU issue view of the same kernel: wave64 VALU instructions per launch (PMC SQ_INSTS_VALU, if
        # measured) x 2 issue cycles (SIMD-32) over the SIMD-cycles of its launch at the 2.4 GHz peak clock
        valu = None
        vpath = os.path.join(ROOT, "profiles", "valu_insts.json")
        if os.path.exists(vpath):
            n_inst = json.load(open(vpath)).get(dom)
            if n_inst:
                valu = {"insts_per_launch": n_inst, "issue_frac": n_inst * 2.0 / (256 * 4 * 2.4e9 * kern[dom] * 1e-3)}
        ms_per_step = elapsed / args.steps * 1e3
        out = {
            "metric": METRIC, "value": world * P / (elapsed / args.steps) / 1e6, "unit": "Msplats/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "cfg1: 1M synthetic Gaussians (seed 0), 1 view 1920x1080 per GPU, "
                                   "GaussianRasterizer forward+backward, colors_precomp + scale/rotation path",
                       "gaussians": P, "image": f"{W}x{H}", "tile_instances": I,
                       "visible": int((radii > 0).sum().item()),
                       "parallelism": "1 view per GPU (mv sharding)" + (
                           f", RCCL all-reduce(SUM) of {P}x14 fp32 per-Gaussian grads per step" if world > 1 else "")},
            "roofline": {"kernel": dom, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": ab, "avg_launch_ms": kern[dom], "valu": valu,
                         "note": "blend kernels are FP32-VALU/exp-issue bound at this density (SURVEY.md 8d); "
                                 "the HBM fraction is reported as the contract asks"},
            "kernel_ms": {k: round(v, 4) for k, v in sorted(kern.items(), key=lambda kv: -kv[1])},
            "hbm_gbs_all_kernels": sum(algorithmic_bytes(k, P, I, npix) for k in kern) / (sum(kern.values()) * 1e-3) / 1e9,
        }
        if allreduce_info is not None:
            out["allreduce"] = allreduce_info
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"], out["psnr_match_db"] = cpu_baseline(g, cam, dev)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
