"""GPU tests at the sizes BASELINE.json names beyond configs[1]:

  configs[2]  5 M anchors + tri-plane features (plane_size 2800, 15 channels, activate_level 2), 1 view 1080p:
              prefilter_voxel radii bit-exact vs the oracle at N = 5 M; the ~15 M neural Gaussians the anchor path
              produces go through the rasterizer AND the oracle (OpenMP build: the forward is per-pixel independent, so
              its integers do not depend on the thread count; the backward sums with atomics, far inside the 1e-4 bar):
              every integer bit-exact, image, gradients; then the whole render() path forward + backward (finite
              gradients on every parameter, bit-reproducible image).
  configs[3]/[4] (multi-GPU) cannot run on the one GPU of the test box; their code path -- bench.py starting N ranks
              itself, the sharded step, the in-place gradient exchange, the statistics broadcast -- runs here with two
              ranks on ONE device over gloo at a reduced anchor count.
"""
import json
import math
import os
import subprocess
import sys
import types

import numpy as np
import pytest
import torch

from util import oracle_settings
from test_gpu_parity import _check_forward, _check_grads, _run_gpu

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cfg2_5M_anchors_plane2800(oracle):
    from splatco_amd.renderer import generate_neural_gaussians, prefilter_voxel, render
    from splatco_amd.synthetic import ANCHOR_CONFIGS, synthetic_anchor_model, synthetic_views
    dev = torch.device("cuda:0")
    N, _, seed = ANCHOR_CONFIGS["cfg2"]
    assert N == 5_000_000
    pc = synthetic_anchor_model(N, seed, dev, plane_size=2800, num_channels=15, activate_level=2)
    assert pc.feat_planes._feat.k0s[3].xy_plane.shape[-1] == 2800 and pc.feat_planes._feat.activate_level == 2
    cam = synthetic_views(1)[0]
    pipe = types.SimpleNamespace(debug=False, compute_cov3D_python=False)
    bg = torch.ones(3, device=dev)
    st = oracle_settings(oracle, cam, np.ones(3, np.float32))
    oracle.use_threads(True)
    try:
        # ---- a2: anchor visibility at N = 5 M, bit-exact (same fp32 inputs on both sides)
        vis = prefilter_voxel(cam.to(dev), pc, pipe, bg)
        with torch.no_grad():
            want = oracle.visible_filter(st, pc.get_anchor.cpu().numpy(), pc.get_scaling[:, :3].cpu().numpy(),
                                         pc.get_rotation.cpu().numpy())
            from splatco_amd.rasterizer import GaussianRasterizer
            from splatco_amd.renderer import _settings
            got = GaussianRasterizer(_settings(cam.to(dev), bg, 1.0, False)).visible_filter(
                means3D=pc.get_anchor, scales=pc.get_scaling[:, :3], rotations=pc.get_rotation)
        assert np.array_equal(got.cpu().numpy(), want), "visible_filter radii at N = 5M"
        assert np.array_equal(vis.cpu().numpy(), want > 0)
        V = int(vis.sum())
        assert V > 4_000_000
        # ---- a3: the anchor path; a5/a6 on its output against the oracle
        with torch.no_grad():
            xyz, color, opacity, scaling, rot, neural_opacity, mask = generate_neural_gaussians(cam.to(dev), pc, vis, is_training=True)
        P = xyz.shape[0]
        print(f"[cfg2] visible anchors {V}, neural Gaussians {P}")
        assert P > 10_000_000 and int(mask.sum()) == P
        g = dict(means3D=xyz.cpu().numpy(), scales=scaling.cpu().numpy(), rotations=rot.cpu().numpy(),
                 opacities=opacity.cpu().numpy(), colors=color.cpu().numpy(), bg=np.ones(3, np.float32))
        del xyz, color, opacity, scaling, rot, neural_opacity, mask
        f = oracle.forward(st, g["means3D"], g["opacities"], g["scales"], g["rotations"], colors_precomp=g["colors"])
        tile_n = f["ranges"][:, 1].astype(np.int64) - f["ranges"][:, 0]
        print(f"[cfg2] tile instances {f['num_rendered']}, largest tile {tile_n.max()} (merge-path passes above 4096)")
        assert f["num_rendered"] > 15_000_000 and tile_n.max() > 4096
        dL = np.random.default_rng(2).standard_normal((3, cam.image_height, cam.image_width)).astype(np.float32)
        o = _run_gpu(cam, g, dL=dL, ref=f)
        _check_forward(f, o, st)
        b = oracle.backward(st, f, o["dL_eff"], g["means3D"], g["scales"], g["rotations"], colors_precomp=g["colors"])
        _check_grads(o["grads"], b, ["means3D", "means2D", "colors_precomp", "opacities", "scales", "rotations"])
        del f, b, o, g
    finally:
        oracle.use_threads(False)
    # ---- a4: the whole render() path at this size, forward + backward
    pc.train()
    out = render(cam.to(dev), pc, pipe, bg, visible_mask=vis, retain_grad=True)
    target = torch.rand(3, cam.image_height, cam.image_width, device=dev)
    ((out["render"] - target).abs().mean() + 0.01 * out["scaling"].prod(dim=1).mean()).backward()
    assert out["render"].shape == (3, 1080, 1920) and torch.isfinite(out["render"]).all()
    gvs = out["viewspace_points"].grad
    assert gvs is not None and torch.isfinite(gvs).all() and torch.all(gvs[:, 2] == 0) and gvs[:, :2].abs().sum() > 0
    planes = pc.feat_planes._feat.k0s
    for name, p in [("anchor", pc._anchor), ("offset", pc._offset), ("feat", pc._anchor_feat), ("scaling", pc._scaling),
                    ("mlp_cov", pc.mlp_cov[0].weight), ("plane L0 (attention)", planes[0].xy_plane),
                    ("plane L1", planes[1].xz_plane), ("plane L2 (1400)", planes[2].yz_plane)]:
        assert p.grad is not None and torch.isfinite(p.grad).all() and p.grad.abs().sum() > 0, name
    assert planes[3].xy_plane.grad is None            # the 2800^2 level is never sampled
    with torch.no_grad():
        again = render(cam.to(dev), pc, pipe, bg, visible_mask=vis)["render"]
    assert torch.equal(again, out["render"].detach())                     # bit-reproducible at 15 M Gaussians


def test_training_loop_learns_a_teacher_scene():
    """Rows a1 - a8 together, as train.py:171-311 strings them: mv = 2 views drawn at random from six, targets rendered from a
    TEACHER scene of another seed, collaborative_step (prefilter, render, L1 + SSIM + scaling regulariser, the cross-view
    consistency term from iteration 30 on, one backward, the total-variation term every 4th iteration, densification
    statistics, FusedAdam) for 120 iterations with adjust_anchor (grow + prune) at 100 -- the image error must FALL (the
    400-iteration soak of tools/exp/soak_train.py goes from 18.7 to 29 - 30 dB, profiles/r05_soak_train.txt), every parameter
    stays finite, the anchor set changes and the loop keeps stepping on the new tensors, and once the allocator's pool is
    warm no step asks the device for memory."""
    import random
    from splatco_amd.adam import FusedAdam
    from splatco_amd.densify import AnchorDensifier
    from splatco_amd.renderer import prefilter_voxel, render
    from splatco_amd.synthetic import synthetic_anchor_model, synthetic_views
    from splatco_amd.train_step import collaborative_step
    dev = torch.device("cuda:0")
    W, H, N = 480, 270, 100_000
    pipe = types.SimpleNamespace(debug=False, compute_cov3D_python=False)
    bg = torch.ones(3, device=dev)
    views = [v.to(dev) for v in synthetic_views(6, W, H)]
    teacher = synthetic_anchor_model(N, 101, dev, plane_size=512)
    teacher.eval()
    with torch.no_grad():
        gts = [render(v, teacher, pipe, bg, visible_mask=prefilter_voxel(v, teacher, pipe, bg))["render"].clamp(0, 1).clone() for v in views]
    del teacher
    pc = synthetic_anchor_model(N, 7, dev, plane_size=512)
    groups = [{"params": [getattr(pc, "_" + n)], "lr": lr, "name": n}
              for n, lr in (("anchor", 0.0), ("offset", 1e-3), ("anchor_feat", 7.5e-3), ("scaling", 7e-3))]
    groups.append({"params": [p for n, p in pc.named_parameters() if not n.startswith("_") and p.requires_grad], "lr": 2e-3,
                   "name": "mlp_and_feat_planes"})
    opt = FusedAdam(groups, eps=1e-15)
    den = AnchorDensifier(pc, opt, voxel_size=0.01, seed=3)
    rng = random.Random(0)

    def psnr(i):
        with torch.no_grad():
            img = render(views[i], pc, pipe, bg, visible_mask=prefilter_voxel(views[i], pc, pipe, bg))["render"].clamp(0, 1)
            return float(10 * torch.log10(1.0 / ((img - gts[i]) ** 2).mean()))

    before = [psnr(i) for i in range(6)]
    first, last, allocs = [], [], None
    for it in range(1, 121):
        pick = rng.sample(range(6), 2)
        loss, out, _ = collaborative_step(pc, [views[i] for i in pick], [gts[i] for i in pick], pipe, bg, optimizer=opt, densifier=den,
                                          consistency_weight=0.05 if it > 30 else 0.0, iteration=it, tv_weight=4e-7)
        (first if it <= 10 else last if it > 110 else []).append(loss.detach())
        if it == 60:
            allocs = torch.cuda.memory_stats(dev)["num_device_alloc"]
        if it == 99:
            # (what this guards against is one allocation PER STEP -- 39 here -- as the drifting instance count caused before
            # large scratch came in size classes; a handful may still happen when a size class is crossed)
            assert torch.cuda.memory_stats(dev)["num_device_alloc"] - allocs <= 8, "steady-state steps keep asking the device for memory"
        if it == 100:
            den.adjust_anchor(iteration=100, check_interval=100, grad_threshold=0.0002)
            assert pc._anchor.shape[0] != N, "adjust_anchor neither grew nor pruned: the statistics did not arrive"
    after = [psnr(i) for i in range(6)]
    print(f"[teacher scene] PSNR of the six views {[round(b, 1) for b in before]} -> {[round(a, 1) for a in after]} dB; "
          f"loss {float(torch.stack(first).mean()):.4f} -> {float(torch.stack(last).mean()):.4f}; anchors {N} -> {pc._anchor.shape[0]}")
    assert all(bool(torch.isfinite(p).all()) for p in pc.parameters())
    assert float(torch.stack(last).mean()) < 0.6 * float(torch.stack(first).mean())
    assert sum(after) / 6 > sum(before) / 6 + 3.0 and min(a - b for a, b in zip(after, before)) > 1.0


def test_bench_sharded_optimizer_and_dry_run():
    """`bench.py --gpus 2 --config cfg3 --exchange rs_ag --optimizer sharded` (adam.ShardedFusedAdam: reduce-scatter, Adam on
    this rank's half, all-gather of the parameters) prints its line; `--dry-run-ranks` prints the collective sequence of a
    step instead, identical on both ranks, and no timing."""
    env = dict(os.environ, SPLATCO_BENCH_ONE_DEVICE="1", SPLATCO_BENCH_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "cfg3", "--anchors", "200000",
            "--exchange", "rs_ag", "--optimizer", "sharded"]
    r = subprocess.run(base + ["--steps", "2", "--warmup", "2"], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["value"] > 0 and "ShardedFusedAdam" in out["config"]["optimizer"]
    assert out["allreduce"]["mode"] == "rs_ag" and out["allreduce"]["ms"] > 0 and "exchange" not in out
    assert "every 4th step" in out["config"]["tv"] and out["tv_pass"]["ms"] > 0
    d = subprocess.run(base + ["--dry-run-ranks"], capture_output=True, text=True, env={k: v for k, v in env.items() if not k.startswith("SPLATCO_BENCH")},
                       timeout=900)
    assert d.returncode == 0, d.stdout[-2000:] + d.stderr[-4000:]
    seqs = [json.loads(l[l.index('{"dry_run_ranks"'):]) for l in d.stdout.splitlines() if '{"dry_run_ranks"' in l]
    assert len(seqs) == 2 and all(q["identical_on_all_ranks"] and q["dry_run_ranks"] == 2 for q in seqs)
    names = {e["collective"] for e in seqs[0]["sequence"]}
    assert names == {"reduce_scatter_tensor", "all_gather_into_tensor", "broadcast"}, names
    n_rs = sum(e["count"] for e in seqs[0]["sequence"] if e["collective"] == "reduce_scatter_tensor")
    n_ag = sum(e["count"] for e in seqs[0]["sequence"] if e["collective"] == "all_gather_into_tensor")
    assert n_rs == n_ag > 0 and "UNMEASURED" in seqs[0]["status"] and not any(l.startswith('{"metric"') for l in d.stdout.splitlines())


@pytest.mark.parametrize("config", ["cfg1", "cfg3"])
def test_bench_starts_its_own_ranks(config):
    """`python bench.py --gpus 2` with no launcher starts two ranks itself (configs[3]/[4] code path; both ranks on the
    one device of this box, gloo instead of RCCL, reduced anchor count) and prints ONE JSON line with the all-reduce
    bookkeeping."""
    env = dict(os.environ, SPLATCO_BENCH_ONE_DEVICE="1", SPLATCO_BENCH_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "2", "--config", config]
    if config != "cfg1":
        cmd += ["--anchors", "200000"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["allreduce"]["bytes"] > 0 and out["allreduce"]["ms"] > 0
    assert out["roofline"]["achieved"] >= 0 and config in out["config"]["workload"]
    # what a first multi-GPU run is judged by: bus rate against one xGMI link and against all of them, the exchange the
    # step could not hide (the same steps once more with the collectives switched off), who took part
    ar, ex = out["allreduce"], out["exchange"]
    assert ar["busbw_GBps"] > 0 and ar["frac_of_one_link"] > 0 and ar["frac_of_all_links"] > 0 and ar["xgmi_link_peak_GBps"] == 153.0
    assert ex["ms_per_step_with_exchange"] > 0 and ex["ms_per_step_without_exchange"] > 0
    assert abs(ex["exposed_ms"] - (ex["ms_per_step_with_exchange"] - ex["ms_per_step_without_exchange"])) < 1e-9
    assert out["ranks"] == {"world_size": 2, "backend": "gloo", "rccl_version": None, "devices": out["ranks"]["devices"]}
    assert "cfg2" not in out and "cpu_baseline" not in out          # N = 1 only
    if config != "cfg1":
        assert ar["units"] >= 2 and ar["anchor_ranges"] >= 1 and ar["issue_order_agreed"] is True
    # the launcher's own timeout: ranks that do not finish are killed and the exit code says so (124), never a held node
    slow = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", config, "--timeout", "0.5"]
                          + (["--anchors", "200000"] if config != "cfg1" else []), capture_output=True, text=True, env=env, timeout=300)
    assert slow.returncode == 124 and "did not finish" in slow.stderr
    # a launcher that started another number of ranks is an error, not a silent N = 1 run
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                         env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), timeout=300)
    assert bad.returncode != 0 and "WORLD_SIZE" in (bad.stdout + bad.stderr)
