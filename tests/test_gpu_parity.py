"""GPU parity tests: the HIP path (through the C-ABI, via the host operator) against the CPU oracle.

Bars (SURVEY.md 8c / BASELINE.md 2):
  * integers -- radii, tiles_touched, point_offsets, ranges, point_list: bit-exact;
    n_contrib: bit-exact wherever no exp()-dependent decision of the pixel lies within 1e-5
    (relative) of its threshold (the spec allows exp() 2 ulp), and >= 99.9 % of pixels overall;
  * image: max-abs <= 1e-4 and PSNR(build, oracle) >= 80 dB (utils/image_utils.py:17-19);
  * gradients: rel-L2 <= 1e-4 per tensor vs the fp32 oracle, checked UNCONDITIONALLY: pixels whose
    n_contrib differs from the oracle's (exp() within 2 ulp of a threshold: at most 0.1 % of the image, the
    count is printed) get dL/dpixel = 0 on BOTH sides, everything else is compared;
    bit-reproducible run to run.
"""
import math
import os

import numpy as np
import pytest
import torch

from util import oracle_settings, psnr, rel_l2, small_scene, stress_scene
from splatco_amd.synthetic import synthetic_camera, synthetic_gaussians

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "the gpu tests need an MI355X"
    return torch.device("cuda:0")


def _settings(cam, bg, scale_modifier=1.0, sh_degree=1, debug=False):
    from splatco_amd.rasterizer import GaussianRasterizationSettings
    d = _dev()
    return GaussianRasterizationSettings(
        image_height=cam.image_height, image_width=cam.image_width, tanfovx=math.tan(cam.FoVx * 0.5),
        tanfovy=math.tan(cam.FoVy * 0.5), bg=torch.tensor(bg, dtype=torch.float32, device=d),
        scale_modifier=scale_modifier, viewmatrix=cam.world_view_transform.to(d),
        projmatrix=cam.full_proj_transform.to(d), sh_degree=sh_degree, campos=cam.camera_center.to(d),
        prefiltered=False, debug=debug)


def _t(a, grad=False):
    return None if a is None else torch.tensor(np.asarray(a), dtype=torch.float32, device=_dev(), requires_grad=grad)


EXCUSED_MAX = 1e-3     # largest fraction of pixels whose n_contrib may differ from the oracle's (threshold pixels)


def _run_gpu(cam, g, scale_modifier=1.0, sh_degree=1, shs=None, cov=None, dL=None, debug=False, ref=None):
    """Forward (+ backward when dL is given) through the operator; returns numpy results.
    ref: the oracle's forward results.  Pixels whose n_contrib differs from ref's are EXCUSED from the gradient
    comparison by zeroing dL/dpixel there (out["dL_eff"] is what the backward ran on -- the oracle's backward must
    be given the same array); their count is bounded and reported."""
    from splatco_amd import rasterizer as R
    from splatco_amd import _C
    rs = _settings(cam, g["bg"], scale_modifier, sh_degree, debug)
    grad = dL is not None
    m = _t(g["means3D"], grad)
    o = _t(g["opacities"], grad)
    s = _t(g["scales"], grad) if cov is None else None
    r = _t(g["rotations"], grad) if cov is None else None
    cv = _t(cov, grad)
    c = _t(g["colors"], grad) if shs is None else None
    sh = _t(shs, grad)
    m2d = torch.zeros_like(m, requires_grad=True) + 0
    if grad:
        m2d.retain_grad()
    # call the autograd function directly so that the saved state is reachable for the getters
    cs = R._CSettings(rs)
    out = {}
    with torch.no_grad():
        color, radii, st = R.rasterize_forward(cs, m.detach(), o.detach(), None if s is None else s.detach(),
                                               None if r is None else r.detach(), None if cv is None else cv.detach(),
                                               None if sh is None else sh.detach(), None if c is None else c.detach())
    out["color"], out["radii"], out["num_rendered"] = color.cpu().numpy(), radii.cpu().numpy(), st.I
    out["tiles_touched"] = st.debug(_C.DBG_TILES_TOUCHED).cpu().numpy().view(np.uint32)
    out["ranges"] = st.debug(_C.DBG_RANGES).cpu().numpy().view(np.uint32)
    out["n_contrib"] = st.debug(_C.DBG_N_CONTRIB).cpu().numpy().view(np.uint32)
    out["final_T"] = st.debug(_C.DBG_FINAL_T).cpu().numpy()
    if st.I:
        out["point_offsets"] = st.debug(_C.DBG_POINT_OFFSETS).cpu().numpy().view(np.uint32)
        out["point_list"] = st.debug(_C.DBG_POINT_LIST).cpu().numpy().view(np.uint32)
    if grad:
        dL_eff = np.array(dL, dtype=np.float32, copy=True)
        if ref is not None:
            excused = out["n_contrib"] != ref["n_contrib"]
            dL_eff[:, excused] = 0.0
            out["excused"] = int(excused.sum())
            print(f"[parity] pixels excused from the gradient comparison (n_contrib differs at a threshold): "
                  f"{out['excused']} of {excused.size} ({excused.mean():.2e})")
            assert excused.mean() <= EXCUSED_MAX, f"{out['excused']} pixels differ in n_contrib"
        out["dL_eff"] = dL_eff
        rast = R.GaussianRasterizer(rs)
        img, rad2 = rast(means3D=m, means2D=m2d, opacities=o, shs=sh, colors_precomp=c, scales=s, rotations=r,
                         cov3D_precomp=cv)
        same = lambda a, b: torch.equal(a.isnan(), b.isnan()) and torch.equal(a.nan_to_num(nan=0.0), b.nan_to_num(nan=0.0))
        assert torch.equal(rad2, radii) and same(img, color)  # deterministic forward (NaN where NaN: non-finite input tests)
        (img * _t(dL_eff)).sum().backward()
        gr = dict(means3D=m.grad, means2D=m2d.grad, opacities=o.grad)
        if cov is None:
            gr.update(scales=s.grad, rotations=r.grad)
        else:
            gr.update(cov3D_precomp=cv.grad)
        gr.update(sh=sh.grad) if shs is not None else gr.update(colors_precomp=c.grad)
        out["grads"] = {k: v.cpu().numpy() for k, v in gr.items()}
    torch.cuda.synchronize()
    return out


def _check_forward(f, o, st, full_ncontrib=True):
    assert np.array_equal(o["radii"], f["radii"]), "radii"
    assert np.array_equal(o["tiles_touched"], f["tiles_touched"]), "tiles_touched"
    assert o["num_rendered"] == f["num_rendered"]
    assert np.array_equal(o["ranges"].astype(np.int64)[f["ranges"][:, 1] > f["ranges"][:, 0]],
                          f["ranges"].astype(np.int64)[f["ranges"][:, 1] > f["ranges"][:, 0]]), "ranges"
    empty = f["ranges"][:, 1] == f["ranges"][:, 0]
    assert np.all((o["ranges"][:, 1] == o["ranges"][:, 0])[empty])
    if f["num_rendered"]:
        assert np.array_equal(o["point_offsets"].astype(np.uint64), f["point_offsets"]), "point_offsets"
        assert np.array_equal(o["point_list"], f["point_list"]), "point_list (sorted ids)"
    # margin = the pixel's smallest relative distance of an exp()-dependent decision (alpha vs 1/255,
    # T' vs 1e-4 -- for ANY splat of its list, not only the last) from its threshold; the spec gives
    # exp() 2 ulp, so only pixels with margin > 1e-5 are required to take identical decisions
    safe = f["margin"] > 1e-5
    assert np.array_equal(o["n_contrib"][safe], f["n_contrib"][safe]), "n_contrib away from thresholds"
    assert (o["n_contrib"] == f["n_contrib"]).mean() >= 0.999
    cmp_px = safe & (o["n_contrib"] == f["n_contrib"])
    print(f"[parity] image compared on {cmp_px.mean():.4%} of the pixels ({(~safe).sum()} within 1e-5 of an exp() "
          f"threshold, {(o['n_contrib'] != f['n_contrib']).sum()} with a different n_contrib)")
    assert cmp_px.mean() >= 0.98
    assert np.abs(o["color"] - f["color"])[:, cmp_px].max() <= 1e-4, "image max-abs"
    assert np.abs(o["final_T"] - f["final_T"])[cmp_px].max() <= 1e-5
    assert psnr(o["color"], f["color"]) >= 80.0


GRAD_TOL = 1e-4       # rel-L2 per gradient tensor vs the fp32 oracle


def _check_grads(gg, b, names, tol=GRAD_TOL):
    errs = {n: rel_l2(gg[n], b[n]) for n in names}
    print("[parity] gradient rel-L2 vs the fp32 oracle: " + ", ".join(f"{n} {e:.2e}" for n, e in errs.items()))
    for n in names:
        assert gg[n].shape == b[n].shape, n
        assert errs[n] <= tol, (n, errs[n])


def test_visible_filter_bit_exact(oracle):
    from splatco_amd.rasterizer import GaussianRasterizer
    for cam, g in (small_scene(P=5000, W=200, H=120, spread=3.0),
                   (synthetic_camera(400, 400), synthetic_gaussians(10_000, 400, 400, 0))):
        st = oracle_settings(oracle, cam, g["bg"])
        want = oracle.visible_filter(st, g["means3D"], g["scales"], g["rotations"])
        rast = GaussianRasterizer(_settings(cam, g["bg"]))
        got = rast.visible_filter(means3D=_t(g["means3D"]), scales=_t(g["scales"]), rotations=_t(g["rotations"]))
        assert got.dtype == torch.int32
        assert np.array_equal(got.cpu().numpy(), want)
        assert 0 < (want > 0).sum()
        vis = rast.markVisible(_t(g["means3D"]))
        assert np.array_equal(vis.cpu().numpy(), oracle.mark_visible(st, g["means3D"]))


@pytest.mark.parametrize("scene", ["small_offaxis", "cfg0_10k_400x400", "ragged_130x70", "many_tiles_2064x2050",
                                   "huge_tiles_3360x3104"])
def test_forward_backward_colors_path(oracle, scene):
    if scene == "small_offaxis":
        cam, g = small_scene(P=400, W=96, H=64, spread=1.5)
    elif scene == "cfg0_10k_400x400":
        cam, g = synthetic_camera(400, 400), synthetic_gaussians(10_000, 400, 400, 0)
    elif scene == "many_tiles_2064x2050":  # 16641 tiles: per-tile LDS histogram beyond the default 64 KB of dynamic LDS
        cam, g = synthetic_camera(2064, 2050), synthetic_gaussians(20_000, 2064, 2050, 9)
    elif scene == "huge_tiles_3360x3104":  # 40740 tiles > LDS_HIST_MAX_TILES: the global-atomic counting fallback
        cam, g = synthetic_camera(3360, 3104), synthetic_gaussians(20_000, 3360, 3104, 10)
    else:  # image size not a multiple of the tile size
        cam, g = synthetic_camera(130, 70), synthetic_gaussians(1500, 130, 70, 4)
    st = oracle_settings(oracle, cam, g["bg"])
    f = oracle.forward(st, g["means3D"], g["opacities"], g["scales"], g["rotations"], colors_precomp=g["colors"])
    rng = np.random.default_rng(1)
    dL = rng.standard_normal((3, cam.image_height, cam.image_width)).astype(np.float32)
    o = _run_gpu(cam, g, dL=dL, ref=f)
    _check_forward(f, o, st)
    b = oracle.backward(st, f, o["dL_eff"], g["means3D"], g["scales"], g["rotations"], colors_precomp=g["colors"])
    _check_grads(o["grads"], b, ["means3D", "means2D", "colors_precomp", "opacities", "scales", "rotations"])
    assert np.all(o["grads"]["means2D"][:, 2] == 0)
    # determinism: a second run is bit-identical (no floating-point atomics anywhere)
    o2 = _run_gpu(cam, g, dL=dL, ref=f)
    for k in o["grads"]:
        assert np.array_equal(o["grads"][k], o2["grads"][k]), k
    assert np.array_equal(o["color"], o2["color"])


@pytest.mark.parametrize("scene", ["cfg0_10k_400x400", "ragged_130x70", "opaque_pile", "merge_path_tile", "large_rects"])
def test_deep_list_variants_forced_on_small_scenes(oracle, scene):
    """The variants the library selects for deep tile lists (more than 8192 entries per tile on average: 20 M anchors) --
    the tile sort without the gm_index array, the blend backward deriving a record's place from gm_base, per-Gaussian
    record flags that let preprocess_backward skip Gaussians without any record -- forced on scenes the oracle can check
    (scr_debug_force_deep_lists): same integers, same image, gradients to 1e-4, and bit-identical to the default variants
    (the skipped terms are +0)."""
    from splatco_amd import _C
    rng = np.random.default_rng(3)
    if scene == "cfg0_10k_400x400":
        cam, g = synthetic_camera(400, 400), synthetic_gaussians(10_000, 400, 400, 0)
    elif scene == "ragged_130x70":
        cam, g = synthetic_camera(130, 70), synthetic_gaussians(1500, 130, 70, 4)
    elif scene == "large_rects":
        # faint discs whose tile rects hold hundreds of tiles (260 in the image), most of them out of reach of the
        # alpha >= 1/255 ellipse: the per-tile record verdicts beyond a rect's first 32 tiles do not fit live_bits -- the
        # plan reports SCR_PLAN_LARGE_RECTS, scr_backward clears those records and preprocess_backward sums them all
        cam, g = synthetic_camera(320, 200), synthetic_gaussians(1200, 320, 200, 6)
        big = rng.choice(1200, 300, replace=False)
        g["scales"][big] *= rng.uniform(8, 40, (300, 1)).astype(np.float32) * np.array([[1.0, 0.5, 0.7]], np.float32)
        g["opacities"][big] = rng.uniform(0.01, 0.3, (300, 1)).astype(np.float32)     # faint: the reachable ellipse is a fraction of the 3-sigma rect
    else:
        # many opaque splats over a few tiles: the pixels finish after a few dozen entries of lists of thousands, so most
        # rounds are cut and most Gaussians never get a record ("merge_path_tile": one tile beyond a sort chunk of 8192)
        P = 6000 if scene == "opaque_pile" else 20000
        cam, g = synthetic_camera(96, 64), synthetic_gaussians(P, 96, 64, seed=8)
        tx, ty = math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2)
        z = rng.uniform(2, 6, P).astype(np.float32)
        span = 40 if scene == "opaque_pile" else 12
        px, py = rng.uniform(30, 30 + span, P), rng.uniform(20, 20 + span * 0.6, P)
        g["means3D"] = np.stack([((2 * px + 1) / 96 - 1) * tx * z, ((2 * py + 1) / 64 - 1) * ty * z, z], 1).astype(np.float32)
        g["scales"] = (np.full((P, 3), 0.02, np.float32) * z[:, None]).astype(np.float32)
        g["opacities"] = np.full((P, 1), 0.9, np.float32)
    st = oracle_settings(oracle, cam, g["bg"])
    f = oracle.forward(st, g["means3D"], g["opacities"], g["scales"], g["rotations"], colors_precomp=g["colors"])
    dL = rng.standard_normal((3, cam.image_height, cam.image_width)).astype(np.float32)
    base = _run_gpu(cam, g, dL=dL, ref=f)
    _C.check(_C.lib.scr_debug_force_deep_lists(1))
    try:
        o = _run_gpu(cam, g, dL=dL, ref=f)
    finally:
        _C.check(_C.lib.scr_debug_force_deep_lists(-1))
    _check_forward(f, o, st)
    b = oracle.backward(st, f, o["dL_eff"], g["means3D"], g["scales"], g["rotations"], colors_precomp=g["colors"])
    _check_grads(o["grads"], b, ["means3D", "means2D", "colors_precomp", "opacities", "scales", "rotations"])
    assert np.array_equal(o["color"], base["color"]) and np.array_equal(o["point_list"], base["point_list"])
    for k in o["grads"]:
        assert np.array_equal(o["grads"][k], base["grads"][k]), k
    if scene == "large_rects":
        # debug mode (pipe.debug): every kernel -- the far-record clearing included -- is followed by a stream synchronisation
        # and an error check; same bits
        dbg = _run_gpu(cam, g, dL=dL, ref=f, debug=True)
        assert np.array_equal(dbg["color"], base["color"]) and all(np.array_equal(dbg["grads"][k], base["grads"][k]) for k in base["grads"])
        from splatco_amd import rasterizer as R
        t = lambda a: torch.tensor(a, device=_dev())
        _, _, stt = R.rasterize_forward(R._CSettings(_settings(cam, g["bg"])), t(g["means3D"]), t(g["opacities"]), t(g["scales"]),
                                        t(g["rotations"]), None, None, t(g["colors"]))
        assert stt.flags & _C.PLAN_LARGE_RECTS
        qm = stt.debug(_C.DBG_QMASK).cpu().numpy()
        print(f"[deep] large_rects: {(f['tiles_touched'] > 32).sum()} Gaussians with more than 32 tiles (largest {f['tiles_touched'].max()}), "
              f"{(qm == 0).mean():.0%} of the instances out of reach of every quadrant")
        assert (f["tiles_touched"] > 32).sum() >= 100 and (qm == 0).mean() > 0.3
    elif scene != "cfg0_10k_400x400" and scene != "ragged_130x70":
        tile_n = f["ranges"][:, 1].astype(np.int64) - f["ranges"][:, 0]
        assert tile_n.max() > (8192 if scene == "merge_path_tile" else 1024)
        never = (o["grads"]["opacities"][:, 0] == 0) & (f["radii"] > 0)
        print(f"[deep] {scene}: largest tile {tile_n.max()}, visible Gaussians without any gradient: {never.sum()} of {(f['radii'] > 0).sum()}")
        assert never.sum() > 0.3 * (f["radii"] > 0).sum()            # the flags have something to skip


def test_sh_and_cov_paths(oracle):
    cam, g = small_scene(P=300, W=96, H=64, spread=1.5, seed=8)
    rng = np.random.default_rng(5)
    dL = rng.standard_normal((3, 64, 96)).astype(np.float32)
    shs = (rng.standard_normal((300, 16, 3)) * 0.4).astype(np.float32)
    for deg in (0, 1, 2, 3):
        st = oracle_settings(oracle, cam, g["bg"], scale_modifier=0.9, sh_degree=deg)
        f = oracle.forward(st, g["means3D"], g["opacities"], g["scales"], g["rotations"], shs=shs)
        o = _run_gpu(cam, g, scale_modifier=0.9, sh_degree=deg, shs=shs, dL=dL, ref=f)
        _check_forward(f, o, st)
        b = oracle.backward(st, f, o["dL_eff"], g["means3D"], g["scales"], g["rotations"], shs=shs)
        _check_grads(o["grads"], b, ["means3D", "means2D", "sh", "opacities", "scales", "rotations"])
    A = rng.standard_normal((300, 3, 3)) * 0.15
    S = A @ A.transpose(0, 2, 1) + 1e-3 * np.eye(3)
    cov = np.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], 1).astype(np.float32)
    st = oracle_settings(oracle, cam, g["bg"])
    f = oracle.forward(st, g["means3D"], g["opacities"], cov3D_precomp=cov, colors_precomp=g["colors"])
    o = _run_gpu(cam, g, cov=cov, dL=dL, ref=f)
    _check_forward(f, o, st)
    b = oracle.backward(st, f, o["dL_eff"], g["means3D"], cov3D_precomp=cov, colors_precomp=g["colors"])
    _check_grads(o["grads"], b, ["means3D", "means2D", "colors_precomp", "opacities", "cov3D_precomp"])


def test_edge_cases(oracle):
    from splatco_amd.rasterizer import GaussianRasterizer
    cam = synthetic_camera(64, 48)
    bg = np.array([0.3, 0.4, 0.5], np.float32)
    rast = GaussianRasterizer(_settings(cam, bg, debug=True))
    d = _dev()
    # empty input: background, no launch
    z = lambda *s: torch.zeros(*s, device=d)
    img, radii = rast(means3D=z(0, 3), means2D=z(0, 3), opacities=z(0, 1), colors_precomp=z(0, 3), scales=z(0, 3),
                      rotations=z(0, 4))
    assert radii.numel() == 0 and torch.allclose(img[:, 3, 5].cpu(), torch.tensor(bg))
    # everything culled (behind the camera): I = 0
    m = torch.tensor([[0.0, 0.0, -5.0], [0.0, 0.0, 0.1]], device=d, requires_grad=True)
    img, radii = rast(means3D=m, means2D=torch.zeros_like(m), opacities=torch.full((2, 1), 0.5, device=d),
                      colors_precomp=torch.ones(2, 3, device=d), scales=torch.full((2, 3), 0.1, device=d),
                      rotations=torch.tensor([[1.0, 0, 0, 0]] * 2, device=d))
    assert radii.tolist() == [0, 0] and torch.allclose(img[:, 10, 10].cpu(), torch.tensor(bg))
    img.sum().backward()
    assert torch.all(m.grad == 0)
    # argument validation: the reference operator's error convention
    with pytest.raises(Exception):
        rast(means3D=z(1, 3), means2D=z(1, 3), opacities=z(1, 1), scales=z(1, 3), rotations=z(1, 4))
    with pytest.raises(Exception):
        rast(means3D=z(1, 3), means2D=z(1, 3), opacities=z(1, 1), colors_precomp=z(1, 3))


def test_debug_mode_dumps_the_inputs_of_a_failed_call(tmp_path, monkeypatch):
    """settings.debug (pipe.debug, --debug_from): a failing operator call leaves its inputs in snapshot_fw.dump before
    the error propagates; without debug nothing is written."""
    from splatco_amd.rasterizer import GaussianRasterizer
    monkeypatch.chdir(tmp_path)
    cam = synthetic_camera(64, 48)
    d = _dev()
    z = lambda *s: torch.rand(*s, device=d)
    for debug in (False, True):
        rast = GaussianRasterizer(_settings(cam, np.zeros(3, np.float32), sh_degree=7, debug=debug))   # degree 7: refused by the C-ABI
        with pytest.raises(RuntimeError, match="sh_degree"):
            rast(means3D=z(5, 3), means2D=z(5, 3), opacities=z(5, 1), shs=z(5, 16, 3), scales=z(5, 3), rotations=z(5, 4))
        assert os.path.exists("snapshot_fw.dump") == debug
    blob = torch.load("snapshot_fw.dump", weights_only=False)
    assert blob["means3D"].shape == (5, 3) and blob["sh"].shape == (5, 16, 3) and blob["raster_settings"]["sh_degree"] == 7


@pytest.mark.parametrize("P", [1500, 3000, 9000])
def test_depth_ties_and_huge_tile(oracle, P):
    """Exact depth ties (stable order by id) and one tile holding more instances than one sort
    chunk (1024): exercises the chunk sort + 1..4 rank-merge passes, odd and even pass counts,
    a partner-less last run."""
    cam = synthetic_camera(96, 64)
    rng = np.random.default_rng(12)
    g = synthetic_gaussians(P, 96, 64, seed=6)
    # pile everything onto the neighbourhood of one pixel, tiny footprints, few distinct depths
    tx, ty = math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2)
    z = rng.choice(np.linspace(2, 6, 37), P).astype(np.float32)
    px, py = rng.uniform(34, 44, P), rng.uniform(34, 44, P)
    g["means3D"] = np.stack([((2 * px + 1) / 96 - 1) * tx * z, ((2 * py + 1) / 64 - 1) * ty * z, z], 1).astype(np.float32)
    g["scales"] = np.full((P, 3), 0.004, np.float32) * z[:, None]
    g["opacities"] = np.full((P, 1), 0.02, np.float32)
    st = oracle_settings(oracle, cam, g["bg"])
    f = oracle.forward(st, g["means3D"], g["opacities"], g["scales"], g["rotations"], colors_precomp=g["colors"])
    assert (f["ranges"][:, 1].astype(np.int64) - f["ranges"][:, 0]).max() > 0.9 * P
    dL = rng.standard_normal((3, 64, 96)).astype(np.float32)
    o = _run_gpu(cam, g, dL=dL, ref=f)
    _check_forward(f, o, st)
    b = oracle.backward(st, f, o["dL_eff"], g["means3D"], g["scales"], g["rotations"], colors_precomp=g["colors"])
    _check_grads(o["grads"], b, ["means3D", "means2D", "colors_precomp", "opacities", "scales", "rotations"])


@pytest.mark.parametrize("P", [65, 129, 257, 320, 321, 384, 385, 513, 576, 577, 640, 641, 768, 769, 1000, 1024, 1025, 1100])
def test_tile_sizes_around_every_wave_sort_path(oracle, P):
    """One tile holding (about) P instances for P on both sides of every size at which the one-wave tile sort changes its
    path (a single network of 64 E keys, or two networks + an LDS merge for tiles just above a power of two:
    binning.hip wave_sort_any): sorted lists bit-exact."""
    cam = synthetic_camera(96, 64)
    rng = np.random.default_rng(P)
    g = synthetic_gaussians(P, 96, 64, seed=3)
    tx, ty = math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2)
    z = rng.uniform(2, 6, P).astype(np.float32)
    z[rng.integers(0, P, P // 8)] = 3.0            # some exact depth ties (order by id)
    px, py = rng.uniform(37, 42, P), rng.uniform(37, 42, P)      # inside tile (2, 2), footprints of about a pixel
    g["means3D"] = np.stack([((2 * px + 1) / 96 - 1) * tx * z, ((2 * py + 1) / 64 - 1) * ty * z, z], 1).astype(np.float32)
    g["scales"] = np.full((P, 3), 0.002, np.float32) * z[:, None]
    g["opacities"] = np.full((P, 1), 0.02, np.float32)
    st = oracle_settings(oracle, cam, g["bg"])
    f = oracle.forward(st, g["means3D"], g["opacities"], g["scales"], g["rotations"], colors_precomp=g["colors"])
    counts = f["ranges"][:, 1].astype(np.int64) - f["ranges"][:, 0]
    assert counts.max() == P, "the scene is meant to put every Gaussian into one tile"
    o = _run_gpu(cam, g, ref=f)
    _check_forward(f, o, st)


@pytest.mark.parametrize("P", [1, 63, 4097])
def test_odd_sizes_and_giant_splats(oracle, P):
    """P = 1, P not a multiple of any workgroup shape, and Gaussians that cover the whole image
    (tiles_touched = every tile; long per-thread tile loops, every quadrant mask bit set)."""
    cam = synthetic_camera(400, 304)
    g = synthetic_gaussians(P, 400, 304, seed=21)
    n_big = min(P, 5)
    g["scales"][:n_big] = np.array([3.0, 2.0, 0.5], np.float32)     # hundreds of pixels across
    g["opacities"][:n_big] = 0.3
    st = oracle_settings(oracle, cam, g["bg"])
    f = oracle.forward(st, g["means3D"], g["opacities"], g["scales"], g["rotations"], colors_precomp=g["colors"])
    assert f["tiles_touched"][:n_big].max() >= 0.5 * st.grid[0] * st.grid[1]
    rng = np.random.default_rng(4)
    dL = rng.standard_normal((3, 304, 400)).astype(np.float32)
    o = _run_gpu(cam, g, dL=dL, ref=f)
    _check_forward(f, o, st)
    b = oracle.backward(st, f, o["dL_eff"], g["means3D"], g["scales"], g["rotations"], colors_precomp=g["colors"])
    _check_grads(o["grads"], b, ["means3D", "means2D", "colors_precomp", "opacities", "scales", "rotations"])


def test_full_size_cfg1_1M_1080p(oracle):
    """BASELINE.json configs[1]: 1M Gaussians, 1920x1080, forward + backward, against the oracle
    (about a minute of single-thread CPU), plus size-independent properties."""
    W, H, P = 1920, 1080, 1_000_000
    cam, g = synthetic_camera(W, H), synthetic_gaussians(P, W, H, 0)
    st = oracle_settings(oracle, cam, g["bg"])
    f = oracle.forward(st, g["means3D"], g["opacities"], g["scales"], g["rotations"], colors_precomp=g["colors"])
    rng = np.random.default_rng(1)
    dL = rng.standard_normal((3, H, W)).astype(np.float32)
    o = _run_gpu(cam, g, dL=dL, ref=f)
    _check_forward(f, o, st)
    # properties: sortedness of every tile list by (depth, id); checksum of per-tile ids
    depth = f["depth"].astype(np.float32).view(np.uint32).astype(np.uint64)
    key = (depth[o["point_list"]] << np.uint64(32)) | o["point_list"].astype(np.uint64)
    tile_of = np.repeat(np.arange(o["ranges"].shape[0]), (o["ranges"][:, 1] - o["ranges"][:, 0]).astype(np.int64))
    full = (tile_of.astype(np.uint64) << np.uint64(52)) ^ key  # tiles ascending, then key ascending
    assert np.all(np.diff(tile_of) >= 0)
    same_tile = tile_of[1:] == tile_of[:-1]
    assert np.all(key[1:][same_tile] > key[:-1][same_tile])
    del full
    b = oracle.backward(st, f, o["dL_eff"], g["means3D"], g["scales"], g["rotations"], colors_precomp=g["colors"])
    _check_grads(o["grads"], b, ["means3D", "means2D", "colors_precomp", "opacities", "scales", "rotations"])
    # linearity of the backward in dL/dcolor: grads(2*dL) == 2*grads(dL) exactly (power-of-two scale)
    o2 = _run_gpu(cam, g, dL=2 * dL, ref=f)
    for k in ("means3D", "opacities", "colors_precomp"):
        assert np.array_equal(o2["grads"][k], 2 * o["grads"][k]), k


STRESS_NAMES = ["means3D", "means2D", "colors_precomp", "opacities", "scales", "rotations"]
EPS32 = 2.0 ** -24     # unit roundoff of binary32


def _fp64_on_fp32_records(oracle, st, f):
    """The fp64 oracle evaluated on the fp32 oracle's per-Gaussian records (screen position, conic, colour -- the
    48-byte splat records, which the device reproduces bit for bit) and integer decisions (visibility, tile lists,
    sort order): blend forward, blend backward and the projection chain in binary64.  What remains between this and
    an fp32 evaluation is exactly what the device and the fp32 oracle are free to do differently: the rounding of
    the per-pixel terms, the order of the sums, the rounding inside the chain."""
    pre = {k: (f[k].astype(np.float64) if f[k].dtype == np.float32 else f[k])
           for k in ("radii", "xy", "depth", "cov3D", "conic_opacity", "rgb", "clamped", "tiles_touched", "rect")}
    bins = {k: f[k] for k in ("point_offsets", "num_rendered", "keys_sorted", "point_list", "ranges")}
    out = dict(pre)
    out.update(bins)
    out.update(oracle.blend_forward(st, pre, bins, f64=True))
    return out


CERTIFY = 1e-5         # a gradient row is "pinned by binary32" when three independent signals all say so (row-relative):
                       #  (1) its CONDITIONING: the fp64 gradient of the row moves by less than this when the binary32
                       #      factors the backward multiplies and sums are moved by one ulp (_one_ulp_response);
                       #  (2) the scalar fp32 oracle (sequential sums, centred moments) lands this close to fp64;
                       #  (3) so does the fp32 oracle in the DEVICE's formulation (raw moments per 8x8 quadrant, shifted).
ULP32 = 2.0 ** -23
# Rounds 3-4 used signal (2) alone: a row can satisfy it by chance, and 1 of 350 scenes failed on such rows while the device
# was 3x closer to fp64 than the oracle overall (profiles/r04_stress_more.txt).  Signal (1) alone is blind to the roundings
# INSIDE an evaluation (a needle's determinant a c - b^2 cancels between two rounded products; sums over 10^4 pixels round at
# every step): with it alone 54 of 350 scenes failed on rows where BOTH fp32 evaluations sat at 1e-4 .. 3e-3 from fp64
# (gpurun log r05_stress_350, first criterion).  A row is certified only when its mathematics and both fp32 witnesses agree;
# every signal can only REMOVE rows from the certified set, and the size of what is left is bounded below (UNPINNED_MAX).
# On the certified rows the device must be within GRAD_TOL of the fp64 gradient; on the others: (device's distance from fp64)
# / max(fp32 oracle's distance, the one-ulp response there, the 2-ulp noise floor of the fp32 projection chain), L2 over the
# set.  profiles/r05_stress.txt holds the 350 scenes under this rule.
UNPINNED_RATIO_MAX = 3.0


def _one_ulp_response(oracle, st, f64, dL_eff, g, b64, trials=6):
    """Per tensor, per Gaussian: the largest row-relative change of the fp64 gradient over `trials` random one-ulp
    (binary32, random sign) perturbations of the factors the backward multiplies and sums: the splat records' conic,
    opacity and colour, the forward's transmittances, dL/dpixel, and the Gaussian's own means / scales / quaternion.  A
    backward-stable fp32 evaluation returns the exact gradient of inputs perturbed like this; a row whose exact gradient
    barely moves is one binary32 can pin, a row that moves by per cent is one no fp32 evaluation can."""
    prng = np.random.default_rng(2024)
    rel = lambda a_: np.asarray(a_, np.float64) * (1.0 + ULP32 * prng.choice([-1.0, 1.0], np.shape(a_)))
    step = lambda a_: np.nextafter(a_.astype(np.float32), np.where(prng.random(a_.shape) < 0.5, -np.inf, np.inf).astype(np.float32))
    resp, dist = {}, {}
    for _ in range(trials):
        fp = dict(f64)
        for k in ("conic_opacity", "rgb", "final_T"):      # not the screen position: pixel - xy is formed without rounding
            fp[k] = rel(f64[k])                            # beyond an ulp of the DIFFERENCE, and the fp64 yardstick uses the same xy
        bp = oracle.backward(st, fp, rel(dL_eff), step(g["means3D"]), step(g["scales"]), step(g["rotations"]),
                             colors_precomp=g["colors"], f64=True)
        for k in STRESS_NAMES:
            d = np.linalg.norm(np.asarray(bp[k], np.float64).reshape(len(b64[k]), -1) - b64[k].reshape(len(b64[k]), -1), axis=1)
            resp[k] = np.maximum(resp.get(k, 0.0), d)
    return resp        # absolute row norms of the change


def stress_case(oracle, rng, verbose=False, info=None):
    """One randomised scene against the oracle.  Forward: the bars of _check_forward.  Gradients, per tensor:
      (a) rel-L2 <= 1e-4 vs the fp32 oracle over ALL Gaussians -- or, where that fails,
      (b) the bar, against the fp64 evaluation, over the Gaussians whose gradient binary32 pins at all (CERTIFY above: the
          row's conditioning and two independent fp32 evaluations agree).  The stress set contains needle-like Gaussians
          (anisotropy up to 300:1 over hundreds of tiles) whose gradient sums cancel to a few per cent and whose
          covariance chain divides by a vanishing determinant: ANY fp32 evaluation -- the scalar fp32 oracle included --
          is 1e-3 .. 1e-2 off on those rows, so no fp32 tolerance can hold there.  The fraction of such rows is
          printed and bounded, and on them the device must still be no further from fp64 (L2 over the set) than
          UNPINNED_RATIO_MAX times the largest of (i) the fp32 oracle's own distance, (ii) what the one-ulp perturbations
          did to the fp64 gradient there and (iii) the noise floor of the fp32 projection chain (the oracle's own chain with
          its inputs, the per-Gaussian screen-space sums, jiggled by +-2 ulp).  A wrong term is O(1) on EVERY row, rounding is not.
    Pixels whose n_contrib differs between any two of the three evaluations are excused on all sides (bounded,
    printed).  Returns a report line; raises AssertionError otherwise."""
    cam, g, sm = stress_scene(rng)
    st = oracle_settings(oracle, cam, g["bg"], scale_modifier=sm)
    args = (g["means3D"], g["opacities"], g["scales"], g["rotations"])
    f = oracle.forward(st, *args, colors_precomp=g["colors"])
    dL = rng.standard_normal((3, cam.image_height, cam.image_width)).astype(np.float32)
    o = _run_gpu(cam, g, scale_modifier=sm, dL=dL, ref=f)
    _check_forward(f, o, st)
    dL_eff = o["dL_eff"]
    b = oracle.backward(st, f, dL_eff, g["means3D"], g["scales"], g["rotations"], colors_precomp=g["colors"])
    e32 = {k: rel_l2(o["grads"][k], b[k]) for k in STRESS_NAMES}
    worst = max(e32.values())
    note, fails = "", []
    if worst > GRAD_TOL or verbose:
        f64 = _fp64_on_fp32_records(oracle, st, f)
        more = (f64["n_contrib"] != f["n_contrib"]) & (dL_eff != 0).any(axis=0)
        if more.any():       # threshold pixels of the fp64 blend: excuse them on all three sides and redo
            assert (more.sum() + o["excused"]) / more.size <= EXCUSED_MAX
            dL2 = dL.copy()
            dL2[:, (o["n_contrib"] != f["n_contrib"]) | (f64["n_contrib"] != f["n_contrib"])] = 0.0
            o = _run_gpu(cam, g, scale_modifier=sm, dL=dL2)
            dL_eff = o["dL_eff"]
            b = oracle.backward(st, f, dL_eff, g["means3D"], g["scales"], g["rotations"], colors_precomp=g["colors"])
            e32 = {k: rel_l2(o["grads"][k], b[k]) for k in STRESS_NAMES}
        b64 = oracle.backward(st, f64, dL_eff, g["means3D"], g["scales"], g["rotations"], colors_precomp=g["colors"], f64=True)
        vis = f["radii"] > 0
        # noise floor of the fp32 chain: +-2 ulp relative jiggles (random signs) of the fp32 screen-space sums, three trials
        gm2, gconic, gop, gcol = oracle.blend_backward(st, f, dL_eff)
        prng = np.random.default_rng(12345)
        jig = lambda a_: (a_.astype(np.float64) * (1.0 + 4.0 * EPS32 * prng.choice([-1.0, 1.0], a_.shape))).astype(np.float32)
        trials = [oracle.preprocess_backward(st, f, jig(gm2), jig(gconic), gcol, g["means3D"], g["scales"], g["rotations"])
                  for _ in range(3)]
        b64 = {k: np.asarray(b64[k], np.float64) for k in STRESS_NAMES}
        response = _one_ulp_response(oracle, st, f64, dL_eff, g, b64)
        rm2, rconic, _, rcol = oracle.blend_backward(st, f, dL_eff, raw_moments=True)
        braw = oracle.preprocess_backward(st, f, rm2, rconic, rcol, g["means3D"], g["scales"], g["rotations"])
        for k in STRESS_NAMES:
            if e32[k] <= GRAD_TOL and not verbose:
                continue
            dev, o32, ref = o["grads"][k].astype(np.float64), b[k].astype(np.float64), b64[k]
            # the three signals of CERTIFY (rows without a gradient: certified)
            rn = np.linalg.norm(ref.reshape(len(ref), -1), axis=1)
            row = lambda a_: np.linalg.norm((np.asarray(a_, np.float64) - ref).reshape(len(ref), -1), axis=1)
            pinned = (response[k] <= CERTIFY * rn) & (row(o32) <= CERTIFY * rn)
            if braw.get(k) is not None:      # ... and the fp32 oracle in the DEVICE's formulation of the screen-space sums (raw
                pinned &= row(braw[k]) <= CERTIFY * rn       # moments per 8x8 quadrant, shifted to the splat's centre) is a third witness
            e_pinned, e_pinned_o32 = rel_l2(dev[pinned], ref[pinned]), rel_l2(o32[pinned], ref[pinned])
            loose = ~pinned
            d_dev, d_o32 = np.linalg.norm(dev[loose] - ref[loose]), np.linalg.norm(o32[loose] - ref[loose])
            d_noise = max(float(np.linalg.norm((tr[k].astype(np.float64) - o32)[loose])) for tr in trials) \
                if trials[0].get(k) is not None else 0.0
            d_noise = max(d_noise, float(np.linalg.norm(response[k][loose])))
            note += (f" {k}: {e32[k]:.1e} over all rows; {loose.sum()} of {vis.sum()} visible rows not pinned by fp32 "
                     f"(device {d_dev / max(np.linalg.norm(ref[loose]), 1e-300):.1e}, fp32 oracle "
                     f"{d_o32 / max(np.linalg.norm(ref[loose]), 1e-300):.1e}, one-ulp response / noise floor "
                     f"{d_noise / max(np.linalg.norm(ref[loose]), 1e-300):.1e} from fp64 there), pinned rows: device {e_pinned:.1e}, "
                     f"fp32 oracle {e_pinned_o32:.1e} from fp64;")
            if e32[k] > GRAD_TOL:
                ratio = d_dev / max(d_o32, d_noise, 1e-300)
                if info is not None:
                    info.setdefault("branch_b", {})[k] = {"all_rows": e32[k], "unpinned_rows": int(loose.sum()),
                                                           "visible_rows": int(vis.sum()), "pinned_rel_l2": e_pinned,
                                                           "ratio": ratio, "ratio_vs_oracle": d_dev / max(d_o32, 1e-300)}
                if not e_pinned <= GRAD_TOL:
                    fails.append((k, "rows binary32 pins (one-ulp response <= 1e-5): device vs fp64", e_pinned))
                if not ratio <= UNPINNED_RATIO_MAX:
                    fails.append((k, "ill-conditioned rows vs fp64: device", d_dev, "fp32 oracle", d_o32, "noise floor", d_noise, "ratio", ratio))
    ranges = f["ranges"]
    if info is not None:
        info.update({"P": int(g["means3D"].shape[0]), "visible": int((f["radii"] > 0).sum()), "worst": float(worst),
                     "image": f"{cam.image_width}x{cam.image_height}", "I": int(f["num_rendered"])})
    line = (f"P={g['means3D'].shape[0]} {cam.image_width}x{cam.image_height} I={f['num_rendered']} "
            f"max tile={(ranges[:, 1].astype(np.int64) - ranges[:, 0]).max()} vis={(f['radii'] > 0).sum()} "
            f"worst grad {worst:.1e}" + (" |" + note if note else ""))
    assert not fails, (fails, line)
    return line


# How far the stress set may lean on branch (b) of stress_case.  Measured over 350 scenes (seeds 0-34) on the round-5 kernels
# under the three-signal rule (profiles/r05_stress.txt): 168 scenes use the branch for at least one tensor, at most 7 of the ten
# scenes of a seed; the rows binary32 does not pin are 8 % of the visible Gaussians in the median and at most 36 % in scenes with
# 1000 or more of them, up to 54 % in the ten tiny ones (a few hundred Gaussians, most of them needles).  The bars below are those
# maxima with head-room; a kernel change that pushes more of the set into the branch fails here instead of passing silently.
UNPINNED_MAX = 0.42            # of the visible Gaussians, scenes with >= 1000 of them
UNPINNED_MAX_SMALL = 0.62      # scenes with fewer
B_SCENES_MAX = 8               # of the ten scenes of a seed


@pytest.mark.parametrize("seed", [0, 1, 2, 3, 4])
def test_randomised_stress_scenes(oracle, seed):
    """50 seeded scenes of the randomised stress set (tools/stress_parity.py runs more of the same).  Every scene's
    outcome -- worst rel-L2 over all rows, which tensors needed branch (b), how many rows fp32 does not pin -- goes to
    the report file named by SPLATCO_STRESS_REPORT (committed as profiles/r04_stress.txt), and the use of branch (b) is
    bounded by asserts."""
    rng = np.random.default_rng(seed)
    lines, b_scenes, over, worst_ratio = [], 0, [], 0.0
    for it in range(10):
        info = {}
        print(f"[stress {seed}/{it}] " + stress_case(oracle, rng, info=info))
        bb = info.get("branch_b", {})
        frac = max((v["unpinned_rows"] / max(v["visible_rows"], 1) for v in bb.values()), default=0.0)
        b_scenes += bool(bb)
        lines.append(f"seed {seed} scene {it}: P={info['P']} {info['image']} I={info['I']} visible={info['visible']} worst rel-L2 over all rows "
                     f"{info['worst']:.2e} -> " + ("branch (a): 1e-4 over all rows" if not bb else
                     "branch (b) for " + ", ".join(f"{k} (all rows {v['all_rows']:.1e}, pinned rows {v['pinned_rel_l2']:.1e}, "
                                                   f"{v['unpinned_rows']} of {v['visible_rows']} rows unpinned, there device / max(oracle, noise) "
                                                   f"= {v['ratio']:.2f}, device / oracle = {v['ratio_vs_oracle']:.2f})" for k, v in bb.items())))
        worst_ratio = max([worst_ratio] + [v["ratio"] for v in bb.values()])
        over.append((it, frac)) if frac > (UNPINNED_MAX if info["visible"] >= 1000 else UNPINNED_MAX_SMALL) else None
    lines.append(f"seed {seed}: {b_scenes} of 10 scenes used branch (b); worst device / max(oracle, noise) ratio on unpinned rows {worst_ratio:.2f}")
    path = os.environ.get("SPLATCO_STRESS_REPORT")
    if path:
        with open(path, "a") as fh:
            fh.write("\n".join(lines) + "\n")
    assert not over, (seed, "scenes whose fraction of visible rows not pinned by binary32 exceeds the bar", over)
    assert b_scenes <= B_SCENES_MAX, (seed, b_scenes)


def _same_with_nonfinite(a, b, tol, what, exact=True):
    """a == b where NaN must sit where NaN sits and +-Inf where the same Inf sits (exact=False: a value that is not finite
    where one is not finite -- sums over pixels of +-Inf terms are Inf or NaN depending on their order); the finite rest
    within tol (max-abs relative to the largest finite magnitude)."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    if exact:
        assert np.array_equal(np.isnan(a), np.isnan(b)), (what, "NaN positions", int(np.isnan(a).sum()), int(np.isnan(b).sum()))
        inf = np.isinf(b)
        assert np.array_equal(np.isinf(a), inf) and np.array_equal(a[inf], b[inf]), (what, "Inf positions")
    else:
        bad = np.isfinite(a) != np.isfinite(b)
        assert not bad.any(), (what, "finite on one side only", int(bad.sum()), np.argwhere(bad)[:5].tolist(), a[bad][:5], b[bad][:5])
    fin = np.isfinite(b)
    if fin.any():
        scale = max(np.abs(b[fin]).max(), 1e-30)
        assert np.abs(a[fin] - b[fin]).max() <= tol * scale, (what, float(np.abs(a[fin] - b[fin]).max() / scale))


@pytest.mark.parametrize("case", ["nan_colour_hidden", "nan_colour_partly_visible", "inf_colour", "inf_opacity", "nan_opacity",
                                  "nan_mean_scale_rotation"])
def test_non_finite_inputs_follow_the_reference_skip_semantics(oracle, case):
    """The reference SKIPS a splat at every pixel it does not contribute to (alpha < 1/255, power > 0, pixel already
    opaque), so a colour that is NaN / Inf reaches only the pixels its splat contributes to.  The fast blend kernels carry
    non-contributing splats with alpha 0; calls whose plan reports SCR_PLAN_NONFINITE_COLOUR run the select-based
    instantiations instead.  Device == oracle, NaN for NaN and Inf for Inf, forward and every gradient; Gaussians with a
    NaN mean / scale / rotation are culled on both sides."""
    from splatco_amd import _C, rasterizer as R
    cam, g = small_scene(P=400, W=160, H=96, seed=11)
    g = {k: np.array(v, copy=True) for k, v in g.items()}
    rng = np.random.default_rng(5)
    f0 = oracle.forward(oracle_settings(oracle, cam, g["bg"]), g["means3D"], g["opacities"], g["scales"], g["rotations"],
                        colors_precomp=g["colors"])
    vis = np.nonzero(f0["radii"] > 0)[0]
    order = vis[np.argsort(f0["depth"][vis])]
    front, back = order[: len(order) // 8], order[-len(order) // 3:]
    expect_flag = False
    if case == "nan_colour_hidden":
        # an opaque wall in front of everything, NaN colours far behind it: no pixel of the wall's footprint reaches them
        wall = front[0]
        g["means3D"][wall] = [0.1, 0.05, -2.0]
        g["scales"][wall] = [3.0, 3.0, 0.01]
        g["rotations"][wall] = [1, 0, 0, 0]
        g["opacities"][wall] = 1.0
        g["colors"][back[:20], 1] = np.nan
        expect_flag = True
    elif case == "nan_colour_partly_visible":
        g["colors"][order[::7], rng.integers(0, 3, len(order[::7]))] = np.nan
        expect_flag = True
    elif case == "inf_colour":
        g["colors"][order[3::9], 0] = np.inf
        g["colors"][order[5::11], 2] = -np.inf
        expect_flag = True
    elif case == "inf_opacity":
        g["opacities"][order[4::10]] = np.inf
    elif case == "nan_opacity":
        g["opacities"][order[2::10]] = np.nan
    else:
        g["means3D"][order[0], 1] = np.nan
        g["scales"][order[1], 2] = np.nan
        g["rotations"][order[2], 0] = np.nan
        g["scales"][order[3], 0] = np.inf
    st = oracle_settings(oracle, cam, g["bg"])
    f = oracle.forward(st, g["means3D"], g["opacities"], g["scales"], g["rotations"], colors_precomp=g["colors"])
    dL = rng.standard_normal((3, cam.image_height, cam.image_width)).astype(np.float32)
    o = _run_gpu(cam, g, dL=dL, ref=f, debug=(case == "inf_colour"))      # (one case also under pipe.debug: the SAFE kernels with a sync + check behind each)
    assert np.array_equal(o["radii"], f["radii"]) and np.array_equal(o["tiles_touched"], f["tiles_touched"])
    assert o["num_rendered"] == f["num_rendered"] and np.array_equal(o["point_list"], f["point_list"])
    if case == "nan_mean_scale_rotation":
        assert (o["radii"][order[:3]] == 0).all(), "NaN mean / scale / rotation: culled"
    same_n = o["n_contrib"] == f["n_contrib"]
    assert same_n.mean() >= 0.999
    _same_with_nonfinite(o["color"][:, same_n], f["color"][:, same_n], 1e-4, "image")
    if case == "nan_colour_hidden":
        wall_px = f["n_contrib"] <= (np.nonzero(f["point_list"] == wall)[0].size and f["n_contrib"])    # all: documentation
        assert np.isfinite(f["color"]).mean() > 0.5, "the wall hides the NaN colours from most pixels in the reference semantics"
    b = oracle.backward(st, f, o["dL_eff"], g["means3D"], g["scales"], g["rotations"], colors_precomp=g["colors"])
    odd = ~np.isfinite(g["opacities"]).reshape(-1)
    if odd.any():
        # The one documented difference: the gradient with respect to an opacity that is itself NaN / Inf.  The reference sums
        # G dL/dalpha (finite); the device sums Y = (opacity G) dL/dalpha for all its moments and divides the total by the
        # opacity once per Gaussian (blend.hip / preprocess_backward_kernel): Inf / Inf.  Every other output -- the image, the
        # other Gaussians' gradients, this Gaussian's other gradients -- is compared below.
        contributing = odd & (np.abs(b["opacities"]).reshape(-1) > 0)
        assert np.isnan(o["grads"]["opacities"].reshape(-1)[contributing]).all() and contributing.any()
        o["grads"]["opacities"] = np.where(odd[:, None], b["opacities"], o["grads"]["opacities"])
    for k in STRESS_NAMES:
        # max-abs over the tensor (rows mix 1e-6 .. 1 magnitudes).  NaN colours: NaN exactly where the oracle has NaN; with
        # infinities in play a per-Gaussian SUM over pixels of +Inf and -Inf terms is Inf or NaN depending on the order
        _same_with_nonfinite(o["grads"][k], b[k], 2e-3, k, exact=case.startswith("nan_colour") or case.startswith("nan_mean"))
    # the plan flag is what selected the kernels
    rs = _settings(cam, g["bg"])
    t = lambda a: torch.tensor(a, device=_dev())
    _, _, stt = R.rasterize_forward(R._CSettings(rs), t(g["means3D"]), t(g["opacities"]), t(g["scales"]), t(g["rotations"]), None,
                                    None, t(g["colors"]))
    assert bool(stt.flags & _C.PLAN_NONFINITE_COLOUR) == expect_flag


def test_plan_flags_are_reported_by_any_lane(oracle):
    """The plan flags are raised by ONE lane per wave: it must be an active one.  Every Gaussian that sits in lane 0 of its
    wave is culled here (behind the camera) and a single Gaussian in the middle of a wave carries the NaN colour / the large
    rect: the flags must still arrive."""
    from splatco_amd import _C, rasterizer as R
    cam, g = small_scene(P=512, W=320, H=200, seed=12)
    g = {k: np.array(v, copy=True) for k, v in g.items()}
    g["means3D"][::64] = [0.0, 0.0, -50.0]                      # lane 0 of every wave: behind the camera
    g["scales"] *= 0.2                                          # no rect of more than 32 tiles to begin with
    t = lambda a: torch.tensor(a, device=_dev())
    run = lambda: R.rasterize_forward(R._CSettings(_settings(cam, g["bg"])), t(g["means3D"]), t(g["opacities"]), t(g["scales"]),
                                      t(g["rotations"]), None, None, t(g["colors"]))[2]
    st = run()
    assert st.flags == 0 and (st.radii.cpu().numpy()[::64] == 0).all()
    f = oracle.forward(oracle_settings(oracle, cam, g["bg"]), g["means3D"], g["opacities"], g["scales"], g["rotations"],
                       colors_precomp=g["colors"])
    vis = np.nonzero((f["radii"] > 0) & (np.arange(512) % 64 > 8))[0]
    g["colors"][vis[3], 1] = np.nan
    assert run().flags == _C.PLAN_NONFINITE_COLOUR
    g["scales"][vis[10]] = [2.0, 1.5, 1.0]                      # a splat over most of the 260 tiles
    assert run().flags == _C.PLAN_NONFINITE_COLOUR | _C.PLAN_LARGE_RECTS
    g["colors"][vis[3], 1] = 0.5
    assert run().flags == _C.PLAN_LARGE_RECTS


def test_backward_takes_the_large_rect_verdict_from_the_forward_not_from_its_argument():
    """scr_backward sums records 32.. of a large rect unconditionally; whether they were cleared first used to depend on
    the plan_flags ARGUMENT (a caller handing back 0, or another forward's flags, had uninitialised scratch summed into
    the gradients without an error).  The verdict is now read from geom_buf on the device: a backward with stale flags
    gives the same bits; debug mode refuses the mismatch."""
    from splatco_amd import _C
    from splatco_amd.rasterizer import GaussianRasterizer
    dev = _dev()
    cam, g = small_scene(P=512, W=320, H=200, seed=12)
    g = {k: np.array(v, copy=True) for k, v in g.items()}
    g["scales"] *= 0.2
    g["scales"][[40, 41, 300]] = [2.0, 1.5, 1.0]                # three splats over most of the 260 tiles
    dL = torch.tensor(np.random.default_rng(3).standard_normal((3, 200, 320)).astype(np.float32), device=dev)

    def run(stale, debug=False):
        ins = [_t(g[k], True) for k in ("means3D", "opacities", "colors", "scales", "rotations")]
        m2d = torch.zeros(512, 3, device=dev, requires_grad=True)
        img, _ = GaussianRasterizer(_settings(cam, g["bg"], debug=debug))(
            means3D=ins[0], means2D=m2d, opacities=ins[1], colors_precomp=ins[2], scales=ins[3], rotations=ins[4])
        st = img.grad_fn.state
        assert st.flags == _C.PLAN_LARGE_RECTS
        if stale:
            st.flags = 0
        # whatever the backward's scratch allocation returns has NaN in it
        poison = torch.full((64 << 20,), float("nan"), device=dev)
        del poison
        (img * dL).sum().backward()
        return [t.grad.clone() for t in ins + [m2d]]

    want = run(False)
    got = run(True)
    for a, b in zip(got, want):
        assert torch.isfinite(a).all() and torch.equal(a, b)
    with pytest.raises(RuntimeError, match="plan_flags"):
        run(True, debug=True)
