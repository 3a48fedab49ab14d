#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by IMPORTING the reference's own Python
(/root/reference, build container only -- the GPU box has no reference) and recording seeded
inputs -> outputs.  Only arrays are written; no reference source text is stored.

Recipe = SURVEY.md Appendix B: stub the native modules that are absent from the reference tree
(diff_gaussian_rasterization, torch_scatter, simple_knn._C, plyfile, cv2, _gridcreater,
_gridencoder, kornia, jaxtyping), make .cuda() the identity, and patch the one helper that
hard-codes device='cuda' (FeaturePlanes.get_offsets, scene/gaussian_model.py:171-180).

Fixtures (all float32 / int, a few hundred KB in total):
  cameras.npz    getWorld2View2 / getProjectionMatrix / camera tensors for 3 seeded poses
                 (utils/graphics_utils.py:38-71, scene/cameras.py:54-58)
  losses.npz     l1_loss / ssim / psnr on seeded [3,64,64] pairs (utils/loss_utils.py, utils/image_utils.py)
  planegrid.npz  PlaneGrid with and without TriPlaneAttention, [1000,3] -> [1000,15] / [1000,30]
                 (scene/grids.py:102-201)
  neural_gaussians.npz  generate_neural_gaussians, N=512 anchors, plane_size=40, num_channels=15,
                 appearance_dim=0, Q0=0, train and eval variants, activate_level 0 and 2
                 (gaussian_renderer/__init__.py:18-116) + every state_dict it needs
  training_statis.npz  GaussianModel.training_statis on small seeded masks (scene/gaussian_model.py:761-782)
  neural_gaussians_app.npz  the same call with the reference's CODE defaults that the README command line switches off:
                 appearance_dim = 32 (arguments/__init__.py:76; per-camera code on the colour head,
                 gaussian_renderer/__init__.py:55-58,76-80, scene/embedding.py); N = 256 anchors, camera uid 3 of 5
  ply_layout.npz what the reference's save_ply (scene/gaussian_model.py:640-673) hands to plyfile for the model of
                 neural_gaussians.npz: the structured array's field names, its bytes and the element name, recorded by
                 a stand-in for PlyElement.describe / PlyData.write (plyfile is not installed here); and what the
                 reference's load_ply_sparse_gaussian (:675-712) makes of that element
  tv.npz         PlaneGrid.total_variation_add_grad (scene/grids.py:240-250) with and without TriPlaneAttention, a
                 cubic 24^3 grid and a 37 x 19 x 30 one (odd row lengths), plane values scaled so that neighbour
                 differences fall on both sides of the smooth-L1 knee, into empty and into existing .grad; and
                 GaussianLearner.tv_loss (scene/gaussian_model.py:217-220) at activate_level 0 and 2
  densify.npz    GaussianModel.adjust_anchor (anchor_growing + prune_anchor, optimizer state surgery) and
                 compute_curvature on small seeded models (scene/gaussian_model.py:784-997,1092-1110)
"""
import argparse
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, path))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def install_stubs():
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)

    def stub(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    class _Dummy:
        def __init__(self, *a, **k):
            pass

    class _Shaped:
        def __class_getitem__(cls, item):
            return cls

    stub("diff_gaussian_rasterization", GaussianRasterizationSettings=_Dummy, GaussianRasterizer=_Dummy)
    stub("torch_scatter", scatter_max=None)
    stub("simple_knn")
    stub("simple_knn._C", distCUDA2=None)
    stub("plyfile", PlyData=_Dummy, PlyElement=_Dummy)
    stub("cv2")
    stub("_gridcreater")
    stub("_gridencoder")
    stub("kornia", create_meshgrid=None)
    stub("jaxtyping", Shaped=_Shaped)
    torch.nn.Module.cuda = lambda self, *a, **k: self
    torch.Tensor.cuda = lambda self, *a, **k: self


def f32(t):
    return t.detach().cpu().numpy().astype(np.float32)


def sd_np(prefix, module):
    return {f"{prefix}{k}": v.detach().cpu().numpy() for k, v in module.state_dict().items()}


def make_cameras():
    gu = _load("utils/graphics_utils.py", "ref_graphics_utils")
    rng = np.random.default_rng(0)
    out = {}
    for i in range(3):
        q = rng.standard_normal(4)
        q /= np.linalg.norm(q)
        w, x, y, z = q
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                      [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                      [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
        T = rng.uniform(-2, 2, 3)
        trans = rng.uniform(-0.5, 0.5, 3) if i == 2 else np.zeros(3)
        scale = 1.5 if i == 2 else 1.0
        fovx, fovy = float(rng.uniform(0.6, 1.4)), float(rng.uniform(0.5, 1.2))
        w2v = gu.getWorld2View2(R, T, trans, scale)
        proj = gu.getProjectionMatrix(znear=0.01, zfar=100.0, fovX=fovx, fovY=fovy)
        wvt = torch.tensor(w2v).transpose(0, 1)                       # scene/cameras.py:54
        pm = proj.transpose(0, 1)                                     # :55
        full = wvt.unsqueeze(0).bmm(pm.unsqueeze(0)).squeeze(0)       # :56
        center = wvt.inverse()[3, :3]                                 # :58
        out.update({f"R{i}": R, f"T{i}": T, f"trans{i}": trans, f"scale{i}": np.float64(scale),
                    f"fovx{i}": np.float64(fovx), f"fovy{i}": np.float64(fovy), f"w2v{i}": w2v,
                    f"proj{i}": f32(proj), f"world_view_transform{i}": f32(wvt),
                    f"full_proj_transform{i}": f32(full), f"camera_center{i}": f32(center)})
    pts = torch.tensor(rng.standard_normal((20, 3)), dtype=torch.float32)
    out["points"] = f32(pts)
    out["points_ndc0"] = f32(gu.geom_transform_points(pts, torch.tensor(out["full_proj_transform0"])))
    np.savez_compressed(os.path.join(OUT, "cameras.npz"), **out)


def make_losses():
    lu = _load("utils/loss_utils.py", "ref_loss_utils")
    iu = _load("utils/image_utils.py", "ref_image_utils")
    g = torch.Generator().manual_seed(0)
    out = {}
    for i in range(3):
        a = torch.rand(3, 64, 64, generator=g)
        b = (a + 0.1 * (i + 1) * torch.randn(3, 64, 64, generator=g)).clamp(0, 1)
        out.update({f"a{i}": f32(a), f"b{i}": f32(b), f"l1_{i}": f32(lu.l1_loss(a, b)), f"ssim_{i}": f32(lu.ssim(a, b)),
                    f"psnr_{i}": f32(iu.psnr(a, b))})
    np.savez_compressed(os.path.join(OUT, "losses.npz"), **out)


def make_planegrid():
    grids = _load("scene/grids.py", "ref_grids")
    out = {}
    g = torch.Generator().manual_seed(1)
    xyz = (torch.rand(1000, 3, generator=g) * 4.4 - 2.2)     # some samples fall outside [-2,2] (zeros padding)
    out["xyz"] = f32(xyz)
    for name, ta in (("plain", False), ("ta", True)):
        torch.manual_seed(2)
        pg = grids.PlaneGrid(15, [24, 24, 24], [-2.0, -2.0, -2.0], [2.0, 2.0, 2.0], config={"factor": 1}, TAflag=ta)
        with torch.no_grad():
            y = pg(xyz, 0)
        out.update(sd_np(f"{name}.", pg))
        out[f"{name}.out"] = f32(y)
        # training-time plane noise (GaussianLearner.Q0 = 0.03, scene/gaussian_model.py:187,213): CPU generator
        # seeded with 7 right before the call
        torch.manual_seed(7)
        with torch.no_grad():
            out[f"{name}.out_q003"] = f32(pg(xyz, 0.03))
    np.savez_compressed(os.path.join(OUT, "planegrid.npz"), **out)


def make_tv():
    """The total-variation term of train.py:242-243 as the reference computes it (autograd through six smooth-L1 sums)."""
    grids = _load("scene/grids.py", "ref_grids")
    out = {}
    names = ("xy_plane", "xz_plane", "yz_plane")
    for tag, ws, ta, scale, w in (("cube_plain", [24, 24, 24], False, 1.0, 4e-7), ("cube_ta", [24, 24, 24], True, 6.0, 1e-7),
                                  ("odd_plain", [37, 19, 30], False, 8.0, 4e-7), ("odd_ta", [37, 19, 30], True, 1.0, 2.5e-3)):
        torch.manual_seed(5)
        pg = grids.PlaneGrid(15, ws, [-2.0, -2.0, -2.0], [2.0, 2.0, 2.0], config={"factor": 1}, TAflag=ta)
        g = torch.Generator().manual_seed(6)
        with torch.no_grad():
            for n in names:
                getattr(pg, n).mul_(scale)
        out[f"{tag}.w"] = np.float64(w)
        for n in names:
            out[f"{tag}.{n}"] = f32(getattr(pg, n))
        pg.total_variation_add_grad(w)                       # into empty .grad
        for n in names:
            out[f"{tag}.{n}.grad"] = f32(getattr(pg, n).grad)
        assert all(p.grad is None for p in pg.parameters() if all(p is not getattr(pg, n) for n in names))
        for n in names:                                      # into an existing .grad (what backward() left there)
            prior = torch.randn(getattr(pg, n).shape, generator=g) * 1e-6
            out[f"{tag}.{n}.prior"] = f32(prior)
            getattr(pg, n).grad = prior.clone()
        pg.total_variation_add_grad(w)
        for n in names:
            out[f"{tag}.{n}.grad_acc"] = f32(getattr(pg, n).grad)
    # GaussianLearner.tv_loss: which grids, which weights (plane_size 40 -> grids of 10, 10, 20, 40)
    for level in (0, 2):
        pc, _, _ = _ref_model([], 8, seed=21)
        pc.feat_planes._feat.activate_level = level
        with torch.no_grad():
            for gi, grid in enumerate(pc.feat_planes._feat.k0s):
                for n in names:
                    getattr(grid, n).mul_(3.0 + gi)
        pc.feat_planes.tv_loss(4e-7)
        for gi, grid in enumerate(pc.feat_planes._feat.k0s):
            for n in names:
                p = getattr(grid, n)
                if level == 0:
                    out[f"learner.k0s.{gi}.{n}"] = f32(p)
                out[f"learner.level{level}.k0s.{gi}.{n}.grad"] = f32(p.grad) if p.grad is not None else np.zeros(0, np.float32)
    np.savez_compressed(os.path.join(OUT, "tv.npz"), **out)


def make_neural_gaussians():
    import scene.gaussian_model as gm
    import gaussian_renderer as gr
    from arguments import ModelParams

    def get_offsets_cpu(self, resolutions_list, dim=3):
        offsets_list, offsets = [0], 0
        for r in resolutions_list:
            offsets += r ** dim
            offsets_list.append(offsets)
        return torch.tensor(resolutions_list, dtype=torch.int), torch.tensor(offsets_list, dtype=torch.int)

    gm.FeaturePlanes.get_offsets = get_offsets_cpu
    parser = argparse.ArgumentParser()
    mp = ModelParams(parser)
    args = parser.parse_args(["--num_channels", "15", "--plane_size", "40", "--appearance_dim", "0"])
    ds = mp.extract(args)
    torch.manual_seed(0)
    pc = gm.GaussianModel(ds.feat_dim, ds.n_offsets, ds.voxel_size, ds.update_depth, ds.update_init_factor,
                          ds.update_hierachy_factor, ds.use_feat_bank, ds.appearance_dim, ds.ratio,
                          ds.add_opacity_dist, ds.add_cov_dist, ds.add_color_dist, model_params=ds)
    N, k = 512, ds.n_offsets
    g = torch.Generator().manual_seed(3)
    pc._anchor = (torch.rand(N, 3, generator=g) * 3.6 - 1.8)
    pc._offset = torch.randn(N, k, 3, generator=g) * 0.5
    pc._anchor_feat = torch.randn(N, 32, generator=g) * 0.5
    pc._scaling = torch.randn(N, 6, generator=g) * 0.3 - 3.0
    pc.feat_planes.Q0 = 0                     # render.py:79 -- no plane noise
    # non-trivial MLP / BN parameters so that the opacity mask is mixed
    with torch.no_grad():
        for m in list(pc.mlp_opacity) + list(pc.mlp_cov) + list(pc.mlp_color):
            if isinstance(m, torch.nn.Linear):
                m.weight.mul_(3.0)
    cam = types.SimpleNamespace(camera_center=torch.tensor([0.3, -0.2, -4.5]), uid=0)
    vis = torch.rand(N, generator=g) > 0.25
    out = {"anchor": f32(pc._anchor), "offset": f32(pc._offset), "anchor_feat": f32(pc._anchor_feat),
           "scaling": f32(pc._scaling), "camera_center": f32(cam.camera_center), "visible_mask": vis.numpy(),
           "n_offsets": np.int64(k)}
    out.update(sd_np("mlp_opacity.", pc.mlp_opacity))
    out.update(sd_np("mlp_cov.", pc.mlp_cov))
    out.update(sd_np("mlp_color.", pc.mlp_color))
    fp = {f"feat_planes.{kk}": v for kk, v in
          ((kk, v.detach().cpu().numpy()) for kk, v in pc.feat_planes.state_dict().items())
          if "num_batches_tracked" not in kk}
    out.update(fp)
    names = ["xyz", "color", "opacity", "scaling", "rot", "neural_opacity", "mask"]
    for level in (0, 2):
        pc.feat_planes._feat.activate_level = level
        for training in (True, False):
            # BatchNorm layers stay in train mode: outputs depend on the batch
            pc.mlp_opacity.train(training); pc.mlp_cov.train(training); pc.mlp_color.train(training)
            with torch.no_grad():
                res = gr.generate_neural_gaussians(cam, pc, vis, is_training=training)
            tag = f"L{level}_{'train' if training else 'eval'}"
            for nme, v in zip(names, res):
                out[f"{tag}.{nme}"] = v.numpy() if v.dtype == torch.bool else f32(v)
    np.savez_compressed(os.path.join(OUT, "neural_gaussians.npz"), **out)

    # ---- checkpoint files written by the reference's OWN code for this model: chkpnt<N>.pth = torch.save(capture())
    # (train.py:313-316, scene/gaussian_model.py:368-372) and checkpoints.pth = save_mlp_checkpoints(mode 'unite')
    # (scene/gaussian_model.py:1045-1066).  Binary data files (tensors), the layout test of splatco_amd.scene_io.
    ref_dir = os.path.join(OUT, "ref_scene")
    os.makedirs(ref_dir, exist_ok=True)
    pc.setup_contractor([0.1, -0.2, 0.3], [4.0, 5.0, 6.0], False)
    torch.save(pc.capture(), os.path.join(ref_dir, "chkpnt7.pth"))
    pc.save_mlp_checkpoints(ref_dir, mode="unite")

    # ---- training_statis (scene/gaussian_model.py:761-782) on the same model object
    g = torch.Generator().manual_seed(5)
    Nn = 40
    pc.opacity_accum = torch.zeros(Nn, 1)
    pc.anchor_demon = torch.zeros(Nn, 1)
    pc.offset_gradient_accum = torch.zeros(Nn * k, 1)
    pc.offset_denom = torch.zeros(Nn * k, 1)
    anchor_vis = torch.rand(Nn, generator=g) > 0.4
    V = int(anchor_vis.sum())
    neural_opacity = torch.randn(V * k, 1, generator=g)
    sel = (neural_opacity > 0).view(-1)
    P = int(sel.sum())
    update_filter = torch.rand(P, generator=g) > 0.3
    vsp = types.SimpleNamespace(grad=torch.randn(P, 3, generator=g))
    pc.training_statis(vsp, neural_opacity, update_filter, sel, anchor_vis)
    np.savez_compressed(os.path.join(OUT, "training_statis.npz"), anchor_visible_mask=anchor_vis.numpy(),
                        neural_opacity=f32(neural_opacity), offset_selection_mask=sel.numpy(),
                        update_filter=update_filter.numpy(), viewspace_grad=f32(vsp.grad), n_offsets=np.int64(k),
                        opacity_accum=f32(pc.opacity_accum), anchor_demon=f32(pc.anchor_demon),
                        offset_gradient_accum=f32(pc.offset_gradient_accum), offset_denom=f32(pc.offset_denom))


def _ref_model(extra_args, N, seed):
    import scene.gaussian_model as gm
    from arguments import ModelParams

    def get_offsets_cpu(self, resolutions_list, dim=3):
        offsets_list, offsets = [0], 0
        for r in resolutions_list:
            offsets += r ** dim
            offsets_list.append(offsets)
        return torch.tensor(resolutions_list, dtype=torch.int), torch.tensor(offsets_list, dtype=torch.int)

    gm.FeaturePlanes.get_offsets = get_offsets_cpu
    parser = argparse.ArgumentParser()
    mp = ModelParams(parser)
    ds = mp.extract(parser.parse_args(["--num_channels", "15", "--plane_size", "40"] + extra_args))
    torch.manual_seed(seed)
    pc = gm.GaussianModel(ds.feat_dim, ds.n_offsets, ds.voxel_size, ds.update_depth, ds.update_init_factor,
                          ds.update_hierachy_factor, ds.use_feat_bank, ds.appearance_dim, ds.ratio,
                          ds.add_opacity_dist, ds.add_cov_dist, ds.add_color_dist, model_params=ds)
    k = ds.n_offsets
    g = torch.Generator().manual_seed(seed + 3)
    pc._anchor = (torch.rand(N, 3, generator=g) * 3.6 - 1.8)
    pc._offset = torch.randn(N, k, 3, generator=g) * 0.5
    pc._anchor_feat = torch.randn(N, 32, generator=g) * 0.5
    pc._scaling = torch.randn(N, 6, generator=g) * 0.3 - 3.0
    pc._rotation = torch.randn(N, 4, generator=g)
    pc._opacity = torch.randn(N, 1, generator=g)
    pc.feat_planes.Q0 = 0
    with torch.no_grad():
        for m in list(pc.mlp_opacity) + list(pc.mlp_cov) + list(pc.mlp_color):
            if isinstance(m, torch.nn.Linear):
                m.weight.mul_(3.0)
    return pc, ds, g


def make_neural_gaussians_appearance():
    """appearance_dim = 32 is the default of the reference's argument parser; the embedding itself only exists after
    GaussianModel.set_appearance(num_cameras), which nothing in the reference calls (so the default command line stops
    at gaussian_renderer/__init__.py:58 with `None(camera_indicies)`): the fixture calls it as a dataset loader would."""
    import gaussian_renderer as gr
    out = {}
    names = ["xyz", "color", "opacity", "scaling", "rot", "neural_opacity", "mask"]
    # (use_feat_bank = True cannot be captured: the reference concatenates ob_view, ob_dist AND the 64 geo features into the
    # feature-bank MLP's input, gaussian_renderer/__init__.py:41-43, which was built for 3 + 1 columns,
    # scene/gaussian_model.py:308-309 -- the branch stops with a shape error, checked here on 2026-10-02)
    for case, extra in (("app", []),):
        pc, ds, g = _ref_model(extra, 256, 20 + len(extra))
        assert ds.appearance_dim == 32 and pc.use_feat_bank == bool(extra)
        pc.set_appearance(5)
        cam = types.SimpleNamespace(camera_center=torch.tensor([0.3, -0.2, -4.5]), uid=3)
        vis = torch.rand(256, generator=g) > 0.25
        pre = case + "."
        out.update({pre + "anchor": f32(pc._anchor), pre + "offset": f32(pc._offset), pre + "anchor_feat": f32(pc._anchor_feat),
                    pre + "scaling": f32(pc._scaling), pre + "camera_center": f32(cam.camera_center),
                    pre + "uid": np.int64(cam.uid), pre + "visible_mask": vis.numpy(), pre + "n_offsets": np.int64(ds.n_offsets)})
        out.update(sd_np(pre + "mlp_opacity.", pc.mlp_opacity))
        out.update(sd_np(pre + "mlp_cov.", pc.mlp_cov))
        out.update(sd_np(pre + "mlp_color.", pc.mlp_color))
        out.update(sd_np(pre + "embedding_appearance.", pc.embedding_appearance))
        if pc.use_feat_bank:
            out.update(sd_np(pre + "mlp_feature_bank.", pc.mlp_feature_bank))
        out.update({f"{pre}feat_planes.{kk}": v.detach().cpu().numpy() for kk, v in pc.feat_planes.state_dict().items()
                    if "num_batches_tracked" not in kk})
        pc.feat_planes._feat.activate_level = 2
        for training in (True, False):
            pc.mlp_opacity.train(training); pc.mlp_cov.train(training); pc.mlp_color.train(training)
            with torch.no_grad():
                res = gr.generate_neural_gaussians(cam, pc, vis, is_training=training)
            tag = f"{pre}{'train' if training else 'eval'}"
            for nme, v in zip(names, res):
                out[f"{tag}.{nme}"] = v.numpy() if v.dtype == torch.bool else f32(v)
    np.savez_compressed(os.path.join(OUT, "neural_gaussians_app.npz"), **out)


def make_ply_layout():
    """save_ply / load_ply_sparse_gaussian of the reference with a recording stand-in for plyfile."""
    import scene.gaussian_model as gm
    rec = {}

    class Element:
        def __init__(self, data, name):
            self.data, self.name = data, name
            self.properties = [types.SimpleNamespace(name=n) for n in data.dtype.names]

        def __getitem__(self, key):
            return self.data[key]

    class PlyElement:
        @staticmethod
        def describe(data, name):
            rec["element"] = Element(np.array(data, copy=True), name)
            return rec["element"]

    class PlyData:
        def __init__(self, elements):
            self.elements = list(elements)

        def write(self, path):
            rec["written_to"] = path

        @staticmethod
        def read(path):
            return PlyData([rec["element"]])

    gm.PlyElement, gm.PlyData = PlyElement, PlyData
    gm.mkdir_p = lambda p: None
    pc, ds, _ = _ref_model(["--appearance_dim", "0"], 37, 40)
    pc.save_ply("/nonexistent/point_cloud/iteration_7/point_cloud.ply")
    el = rec["element"]
    assert rec["written_to"].endswith("point_cloud.ply") and el.name == "vertex"
    out = {"element_name": np.array(el.name), "field_names": np.array(el.data.dtype.names),
           "field_formats": np.array([el.data.dtype.fields[n][0].str for n in el.data.dtype.names]),
           "itemsize": np.int64(el.data.dtype.itemsize), "payload": np.frombuffer(el.data.tobytes(), np.uint8),
           "anchor": f32(pc._anchor), "offset": f32(pc._offset), "anchor_feat": f32(pc._anchor_feat),
           "scaling": f32(pc._scaling), "rotation": f32(pc._rotation), "opacity": f32(pc._opacity),
           "n_offsets": np.int64(ds.n_offsets)}
    # the reference's reader on the element its writer produced (torch.tensor(..., device="cuda") -> host)
    real_tensor = torch.tensor
    torch.tensor = lambda *a, **k: real_tensor(*a, **{kk: v for kk, v in k.items() if kk != "device"})
    try:
        other = gm.GaussianModel.__new__(gm.GaussianModel)
        other.load_ply_sparse_gaussian("ignored")
    finally:
        torch.tensor = real_tensor
    for n in ("anchor", "offset", "anchor_feat", "scaling", "rotation", "opacity"):
        out["loaded." + n] = f32(getattr(other, "_" + n))
        out["loaded." + n + ".requires_grad"] = np.bool_(getattr(other, "_" + n).requires_grad)
    np.savez_compressed(os.path.join(OUT, "ply_layout.npz"), **out)


def make_densify():
    """GaussianModel.adjust_anchor / anchor_growing / prune_anchor / compute_curvature
    (scene/gaussian_model.py:784-997,1092-1110) on a small seeded model, CPU.  The reference hard-codes
    device='cuda' in torch.zeros / torch.ones calls and takes scatter_max from torch_scatter: both are
    patched (device kwarg dropped; scatter_max restated with scatter_reduce amax)."""
    import scene.gaussian_model as gm

    def drop_cuda(fn):
        def wrapped(*a, **k):
            if str(k.get("device", "")).startswith("cuda"):
                k.pop("device")
            return fn(*a, **k)
        return wrapped

    torch.zeros, torch.ones = drop_cuda(torch.zeros), drop_cuda(torch.ones)

    def scatter_max(src, index, dim=0):
        out = torch.zeros(int(index.max()) + 1, src.shape[1], dtype=src.dtype)
        out = out.scatter_reduce(0, index, src, "amax", include_self=False)
        return out, None

    gm.scatter_max = scatter_max
    out = {}
    for case, (iteration, N) in enumerate(((1700, 60), (1600, 48))):
        k, F_ = 4, 8
        pc = gm.GaussianModel.__new__(gm.GaussianModel)
        torch.nn.Module.__init__(pc) if isinstance(pc, torch.nn.Module) else None
        pc.n_offsets, pc.feat_dim, pc.voxel_size = k, F_, 0.01
        pc.update_depth, pc.update_init_factor, pc.update_hierachy_factor = 3, 16, 4
        pc.scaling_activation = torch.exp
        g = torch.Generator().manual_seed(11 + case)
        anchor = torch.round((torch.rand(N, 3, generator=g) * 2 - 1) / 0.16) * 0.16      # on the coarsest grid
        params = {"anchor": anchor, "offset": torch.randn(N, k, 3, generator=g) * 0.5,
                  "anchor_feat": torch.randn(N, F_, generator=g), "opacity": torch.full((N, 1), -2.1972246),
                  "scaling": torch.randn(N, 6, generator=g) * 0.5 - 2.0, "rotation": torch.randn(N, 4, generator=g)}
        for name, v in params.items():
            setattr(pc, "_" + name, torch.nn.Parameter(v.clone()))
        groups = [{"params": [getattr(pc, "_" + name)], "lr": 1e-3, "name": name} for name in params]
        pc.optimizer = torch.optim.Adam(groups, lr=0.0, eps=1e-15)
        # one optimizer step so that exp_avg / exp_avg_sq exist (except for "rotation": state-less branch)
        for name in params:
            if name != "rotation":
                getattr(pc, "_" + name).grad = torch.randn(getattr(pc, "_" + name).shape, generator=g)
        pc.optimizer.step()
        pc.offset_gradient_accum = torch.rand(N * k, 1, generator=g) * 0.05
        pc.offset_denom = torch.randint(0, 90, (N * k, 1), generator=g).float()
        pc.opacity_accum = torch.rand(N, 1, generator=g) * 2.0
        pc.anchor_demon = torch.randint(60, 120, (N, 1), generator=g).float()
        pre = f"c{case}."
        out[pre + "iteration"], out[pre + "n_offsets"] = np.int64(iteration), np.int64(k)
        for name in params:
            out[pre + "in." + name] = f32(getattr(pc, "_" + name))
            st = pc.optimizer.state.get(getattr(pc, "_" + name), None)
            if st is not None:
                out[pre + "in.exp_avg." + name], out[pre + "in.exp_avg_sq." + name] = f32(st["exp_avg"]), f32(st["exp_avg_sq"])
        for name in ("offset_gradient_accum", "offset_denom", "opacity_accum", "anchor_demon"):
            out[pre + "in." + name] = f32(getattr(pc, name))
        torch.manual_seed(100 + case)                     # the random pick of anchor_growing (:844)
        out[pre + "seed"] = np.int64(100 + case)
        if iteration == 1600:
            out[pre + "curvature"] = f32(pc.compute_curvature(pc.get_anchor))
        with torch.no_grad():                             # train.py:243 runs the densification block under no_grad
            pc.adjust_anchor(iteration=iteration, check_interval=100, success_threshold=0.8, grad_threshold=0.0002,
                             min_opacity=0.005)
        for name in params:
            out[pre + "out." + name] = f32(getattr(pc, "_" + name))
            st = pc.optimizer.state.get(getattr(pc, "_" + name), None)
            if st is not None:
                out[pre + "out.exp_avg." + name], out[pre + "out.exp_avg_sq." + name] = f32(st["exp_avg"]), f32(st["exp_avg_sq"])
        for name in ("offset_gradient_accum", "offset_denom", "opacity_accum", "anchor_demon", "max_radii2D"):
            out[pre + "out." + name] = f32(getattr(pc, name))
        print("densify case", case, "anchors", N, "->", pc._anchor.shape[0])
    np.savez_compressed(os.path.join(OUT, "densify.npz"), **out)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    install_stubs()
    makers = {"cameras": make_cameras, "losses": make_losses, "planegrid": make_planegrid,
              "neural_gaussians": make_neural_gaussians, "neural_gaussians_app": make_neural_gaussians_appearance,
              "ply_layout": make_ply_layout, "tv": make_tv, "densify": make_densify}
    for name in (sys.argv[1:] or list(makers)):          # densify patches torch.zeros / torch.ones: keep it last
        makers[name]()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))
