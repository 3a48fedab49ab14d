"""CPU tests of the rows next to the hot path that are pure host logic: anchor densification
bookkeeping (tests/golden/densify.npz, captured from the reference's adjust_anchor) and the
anchor PLY / checkpoint formats."""
import os
import types

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(__file__), "golden")
NAMES = ("anchor", "offset", "anchor_feat", "opacity", "scaling", "rotation")


def _case(d, case):
    from splatco_amd.densify import AnchorDensifier
    pre = f"c{case}."
    k = int(d[pre + "n_offsets"])
    m = types.SimpleNamespace(n_offsets=k, feat_dim=d[pre + "in.anchor_feat"].shape[1])
    groups = []
    for n in NAMES:
        p = torch.nn.Parameter(torch.tensor(d[pre + "in." + n]))
        setattr(m, "_" + n, p)
        groups.append({"params": [p], "lr": 1e-3, "name": n})
    opt = torch.optim.Adam(groups, lr=0.0, eps=1e-15)
    for n in NAMES:
        if pre + "in.exp_avg." + n in d.files:
            opt.state[getattr(m, "_" + n)] = {"step": torch.tensor(1.0), "exp_avg": torch.tensor(d[pre + "in.exp_avg." + n]),
                                              "exp_avg_sq": torch.tensor(d[pre + "in.exp_avg_sq." + n])}
    den = AnchorDensifier(m, opt, voxel_size=0.01, update_depth=3, update_init_factor=16, update_hierachy_factor=4)
    for n in ("offset_gradient_accum", "offset_denom", "opacity_accum", "anchor_demon"):
        setattr(den, n, torch.tensor(d[pre + "in." + n]))
    return m, opt, den, pre


@pytest.mark.parametrize("case", [0, 1])
def test_adjust_anchor_matches_reference(case):
    """scene/gaussian_model.py:929-997 incl. anchor_growing, prune_anchor and the Adam-state surgery; case 1
    runs the curvature branch (iteration 1600).  Same torch seed -> same random pick -> identical anchors."""
    d = np.load(os.path.join(GOLD, "densify.npz"))
    m, opt, den, pre = _case(d, case)
    torch.manual_seed(int(d[pre + "seed"]))
    den.adjust_anchor(iteration=int(d[pre + "iteration"]), check_interval=100, success_threshold=0.8,
                      grad_threshold=0.0002, min_opacity=0.005)
    for n in NAMES:
        got = getattr(m, "_" + n).detach().numpy()
        assert got.shape == d[pre + "out." + n].shape, n
        np.testing.assert_allclose(got, d[pre + "out." + n], rtol=1e-6, atol=1e-7, err_msg=n)
        if pre + "out.exp_avg." + n in d.files:
            st = opt.state[getattr(m, "_" + n)]
            np.testing.assert_allclose(st["exp_avg"].numpy(), d[pre + "out.exp_avg." + n], rtol=1e-6, atol=1e-9)
            np.testing.assert_allclose(st["exp_avg_sq"].numpy(), d[pre + "out.exp_avg_sq." + n], rtol=1e-6, atol=1e-12)
        else:
            assert getattr(m, "_" + n) not in opt.state
    for n in ("offset_gradient_accum", "offset_denom", "opacity_accum", "anchor_demon", "max_radii2D"):
        np.testing.assert_allclose(getattr(den, n).numpy(), d[pre + "out." + n], rtol=1e-6, atol=1e-7, err_msg=n)
    assert m._anchor.shape[0] > d[pre + "in.anchor"].shape[0] // 2


def test_sort_anchors_permutes_parameters_moments_and_accumulators():
    """AnchorDensifier.sort_anchors (Morton order; not in the reference): every per-anchor tensor -- parameters, Adam
    moments, the four accumulators -- moves with the same permutation, and adjust_anchor afterwards gives the reference's
    anchors in the permuted order (same set of rows)."""
    from splatco_amd.scene_model import morton_order
    d = np.load(os.path.join(GOLD, "densify.npz"))
    m, opt, den, pre = _case(d, 0)
    k = den.n_offsets
    before = {n: getattr(m, "_" + n).detach().clone() for n in NAMES}
    mom = {n: opt.state[getattr(m, "_" + n)]["exp_avg"].clone() for n in NAMES if getattr(m, "_" + n) in opt.state}
    acc = {n: getattr(den, n).clone() for n in ("offset_gradient_accum", "offset_denom", "opacity_accum", "anchor_demon")}
    perm = den.sort_anchors()
    assert torch.equal(perm, morton_order(before["anchor"])) and sorted(perm.tolist()) == list(range(len(perm)))
    code = lambda a: morton_order(a)        # sorted input -> identity permutation (stable sort)
    assert torch.equal(code(m._anchor), torch.arange(len(perm)))
    for n in NAMES:
        p = getattr(m, "_" + n)
        assert torch.equal(p.detach(), before[n][perm]), n
        assert p is [g for g in opt.param_groups if g["name"] == n][0]["params"][0]
        if n in mom:
            assert torch.equal(opt.state[p]["exp_avg"], mom[n][perm]), n
    assert torch.equal(den.opacity_accum, acc["opacity_accum"][perm]) and torch.equal(den.anchor_demon, acc["anchor_demon"][perm])
    assert torch.equal(den.offset_denom.view(-1, k), acc["offset_denom"].view(-1, k)[perm])
    assert torch.equal(den.offset_gradient_accum.view(-1, k), acc["offset_gradient_accum"].view(-1, k)[perm])


def test_compute_curvature_matches_reference():
    from splatco_amd.densify import compute_curvature
    d = np.load(os.path.join(GOLD, "densify.npz"))
    got = compute_curvature(torch.tensor(d["c1.in.anchor"]))
    np.testing.assert_allclose(got.numpy(), d["c1.curvature"], rtol=2e-4, atol=1e-6)


def _small_model(seed=0, N=37):
    from splatco_amd.scene_model import AnchorGaussianModel
    torch.manual_seed(seed)
    pc = AnchorGaussianModel(feat_dim=32, n_offsets=10, plane_size=40, num_channels=15)
    pc.set_anchors(torch.randn(N, 3), torch.randn(N, 10, 3), torch.randn(N, 32), torch.randn(N, 6), torch.randn(N, 4))
    return pc


def test_ply_layout_and_round_trip(tmp_path):
    """scene/gaussian_model.py:640-712: property names / order, the (N,3,k) offset flattening, binary
    little-endian float32 payload; load(save(x)) == x bit for bit."""
    from splatco_amd import scene_io
    pc = _small_model()
    path = str(tmp_path / "point_cloud" / "iteration_7" / "point_cloud.ply")
    scene_io.save_ply(pc, path)
    raw = open(path, "rb").read()
    head, payload = raw.split(b"end_header\n", 1)
    lines = head.decode().split("\n")
    assert lines[:3] == ["ply", "format binary_little_endian 1.0", "element vertex 37"]
    names = [l.split()[-1] for l in lines[3:] if l]
    assert names[:6] == ["x", "y", "z", "nx", "ny", "nz"] and names[6] == "f_offset_0" and names[35] == "f_offset_29"
    assert names[36] == "f_anchor_feat_0" and names[68] == "opacity" and names[69:75] == [f"scale_{i}" for i in range(6)]
    assert names[75:] == [f"rot_{i}" for i in range(4)] and len(payload) == 37 * 79 * 4
    row0 = np.frombuffer(payload[:79 * 4], "<f4")
    np.testing.assert_array_equal(row0[:3], pc._anchor[0].detach().numpy())
    # f_offset_{c*k + j} = offset[n, j, c]  (transpose(1,2).flatten, :660)
    np.testing.assert_array_equal(row0[6:36].reshape(3, 10), pc._offset[0].detach().numpy().T)
    other = _small_model(seed=1, N=5)
    scene_io.load_ply_sparse_gaussian(other, path)
    for n in ("_anchor", "_offset", "_anchor_feat", "_opacity", "_scaling", "_rotation"):
        assert torch.equal(getattr(other, n), getattr(pc, n)), n
    assert other._anchor.requires_grad and other._rotation.requires_grad and other._opacity.requires_grad   # as :706-712


def test_ply_equals_what_the_reference_hands_to_plyfile(tmp_path):
    """tests/golden/ply_layout.npz: the structured array the REFERENCE's save_ply (scene/gaussian_model.py:640-673) built
    and passed to PlyElement.describe(elements, 'vertex') / PlyData([el]).write(path), recorded by a stand-in for the
    absent plyfile package (tools/make_golden.py), plus what the reference's load_ply_sparse_gaussian (:675-712) made of
    that element.  save_ply here must write the same properties in the same order with the same bytes, and the loader
    must return the reference loader's tensors."""
    from splatco_amd import scene_io
    from splatco_amd.scene_model import AnchorGaussianModel
    d = np.load(os.path.join(GOLD, "ply_layout.npz"))
    t = lambda n: torch.tensor(d[n])
    pc = AnchorGaussianModel(feat_dim=32, n_offsets=int(d["n_offsets"]), plane_size=8, num_channels=15)
    pc.set_anchors(t("anchor"), t("offset"), t("anchor_feat"), t("scaling"), t("rotation"), t("opacity"))
    path = str(tmp_path / "point_cloud" / "iteration_7" / "point_cloud.ply")
    scene_io.save_ply(pc, path)
    head, payload = open(path, "rb").read().split(b"end_header\n", 1)
    lines = [l for l in head.decode().split("\n") if l]
    N = d["anchor"].shape[0]
    assert lines[:3] == ["ply", "format binary_little_endian 1.0", f"element {str(d['element_name'])} {N}"]
    assert [l.split()[-1] for l in lines[3:]] == [str(n) for n in d["field_names"]]           # names AND order
    assert all(l.split()[:2] == ["property", "float"] for l in lines[3:]) and set(str(f) for f in d["field_formats"]) == {"<f4"}
    assert int(d["itemsize"]) == 4 * len(d["field_names"])
    assert payload == d["payload"].tobytes()                                                   # the reference's bytes
    other = AnchorGaussianModel(feat_dim=32, n_offsets=int(d["n_offsets"]), plane_size=8, num_channels=15)
    scene_io.load_ply_sparse_gaussian(other, path)
    for n in ("anchor", "offset", "anchor_feat", "scaling", "rotation", "opacity"):
        got = getattr(other, "_" + n)
        assert np.array_equal(got.detach().numpy(), d["loaded." + n]), n
        assert bool(got.requires_grad) == bool(d["loaded." + n + ".requires_grad"]), n
        assert np.array_equal(d["loaded." + n], d[n]), n                                       # and the reference round-trips itself


def test_ply_reader_accepts_ascii_and_permuted_properties(tmp_path):
    from splatco_amd import scene_io
    p = tmp_path / "a.ply"
    p.write_text("ply\nformat ascii 1.0\ncomment made by hand\nelement vertex 2\nproperty float z\nproperty double x\n"
                 "property float y\nend_header\n3 1 2\n6 4 5\n")
    v = scene_io.read_ply_vertices(str(p))
    np.testing.assert_array_equal(np.stack([v["x"], v["y"], v["z"]], 1), [[1, 2, 3], [4, 5, 6]])
    with pytest.raises(ValueError):
        (tmp_path / "b.ply").write_text("plx\n")
        scene_io.read_ply_vertices(str(tmp_path / "b.ply"))


def test_scene_save_load_round_trip(tmp_path):
    """Directory layout of Scene.save / train.py:316 and the reload path of scene/__init__.py:80-94."""
    from splatco_amd import scene_io
    pc = _small_model()
    scene_io.save_scene(pc, str(tmp_path), 30000)
    assert sorted(os.listdir(tmp_path / "point_cloud" / "iteration_30000")) == ["checkpoints.pth", "point_cloud.ply"]
    ck = torch.load(tmp_path / "point_cloud" / "iteration_30000" / "checkpoints.pth", weights_only=True)
    assert sorted(ck) == ["color_mlp", "cov_mlp", "opacity_mlp"] and sorted(ck["cov_mlp"]) == ["0.bias", "0.weight", "2.bias", "2.weight"]
    other = _small_model(seed=5, N=3)
    scene_io.load_scene(other, str(tmp_path), 30000)
    a, b = pc.state_dict(), other.state_dict()
    assert a.keys() == b.keys()
    for k_ in a:
        assert torch.equal(a[k_], b[k_]), k_


def test_checkpoints_written_by_the_reference_load_and_match(tmp_path):
    """tests/golden/ref_scene/ holds chkpnt7.pth (= torch.save(GaussianModel.capture()), train.py:313-316,
    scene/gaussian_model.py:368-372) and checkpoints.pth (save_mlp_checkpoints 'unite', :1045-1066) written by the
    REFERENCE's own code for the model of neural_gaussians.npz (tools/make_golden.py).  They must load through
    splatco_amd.scene_io into our modules strictly, reproduce the fixture's weights, and what save_scene writes for
    that model must have the same layout: same keys, shapes, dtypes and values, contractor bounds included."""
    from splatco_amd import scene_io
    from test_host_golden import _model_from_fixture
    d = np.load(os.path.join(GOLD, "neural_gaussians.npz"))
    ref_dir = os.path.join(GOLD, "ref_scene")
    pc = _model_from_fixture(d)                                   # weights from the .npz arrays
    other = _small_model(seed=9, N=4)
    other.feat_planes = type(pc.feat_planes)(40, 15)              # plane_size of the fixture
    scene_io.load_mlp_checkpoints(other, ref_dir)                 # <- reference-written file
    for name in ("mlp_opacity", "mlp_cov", "mlp_color"):
        a, b = getattr(pc, name).state_dict(), getattr(other, name).state_dict()
        assert a.keys() == b.keys()
        assert all(torch.equal(a[k_], b[k_]) for k_ in a), name
    ck = torch.load(os.path.join(ref_dir, "chkpnt7.pth"), map_location="cpu", weights_only=True)
    assert isinstance(ck, tuple) and len(ck) == 2
    missing, unexpected = other.feat_planes.load_state_dict(ck[0], strict=False)
    assert not unexpected and not missing, (missing, unexpected)    # our module tree IS the reference's key set
    want = pc.feat_planes.state_dict()
    for k_, v in other.feat_planes.state_dict().items():
        if "num_batches_tracked" not in k_:
            assert torch.equal(v, want[k_]), k_
    assert sorted(ck[1]) == ["xyz_max", "xyz_min"]
    # what we write for the same model == what the reference wrote
    pc.contractor_state = dict(ck[1])
    pc.set_anchors(torch.tensor(d["anchor"]), torch.tensor(d["offset"]), torch.tensor(d["anchor_feat"]), torch.tensor(d["scaling"]))
    scene_io.save_scene(pc, str(tmp_path), 7)
    mine = torch.load(tmp_path / "chkpnt7.pth", map_location="cpu", weights_only=True)
    assert mine[0].keys() == ck[0].keys() and mine[1].keys() == ck[1].keys()
    for k_ in ck[0]:
        if "num_batches_tracked" not in k_:
            assert mine[0][k_].dtype == ck[0][k_].dtype and torch.equal(mine[0][k_], ck[0][k_]), k_
    assert all(torch.equal(mine[1][k_], ck[1][k_]) for k_ in ck[1])
    mine_mlp = torch.load(tmp_path / "point_cloud" / "iteration_7" / "checkpoints.pth", weights_only=True)
    ref_mlp = torch.load(os.path.join(ref_dir, "checkpoints.pth"), weights_only=True)
    assert mine_mlp.keys() == ref_mlp.keys()
    for k_ in ref_mlp:
        assert mine_mlp[k_].keys() == ref_mlp[k_].keys() and all(torch.equal(mine_mlp[k_][j], ref_mlp[k_][j]) for j in ref_mlp[k_])
    back = _small_model(seed=3, N=2)
    back.feat_planes = type(pc.feat_planes)(40, 15)
    scene_io.load_scene(back, str(tmp_path), 7)
    assert sorted(back.contractor_state) == ["xyz_max", "xyz_min"] and torch.equal(back.contractor_state["xyz_min"], ck[1]["xyz_min"])
