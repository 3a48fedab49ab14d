"""Drop-in for the reference's gaussian_renderer module: prefilter_voxel / generate_neural_gaussians
/ render with the same signatures, result-dict keys and autograd behaviour
(gaussian_renderer/__init__.py:18-244), calling the MI355X rasterizer operator.

`pc` is any object with the GaussianModel attributes the reference reads (e.g.
splatco_amd.scene_model.AnchorGaussianModel or the reference's own GaussianModel).
"""
import math

import torch

from ._C import stage as _C_stage        # opt-in roctx ranges (nothing when off)
from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer


def _mlp_heads(pc, x):
    """mlp_opacity / mlp_color / mlp_cov (scene/gaussian_model.py:315-337) on a shared input."""
    heads = (pc.get_opacity_mlp, pc.get_color_mlp, pc.get_cov_mlp)
    first = [h[0] for h in heads]
    if not all(isinstance(h, torch.nn.Sequential) and isinstance(h[0], torch.nn.Linear) and isinstance(h[1], torch.nn.ReLU)
               for h in heads) or len({f.out_features for f in first}) != 1:
        return tuple(h(x) for h in heads)
    from .scene_model import TallLinear
    w = torch.cat([f.weight for f in first], dim=0)
    b = torch.cat([f.bias for f in first], dim=0)
    hdn = torch.relu_(TallLinear.apply_weights(x, w, b))
    # split, not three slices: its backward is one concatenation of the three hidden gradients, where a slice's
    # backward materialises a zero-padded [V, 96] tensor per head and sums them
    parts = hdn.split(first[0].out_features, dim=1)
    return tuple(h[2:](x_) for h, x_ in zip(heads, parts))


def _parts_capable():
    from .scene_model import GaussianLearner
    return GaussianLearner


def generate_neural_gaussians(viewpoint_camera, pc, visible_mask=None, is_training=False, expand=None, fused_heads=True):
    """Anchors -> neural Gaussians (gaussian_renderer/__init__.py:18-116), same op order.
    The mask / compaction / post-processing block (:68-111) is the single HIP op of splatco_amd.expand
    (device tensors only, no CPU path).  expand: test hook -- a callable with expand_compact's signature
    (tests/torch_restatements.py holds the torch op chain the kernel is checked against).  fused_heads=False keeps
    the MLP heads as rocBLAS GEMMs (the checker of csrc/mlp_heads.hip)."""
    if visible_mask is None:
        visible_mask = torch.ones(pc.get_anchor.shape[0], dtype=torch.bool, device=pc.get_anchor.device)
    # `t[visible_mask]` four times (:23-29) = four mask->index conversions (each a host sync) and
    # four sort-based index_put backwards; one nonzero + index_select gives the same rows
    from .expand import visible_indices
    idx = visible_indices(visible_mask)             # on the GPU csrc/expand.hip: count / scan / write, one host read of the count
    from . import anchor_gather as _ag
    g_fea, presampled = None, None
    if _ag.fused_gather_taken(pc, fused_heads):
        sink = getattr(pc, "_grad_sink", None)
        if (sink is not None and sink.ranges and sink.on_range is not None and torch.is_grad_enabled()
                and hasattr(pc.feat_planes, "presample")):
            # A multi-rank training step (the gradient sink hands the per-anchor gradients to the exchange range by range):
            # the tri-plane features are sampled BEFORE the gather is applied -- autograd then runs the gather's backward
            # before the tri-plane / attention backward, and the exchange of the per-anchor gradients overlaps them
            # (FeaturePlanes.presample; the sample positions are detached in the reference, scene/gaussian_model.py:210).
            # Same kernels, same values; one extra [V,3] index_select.
            presampled = pc.feat_planes.presample(pc._anchor.detach().index_select(0, idx))
        # the four gathers, exp(_scaling) and the [V,71] concatenation of :23-31 as one pass (csrc/anchor_gather.hip)
        feat, anchor, grid_offsets, grid_scaling, g_fea = _ag.gather_anchors(pc, idx)
    else:
        if getattr(pc, "_grad_sink", None) is not None and torch.is_grad_enabled():
            # autograd would ADD these gradients into arena memory that zero() skipped for the sink
            raise RuntimeError("a gradient sink is attached (train_step.collaborative_step with a GradArena) but this "
                               "render does not take the fused anchor gather (fused_heads=False or a model whose "
                               "get_scaling is not exp(_scaling)): detach the sink or render through the fused path")
        feat = pc._anchor_feat.index_select(0, idx)
        anchor = pc.get_anchor.index_select(0, idx)
        grid_offsets = pc._offset.index_select(0, idx)
        grid_scaling = pc.get_scaling.index_select(0, idx)
    V, k = anchor.shape[0], pc.n_offsets
    if getattr(pc, "use_feat_bank", False):
        # dead in the reference too: its feature-bank MLP is built for 3 + 1 inputs (scene/gaussian_model.py:308-309) and
        # fed 3 + 1 + 64 columns (gaussian_renderer/__init__.py:41-43) -- the branch stops with a shape error there
        raise NotImplementedError("use_feat_bank: the reference's own branch cannot run (4-input MLP fed 68 columns)")
    appearance_dim = int(getattr(pc, "appearance_dim", 0) or 0)
    from . import mlp_heads as _mh
    # the benchmarked configuration (README.md:93: --appearance_dim 0, no feature bank, no distance inputs): the three
    # heads read the same [V, 99] input.  The reference's CODE default is appearance_dim = 32
    # (arguments/__init__.py:76): that runs through the framework's GEMMs below, op for op as
    # gaussian_renderer/__init__.py:40-93
    plain = not (pc.add_opacity_dist or pc.add_color_dist or pc.add_cov_dist or appearance_dim > 0)
    if g_fea is None:
        g_fea = torch.concat((feat, anchor, grid_offsets.reshape(V, -1), grid_scaling), dim=1)
    use_fused = plain and fused_heads and _mh.supported(pc, feat, feat, feat)
    if use_fused and not isinstance(pc.feat_planes, _parts_capable()):
        # someone else's feature planes (e.g. the reference's own GaussianModel): geo_fea arrives concatenated
        geo_fea = pc.feat_planes.inference(anchor, g_fea, 0)
        neural_opacity, color, scale_rot = _mh.mlp_heads(pc, feat, anchor, viewpoint_camera.camera_center, geo_fea)
        ob_view = None
    elif use_fused:
        # the three heads as ONE fp32-MFMA kernel per direction (csrc/mlp_heads.hip): x = cat(feat, ob_view, geo_fea)
        # (:58-60) is never built, ob_view (:34-38) is computed inside, and geo_fea is read as the two matrices
        # FeaturePlanes' two GEMMs leave behind (its cat, scene/gaussian_model.py:166, is skipped as well)
        if presampled is not None:
            geo_a, geo_b = pc.feat_planes.inference(anchor, g_fea, 0, parts=True, presampled=presampled)
        else:
            geo_a, geo_b = pc.feat_planes.inference(anchor, g_fea, 0, parts=True)
        neural_opacity, color, scale_rot = _mh.mlp_heads(pc, feat, anchor, viewpoint_camera.camera_center, geo_a, geo_b)
        ob_view = None
    else:
        geo_fea = pc.feat_planes.inference(anchor, g_fea, 0)
        ob_view = anchor - viewpoint_camera.camera_center
        ob_dist = ob_view.norm(dim=1, keepdim=True)
        ob_view = ob_view / ob_dist
    if ob_view is None:
        pass
    elif plain:
        # the three heads read the same [V, 99] input (:62-93 with the default flags): their first layers run
        # as ONE GEMM with stacked weights (one pass over the input instead of three, forward and backward)
        cat_local_view_wodist = torch.cat([feat, ob_view, geo_fea], dim=1)
        neural_opacity, color, scale_rot = _mlp_heads(pc, cat_local_view_wodist)
    else:
        cat_local_view = torch.cat([feat, ob_view, ob_dist, geo_fea], dim=1)
        cat_local_view_wodist = torch.cat([feat, ob_view, geo_fea], dim=1)
        neural_opacity = pc.get_opacity_mlp(cat_local_view if pc.add_opacity_dist else cat_local_view_wodist)
        color_in = cat_local_view if pc.add_color_dist else cat_local_view_wodist
        if appearance_dim > 0:                               # per-camera appearance code on the colour head (:55-58,76-80)
            camera_indicies = torch.full((V,), int(viewpoint_camera.uid), dtype=torch.long, device=anchor.device)
            color_in = torch.cat([color_in, pc.get_appearance(camera_indicies)], dim=1)
        color = pc.get_color_mlp(color_in)
        scale_rot = pc.get_cov_mlp(cat_local_view if pc.add_cov_dist else cat_local_view_wodist)
    neural_opacity = neural_opacity.reshape([-1, 1])
    color = color.reshape([V * k, 3])
    scale_rot = scale_rot.reshape([V * k, 7])
    if expand is None:
        if not anchor.is_cuda:
            raise RuntimeError("generate_neural_gaussians needs device tensors: the expansion / compaction step is a "
                               "HIP kernel (csrc/expand.hip) and has no CPU path")
        from .expand import expand_compact as expand
    xyz, color, opacity, scaling, rot, mask = expand(neural_opacity, color, scale_rot, grid_offsets, grid_scaling,
                                                     anchor, k)
    if is_training:
        return xyz, color, opacity, scaling, rot, neural_opacity, mask
    return xyz, color, opacity, scaling, rot


def _settings(viewpoint_camera, bg_color, scaling_modifier, debug):
    return GaussianRasterizationSettings(
        image_height=int(viewpoint_camera.image_height), image_width=int(viewpoint_camera.image_width),
        tanfovx=math.tan(viewpoint_camera.FoVx * 0.5), tanfovy=math.tan(viewpoint_camera.FoVy * 0.5),
        bg=bg_color, scale_modifier=scaling_modifier, viewmatrix=viewpoint_camera.world_view_transform,
        projmatrix=viewpoint_camera.full_proj_transform, sh_degree=1, campos=viewpoint_camera.camera_center,
        prefiltered=False, debug=debug)


class _ZeroCarrier(torch.autograd.Function):
    """zeros_like(like) as a non-leaf of the autograd graph: what `torch.zeros_like(xyz, requires_grad=True) + 0`
    (gaussian_renderer/__init__.py:133) produces, in one fill instead of a fill and an add.  `anchor` is an empty leaf that
    only ties the result into the graph; nothing flows back into it."""

    @staticmethod
    def forward(ctx, anchor, like, dtype):
        ctx.set_materialize_grads(False)
        out = torch.zeros_like(like, dtype=dtype)
        import weakref
        ctx.out_ref = weakref.ref(out)
        return out

    @staticmethod
    def backward(ctx, grad):
        # `retain_grad()` on the result makes autograd CLONE the incoming gradient into .grad (187 MB and 0.12 ms at
        # configs[2], 0.6 ms at configs[4]); keep_grad() below asks this node to hand the tensor over as it is instead --
        # the rasterizer's backward allocates dL/dmeans2D as a tensor of its own for exactly that.  One backward per
        # render (train.py:240): a second one REPLACES the gradient where retain_grad would add.
        out = ctx.out_ref()
        if out is not None and grad is not None and getattr(out, "_scr_keep_grad", False):
            out.grad = grad
        return None, None, None


def keep_grad(screenspace_points):
    """screenspace_points.retain_grad() (gaussian_renderer/__init__.py:134-138) without the copy: after backward(),
    .grad is the rasterizer's dL/dmeans2D itself."""
    if not screenspace_points.requires_grad:          # under no_grad (as the reference's `try: retain_grad()`)
        return
    if isinstance(getattr(screenspace_points, "grad_fn", None), _ZeroCarrier._backward_cls):
        screenspace_points._scr_keep_grad = True
    else:
        screenspace_points.retain_grad()


def render(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, visible_mask=None, retain_grad=False):
    """gaussian_renderer/__init__.py:118-188.  Background tensor must be on the GPU."""
    is_training = pc.get_color_mlp.training
    with _C_stage("generate_neural_gaussians"):
        out = generate_neural_gaussians(viewpoint_camera, pc, visible_mask, is_training=is_training)
    xyz, color, opacity, scaling, rot = out[:5]
    # zero tensor that carries the screen-space mean gradient back to the caller (:133-138): a NON-LEAF that requires grad
    # (the reference's `zeros_like(..., requires_grad=True) + 0`), whose .grad is populated only with retain_grad (:134-138,
    # train.py:185-186) -- same contract, without the extra pass over [P, 3] that the `+ 0` costs (_ZeroCarrier)
    screenspace_points = _ZeroCarrier.apply(torch.empty(0, device=xyz.device, requires_grad=True), xyz, pc.get_anchor.dtype)
    if retain_grad:
        keep_grad(screenspace_points)
    rasterizer = GaussianRasterizer(raster_settings=_settings(viewpoint_camera, bg_color, scaling_modifier, pipe.debug))
    with _C_stage("rasterize"):
        rendered_image, radii = rasterizer(means3D=xyz, means2D=screenspace_points, shs=None, colors_precomp=color,
                                           opacities=opacity, scales=scaling, rotations=rot, cov3D_precomp=None)
    res = {"render": rendered_image, "viewspace_points": screenspace_points, "visibility_filter": radii > 0,
           "radii": radii}
    if is_training:
        res.update({"selection_mask": out[6], "neural_opacity": out[5], "scaling": scaling})
    return res


def prefilter_voxel(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, override_color=None):
    """Anchor visibility (gaussian_renderer/__init__.py:191-244): radii_pure > 0."""
    rasterizer = GaussianRasterizer(raster_settings=_settings(viewpoint_camera, bg_color, scaling_modifier, pipe.debug))
    if pipe.compute_cov3D_python:
        # the reference would crash here (scales is None at :239); only the
        # scale / rotation path is live
        raise NotImplementedError("compute_cov3D_python is not a live path of the reference's prefilter_voxel")
    with torch.no_grad():
        from .scene_model import AnchorGaussianModel
        if type(pc).get_scaling is AnchorGaussianModel.get_scaling:
            scales = torch.exp(pc._scaling[:, :3])      # = get_scaling[:, :3] without exp of the other three columns + a strided copy
        else:
            scales = pc.get_scaling[:, :3]
        radii_pure = rasterizer.visible_filter(means3D=pc.get_anchor, scales=scales,
                                               rotations=pc.get_rotation, cov3D_precomp=None)
    return radii_pure > 0
