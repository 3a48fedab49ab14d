"""Anchor densification bookkeeping: the consumer side of the rasterizer's
dL/dmeans2D output, restated from GaussianModel.training_statis / adjust_anchor / anchor_growing /
prune_anchor / compute_curvature (scene/gaussian_model.py:761-782, 784-997, 1092-1110) with the same
semantics (including the reference's quirks, noted inline) but without its host-side loops:

  * the duplicate test of new voxels against the existing anchors is a sort-based set membership
    (torch.unique over the stacked voxel coordinates) instead of the chunked O(N*M) broadcast compare
    (:870-879);
  * the curvature pass builds all k-neighbourhood covariances at once and calls one batched
    eigvalsh instead of a Python loop over the anchors (:1099-1108).

Everything stays on the model's device; the only host synchronisations are the data-dependent shapes.
Fixture: tests/golden/densify.npz (captured from the reference's own Python by tools/make_golden.py)."""
import torch
from torch import nn


# optimizer param-group name -> model attribute (scene/gaussian_model.py:520-531)
_PARAMS = {"anchor": "_anchor", "offset": "_offset", "anchor_feat": "_anchor_feat", "opacity": "_opacity",
           "scaling": "_scaling", "rotation": "_rotation"}
_SKIP = ("mlp", "conv", "feat_base", "embedding", "feat_planes")


def inverse_sigmoid(x):                      # utils/general_utils.py:17-18
    return torch.log(x / (1 - x))


def _knn_grid(pts, per_cell=2.0):
    """Uniform grid over the bounding box with about `per_cell` points per cell; returns (lo[3], h, n[3])."""
    lo, hi = pts.amin(dim=0), pts.amax(dim=0)
    ext = (hi - lo).double()
    ext = torch.maximum(ext, ext.max().clamp_min(1e-30) * 1e-6)
    N = pts.shape[0]
    h = float((ext.prod() * per_cell / N) ** (1.0 / 3.0))
    n = torch.ceil(ext / h).clamp_min(1)
    over = float(n.prod()) / (4.0 * N + 64)              # flat / degenerate clouds: bound the cell count
    if over > 1.0:
        h *= over ** (1.0 / 3.0)
        n = torch.ceil(ext / h).clamp_min(1)
    return lo, h, [int(v) for v in n.tolist()]


def _knn_indices(points, k):
    """indices [N, k] of the k nearest OTHER points, nearest first (column 0 of the k+1 query dropped,
    as :1096-1100 does).  Device tensors: grid-bucketed exact search in csrc/densify.hip (any N; the reference
    goes to sklearn on the host); CPU tensors (the golden tests): brute force."""
    N = points.shape[0]
    if points.is_cuda:
        import ctypes as C
        from . import _C
        from .rasterizer import _stream
        if k > 16:
            raise ValueError("k <= 16")
        pts = points.detach().float().contiguous()
        lo, h, (nx, ny, nz) = _knn_grid(pts)
        cell = ((pts - lo) / h).floor().long()
        cell = torch.minimum(cell.clamp_min(0), torch.tensor([nx - 1, ny - 1, nz - 1], device=pts.device))
        key = (cell[:, 2] * ny + cell[:, 1]) * nx + cell[:, 0]
        skey, order = torch.sort(key)
        ncell = nx * ny * nz
        cell_start = torch.searchsorted(skey, torch.arange(ncell + 1, device=pts.device)).int()
        spts = pts.index_select(0, order).contiguous()
        out = torch.empty(N, k, dtype=torch.long, device=pts.device)
        grid = (C.c_float * 7)(float(lo[0]), float(lo[1]), float(lo[2]), h, nx, ny, nz)
        with torch.cuda.device(pts.device):
            _C.check(_C.lib.scr_knn(N, k, grid, spts.data_ptr(), order.data_ptr(), cell_start.data_ptr(), out.data_ptr(),
                                    _stream()))
        return out
    out = torch.empty(N, k, dtype=torch.long, device=points.device)
    step = max(1, (1 << 24) // max(N, 1))
    for s in range(0, N, step):
        d = torch.cdist(points[s:s + step], points)
        out[s:s + step] = d.topk(k + 1, dim=1, largest=False).indices[:, 1:]
    return out


def compute_curvature(points, k=10):
    """lambda_min / sum(lambda) of the covariance of each point's k nearest neighbours (:1092-1110)."""
    pts = points.detach()
    idx = _knn_indices(pts, k)
    if pts.is_cuda:          # covariance + closed-form eigenvalues on the device (fp64), one thread per point
        from . import _C
        from .rasterizer import _stream
        p32 = pts.float().contiguous()
        out = torch.empty(p32.shape[0], dtype=torch.float32, device=p32.device)
        with torch.cuda.device(p32.device):
            _C.check(_C.lib.scr_knn_curvature(p32.shape[0], k, p32.data_ptr(), idx.data_ptr(), out.data_ptr(), _stream()))
        return out
    nb = pts[idx]                                                   # [N,k,3]
    c = nb - nb.mean(dim=1, keepdim=True)
    cov = c.transpose(1, 2) @ c / (k - 1)
    ev = torch.linalg.eigvalsh(cov)                                 # ascending
    return ev[:, 0] / ev.sum(dim=1)


class AnchorDensifier:
    """Owns the four accumulators of the reference's GaussianModel and edits the model's per-anchor
    parameters + their Adam state in place of `adjust_anchor`."""

    def __init__(self, model, optimizer, voxel_size=0.001, update_depth=3, update_init_factor=16,
                 update_hierachy_factor=4, seed=None, rand=None):
        """seed: draw the random candidate picks of anchor_growing (:843-844, torch.rand_like on the global
        generator in the reference) from a private generator seeded with it -- every rank of the sharded --mv step
        passes the same seed, so that the replicas grow identical anchor sets."""
        self.model, self.optimizer = model, optimizer
        self.rand = rand                  # test hook: rand(shape, device) -> uniform [0,1) floats (e.g. the CPU stream of a fixture)
        self.generator = None
        if seed is not None:
            self.generator = torch.Generator(device=model._anchor.device)
            self.generator.manual_seed(int(seed))
        self.voxel_size, self.update_depth = voxel_size, update_depth
        self.update_init_factor, self.update_hierachy_factor = update_init_factor, update_hierachy_factor
        self.n_offsets, self.feat_dim = model.n_offsets, model.feat_dim
        N, dev = model._anchor.shape[0], model._anchor.device
        self.opacity_accum = torch.zeros(N, 1, device=dev)          # :513-518
        self.offset_gradient_accum = torch.zeros(N * self.n_offsets, 1, device=dev)
        self.offset_denom = torch.zeros(N * self.n_offsets, 1, device=dev)
        self.anchor_demon = torch.zeros(N, 1, device=dev)
        self.max_radii2D = torch.zeros(N, device=dev)

    # ---- :761-782
    def training_statis(self, viewspace_point_tensor, opacity, update_filter, offset_selection_mask, anchor_visible_mask):
        from . import stats
        stats.training_statis(self.opacity_accum, self.anchor_demon, self.offset_gradient_accum, self.offset_denom,
                              self.n_offsets, viewspace_point_tensor.grad, opacity, update_filter, offset_selection_mask,
                              anchor_visible_mask)

    def apply_statis(self, visible_index, inc_opacity, inc_grad):
        """Adds one view's increments (stats.statis_increments, possibly computed on another rank)."""
        from . import stats
        stats.statis_apply(self.opacity_accum, self.anchor_demon, self.offset_gradient_accum, self.offset_denom,
                           self.n_offsets, visible_index, inc_opacity, inc_grad)

    # ---- optimizer surgery (:738-759, 784-818)
    def _groups(self):
        for group in self.optimizer.param_groups:
            if any(s in group["name"] for s in _SKIP):
                continue
            if group["name"] in _PARAMS:
                assert len(group["params"]) == 1
                yield group

    def _replace(self, group, new_value, state_edit):
        old = group["params"][0]
        stored = self.optimizer.state.get(old, None)
        if stored is not None:
            stored["exp_avg"], stored["exp_avg_sq"] = state_edit(stored["exp_avg"]), state_edit(stored["exp_avg_sq"])
            del self.optimizer.state[old]
        param = nn.Parameter(new_value.requires_grad_(True))
        group["params"][0] = param
        if stored is not None:
            self.optimizer.state[param] = stored
        setattr(self.model, _PARAMS[group["name"]], param)
        return param

    def cat_tensors_to_optimizer(self, tensors_dict):
        for group in self._groups():
            ext = tensors_dict[group["name"]]
            self._replace(group, torch.cat((group["params"][0].detach(), ext), dim=0),
                          lambda s: torch.cat((s, torch.zeros_like(ext)), dim=0))

    def prune_anchor(self, mask):
        keep = ~mask
        for group in self._groups():
            p = self._replace(group, group["params"][0].detach()[keep], lambda s: s[keep])
            if group["name"] == "scaling":                   # reference quirk: clamps the RAW (log) values (:803-806)
                with torch.no_grad():
                    p[:, 3:].clamp_(max=0.05)

    @torch.no_grad()
    def sort_anchors(self, perm=None):
        """Puts the anchors -- parameters, Adam moments and the four densification accumulators -- in Morton order
        (scene_model.morton_order): new row i = old row perm[i].  Not in the reference (the anchor
        order carries no meaning there); call it after adjust_anchor to keep 64 consecutive anchors spatial neighbours."""
        if perm is None:
            from .scene_model import morton_order
            perm = morton_order(self.model._anchor)
        k = self.n_offsets
        for group in self._groups():
            self._replace(group, group["params"][0].detach()[perm].contiguous(), lambda s: s[perm].contiguous())
        self.opacity_accum = self.opacity_accum[perm]
        self.anchor_demon = self.anchor_demon[perm]
        self.offset_gradient_accum = self.offset_gradient_accum.view(-1, k)[perm].reshape(-1, 1)
        self.offset_denom = self.offset_denom.view(-1, k)[perm].reshape(-1, 1)
        return perm

    # ---- :826-925
    @torch.no_grad()
    def anchor_growing(self, grads, threshold, offset_mask):
        m, k = self.model, self.n_offsets
        init_length = m._anchor.shape[0] * k
        for i in range(self.update_depth):
            cur_threshold = threshold * ((self.update_hierachy_factor // 2) ** i)
            candidate_mask = (grads >= cur_threshold) & offset_mask
            if self.rand is not None:
                rand = self.rand(candidate_mask.shape, candidate_mask.device)
            elif self.generator is None:
                rand = torch.rand_like(candidate_mask.float())
            else:
                rand = torch.rand(candidate_mask.shape, device=candidate_mask.device, generator=self.generator)
            rand_mask = rand > (0.5 ** (i + 1))                                          # random pick (:844)
            candidate_mask = candidate_mask & rand_mask
            length_inc = m._anchor.shape[0] * k - init_length
            if length_inc == 0:
                if i > 0:                                     # reference quirk: deeper levels run only after growth
                    continue
            else:
                candidate_mask = torch.cat([candidate_mask, torch.zeros(length_inc, dtype=torch.bool, device=grads.device)])
            scaling = torch.exp(m._scaling)
            all_xyz = m._anchor.unsqueeze(1) + m._offset * scaling[:, :3].unsqueeze(1)
            size_factor = self.update_init_factor // (self.update_hierachy_factor ** i)
            cur_size = self.voxel_size * size_factor
            grid_coords = torch.round(m._anchor / cur_size).int()
            selected_xyz = all_xyz.view(-1, 3)[candidate_mask]
            selected_grid_coords = torch.round(selected_xyz / cur_size).int()
            uniq, inverse = torch.unique(selected_grid_coords, return_inverse=True, dim=0)
            # voxels already holding an anchor: membership through one more sort instead of the
            # chunked [M,1,3] == [1,4096,3] compare of :870-879
            S = uniq.shape[0]
            _, inv_all = torch.unique(torch.cat([uniq, grid_coords], dim=0), return_inverse=True, dim=0)
            taken = torch.zeros(int(inv_all.max()) + 1 if inv_all.numel() else 0, dtype=torch.bool, device=grads.device)
            taken[inv_all[S:]] = True
            fresh = ~taken[inv_all[:S]]
            candidate_anchor = uniq[fresh] * cur_size
            M = candidate_anchor.shape[0]
            if M == 0:
                continue
            new_scaling = torch.log(torch.ones_like(candidate_anchor).repeat([1, 2]).float() * cur_size)
            new_rotation = torch.zeros(M, 4, device=grads.device)
            new_rotation[:, 0] = 1.0
            new_opacities = inverse_sigmoid(0.1 * torch.ones(M, 1, device=grads.device))
            new_feat = m._anchor_feat.unsqueeze(1).repeat([1, k, 1]).view(-1, self.feat_dim)[candidate_mask]
            pooled = torch.zeros(S, self.feat_dim, device=grads.device, dtype=new_feat.dtype)
            pooled = pooled.scatter_reduce(0, inverse.unsqueeze(1).expand(-1, self.feat_dim), new_feat, "amax",
                                           include_self=False)                          # scatter_max (:895)
            new_offsets = torch.zeros(M, k, 3, device=grads.device)
            self.anchor_demon = torch.cat([self.anchor_demon, torch.zeros(M, 1, device=grads.device)], dim=0)
            self.opacity_accum = torch.cat([self.opacity_accum, torch.zeros(M, 1, device=grads.device)], dim=0)
            self.cat_tensors_to_optimizer({"anchor": candidate_anchor, "scaling": new_scaling, "rotation": new_rotation,
                                           "anchor_feat": pooled[fresh], "offset": new_offsets,
                                           "opacity": new_opacities})

    # ---- :929-997
    @torch.no_grad()
    def adjust_anchor(self, iteration, check_interval=100, success_threshold=0.8, grad_threshold=0.0002,
                      min_opacity=0.005):
        m, k = self.model, self.n_offsets
        grads = self.offset_gradient_accum / self.offset_denom
        grads[grads.isnan()] = 0.0
        grads_norm = torch.norm(grads, dim=-1)
        offset_mask = (self.offset_denom > check_interval * success_threshold * 0.5).squeeze(dim=1)
        if iteration % 3000 == 0 or iteration == 1600:             # curvature densification (:936-945)
            curvature_mask = compute_curvature(m._anchor).view(m._anchor.shape[0], -1) <= 0.1
            # reference quirk: k stacked copies of the per-anchor mask (not a per-anchor repeat)
            curvature_mask = torch.cat([curvature_mask.squeeze()] * k, dim=0)
            offset_mask = offset_mask | curvature_mask
        self.anchor_growing(grads_norm, grad_threshold, offset_mask)

        dev = self.offset_denom.device
        pad = m._anchor.shape[0] * k - self.offset_denom.shape[0]
        self.offset_denom[offset_mask] = 0
        self.offset_denom = torch.cat([self.offset_denom, torch.zeros(pad, 1, device=dev)], dim=0)
        self.offset_gradient_accum[offset_mask] = 0
        self.offset_gradient_accum = torch.cat([self.offset_gradient_accum, torch.zeros(pad, 1, device=dev)], dim=0)

        prune_mask = (self.opacity_accum < min_opacity * self.anchor_demon).squeeze(dim=1)
        anchors_mask = (self.anchor_demon > check_interval * success_threshold).squeeze(dim=1)
        prune_mask = prune_mask & anchors_mask
        self.offset_denom = self.offset_denom.view(-1, k)[~prune_mask].reshape(-1, 1)
        self.offset_gradient_accum = self.offset_gradient_accum.view(-1, k)[~prune_mask].reshape(-1, 1)
        self.opacity_accum[anchors_mask] = 0
        self.anchor_demon[anchors_mask] = 0
        self.opacity_accum = self.opacity_accum[~prune_mask]
        self.anchor_demon = self.anchor_demon[~prune_mask]
        if prune_mask.shape[0] > 0:
            self.prune_anchor(prune_mask)
        self.max_radii2D = torch.zeros(m._anchor.shape[0], device=dev)
