"""Plain-torch restatements of reference op chains that the product runs as HIP kernels.  TEST
INFRASTRUCTURE ONLY (the checkers of the kernels; each is itself pinned by a golden fixture captured
from the reference's own Python): nothing under splatco_amd/ imports this file.

  expand_torch_chain     gaussian_renderer/__init__.py:68-111   -> checker of csrc/expand.hip
  training_statis_torch  scene/gaussian_model.py:761-782        -> checker of csrc/densify.hip
  tv_add_grad_torch      scene/grids.py:240-250 (closed form)   -> checker of csrc/tv.hip, CPU stand-in in the gloo tests
  adam_apply_torch       torch.optim.Adam's update (no decay)   -> CPU stand-in of csrc/adam.hip in the gloo tests
"""
import torch
import torch.nn.functional as F


def expand_torch_chain(neural_opacity, color, scale_rot, grid_offsets, grid_scaling, anchor, k):
    """mask = neural_opacity > 0; gather the kept candidates; scaling / rot / xyz post-processing."""
    V = anchor.shape[0]
    mask = (neural_opacity > 0.0).view(-1)
    opacity = neural_opacity[mask]
    per_anchor = torch.cat([grid_scaling, anchor], dim=-1)                       # [V, 6 + 3]
    per_cand = per_anchor.unsqueeze(1).expand(V, k, 9).reshape(V * k, 9)
    kept = torch.cat([per_cand, color, scale_rot, grid_offsets.reshape(-1, 3)], dim=-1)[mask]
    gs, anc, color, scale_rot, offsets = kept.split([6, 3, 3, 7, 3], dim=-1)
    scaling = gs[:, 3:] * torch.sigmoid(scale_rot[:, :3])
    rot = F.normalize(scale_rot[:, 3:7])
    xyz = anc + offsets * gs[:, :3]
    return xyz, color, opacity, scaling, rot, mask


def training_statis_torch(opacity_accum, anchor_demon, offset_gradient_accum, offset_denom, n_offsets,
                          viewspace_point_grad, opacity, update_filter, offset_selection_mask, anchor_visible_mask):
    """Boolean-mask formulation of the four accumulator updates (in place)."""
    k = n_offsets
    op = opacity.detach().reshape(-1, k).clamp(min=0).sum(dim=1, keepdim=True)
    opacity_accum[anchor_visible_mask] += op
    anchor_demon[anchor_visible_mask] += 1
    cand_visible = anchor_visible_mask.unsqueeze(1).expand(-1, k).reshape(-1)     # [N*k]
    counted = torch.zeros_like(cand_visible)
    counted[cand_visible] = offset_selection_mask
    sel = counted.clone()
    counted[sel] = update_filter
    offset_gradient_accum[counted] += viewspace_point_grad[update_filter, :2].norm(dim=-1, keepdim=True)
    offset_denom[counted] += 1
    return opacity_accum, anchor_demon, offset_gradient_accum, offset_denom


def statis_increments_torch(n_offsets, viewspace_point_grad, opacity, update_filter, offset_selection_mask):
    """CPU stand-in of stats.statis_increments (csrc/densify.hip) for the gloo tests of the sharded step."""
    k = n_offsets
    inc_op = opacity.detach().reshape(-1, k).clamp(min=0).sum(dim=1)
    sel = offset_selection_mask.reshape(-1).nonzero().squeeze(1)                  # candidate of every Gaussian
    inc_g = torch.full((offset_selection_mask.numel(),), -1.0)
    inc_g[sel[update_filter]] = viewspace_point_grad[update_filter, :2].norm(dim=-1)
    return inc_op.float(), inc_g


def statis_apply_torch(opacity_accum, anchor_demon, offset_gradient_accum, offset_denom, n_offsets, visible_index,
                       inc_opacity, inc_grad):
    k = n_offsets
    opacity_accum[visible_index, 0] += inc_opacity
    anchor_demon[visible_index, 0] += 1
    rows = (visible_index[:, None] * k + torch.arange(k)[None, :]).reshape(-1)
    hit = inc_grad >= 0
    offset_gradient_accum[rows[hit], 0] += inc_grad[hit]
    offset_denom[rows[hit], 0] += 1


def tv_add_grad_torch(entries):
    """CPU stand-in of splatco_amd.tv.tv_add_grad (csrc/tv.hip): the closed-form derivative of the smooth-L1 total
    variation, coef * clamp(neighbour difference, -1, 1), added into plane.grad.  Pinned by tests/golden/tv.npz."""
    from splatco_amd.tv import tv_coef
    with torch.no_grad():
        for p, w in entries:
            c = torch.tensor(tv_coef(w), dtype=p.dtype)
            g = torch.zeros_like(p)
            for dim in (2, 3):
                n = p.shape[dim]
                if n < 2:
                    continue
                h = (p.narrow(dim, 1, n - 1) - p.narrow(dim, 0, n - 1)).clamp(-1, 1) * c
                g.narrow(dim, 1, n - 1).add_(h)
                g.narrow(dim, 0, n - 1).sub_(h)
            p.grad = g if p.grad is None else p.grad.add_(g)


def adam_apply_torch(entries, beta1, beta2, eps, device=None):
    """CPU stand-in of splatco_amd.adam._adam_apply (csrc/adam.hip): the same elementwise update on every
    (param, grad, exp_avg, exp_avg_sq, step_size, bias_correction2_sqrt) entry, in the tensors' own dtype."""
    with torch.no_grad():
        for p, g, m, v, step_size, bc2 in entries:
            m.add_((g - m) * (1.0 - beta1))
            v.mul_(beta2).add_(((1.0 - beta2) * g) * g)
            p.sub_(step_size * (m / (v.sqrt() / bc2 + eps)))
