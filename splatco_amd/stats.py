"""Consumer of the means2D gradient: the densification statistics of
GaussianModel.training_statis (scene/gaussian_model.py:761-782), restated as a pure function over
the four accumulators.  Pins the shape / units contract of the rasterizer's dL/dmeans2D output
(norm over [:, :2] of the NDC-space gradient); fixture tests/golden/training_statis.npz."""
import torch


def training_statis(opacity_accum, anchor_demon, offset_gradient_accum, offset_denom, n_offsets,
                    viewspace_point_grad, opacity, update_filter, offset_selection_mask, anchor_visible_mask):
    temp_opacity = opacity.clone().view(-1).detach()
    temp_opacity[temp_opacity < 0] = 0
    temp_opacity = temp_opacity.view([-1, n_offsets])
    opacity_accum[anchor_visible_mask] += temp_opacity.sum(dim=1, keepdim=True)
    anchor_demon[anchor_visible_mask] += 1
    anchor_visible_mask = anchor_visible_mask.unsqueeze(dim=1).repeat([1, n_offsets]).view(-1)
    combined_mask = torch.zeros_like(offset_gradient_accum, dtype=torch.bool).squeeze(dim=1)
    combined_mask[anchor_visible_mask] = offset_selection_mask
    temp_mask = combined_mask.clone()
    combined_mask[temp_mask] = update_filter
    grad_norm = torch.norm(viewspace_point_grad[update_filter, :2], dim=-1, keepdim=True)
    offset_gradient_accum[combined_mask] += grad_norm
    offset_denom[combined_mask] += 1
    return opacity_accum, anchor_demon, offset_gradient_accum, offset_denom
