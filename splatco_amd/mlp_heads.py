"""The three MLP heads of generate_neural_gaussians as one autograd op (host side of csrc/mlp_heads.hip).

Reference: gaussian_renderer/__init__.py:58-93 with the default flags (add_opacity_dist / add_cov_dist /
add_color_dist False, appearance_dim 0) and scene/gaussian_model.py:315-337 (opacity 99->32->10 Tanh, colour
99->32->30 Sigmoid, cov 99->32->70).  The op takes what the reference concatenates -- feat [V,32], the anchor
positions (ob_view is computed inside), geo_fea [V,64] -- and never builds x, the hidden layer or their
gradients in HBM.  fp32 MFMA on the device; there is no CPU path.
"""
import torch

from . import _C
from .rasterizer import _stream


def supported(pc, feat, geo_a, geo_b=None):
    """The kernel is written for the reference's sizes: feat 32, geo_fea 64, hidden 32, n_offsets 10, plain heads."""
    heads = (pc.get_opacity_mlp, pc.get_color_mlp, pc.get_cov_mlp)
    try:
        ok = all(isinstance(h, torch.nn.Sequential) and isinstance(h[0], torch.nn.Linear) and isinstance(h[1], torch.nn.ReLU)
                 and isinstance(h[2], torch.nn.Linear) for h in heads)
        ok = ok and isinstance(heads[0][3], torch.nn.Tanh) and isinstance(heads[1][3], torch.nn.Sigmoid) and len(heads[2]) == 3
        ok = ok and all(h[0].weight.shape == (32, 99) for h in heads)
        ok = ok and [h[2].weight.shape[0] for h in heads] == [10, 30, 70]
    except (IndexError, AttributeError):
        return False
    geo_ok = (geo_a.shape[1] == 64) if geo_b is None else (geo_a.shape[1] == 32 and geo_b.shape[1] == 32)
    return bool(ok and feat.is_cuda and feat.shape[1] == 32 and geo_ok and feat.dtype == torch.float32)


class _MlpHeads(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, anchor, campos, geo_a, geo_b, w1, b1, w2o, b2o, w2c, b2c, w2v, b2v):
        c = lambda t: t.detach().contiguous().float()
        # feat may be columns 0..31 of the gather's [V,72] matrix (anchor_gather: not written a second time): read in place
        in_place = (feat.dtype == torch.float32 and feat.dim() == 2 and feat.stride(1) == 1 and feat.stride(0) % 4 == 0
                    and feat.data_ptr() % 16 == 0 and feat.shape[0] > 1)
        feat = feat.detach() if in_place else c(feat)
        anchor, campos, geo_a, geo_b = c(anchor), c(campos), c(geo_a), c(geo_b)
        ws = [c(t) for t in (w1, b1, w2o, b2o, w2c, b2c, w2v, b2v)]
        V, dev = feat.shape[0], feat.device
        new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        out_o, out_c, out_v = new(V, 10), new(V, 30), new(V, 70)
        hidden = _C.scratch(_C.lib.scr_mlp_heads_hidden_bytes(V), dev)
        if V:
            with torch.cuda.device(dev):
                _C.check(_C.lib.scr_mlp_heads_forward(V, feat.data_ptr(), feat.stride(0), anchor.data_ptr(), campos.data_ptr(), geo_a.data_ptr(),
                                                      geo_b.data_ptr(), *[t.data_ptr() for t in ws], hidden.data_ptr(), out_o.data_ptr(),
                                                      out_c.data_ptr(), out_v.data_ptr(), _stream()))
        ctx.save_for_backward(feat, anchor, campos, geo_a, geo_b, ws[0], ws[2], ws[4], ws[6], hidden, out_o, out_c)
        return out_o, out_c, out_v

    @staticmethod
    def backward(ctx, g_o, g_c, g_v):
        feat, anchor, campos, geo_a, geo_b, w1, w2o, w2c, w2v, hidden, out_o, out_c = ctx.saved_tensors
        V, dev = feat.shape[0], feat.device
        z = lambda g, n: (torch.zeros(V, n, dtype=torch.float32, device=dev) if g is None else g.contiguous().float())
        g_o, g_c, g_v = z(g_o, 10), z(g_c, 30), z(g_v, 70)
        new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        d_feat, d_anchor, d_geo_a, d_geo_b = new(V, 32), new(V, 3), new(V, 32), new(V, 32)
        d_w1, d_b1 = new(96, 99), new(96)
        d_w2o, d_b2o, d_w2c, d_b2c, d_w2v, d_b2v = new(10, 32), new(10), new(30, 32), new(30), new(70, 32), new(70)
        if V == 0:
            for t in (d_w1, d_b1, d_w2o, d_b2o, d_w2c, d_b2c, d_w2v, d_b2v):
                t.zero_()
        else:
            partial = _C.scratch(_C.lib.scr_mlp_heads_partial_bytes(V), dev)
            with torch.cuda.device(dev):
                _C.check(_C.lib.scr_mlp_heads_backward(
                    V, feat.data_ptr(), feat.stride(0), anchor.data_ptr(), campos.data_ptr(), geo_a.data_ptr(), geo_b.data_ptr(), w1.data_ptr(), w2o.data_ptr(),
                    w2c.data_ptr(), w2v.data_ptr(), hidden.data_ptr(), out_o.data_ptr(), out_c.data_ptr(), g_o.data_ptr(),
                    g_c.data_ptr(), g_v.data_ptr(), partial.data_ptr(), d_feat.data_ptr(), d_anchor.data_ptr(),
                    d_geo_a.data_ptr(), d_geo_b.data_ptr(), d_w1.data_ptr(), d_b1.data_ptr(), d_w2o.data_ptr(), d_b2o.data_ptr(), d_w2c.data_ptr(),
                    d_b2c.data_ptr(), d_w2v.data_ptr(), d_b2v.data_ptr(), _stream()))
        return d_feat, d_anchor, None, d_geo_a, d_geo_b, d_w1, d_b1, d_w2o, d_b2o, d_w2c, d_b2c, d_w2v, d_b2v


def mlp_heads(pc, feat, anchor, camera_center, geo_fea, geo_b=None):
    """(neural_opacity [V,10], color [V,30], scale_rot [V,70]) of the visible anchors.  geo_fea [V,64], or its two halves
    (geo_fea [V,32], geo_b [V,32]) as FeaturePlanes' two GEMMs leave them (no concatenation)."""
    if geo_b is None:
        geo_fea, geo_b = geo_fea[:, :32], geo_fea[:, 32:]
    ho, hc, hv = pc.get_opacity_mlp, pc.get_color_mlp, pc.get_cov_mlp
    w1 = torch.cat([ho[0].weight, hc[0].weight, hv[0].weight], dim=0)
    b1 = torch.cat([ho[0].bias, hc[0].bias, hv[0].bias], dim=0)
    return _MlpHeads.apply(feat, anchor, camera_center, geo_fea, geo_b, w1, b1, ho[2].weight, ho[2].bias, hc[2].weight, hc[2].bias,
                           hv[2].weight, hv[2].bias)
