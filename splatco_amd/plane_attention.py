"""TriPlaneAttention of the level-0 grid on MI355X (host side of csrc/attention.hip).

`attended_pair_planes(xy, xz, yz, ta)` is what PlaneGrid.compute_planes_feat does with its attention module on every
call (scene/grids.py:166-168: `tri = self.TA(cat(xy, xz, yz)); xyA, xzA, yzA = chunk(tri, 3)`) plus the stacking of each
plane on its attended twin that the tri-plane sampler wants: three tensors [1, 2R, H, W] = (plane | attended plane).
ChannelAttention's pools, SpatialAttention's channel mean / max, the 7x7 convolution, both sigmoids and both products
run in csrc/attention.hip; only the 15-number shared MLP (two bias-free 1x1 convolutions, scene/grids.py:26-27) stays
in the framework.  No MIOpen call is left on the path.
"""
import torch
import torch.nn.functional as F

from . import _C
from .rasterizer import _stream


def _channel_mlp(avg, mx, w1, w2):
    """sigmoid(sharedMLP(avg) + sharedMLP(max)) for pooled vectors [C] (scene/grids.py:33-36); the 1x1 convolutions
    are matrix-vector products."""
    a, b = w1.flatten(1), w2.flatten(1)
    return torch.sigmoid(F.linear(F.relu(F.linear(avg, a)), b) + F.linear(F.relu(F.linear(mx, a)), b))


class _AttendedPairs(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xy, xz, yz, w1, w2, wc):
        with torch.cuda.device(xy.device):     # kernels launch on the CURRENT device: make it the tensors' device
            return _AttendedPairs._forward(ctx, xy, xz, yz, w1, w2, wc)

    @staticmethod
    def _forward(ctx, xy, xz, yz, w1, w2, wc):
        R, H, W = xy.shape[1], xy.shape[2], xy.shape[3]
        dev, C3 = xy.device, 3 * R
        planes = [p.detach().contiguous().float() for p in (xy, xz, yz)]
        scratch = torch.empty(_C.lib.scr_tpa_scratch_bytes(R, H, W), dtype=torch.uint8, device=dev)
        avg, mx = torch.empty(C3, device=dev), torch.empty(C3, device=dev)
        arg = torch.empty(C3, dtype=torch.int32, device=dev)
        _C.check(_C.lib.scr_tpa_stats(R, H, W, *(p.data_ptr() for p in planes), avg.data_ptr(), mx.data_ptr(),
                                      arg.data_ptr(), scratch.data_ptr(), _stream()))
        ca = _channel_mlp(avg, mx, w1.detach().float(), w2.detach().float()).contiguous()
        wcc = wc.detach().contiguous().float()
        s = torch.empty(2, H, W, device=dev)
        am = torch.empty(H * W, dtype=torch.uint8, device=dev)
        sa = torch.empty(H, W, device=dev)
        out = [torch.empty(1, 2 * R, H, W, device=dev) for _ in range(3)]
        _C.check(_C.lib.scr_tpa_forward(R, H, W, *(p.data_ptr() for p in planes), ca.data_ptr(), wcc.data_ptr(),
                                        s.data_ptr(), am.data_ptr(), sa.data_ptr(), *(o.data_ptr() for o in out), _stream()))
        ctx.save_for_backward(*planes, w1, w2, wcc, avg, mx, arg, ca, s, am, sa)
        ctx.dims = (R, H, W)
        return tuple(out)

    @staticmethod
    def backward(ctx, g0, g1, g2):
        with torch.cuda.device(g0.device):
            return _AttendedPairs._backward(ctx, g0, g1, g2)

    @staticmethod
    def _backward(ctx, g0, g1, g2):
        p0, p1, p2, w1, w2, wcc, avg, mx, arg, ca, s, am, sa = ctx.saved_tensors
        R, H, W = ctx.dims
        dev, C3 = p0.device, 3 * R
        gs = [g.contiguous().float() for g in (g0, g1, g2)]
        d = [torch.empty(1, R, H, W, device=dev) for _ in range(3)]
        dca, dw = torch.empty(C3, device=dev), torch.empty(2 * 49, device=dev)
        scratch = torch.empty(_C.lib.scr_tpa_scratch_bytes(R, H, W), dtype=torch.uint8, device=dev)
        _C.check(_C.lib.scr_tpa_backward(R, H, W, p0.data_ptr(), p1.data_ptr(), p2.data_ptr(), ca.data_ptr(), wcc.data_ptr(),
                                         s.data_ptr(), am.data_ptr(), sa.data_ptr(), *(g.data_ptr() for g in gs),
                                         *(t.data_ptr() for t in d), dca.data_ptr(), dw.data_ptr(), scratch.data_ptr(),
                                         _stream()))
        # the 15-number MLP and its sigmoid: re-run under autograd (a dozen tiny kernels)
        with torch.enable_grad():
            a_, m_ = avg.detach().requires_grad_(True), mx.detach().requires_grad_(True)
            w1_, w2_ = w1.detach().float().requires_grad_(True), w2.detach().float().requires_grad_(True)
            davg, dmx, dw1, dw2 = torch.autograd.grad(_channel_mlp(a_, m_, w1_, w2_), (a_, m_, w1_, w2_), dca)
        _C.check(_C.lib.scr_tpa_backward_stats(R, H, W, davg.contiguous().data_ptr(), dmx.contiguous().data_ptr(),
                                               arg.data_ptr(), *(t.data_ptr() for t in d), _stream()))
        return d[0], d[1], d[2], dw1.reshape(w1.shape), dw2.reshape(w2.shape), dw.reshape(1, 2, 7, 7)


def fused_ok(xy, xz, yz, ta):
    """csrc/attention.hip covers the reference's module: equal plane sizes, at most 24 stacked channels, a 7x7 window."""
    conv = ta.sa.conv
    return (xy.is_cuda and xy.shape == xz.shape == yz.shape and 3 * xy.shape[1] <= 24 and tuple(conv.kernel_size) == (7, 7)
            and tuple(conv.padding) == (3, 3) and conv.bias is None and xy.dtype == torch.float32)


def attended_pair_planes(xy, xz, yz, ta):
    """(cat(xy, xyA), cat(xz, xzA), cat(yz, yzA)) with (xyA, xzA, yzA) = chunk(ta(cat(xy, xz, yz)), 3)."""
    return _AttendedPairs.apply(xy, xz, yz, ta.ca.sharedMLP[0].weight, ta.ca.sharedMLP[2].weight, ta.sa.conv.weight)
