"""Host-side model pieces on the render() path (plain PyTorch; rocBLAS does the tiny GEMMs).

Restates -- does not import -- what gaussian_renderer.generate_neural_gaussians() reads from the
reference's GaussianModel: the tri-plane feature grids (scene/grids.py:22-64,102-201), the
multi-level FeaturePlanes / GaussianLearner (scene/gaussian_model.py:97-220) and the three MLP
heads (scene/gaussian_model.py:315-337).  Module / parameter names mirror the reference so its
state_dicts load unchanged (tests/golden/neural_gaussians.npz was captured from the reference).

Out of scope here: optimiser, densification, PLY io, entropy models, the
dead Spatial_CTX grids.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

NORM_LINEAR_HIP = True       # _NormLinearFn on the GPU: csrc/normlinear.hip (False: the torch ops, kept as the checker's other leg)
FUSE_NORM_LINEAR = True      # FeaturePlanes: fold the train-mode BatchNorms into their Linears (see _NormLinearFn)
# Measured and left off (round 3, cfg2, anchors in Morton order): tri-plane forward 1.58 -> 2.00 ms (60 floats per sample and
# plane keep fewer threads in flight than the line requests saved are worth once the gathers are mostly L2 hits), plane
# backward 2.31 -> 2.25 ms, step 22.7 -> 23.1 ms.  tests/test_gpu_renderer.py keeps the path checked.
STACK_LEVEL0 = os.environ.get("SPLATCO_STACK_LEVEL0", "0") != "0"   # FeaturePlanes: sample the attention grid and the same-size plain grid of level 0 as one stacked grid


class _TallLinearFn(torch.autograd.Function):
    """y = x W^T + b for x with millions of rows and a few dozen columns (one row per anchor).
    Forward and dL/dx are ordinary GEMMs; the weight gradient dW = dy^T x contracts over the row
    dimension (K ~ 10^6, output 32x99): hipBLASLt runs that as ONE tile without split-K (1.7 ms per
    layer at 0.93 M rows on MI355X, 43 % of the anchor path).  Here the contraction is split into
    2048-row slabs (a batched GEMM, one partial per slab) that are then summed: same math, fixed
    summation order, ~15x faster."""
    SLAB = 2048

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return F.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dx = dy @ weight if ctx.needs_input_grad[0] else None
        M, c = x.shape[0], _TallLinearFn.SLAB
        S = M // c
        dw = None
        if ctx.needs_input_grad[1]:
            dw = dy[S * c:].t() @ x[S * c:]
            if S:
                dw = dw + torch.bmm(dy[:S * c].view(S, c, -1).transpose(1, 2), x[:S * c].view(S, c, -1)).sum(0)
        db = dy.sum(0) if ctx.has_bias and ctx.needs_input_grad[2] else None
        return dx, dw, db


class TallLinear(nn.Linear):
    """nn.Linear (same parameters / state_dict keys) with the slab-split weight gradient."""

    @staticmethod
    def apply_weights(x, weight, bias):
        if x.dim() == 2 and x.shape[0] >= 4 * _TallLinearFn.SLAB and x.is_cuda:
            return _TallLinearFn.apply(x, weight, bias)
        return F.linear(x, weight, bias)

    def forward(self, x):
        return TallLinear.apply_weights(x, self.weight, self.bias)


def _tall_contract(a, b):
    """a^T b for two matrices with millions of rows (slab-split, fixed summation order; see _TallLinearFn)."""
    M, c = a.shape[0], _TallLinearFn.SLAB
    S = M // c
    out = a[S * c:].t() @ b[S * c:]
    if S:
        out = out + torch.bmm(a[:S * c].view(S, c, -1).transpose(1, 2), b[:S * c].view(S, c, -1)).sum(0)
    return out


def _col_var_mean(x):
    """Biased variance and mean of every column of a tall matrix, two-level (per 2048-row slab, then Chan's
    combination).  torch.var_mean(x, dim=0) on ROCm takes 150 ms for [4.6 M, 71] fp32 (0.5 ms for 60 columns)
    and loses 3 digits; this is 0.45 ms for either width (tools/exp/colstats_probe.py)."""
    V, d = x.shape
    c = _TallLinearFn.SLAB
    S = V // c
    if S == 0:
        return torch.var_mean(x, dim=0, unbiased=False)
    v_s, m_s = torch.var_mean(x[:S * c].view(S, c, d), dim=1, unbiased=False)
    r = V - S * c
    tot = m_s.sum(0) * c
    if r:
        v_r, m_r = torch.var_mean(x[S * c:], dim=0, unbiased=False)
        tot = tot + m_r * r
    mean = tot / V
    acc = c * (v_s + (m_s - mean) ** 2).sum(0)
    if r:
        acc = acc + r * (v_r + (m_r - mean) ** 2)
    return acc / V, mean


class _NormLinearFn(torch.autograd.Function):
    """y = xhat G^T + c with xhat = (x - mean(x)) / sqrt(var(x) + eps), the batch statistics taken over
    the rows (BatchNorm1d in training mode without its affine part, folded into G and c by the caller).

    Train-mode BatchNorm followed by Linear is linear in xhat, so the normalised copy of x is never
    materialised: forward = one statistics pass + ONE GEMM with column-scaled weights; backward =
    one tall contraction dy^T x (from which the weight gradient AND both BatchNorm reduction terms
    follow on 32 x d matrices), one GEMM dy G' and one fused multiply-add pass.  x has one row per
    anchor (millions) and d <= 131 columns, so every saved pass over [V, d] counts.
    Also returns (mean, biased var) for the running-statistics update."""

    @staticmethod
    def fused_ok(x, G):
        # csrc/normlinear.hip: 32 output features, at most 80 input columns, fp32 rows with unit column stride
        return (NORM_LINEAR_HIP and x.is_cuda and x.dtype == torch.float32 and G.dtype == torch.float32 and x.dim() == 2 and x.shape[0] >= 1
                and G.shape[0] == 32 and x.shape[1] <= 80 and x.stride(1) == 1)

    @staticmethod
    def forward(ctx, x, G, c, eps, col_stats=None, defer=None):
        ctx.fused = _NormLinearFn.fused_ok(x, G)
        ctx.defer = defer if ctx.fused else None      # anchor_gather.DeferredDx: the producer of x forms dx in ITS backward
        if ctx.fused:
            from . import _C
            from .rasterizer import _stream
            V, d = x.shape
            Gc, cc = G.detach().contiguous(), c.detach().contiguous().float()
            y = torch.empty(V, 32, dtype=torch.float32, device=x.device)
            mean, var, inv = (torch.empty(d, dtype=torch.float32, device=x.device) for _ in range(3))
            scratch = _C.scratch(_C.lib.scr_norm_linear_scratch_bytes(V), x.device)
            # column statistics the producer of x formed on the way (anchor_gather): [rows, 2, 80] partial sums about x[0]
            ok_stats = (col_stats is not None and col_stats.is_cuda and col_stats.dtype == torch.float32 and col_stats.dim() == 3
                        and col_stats.shape[1:] == (2, 80) and col_stats.shape[0] > 0 and col_stats.is_contiguous())
            with torch.cuda.device(x.device):
                _C.check(_C.lib.scr_norm_linear_forward(V, d, x.data_ptr(), x.stride(0), Gc.data_ptr(), cc.data_ptr(), float(eps),
                                                        y.data_ptr(), mean.data_ptr(), var.data_ptr(), inv.data_ptr(),
                                                        scratch.data_ptr(), col_stats.data_ptr() if ok_stats else None,
                                                        col_stats.shape[0] if ok_stats else 0, _stream()))
            ctx.save_for_backward(x, Gc, mean, inv)
            ctx.mark_non_differentiable(mean, var)
            return y, mean, var
        var, mean = _col_var_mean(x)
        inv = torch.rsqrt(var + eps)
        Gs = G * inv                                   # scale columns
        y = torch.addmm(c - Gs @ mean, x, Gs.t())
        ctx.save_for_backward(x, G, mean, inv)
        ctx.mark_non_differentiable(mean, var)
        return y, mean, var

    @staticmethod
    def backward(ctx, dy, _dm, _dv):
        x, G, mean, inv = ctx.saved_tensors
        V = x.shape[0]
        if ctx.fused:
            from . import _C
            from .rasterizer import _stream
            d = x.shape[1]
            if dy.dtype != torch.float32 or dy.stride(1) != 1 or dy.stride(0) % 4 or dy.data_ptr() % 16:
                dy = dy.contiguous().float()
            need_dx = ctx.needs_input_grad[0]
            defer = ctx.defer if need_dx and x.stride(1) == 1 and d == getattr(ctx.defer, "width", None) else None
            ldx = (d + 3) // 4 * 4                   # 16-byte aligned rows (the pad columns are never written or read)
            dx = torch.empty(V, ldx, dtype=torch.float32, device=x.device)[:, :d] if need_dx and defer is None else None
            dG = torch.empty(32, d, dtype=torch.float32, device=x.device)
            dc = torch.empty(32, dtype=torch.float32, device=x.device)
            coef = torch.empty(34, 80, dtype=torch.float32, device=x.device) if defer is not None else None
            scratch = _C.scratch(_C.lib.scr_norm_linear_scratch_bytes(V), x.device)
            with torch.cuda.device(x.device):
                _C.check(_C.lib.scr_norm_linear_backward(V, d, x.data_ptr(), x.stride(0), dy.data_ptr(), dy.stride(0), G.data_ptr(),
                                                         mean.data_ptr(), inv.data_ptr(), dx.data_ptr() if dx is not None else None,
                                                         ldx, dG.data_ptr(), dc.data_ptr(), scratch.data_ptr(),
                                                         coef.data_ptr() if coef is not None else None, _stream()))
            if defer is not None:
                # the producer of x (the anchor gather) forms dx = k0 + x k1 + dy Gi inside its own backward kernel: it gets the
                # coefficients, and autograd a gradient of zeros that occupies no memory (anchor_gather.DeferredDx)
                defer.coef, defer.dy, defer.x = coef, dy, x
                dx = torch.zeros((), dtype=torch.float32, device=x.device).expand(V, d)
            return dx, dG, dc, None, None, None
        if dy.stride(1) != 1:      # a column block of a wider gradient (row stride > width) is fine for every op below
            dy = dy.contiguous()
        sdy = dy.sum(0)
        H = (_tall_contract(dy, x) - sdy[:, None] * mean) * inv            # dy^T xhat            [out, d]
        dx = None
        if ctx.needs_input_grad[0]:
            u = G.t() @ sdy                                                  # sum_v dxhat          [d]
            w = (G * H).sum(0)                                               # sum_v dxhat * xhat   [d]
            k1 = -(inv * inv) * w / V
            k0 = -inv * u / V - mean * k1
            dx = torch.addmm(k0, dy, G * inv)
            dx.addcmul_(x, k1)
        return dx, H, sdy, None, None, None


def _ptr_table(tensors):
    import ctypes as C
    return (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


class _NormFold(torch.autograd.Function):
    """(G [32, d], c [32]) of sum_i Linear_i(BatchNorm_i(.)) from the pairs' parameters, as ONE launch per direction
    (csrc/normlinear.hip nl_fold_kernel / nl_fold_backward_kernel) instead of ~15 framework kernels on 32 x 71 numbers
    forward and twice that backward.  tensors = (W_0, b_0, gamma_0, beta_0, W_1, ...)."""

    @staticmethod
    def forward(ctx, d, widths, cols, col_at, *tensors):
        import ctypes as C
        from . import _C
        from .rasterizer import _stream
        L, dev = len(widths), tensors[0].device
        at = None if col_at is None else (C.c_uint8 * d)(*col_at)
        ts = [t.detach().contiguous().float() for t in tensors]
        W, b, ga, be = ts[0::4], ts[1::4], ts[2::4], ts[3::4]
        G, c = torch.empty(32, d, device=dev), torch.empty(32, device=dev)
        wi, ci = (C.c_int32 * L)(*widths), (C.c_int32 * L)(*cols)
        with torch.cuda.device(dev):
            _C.check(_C.lib.scr_norm_fold(L, d, wi, ci, at, _ptr_table(W), _ptr_table(b), _ptr_table(ga), _ptr_table(be),
                                          G.data_ptr(), c.data_ptr(), _stream(dev)))
        ctx.save_for_backward(*W, *ga, *be)
        ctx.meta = (d, tuple(widths), tuple(cols), None if col_at is None else tuple(col_at))
        return G, c

    @staticmethod
    def backward(ctx, dG, dc):
        import ctypes as C
        from . import _C
        from .rasterizer import _stream
        d, widths, cols, col_at = ctx.meta
        L = len(widths)
        at = None if col_at is None else (C.c_uint8 * d)(*col_at)
        saved = ctx.saved_tensors
        W, ga, be = saved[:L], saved[L:2 * L], saved[2 * L:]
        dev = W[0].device
        dG = torch.zeros(32, d, device=dev) if dG is None else dG.contiguous().float()
        dc = torch.zeros(32, device=dev) if dc is None else dc.contiguous().float()
        dW = [torch.empty_like(w) for w in W]
        db = [torch.empty(32, device=dev) for _ in range(L)]
        dga, dbe = [torch.empty_like(g) for g in ga], [torch.empty_like(g) for g in be]
        wi, ci = (C.c_int32 * L)(*widths), (C.c_int32 * L)(*cols)
        with torch.cuda.device(dev):
            _C.check(_C.lib.scr_norm_fold_backward(L, d, wi, ci, at, _ptr_table(W), _ptr_table(ga), _ptr_table(be), dG.data_ptr(),
                                                   dc.data_ptr(), _ptr_table(dW), _ptr_table(db), _ptr_table(dga),
                                                   _ptr_table(dbe), _stream(dev)))
        out = []
        for i in range(L):
            out += [dW[i], db[i], dga[i], dbe[i]]
        return (None, None, None, None, *out)


def _fold_on_device(x, bns, linears):
    return (x.is_cuda and x.dtype == torch.float32 and 1 <= len(bns) <= 4 and x.shape[1] <= 80
            and all(l.out_features == 32 and l.bias is not None and l.weight.dtype == torch.float32 for l in linears)
            and all(bn.affine for bn in bns))


def _norm_linear(x, bns, linears, col_at=None):
    """sum_i Linear_i(BatchNorm_i(x_i)) where x = cat_i(x_i) column-wise (bns[i] normalises its own column
    block; pass the same block twice -- offsets repeat -- is not supported) OR every BatchNorm_i sees the
    whole x (len(bns) == len(linears), all of width x.shape[1]).  Updates the running statistics as
    nn.BatchNorm1d.forward does in training mode.
    col_at (sequence of d ints, a permutation): x keeps the reference's column j at column col_at[j] (FeaturePlanes
    samples two same-size grids as one and gets their features interleaved); G, the batch statistics and the running
    statistics follow."""
    d = x.shape[1]
    shared = all(bn.num_features == d for bn in bns)
    if not shared:
        assert sum(bn.num_features for bn in bns) == d
    widths = [bn.num_features for bn in bns]
    cols = [0] * len(bns) if shared else [sum(widths[:i]) for i in range(len(bns))]
    on_device = _fold_on_device(x, bns, linears)
    if on_device:
        # one launch builds G and c from the 4 L parameter tensors (and one takes their gradients back)
        G, c = _NormFold.apply(d, tuple(widths), tuple(cols), None if col_at is None else tuple(int(v) for v in col_at),
                               *[t for bn, lin in zip(bns, linears) for t in (lin.weight, lin.bias, bn.weight, bn.bias)])
    elif shared:    # same input through every pair: the folded weights simply add up
        G = sum(lin.weight * bn.weight for bn, lin in zip(bns, linears))
        c = sum(lin.weight @ bn.bias + lin.bias for bn, lin in zip(bns, linears))
    else:
        G = torch.cat([lin.weight * bn.weight for bn, lin in zip(bns, linears)], dim=1)
        c = sum(lin.weight @ bn.bias + lin.bias for bn, lin in zip(bns, linears))
    at_idx = None
    if col_at is not None and not on_device:      # G[:, col_at[j]] = G_ref[:, j]
        at_idx = torch.as_tensor(list(col_at), dtype=torch.long, device=x.device)
        G = torch.zeros_like(G).index_copy(1, at_idx, G)
    assert all(bn.eps == bns[0].eps for bn in bns)
    # (statistics the producer of x left on it -- anchor_gather -- are only valid for x as it is: same rows, no column map)
    stats = getattr(x, "_scr_col_stats", None) if col_at is None else None
    if stats is not None and getattr(x, "_scr_col_stats_version", None) != x._version:
        stats = None        # x was edited in place since its producer summed its columns: the built-in statistics pass runs
    defer = getattr(x, "_scr_deferred_dx", None)      # the producer of x (anchor gather / tri-plane sampling) forms dx in ITS backward
    y, mean, var = _NormLinearFn.apply(x, G, c, bns[0].eps, stats, defer)
    with torch.no_grad():
        n = x.shape[0]
        track = [bn for bn in bns if bn.track_running_stats and bn.training]
        if on_device and len(track) == len(bns) and all(bn.momentum is not None for bn in bns) and mean.is_cuda:
            # nn.BatchNorm1d's running-statistics update for all pairs in one launch (six tiny kernels per BatchNorm otherwise)
            import ctypes as C
            from . import _C
            from .rasterizer import _stream
            L = len(bns)
            with torch.cuda.device(x.device):
                _C.check(_C.lib.scr_norm_running_stats(
                    L, d, (C.c_int32 * L)(*widths), (C.c_int32 * L)(*cols),
                    None if col_at is None else (C.c_uint8 * d)(*[int(v) for v in col_at]),
                    (C.c_float * L)(*[float(bn.momentum) for bn in bns]),
                    _ptr_table([bn.running_mean for bn in bns]), _ptr_table([bn.running_var for bn in bns]),
                    _ptr_table([bn.num_batches_tracked for bn in bns]), mean.data_ptr(), var.data_ptr(), n, _stream(x.device)))
            return y
        if col_at is not None:                     # statistics back in the reference's column order
            if at_idx is None:
                at_idx = torch.as_tensor(list(col_at), dtype=torch.long, device=x.device)
            mean, var = mean.index_select(0, at_idx), var.index_select(0, at_idx)
        off = 0
        for bn in bns:
            m, v = (mean, var) if shared else (mean[off:off + bn.num_features], var[off:off + bn.num_features])
            off += 0 if shared else bn.num_features
            if bn.track_running_stats and bn.training:
                bn.num_batches_tracked += 1
                mom = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
                bn.running_mean.mul_(1 - mom).add_(m, alpha=mom)
                bn.running_var.mul_(1 - mom).add_(v * (n / max(n - 1, 1)), alpha=mom)
    return y


class ChannelAttention(nn.Module):            # scene/grids.py:22-36
    def __init__(self, in_planes, ratio=5):
        super().__init__()
        self.sharedMLP = nn.Sequential(nn.Conv2d(in_planes, in_planes // ratio, 1, bias=False), nn.ReLU(),
                                       nn.Conv2d(in_planes // ratio, in_planes, 1, bias=False))

    def forward(self, x):
        # global average / max pool (AdaptiveAvgPool2d(1) / AdaptiveMaxPool2d(1)); amax instead of
        # adaptive_max_pool2d: same value, 2.5 ms -> 0.05 ms on a [1,15,700,700] stack on MI355X
        avg = self.sharedMLP(x.mean(dim=(2, 3), keepdim=True))
        mx = self.sharedMLP(x.amax(dim=(2, 3), keepdim=True))
        return torch.sigmoid(avg + mx)


class SpatialAttention(nn.Module):            # scene/grids.py:38-52
    def __init__(self, kernel_size=7):
        super().__init__()
        self.conv = nn.Conv2d(2, 1, kernel_size, padding=3 if kernel_size == 7 else 1, bias=False)

    def forward(self, x):
        avg = torch.mean(x, dim=1, keepdim=True)
        mx, _ = torch.max(x, dim=1, keepdim=True)
        return torch.sigmoid(self.conv(torch.cat([avg, mx], dim=1)))


class TriPlaneAttention(nn.Module):           # scene/grids.py:55-64
    def __init__(self, planes):
        super().__init__()
        self.ca = ChannelAttention(planes)
        self.sa = SpatialAttention()

    def forward(self, x):
        x = self.ca(x) * x
        return self.sa(x) * x


def _sample(plane, ind_norm, cols):
    # F.grid_sample bilinear, align_corners=True, zeros padding (scene/grids.py:148-150)
    return F.grid_sample(plane, ind_norm[:, :, :, cols], mode="bilinear", align_corners=True).flatten(0, 2).T


class PlaneGrid(nn.Module):                   # scene/grids.py:102-201
    def __init__(self, channels, world_size, xyz_min, xyz_max, TAflag=False, factor=1):
        super().__init__()
        self.channels, self.TAflag = channels, TAflag
        self.register_buffer("xyz_min", torch.as_tensor(xyz_min))
        self.register_buffer("xyz_max", torch.as_tensor(xyz_max))
        X, Y, Z = (int(w) * factor for w in world_size)
        R = channels // 3
        self.xy_plane = nn.Parameter(torch.randn(1, R, X, Y) * 0.1)
        self.xz_plane = nn.Parameter(torch.randn(1, R, X, Z) * 0.1)
        self.yz_plane = nn.Parameter(torch.randn(1, R, Y, Z) * 0.1)
        if TAflag:
            self.TA = TriPlaneAttention(channels)

    def get_dim(self):
        return self.channels * 2 if self.TAflag else self.channels

    def total_variation_add_grad(self, w):
        """scene/grids.py:240-250: adds the gradient of the smooth-L1 total variation of the three planes (weight w, / 6)
        into their .grad, in place (csrc/tv.hip; the attention module takes no part, as there)."""
        from .tv import grid_planes, tv_add_grad
        tv_add_grad([(p, w) for p in grid_planes(self)])

    def fused_ok(self, xyz):
        return self.xy_plane.is_cuda and self.channels // 3 <= 8 and not xyz.requires_grad    # csrc/triplane.hip (2 R <= 16)

    def bounds_key(self):
        """The grid's box as host floats (one device read per change of the buffers): grids with equal boxes sample at
        the same normalised coordinates, which FeaturePlanes then computes once."""
        ver = (self.xyz_min._version, self.xyz_max._version, self.xyz_min.data_ptr(), self.xyz_max.data_ptr())
        if getattr(self, "_bounds_ver", None) != ver:
            self._bounds_ver = ver
            self._bounds = tuple(self.xyz_min.detach().cpu().tolist()) + tuple(self.xyz_max.detach().cpu().tolist())
        return self._bounds

    def box_coords(self, xyz):
        """(xyz - xyz_min) / (xyz_max - xyz_min) * 2 - 1 (scene/grids.py:146) for [..., 3] points; on the device one pass
        (scr_box_coords: the same IEEE operations in the same order) instead of four."""
        p = xyz.reshape(-1, 3)
        if p.is_cuda and p.dtype == torch.float32 and not p.requires_grad and p.shape[0] > 0:
            import ctypes as C
            from . import _C
            from .rasterizer import _stream
            b = self.bounds_key()
            p = p.contiguous()
            out = torch.empty_like(p)
            with torch.cuda.device(p.device):
                _C.check(_C.lib.scr_box_coords(p.shape[0], p.data_ptr(), (C.c_float * 3)(*b[:3]), (C.c_float * 3)(*b[3:]),
                                               out.data_ptr(), _stream(p.device)))
            return out
        return (p - self.xyz_min) / (self.xyz_max - self.xyz_min) * 2 - 1

    def sample_spec(self, xyz, col0=0, ind3=None):
        """(ind [V,3], planes, first output column of every plane) for triplane.multi_triplane_sample: what
        compute_planes_feat samples and where the reference's torch.cat puts it (scene/grids.py:165,181)."""
        R = self.channels // 3
        if ind3 is None:
            ind3 = self.box_coords(xyz)
        if not self.TAflag:
            return ind3, (self.xy_plane, self.xz_plane, self.yz_plane), tuple(col0 + R * j for j in range(3))
        # column order of :181: xy, xyA, xz, xzA, yz, yzA -- a plane and its attended twin are sampled at the same
        # positions and land side by side, so each pair is stacked into ONE plane of 2 R channels: half the random
        # cache lines per sample (the channel-last rows of the pair are adjacent), one launch instead of two
        from . import plane_attention
        if plane_attention.fused_ok(self.xy_plane, self.xz_plane, self.yz_plane, self.TA):
            # the attention module and the stacking as a few streaming passes (csrc/attention.hip), every call (:166-168)
            pairs = plane_attention.attended_pair_planes(self.xy_plane, self.xz_plane, self.yz_plane, self.TA)
            return ind3, pairs, tuple(col0 + 2 * R * j for j in range(3))
        tri = self.TA(torch.cat((self.xy_plane, self.xz_plane, self.yz_plane), dim=1))   # every call (:166-168)
        xyA, xzA, yzA = torch.chunk(tri, 3, dim=1)
        return ind3, (torch.cat((self.xy_plane, xyA), dim=1), torch.cat((self.xz_plane, xzA), dim=1),
                      torch.cat((self.yz_plane, yzA), dim=1)), tuple(col0 + 2 * R * j for j in range(3))

    def forward(self, xyz, Q=0):
        shape = xyz.shape[:-1]
        if self.fused_ok(xyz):
            from .triplane import multi_triplane_sample
            feat = multi_triplane_sample([self.sample_spec(xyz)])
            # training-time uniform noise (:159-164) reaches plain grids only: for the attention grid the reference
            # builds the noised concatenation and then overwrites it at :174-181 with the un-noised samples
            if Q != 0 and not self.TAflag:
                feat = feat + torch.empty_like(feat).uniform_(-0.5, 0.5) * Q
            return feat.reshape(*shape, self.get_dim())
        xyz = xyz.reshape(1, 1, -1, 3)
        ind = (xyz - self.xyz_min) / (self.xyz_max - self.xyz_min) * 2 - 1
        ind = torch.cat([ind, torch.zeros_like(ind[..., [0]])], dim=-1)
        xy, xz, yz = _sample(self.xy_plane, ind, [1, 0]), _sample(self.xz_plane, ind, [2, 0]), _sample(self.yz_plane, ind, [2, 1])
        if not self.TAflag:
            if Q != 0:                                          # training-time uniform noise (:159-164), plain grids only
                xy = xy + torch.empty_like(xy).uniform_(-0.5, 0.5) * Q
                xz = xz + torch.empty_like(xz).uniform_(-0.5, 0.5) * Q
                yz = yz + torch.empty_like(yz).uniform_(-0.5, 0.5) * Q
            return torch.cat([xy, xz, yz], dim=-1).reshape(*shape, self.channels)
        tri = self.TA(torch.cat((self.xy_plane, self.xz_plane, self.yz_plane), dim=1))   # every call (:166-168)
        xyA, xzA, yzA = torch.chunk(tri, 3, dim=1)
        feat = torch.cat([xy, _sample(xyA, ind, [1, 0]), xz, _sample(xzA, ind, [2, 0]), yz, _sample(yzA, ind, [2, 1])], dim=-1)
        return feat.reshape(*shape, self.channels * 2)


class FeaturePlanes(nn.Module):               # scene/gaussian_model.py:97-169
    def __init__(self, world_size, xyz_min, xyz_max, feat_dim=24, out_dim=32):
        super().__init__()
        self.activate_level, self.num_levels, self.level_factor = 0, 3, 0.5
        t_ws = torch.tensor(world_size)
        self.k0s = nn.ModuleList()
        for i in range(self.num_levels):
            cur = (t_ws * self.level_factor ** (self.num_levels - i - 1)).int().tolist()
            if i == 0:
                self.k0s.append(PlaneGrid(feat_dim, cur, xyz_min, xyz_max, TAflag=True))
            self.k0s.append(PlaneGrid(feat_dim, cur, xyz_min, xyz_max))
        self.models, self.CTX_models = nn.ModuleList(), nn.ModuleList()
        for i in range(self.num_levels):
            d = self.k0s[i].get_dim()
            self.models.append(nn.Sequential(nn.BatchNorm1d(d), TallLinear(d, out_dim)))
            self.CTX_models.append(nn.Sequential(nn.BatchNorm1d(71), TallLinear(71, out_dim)))

    def _stackable(self):
        """Level 0's two grids (attention + plain) can be sampled as one: same plane sizes, same box, fused attention."""
        from . import plane_attention
        g0, g1 = self.k0s[0], self.k0s[1]
        return (g0.TAflag and not g1.TAflag and g0.channels == g1.channels and g0.bounds_key() == g1.bounds_key()
                and all(a.shape == b.shape for a, b in ((g0.xy_plane, g1.xy_plane), (g0.xz_plane, g1.xz_plane), (g0.yz_plane, g1.yz_plane)))
                and 3 * (g0.channels // 3) <= 16 and plane_attention.fused_ok(g0.xy_plane, g0.xz_plane, g0.yz_plane, g0.TA))

    def inactive_parameters(self):
        """Parameters of the levels above activate_level: forward() does not touch them, so they never receive a gradient.
        The reference leaves their .grad None and torch.optim skips them; a training loop that pre-allocates gradients
        (multiview.GradArena) should leave them out of its parameter list -- at the default sizes the 2800^2 level is 470 MB
        of zeros per step to clear, to all-reduce and to run Adam over."""
        L = self.activate_level + 1
        return [p for mods in (self.k0s[L:], self.models[L:], self.CTX_models[L:]) for m in mods for p in m.parameters()]

    def _fused_path(self):
        L = self.activate_level + 1
        return FUSE_NORM_LINEAR and all(m[0].training for m in list(self.models[:L]) + list(self.CTX_models[:L]))

    def presample(self, x, Q=0):
        """The sampling half of forward() on its own: (feats, col_at) for forward(..., presampled=...), or None when this
        configuration does not take the fused path.  WHY it exists: autograd runs the backward of what was created LATER
        first.  Sampled before the anchor gather is applied (the coordinates are detached, scene/gaussian_model.py:210, so
        they can be gathered separately), the tri-plane backward -- a quarter of the anchor path's backward -- runs AFTER
        the gather's, i.e. after the per-anchor gradients are final: in a multi-rank step their exchange (99 % of the
        bytes) is then on the wire while the tri-plane and attention backward passes still compute."""
        if not self._fused_path() or x.dim() != 2 or not all(self.k0s[i].fused_ok(x) for i in range(self.activate_level + 1)):
            return None
        return self._sample_all(x, Q)

    def _sample_all(self, x, Q):
        L = self.activate_level + 1
        col_at = None
        # every active grid samples straight into its columns of one matrix (no torch.cat of the grids' outputs)
        from .triplane import multi_triplane_sample
        specs, col, shared, noise_cols = [], 0, {}, []
        first = 0
        if L > 1 and STACK_LEVEL0 and self._stackable():
            # The attention grid and the level-0 plain grid have the same size and the same box: every point
            # samples both at the same texels.  The plain grid's plane is stacked on the attention grid's pair planes
            # (plane | attended twin | level-0 plane: 3 r channels) and the two grids are sampled -- and their
            # gradients scattered -- as ONE: one gather / one record of 15 channels per projection instead of 10 + 5.
            # The samples of projection q land at columns 15 q .. 15 q + 15 where the reference's concatenation
            # (scene/gaussian_model.py:160-166) has grid 0's at 10 q .. and grid 1's at 30 + 5 q ..: col_at tells
            # the BatchNorm-Linear fold, which is free to keep its input columns in any order.
            g0, g1 = self.k0s[0], self.k0s[1]
            r = g1.channels // 3
            ind3, pairs, _ = g0.sample_spec(x, 0)
            shared.setdefault(g0.bounds_key(), ind3)
            stacked = tuple(torch.cat((pr, pl), dim=1) for pr, pl in zip(pairs, (g1.xy_plane, g1.xz_plane, g1.yz_plane)))
            specs.append((ind3, stacked, (0, 3 * r, 6 * r)))
            col_at = [(j // (2 * r)) * 3 * r + j % (2 * r) for j in range(6 * r)]
            col_at += [q * 3 * r + 2 * r + k for q in range(3) for k in range(r)]
            noise_cols = [(q * 3 * r + 2 * r, r) for q in range(3)]
            col, first = 9 * r, 2
        for i in range(first, L):
            # grids with the same box sample at the same normalised coordinates: ONE tensor, which also lets the
            # backward of all grids run as one pass over the points
            key = self.k0s[i].bounds_key()
            spec = self.k0s[i].sample_spec(x, col, shared.get(key))
            shared.setdefault(key, spec[0])
            specs.append(spec)
            if not self.k0s[i].TAflag:
                noise_cols.append((col, self.k0s[i].get_dim()))
            col += self.k0s[i].get_dim()
        if col_at is not None:
            col_at += list(range(len(col_at), col))
        feats = multi_triplane_sample(specs)
        if Q != 0:                       # uniform noise on the plain grids' blocks only (scene/grids.py:159-181)
            for c0, w in noise_cols:
                feats[:, c0:c0 + w] += torch.empty(feats.shape[0], w, device=feats.device).uniform_(-0.5, 0.5) * Q
        return feats, col_at

    def forward(self, x, g_fea, Q=0, parts=False, presampled=None):
        """parts=True: return the two 32-column halves (plane branch, attribute branch) instead of their
        concatenation, when they exist as separate matrices (the fused MLP heads read them as they are).
        presampled: what presample(x, Q) returned for the same x (see there)."""
        L = self.activate_level + 1
        col_at = None
        if self._fused_path():
            # sum_i cat(Linear(BN(feat_i)), Linear(BN(g_fea))) is linear in the normalised inputs: two GEMMs
            if presampled is not None:
                feats, col_at = presampled
            elif x.dim() == 2 and all(self.k0s[i].fused_ok(x) for i in range(L)):
                # every active grid samples straight into its columns of one matrix (no torch.cat of the grids' outputs)
                feats, col_at = self._sample_all(x, Q)
            else:
                feats = torch.cat([self.k0s[i](x, Q) for i in range(L)], dim=1) if L > 1 else self.k0s[0](x, Q)
            a = _norm_linear(feats, [self.models[i][0] for i in range(L)], [self.models[i][1] for i in range(L)], col_at=col_at)
            b = _norm_linear(g_fea, [self.CTX_models[i][0] for i in range(L)], [self.CTX_models[i][1] for i in range(L)])
            return (a, b) if parts else torch.cat((a, b), dim=1)
        res = []
        for i in range(self.activate_level + 1):
            feat = self.k0s[i](x, Q)
            res.append(torch.cat((self.models[i](feat), self.CTX_models[i](g_fea)), dim=1))
        return sum(res)


class GaussianLearner(nn.Module):             # scene/gaussian_model.py:184-220
    def __init__(self, plane_size, num_channels, xyz_min=(-2, -2, -2), xyz_max=(2, 2, 2)):
        super().__init__()
        self.Q0 = 0.03
        self._feat = FeaturePlanes([plane_size] * 3, torch.tensor(xyz_min), torch.tensor(xyz_max), feat_dim=num_channels)
        self.register_buffer("opacity_scale", torch.tensor(10))

    def activate_plane_level(self):
        self._feat.activate_level += 1

    def tv_loss(self, w):
        """scene/gaussian_model.py:217-220: grid `level` of the active levels gets w * 0.5^(2 - level); one launch."""
        from .tv import feature_planes_tv
        feature_planes_tv(self._feat, w)

    def presample(self, xyz):
        """FeaturePlanes.presample at the noise level inference() uses: hand the result to inference(presampled=...)."""
        return self._feat.presample(xyz.detach(), self.Q0)

    def inference(self, xyz, g_fea, Q0, parts=False, presampled=None):
        # the reference ignores the Q0 argument and uses self.Q0 (:209-215); xyz is detached
        out = self._feat(xyz.detach(), g_fea, self.Q0, parts=parts, presampled=presampled)
        if parts and not isinstance(out, tuple):
            out = (out[:, :32], out[:, 32:])
        return out


@torch.no_grad()
def morton_order(anchor, bits=10):
    """Permutation (int64 [N]) that sorts points [N,3] by the 3-D Morton code of their position in their bounding
    box (`bits` per axis, at most 21)."""
    a = anchor.detach()
    if a.shape[0] == 0:
        return torch.empty(0, dtype=torch.long, device=a.device)
    lo, hi = a.amin(0), a.amax(0)
    q = ((a - lo) / (hi - lo).clamp_min(1e-12) * (1 << bits)).long().clamp_(0, (1 << bits) - 1)
    code = torch.zeros(a.shape[0], dtype=torch.long, device=a.device)
    for b in range(bits):
        for ax in range(3):
            code |= ((q[:, ax] >> b) & 1) << (3 * b + ax)
    return torch.argsort(code, stable=True)


class AppearanceEmbedding(nn.Module):
    """Per-camera appearance code (scene/embedding.py:52-80): a lookup table whose state_dict key is
    `embedding.weight`, as in the reference's checkpoints ('appearance', scene/gaussian_model.py:1052-1059)."""

    def __init__(self, num_cameras, dim):
        super().__init__()
        self.in_dim, self.out_dim = num_cameras, dim
        self.embedding = nn.Embedding(num_cameras, dim)

    def forward(self, idx):
        return self.embedding(idx)


class AnchorGaussianModel(nn.Module):
    """The attributes of the reference's GaussianModel that render() / prefilter_voxel() touch
    (scene/gaussian_model.py:253-337,396-432), with the same names."""

    def __init__(self, feat_dim=32, n_offsets=10, appearance_dim=0, plane_size=2800, num_channels=15,
                 use_feat_bank=False, add_opacity_dist=False, add_cov_dist=False, add_color_dist=False):
        super().__init__()
        if use_feat_bank:
            raise NotImplementedError("use_feat_bank: dead in the reference (its 4-input feature-bank MLP, scene/gaussian_model.py:"
                                      "308-309, is fed 68 columns at gaussian_renderer/__init__.py:41-43 and stops with a shape error)")
        self.feat_dim, self.n_offsets, self.appearance_dim, self.use_feat_bank = feat_dim, n_offsets, appearance_dim, use_feat_bank
        self.add_opacity_dist, self.add_cov_dist, self.add_color_dist = add_opacity_dist, add_cov_dist, add_color_dist
        od, cd, kd = int(add_opacity_dist), int(add_cov_dist), int(add_color_dist)
        self.mlp_opacity = nn.Sequential(TallLinear(feat_dim + 3 + od + 64, feat_dim), nn.ReLU(True),
                                         TallLinear(feat_dim, n_offsets), nn.Tanh())
        self.mlp_cov = nn.Sequential(TallLinear(feat_dim + 3 + cd + 64, feat_dim), nn.ReLU(True),
                                     TallLinear(feat_dim, 7 * n_offsets))
        self.mlp_color = nn.Sequential(TallLinear(feat_dim + 3 + kd + appearance_dim + 64, feat_dim), nn.ReLU(True),
                                       TallLinear(feat_dim, 3 * n_offsets), nn.Sigmoid())
        # off on the benchmarked path (README.md:93 --appearance_dim 0, arguments/__init__.py:57); the reference's code
        # defaults (appearance_dim = 32, arguments/__init__.py:76) run through the unfused heads of renderer.py
        self.embedding_appearance = None                                         # set_appearance(num_cameras), :390-392
        self.feat_planes = GaussianLearner(plane_size, num_channels)
        self._anchor = nn.Parameter(torch.empty(0, 3))
        self._offset = nn.Parameter(torch.empty(0, n_offsets, 3))
        self._anchor_feat = nn.Parameter(torch.empty(0, feat_dim))
        self._scaling = nn.Parameter(torch.empty(0, 6))
        self._rotation = nn.Parameter(torch.empty(0, 4), requires_grad=False)   # requires_grad False in the reference
        self._opacity = nn.Parameter(torch.empty(0, 1), requires_grad=False)    # carried, never read by render()
        self.rotation_activation = F.normalize
        self.contractor_state = {}        # xyz_min / xyz_max of the reference's Conctractor: carried through checkpoints only

    def set_anchors(self, anchor, offset, anchor_feat, scaling, rotation=None, opacity=None):
        N = anchor.shape[0]
        if opacity is None:                                  # inverse_sigmoid(0.1), scene/gaussian_model.py:493
            opacity = torch.full((N, 1), -2.1972246, device=anchor.device)
        self._opacity = nn.Parameter(opacity.float(), requires_grad=False)
        self._anchor = nn.Parameter(anchor.float())
        self._offset = nn.Parameter(offset.float())
        self._anchor_feat = nn.Parameter(anchor_feat.float())
        self._scaling = nn.Parameter(scaling.float())
        if rotation is None:
            rotation = torch.zeros(N, 4, device=anchor.device)
            rotation[:, 0] = 1
        self._rotation = nn.Parameter(rotation.float(), requires_grad=False)

    # the reference's eval()/train() touch only the three MLP heads; BatchNorm in FeaturePlanes
    # stays in train mode forever (scene/gaussian_model.py:350-366)
    def eval(self):
        return self.train(False)

    def train(self, mode=True):
        self.mlp_opacity.train(mode); self.mlp_cov.train(mode); self.mlp_color.train(mode)
        if self.embedding_appearance is not None:
            self.embedding_appearance.train(mode)
        return self

    def set_appearance(self, num_cameras):                   # scene/gaussian_model.py:390-392
        if self.appearance_dim > 0:
            dev = self.mlp_color[0].weight.device
            self.embedding_appearance = AppearanceEmbedding(num_cameras, self.appearance_dim).to(dev)

    # ---- memory layout: anchors in spatial (Morton) order.  Nothing in the reference depends on the order of the
    # anchors (it only breaks exact depth ties); on MI355X it decides whether the tri-plane samples, the gradient
    # scatter and the rasterizer's tile scatter of 64 consecutive anchors touch neighbouring cache lines or 64 random
    # ones.  create_from_pcd leaves them in np.unique (lexicographic) order, anchor_growing appends (scene/gaussian_model.py:
    # 449,913-925); a training loop can restore the locality after every adjust_anchor with AnchorDensifier.sort_anchors.
    def morton_order(self, bits=10):
        return morton_order(self._anchor, bits)

    @torch.no_grad()
    def sort_anchors(self, perm=None):
        """Reorders the per-anchor tensors in place (no optimizer attached: use AnchorDensifier.sort_anchors when
        there is one).  Returns the permutation applied (new row i = old row perm[i])."""
        perm = self.morton_order() if perm is None else perm
        for name in ("_anchor", "_offset", "_anchor_feat", "_scaling", "_rotation", "_opacity"):
            p = getattr(self, name)
            if p.shape[0] == perm.shape[0]:
                p.data = p.data[perm].contiguous()
        return perm

    @property
    def get_anchor(self):
        return self._anchor

    @property
    def get_scaling(self):
        return torch.exp(self._scaling)          # the reference's `1.0 * exp(.)` (scene/gaussian_model.py:397-399): the same bits

    @property
    def get_rotation(self):
        # normalize(_rotation), scene/gaussian_model.py:405-407.  The anchors' rotation never receives a gradient
        # (requires_grad False in the reference): the normalised copy is kept until the parameter is written or replaced
        # instead of two passes over [N, 4] in every prefilter_voxel call.  "Written" = an in-place torch operation (the
        # version counter moves) or a new tensor / storage; a write through `_rotation.data[...]` or a raw kernel is NOT
        # seen -- assign `pc._rot_key = None` after one
        r = self._rotation
        key = (id(r), r._version, r.data_ptr(), tuple(r.shape))
        if getattr(self, "_rot_key", None) != key or r.requires_grad:
            out = self.rotation_activation(r)
            if r.requires_grad:
                return out
            self._rot_key, self._rot_norm = key, out.detach()
        return self._rot_norm

    @property
    def get_appearance(self):
        if self.embedding_appearance is None:
            raise RuntimeError("appearance_dim > 0: call set_appearance(num_cameras) first (the reference does so in "
                               "Scene.__init__, scene/__init__.py)")
        return self.embedding_appearance

    @property
    def get_opacity_mlp(self):
        return self.mlp_opacity

    @property
    def get_cov_mlp(self):
        return self.mlp_cov

    @property
    def get_color_mlp(self):
        return self.mlp_color
