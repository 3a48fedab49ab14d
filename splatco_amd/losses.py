"""Loss / metric functions of the training step around the rasterizer (plain PyTorch), restating
utils/loss_utils.py:17-63 and utils/image_utils.py:14-19; pinned by tests/golden/losses.npz.
psnr() is the parity metric of BASELINE.json ("PSNR-match")."""
from math import exp

import os

import torch
import torch.nn.functional as F


def l1_loss(network_output, gt):
    return torch.abs(network_output - gt).mean()


def l2_loss(network_output, gt):
    return ((network_output - gt) ** 2).mean()


def psnr(img1, img2):
    mse = ((img1 - img2) ** 2).view(img1.shape[0], -1).mean(1, keepdim=True)
    return 20 * torch.log10(1.0 / torch.sqrt(mse))


def _window(window_size, channel, like):
    g = torch.tensor([exp(-(x - window_size // 2) ** 2 / float(2 * 1.5 ** 2)) for x in range(window_size)])
    g = (g / g.sum()).unsqueeze(1)
    w2d = g.mm(g.t()).float().unsqueeze(0).unsqueeze(0)
    return w2d.expand(channel, 1, window_size, window_size).contiguous().to(like)


def ssim(img1, img2, window_size=11, size_average=True):
    channel = img1.size(-3)
    window = _window(window_size, channel, img1)
    pad = window_size // 2
    mu1 = F.conv2d(img1, window, padding=pad, groups=channel)
    mu2 = F.conv2d(img2, window, padding=pad, groups=channel)
    mu1_sq, mu2_sq, mu1_mu2 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    sigma1_sq = F.conv2d(img1 * img1, window, padding=pad, groups=channel) - mu1_sq
    sigma2_sq = F.conv2d(img2 * img2, window, padding=pad, groups=channel) - mu2_sq
    sigma12 = F.conv2d(img1 * img2, window, padding=pad, groups=channel) - mu1_mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    ssim_map = ((2 * mu1_mu2 + C1) * (2 * sigma12 + C2)) / ((mu1_sq + mu2_sq + C1) * (sigma1_sq + sigma2_sq + C2))
    return ssim_map.mean() if size_average else ssim_map.mean(1).mean(1).mean(1)


class _L1Ssim(torch.autograd.Function):
    """(mean |x - y|, mean SSIM(x, y)) of two [C,H,W] device images in one fused HIP pass
    (csrc/ssim.hip); the gradient flows to x only.  11 ms -> ~0.3 ms forward+backward at 1080p."""

    @staticmethod
    def forward(ctx, x, y):
        from . import _C
        from .rasterizer import _stream
        # whether the derivative maps are needed is a property of the INPUT: grad mode is off inside forward, so a
        # converted copy (non-contiguous crop, other dtype) would report requires_grad = False
        need = bool(ctx.needs_input_grad[0])
        x, y = x.contiguous().float(), y.contiguous().float()
        C, H, W = x.shape
        scratch = torch.empty(_C.lib.scr_l1_ssim_scratch_bytes(C, H, W, int(need)), dtype=torch.uint8, device=x.device)
        out = torch.empty(2, dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):      # kernels launch on the CURRENT device: make it the tensors' device
            _C.check(_C.lib.scr_l1_ssim_forward(C, H, W, x.data_ptr(), y.data_ptr(), scratch.data_ptr(), int(need),
                                                out.data_ptr(), _stream(x.device)))
        ctx.save_for_backward(x, y, scratch)
        ctx.have_maps = need
        return out[0], out[1]

    @staticmethod
    def backward(ctx, g_l1, g_ssim):
        from . import _C
        from .rasterizer import _stream
        x, y, scratch = ctx.saved_tensors
        if not ctx.have_maps:
            raise RuntimeError("l1_ssim backward without the derivative maps of the forward pass")
        C, H, W = x.shape
        g_l1 = g_l1.contiguous().float().reshape(1)
        g_ssim = g_ssim.contiguous().float().reshape(1)
        dx = torch.empty_like(x)
        with torch.cuda.device(x.device):
            _C.check(_C.lib.scr_l1_ssim_backward(C, H, W, x.data_ptr(), y.data_ptr(), scratch.data_ptr(),
                                                 g_l1.data_ptr(), g_ssim.data_ptr(), dx.data_ptr(), _stream(x.device)))
        return dx, None


class _PairL1(torch.autograd.Function):
    """mean |(real1 - real2) - (gen1 - gen2)| of four equally shaped device images (the L1 of the cross-view consistency
    term, train.py:208-217) in one pass per direction (csrc/ssim.hip); the gradient flows to gen1 / gen2."""

    @staticmethod
    def forward(ctx, gen1, gen2, real1, real2):
        from . import _C
        from .rasterizer import _stream
        c = lambda t: t.detach().contiguous().float()
        gen1, gen2, real1, real2 = c(gen1), c(gen2), c(real1), c(real2)
        n, dev = gen1.numel(), gen1.device
        scratch = torch.empty(_C.lib.scr_pair_l1_scratch_bytes(n), dtype=torch.uint8, device=dev)
        out = torch.empty(1, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _C.check(_C.lib.scr_pair_l1_forward(n, gen1.data_ptr(), gen2.data_ptr(), real1.data_ptr(), real2.data_ptr(),
                                                scratch.data_ptr(), out.data_ptr(), _stream(dev)))
        ctx.save_for_backward(gen1, gen2, real1, real2)
        return out.reshape(())

    @staticmethod
    def backward(ctx, g):
        from . import _C
        from .rasterizer import _stream
        gen1, gen2, real1, real2 = ctx.saved_tensors
        n, dev = gen1.numel(), gen1.device
        g = g.contiguous().float().reshape(1)
        d1 = torch.empty_like(gen1) if ctx.needs_input_grad[0] else None
        d2 = torch.empty_like(gen2) if ctx.needs_input_grad[1] else None
        with torch.cuda.device(dev):
            _C.check(_C.lib.scr_pair_l1_backward(n, gen1.data_ptr(), gen2.data_ptr(), real1.data_ptr(), real2.data_ptr(), g.data_ptr(),
                                                 None if d1 is None else d1.data_ptr(), None if d2 is None else d2.data_ptr(),
                                                 _stream(dev)))
        return d1, d2, None, None


def pair_l1(gen1, gen2, real1, real2):
    """l1_loss(real1 - real2, gen1 - gen2) (train.py:213): the fused op on the GPU, the reference's ops elsewhere."""
    if gen1.is_cuda and gen1.shape == gen2.shape == real1.shape == real2.shape and gen1.numel() > 0 and not (real1.requires_grad or real2.requires_grad):
        return _PairL1.apply(gen1, gen2, real1, real2)
    return l1_loss(real1 - real2, gen1 - gen2)


def l1_ssim(image, gt_image):
    """Fused (l1_loss, ssim) for [3,H,W] device images; falls back to the torch ops on CPU."""
    if image.is_cuda and image.dim() == 3:
        return _L1Ssim.apply(image, gt_image)
    return l1_loss(image, gt_image), ssim(image, gt_image)


class _ScalingReg(torch.autograd.Function):
    """mean(prod(scaling, dim=1)) for scaling [P,3] on the GPU (csrc/ssim.hip).  torch's prod backward first counts the
    zeros of its input (a compare, an int64 reduction and a host read) -- 2.5 ms per step at 85 M Gaussians."""

    @staticmethod
    def forward(ctx, scaling):
        from . import _C
        from .rasterizer import _stream
        s = scaling.detach().contiguous().float()
        P = s.shape[0]
        scratch = torch.empty(_C.lib.scr_scaling_reg_scratch_bytes(P), dtype=torch.uint8, device=s.device)
        out = torch.empty(1, dtype=torch.float32, device=s.device)
        with torch.cuda.device(s.device):
            _C.check(_C.lib.scr_scaling_reg_forward(P, s.data_ptr(), scratch.data_ptr(), out.data_ptr(), _stream(s.device)))
        ctx.save_for_backward(s)
        return out.reshape(())

    @staticmethod
    def backward(ctx, g):
        from . import _C
        from .rasterizer import _stream
        (s,) = ctx.saved_tensors
        g = g.contiguous().float().reshape(1)
        d = torch.empty_like(s)
        with torch.cuda.device(s.device):
            _C.check(_C.lib.scr_scaling_reg_backward(s.shape[0], s.data_ptr(), g.data_ptr(), d.data_ptr(), _stream(s.device)))
        return d


class _ScalingRegTap(torch.autograd.Function):
    """The same value as _ScalingReg for the `scaling` output of expand.expand_compact, with the gradient sent to the
    expansion's `tap` instead of to `scaling`: csrc/expand.hip's backward kernel adds dL/dreg * d reg / d scaling to the
    rasterizer's dL/dscales while it reads them (at 88 M Gaussians the separate gradient tensor and autograd's
    accumulation pass were 0.9 ms of the cfg4 step)."""

    @staticmethod
    def forward(ctx, tap, scaling):
        from . import _C
        from .rasterizer import _stream
        s = scaling.detach().contiguous().float()
        P = s.shape[0]
        scratch = torch.empty(_C.lib.scr_scaling_reg_scratch_bytes(P), dtype=torch.uint8, device=s.device)
        out = torch.empty(1, dtype=torch.float32, device=s.device)
        with torch.cuda.device(s.device):
            _C.check(_C.lib.scr_scaling_reg_forward(P, s.data_ptr(), scratch.data_ptr(), out.data_ptr(), _stream(s.device)))
        return out.reshape(())

    @staticmethod
    def backward(ctx, g):
        return g.reshape(1), None


def scaling_reg(scaling):
    """mean(prod(scaling, dim=1)) (train.py:192-196)."""
    if scaling.is_cuda and scaling.dim() == 2 and scaling.shape[1] == 3 and scaling.shape[0] > 0:
        tap = None if os.environ.get("SPLATCO_NO_REG_TAP") else getattr(scaling, "_scr_reg_tap", None)
        if tap is not None and tap.requires_grad and scaling.dtype == torch.float32:
            return _ScalingRegTap.apply(tap, scaling)
        return _ScalingReg.apply(scaling)
    return scaling.prod(dim=1).mean()


def view_loss(image, gt_image, scaling, lambda_dssim=0.2):
    """Per-view training loss (train.py:192-196): 0.8 L1 + 0.2 (1 - SSIM) + 0.01 mean(prod(scaling))."""
    l1, s = l1_ssim(image, gt_image)
    return (1.0 - lambda_dssim) * l1 + lambda_dssim * (1.0 - s) + 0.01 * scaling_reg(scaling)
