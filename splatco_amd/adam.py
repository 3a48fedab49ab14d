"""The optimizer of the training step as one streaming HIP pass: a drop-in for the reference's
`torch.optim.Adam(l, lr=0.0, eps=1e-15)` (scene/gaussian_model.py:575; stepped at train.py:310-312).

Same parameter groups, same state layout (`state[p] = {"step", "exp_avg", "exp_avg_sq"}`, `step` a CPU scalar tensor as in
torch's default Adam), so the optimizer surgery of densification (`densify.AnchorDensifier.replace_tensor_to_optimizer`,
`cat_tensors_to_optimizer`, `_prune_optimizer`: scene/gaussian_model.py:738-818) and `state_dict()` / `load_state_dict()`
work on it unchanged and checkpoints interchange with torch.optim.Adam.  What differs is the step: every group is one
launch of `scr_adam_step` (csrc/adam.hip) over parameter, gradient and moments -- at 20 M anchors that is 40 GB of
traffic, priced against the copy probe.  weight_decay / amsgrad / maximize are not part of the reference's optimizer and
are refused."""
import ctypes as C
import math

import torch

from . import _C


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False, maximize=False):
        if weight_decay != 0.0 or amsgrad or maximize:
            raise NotImplementedError("FusedAdam mirrors the reference's Adam(l, lr=0.0, eps=1e-15): no weight_decay / amsgrad / maximize")
        if not (0.0 <= betas[0] < 1.0 and 0.0 <= betas[1] < 1.0) or eps < 0.0 or lr < 0.0:
            raise ValueError(f"bad hyper-parameters: lr {lr}, betas {betas}, eps {eps}")
        # the keys torch.optim.Adam keeps in its groups ride along unchanged, so that a state_dict() of this optimizer
        # loads into torch's and back
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=0.0, amsgrad=False, maximize=False, foreach=None,
                                      capturable=False, differentiable=False, fused=None, decoupled_weight_decay=False))

    def _init_state(self, p):
        st = self.state[p]
        if len(st) == 0:
            st["step"] = torch.tensor(0.0, dtype=torch.float32)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return st

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            if group.get("weight_decay", 0.0) != 0.0 or group.get("amsgrad") or group.get("maximize"):
                raise NotImplementedError("FusedAdam: a group asks for weight_decay / amsgrad / maximize")
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            b1, b2 = group["betas"]
            lr = float(group["lr"])
            table = (_C.AdamTensor * len(ps))()
            dev = ps[0].device
            for e, p in zip(table, ps):
                g = p.grad
                st = self._init_state(p)
                m, v = st["exp_avg"], st["exp_avg_sq"]
                if not (p.is_cuda and p.device == dev and g.device == dev and m.device == dev and v.device == dev):
                    raise ValueError("FusedAdam: parameters, gradients and moments of a group must live on one GPU (no CPU path)")
                if not (p.dtype == g.dtype == m.dtype == v.dtype == torch.float32):
                    raise TypeError("FusedAdam: fp32 parameters, gradients and moments only")
                if g.is_sparse or not (p.is_contiguous() and g.is_contiguous() and m.is_contiguous() and v.is_contiguous()):
                    raise ValueError("FusedAdam: dense contiguous tensors only")
                if not (g.shape == p.shape == m.shape == v.shape):
                    raise ValueError(f"FusedAdam: shapes differ: param {tuple(p.shape)}, grad {tuple(g.shape)}, moments {tuple(m.shape)}")
                t = float(st["step"]) + 1.0
                e.param, e.grad, e.exp_avg, e.exp_avg_sq = p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr()
                e.numel = p.numel()
                e.step_size = lr / (1.0 - b1 ** t)                     # doubles, as torch's Python forms them
                e.bias_correction2_sqrt = math.sqrt(1.0 - b2 ** t)
            with torch.cuda.device(dev):
                _C.check(_C.lib.scr_adam_step(len(ps), table, b1, b2, float(group["eps"]), torch.cuda.current_stream(dev).cuda_stream))
            for p in ps:                        # the update is queued for every tensor of the group: the step counts move together
                self.state[p]["step"] += 1
        return loss
