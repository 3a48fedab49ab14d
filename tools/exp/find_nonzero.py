import sys, types, traceback
import torch
sys.path.insert(0, ".")
orig_nonzero = torch.Tensor.nonzero
def traced_nonzero(self, *a, **k):
    if self.is_cuda and self.numel() > 100000:
        print("NONZERO", tuple(self.shape), self.dtype, "".join(traceback.format_stack(limit=4)[:-1]))
    return orig_nonzero(self, *a, **k)
torch.Tensor.nonzero = traced_nonzero
orig_sum = torch.Tensor.sum
def traced_sum(self, *a, **k):
    if self.is_cuda and self.numel() > 100000 and self.dtype in (torch.bool, torch.int64, torch.int32, torch.uint8):
        print("SUM", tuple(self.shape), self.dtype, "".join(traceback.format_stack(limit=4)[:-1]))
    return orig_sum(self, *a, **k)
torch.Tensor.sum = traced_sum
sys.argv = ["step_breakdown.py", "cfg3", "1000000", "1"]
exec(open("tools/step_breakdown.py").read())
