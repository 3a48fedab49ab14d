"""Developer probe: time of the per-view training loss (train.py:192-196) at 1080p."""
import sys, time, torch
sys.path.insert(0, ".")
from splatco_amd.losses import view_loss, ssim, l1_loss
dev = torch.device("cuda:0")
img = torch.rand(3, 1080, 1920, device=dev, requires_grad=True)
gt = torch.rand(3, 1080, 1920, device=dev)
sc = torch.rand(1_000_000, 3, device=dev, requires_grad=True)
def step():
    img.grad = None
    view_loss(img, gt, sc).backward()
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): step()
torch.cuda.synchronize(); print(f"view_loss fwd+bwd at 1080p: {(time.perf_counter()-t0)/10*1e3:.2f} ms")
with torch.no_grad():
    for _ in range(3): ssim(img, gt)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): ssim(img, gt)
    torch.cuda.synchronize(); print(f"ssim fwd only: {(time.perf_counter()-t0)/10*1e3:.2f} ms")
