"""Camera matrix conventions of the hot path (host side, plain numpy/torch).

Restates -- does not import -- the reference's conventions; pinned by tests/golden/cameras.npz,
captured from the reference's own functions (tools/make_golden.py):
  * getWorld2View2         utils/graphics_utils.py:38-49
  * getProjectionMatrix    utils/graphics_utils.py:51-71   (znear .01, zfar 100: scene/cameras.py:48-49)
  * world_view_transform = W2C^T, full_proj_transform = W2C^T @ P^T, camera_center =
    inverse(world_view_transform)[3,:3]                     scene/cameras.py:54-58
"""
import math
from dataclasses import dataclass

import numpy as np
import torch


def get_world2view2(R, t, translate=(0.0, 0.0, 0.0), scale=1.0):
    Rt = np.zeros((4, 4), dtype=np.float64)
    Rt[:3, :3] = np.asarray(R, dtype=np.float64).T
    Rt[:3, 3] = np.asarray(t, dtype=np.float64)
    Rt[3, 3] = 1.0
    C2W = np.linalg.inv(Rt)
    C2W[:3, 3] = (C2W[:3, 3] + np.asarray(translate, dtype=np.float64)) * scale
    return np.linalg.inv(C2W).astype(np.float32)


def get_projection_matrix(znear, zfar, fovX, fovY):
    tx, ty = math.tan(fovX / 2), math.tan(fovY / 2)
    top, right = ty * znear, tx * znear
    bottom, left = -top, -right
    P = torch.zeros(4, 4)
    P[0, 0] = 2.0 * znear / (right - left)
    P[1, 1] = 2.0 * znear / (top - bottom)
    P[0, 2] = (right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


@dataclass
class MiniCam:
    """The attributes gaussian_renderer.render()/prefilter_voxel() read from a camera
    (gaussian_renderer/__init__.py:34,142-156); same names as scene/cameras.py:17-105."""
    image_width: int
    image_height: int
    FoVx: float
    FoVy: float
    world_view_transform: torch.Tensor
    full_proj_transform: torch.Tensor
    camera_center: torch.Tensor
    uid: int = 0
    znear: float = 0.01
    zfar: float = 100.0

    def to(self, device):
        return MiniCam(self.image_width, self.image_height, self.FoVx, self.FoVy,
                       self.world_view_transform.to(device), self.full_proj_transform.to(device),
                       self.camera_center.to(device), self.uid, self.znear, self.zfar)


def make_camera(R, T, FoVx, FoVy, width, height, uid=0, trans=(0.0, 0.0, 0.0), scale=1.0,
                znear=0.01, zfar=100.0):
    """Build the three tensors exactly as scene/cameras.py:54-58 does (on CPU)."""
    wvt = torch.tensor(get_world2view2(R, T, trans, scale)).transpose(0, 1)
    proj = get_projection_matrix(znear, zfar, FoVx, FoVy).transpose(0, 1)
    full = wvt.unsqueeze(0).bmm(proj.unsqueeze(0)).squeeze(0)
    center = wvt.inverse()[3, :3]
    return MiniCam(int(width), int(height), float(FoVx), float(FoVy), wvt.contiguous(),
                   full.contiguous(), center.contiguous(), uid, znear, zfar)


def look_at_camera(eye, target, up, FoVx, width, height, uid=0):
    """Convenience: a camera at `eye` looking at `target` (COLMAP convention: +z forward,
    +y down), FoVy from the aspect ratio."""
    eye, target, up = (np.asarray(v, dtype=np.float64) for v in (eye, target, up))
    f = target - eye
    f /= np.linalg.norm(f)
    r = np.cross(f, up)
    r /= np.linalg.norm(r)
    d = np.cross(f, r)
    R = np.stack([r, d, f], axis=1)          # camera-to-world rotation (columns = cam axes)
    T = -R.T @ eye                           # world-to-camera translation
    FoVy = 2.0 * math.atan(math.tan(FoVx / 2) * height / width)
    return make_camera(R, T, FoVx, FoVy, width, height, uid)
