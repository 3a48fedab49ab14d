"""Drop-in module name: `from diff_gaussian_rasterization import GaussianRasterizationSettings,
GaussianRasterizer` (gaussian_renderer/__init__.py:15) resolves here when this repository is on
sys.path.  The implementation lives in splatco_amd/ (HIP kernels for MI355X behind a C-ABI)."""
from splatco_amd.rasterizer import (GaussianRasterizationSettings, GaussianRasterizer,  # noqa: F401
                                    rasterize_gaussians)
