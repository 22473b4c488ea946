"""Drop-in module: ``from diff_gaussian_rasterization import GaussianRasterizationSettings,
GaussianRasterizer`` (sings/rec/renderer/gs_renderer_single.py:6-9) resolves to the MI355X
implementation in sings_amd when this repository is on sys.path."""
from sings_amd.rasterizer import (  # noqa: F401
    GaussianRasterizationSettings,
    GaussianRasterizer,
    rasterize_gaussians,
    _RasterizeGaussians,
)
