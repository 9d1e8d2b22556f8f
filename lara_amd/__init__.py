"""lara_amd -- MI355X-native hot path of autonomousvision/LaRa.

Only what the hot path needs: the 2D-Gaussian-surfel rasteriser behind the
reference's ``GaussianRasterizer`` / ``GaussianRasterizationSettings`` operator API
(``lara_amd.rasterizer``; also importable as ``diff_surfel_rasterization`` from the repo root),
the camera-matrix helpers that feed it (``lara_amd.cameras``) and seeded synthetic scenes
(``lara_amd.synthetic``).  Kernels live in ``lara_amd/csrc`` (HIP, gfx950) behind the C ABI of
``include/lara2dgs.h``.
"""
from .rasterizer import (GaussianRasterizationSettings, GaussianRasterizer,  # noqa: F401
                         rasterize_gaussians, rasterize_gaussians_views)

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians", "rasterize_gaussians_views"]
