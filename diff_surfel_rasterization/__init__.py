"""Import shim: ``from diff_surfel_rasterization import GaussianRasterizationSettings,
GaussianRasterizer`` -- the exact import LaRa performs at lightning/renderer_2dgs.py:7-10 --
resolves to the MI355X-native operator.  Put the repo root on PYTHONPATH (or install this
directory next to LaRa) and train_lightning.py / eval_all.py / evaluation.py run unchanged."""
from lara_amd.rasterizer import (GaussianRasterizationSettings, GaussianRasterizer,  # noqa: F401
                                 rasterize_gaussians)

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians"]
