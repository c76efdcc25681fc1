"""cnc_amd.nerfacc — the subset of nerfacc 0.5.3 that CNC uses (reference: nerfacc/__init__.py),
with the CUDA extension replaced by HIP kernels for gfx950.  PropNet sampling, pdf utilities and
the camera undistortion kernels are outside the CNC hot path and not provided."""
from .data_specs import RayIntervals, RaySamples
from .estimators.occ_grid import OccGridEstimator
from .grid import ray_aabb_intersect, traverse_grids
from .pack import pack_info
from .scan import exclusive_prod, exclusive_sum, inclusive_prod, inclusive_sum
from .volrend import (
    accumulate_along_rays,
    accumulate_along_rays_,
    render_transmittance_from_alpha,
    render_transmittance_from_density,
    render_visibility_from_alpha,
    render_visibility_from_density,
    render_weight_from_alpha,
    render_weight_from_density,
    rendering,
)

__version__ = "0.5.3+cnc.hip"

__all__ = [
    "__version__", "inclusive_prod", "exclusive_prod", "inclusive_sum", "exclusive_sum",
    "pack_info", "render_visibility_from_alpha", "render_visibility_from_density",
    "render_weight_from_alpha", "render_weight_from_density", "render_transmittance_from_alpha",
    "render_transmittance_from_density", "accumulate_along_rays", "accumulate_along_rays_",
    "rendering", "RayIntervals", "RaySamples", "ray_aabb_intersect", "traverse_grids",
    "OccGridEstimator",
]
